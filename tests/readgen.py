"""Seeded synthetic read SETS (whole FASTQ/FASTA inputs) for the end-to-end runner tests.

tests/golden/make_golden.py writes these inputs to disk, runs the unchanged reference CLI over them
and records the md5 of every output file (tests/golden/runner_goldens.json); the tests regenerate
the same inputs from the same seeds (a sha1 of each input is stored to detect generator drift) and
run porechop_amd.runner over them.  Shapes: native-barcoded reads (Y adapters + reverse barcodes
with their flanks), rapid-barcoded reads (forward barcodes), plain ligation reads with chimeric
junctions, short reads, RNA, lower case, reads without adapters, FASTA, gzip, an Albacore-style
directory tree."""
import gzip
import hashlib
import io
import json
import os
import random

from tests.pairgen import mutate

HERE = os.path.dirname(os.path.abspath(__file__))


def _panel():
    with open(os.path.join(HERE, "golden", "panel.json")) as f:
        return {a["name"]: a for a in json.load(f)}


def _body(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def _quals(rng, n):
    return "".join(chr(33 + rng.randint(2, 40)) for _ in range(n))


Y_TOP = "AATGTACTTCGTTCAGTTACGTATTGCT"
Y_BOTTOM = "GCAATACGTAACTGAACGAAGT"


def native_reads(seed, n, barcodes=(1, 2, 3, 7), with_chimeras=True):
    """Native barcoding: Y_Top + flank + BCxx_rev-side barcode + flank | body | mirror image."""
    rng = random.Random(seed)
    panel = _panel()
    reads = []
    for i in range(n):
        kind = rng.random()
        length = rng.choice([400, 1200, 2500, 4000])
        body = _body(rng, length)
        b = rng.choice(barcodes)
        bs = panel["Barcode %d (reverse)" % b]
        start = "AATGTACTTCGTTCAGTTACGTATTGCTAAGGTTAA" + bs["start"][1] + "CAGCACCT"
        b_end = b if rng.random() < 0.8 else rng.choice(barcodes)          # some reads disagree at the two ends
        be = panel["Barcode %d (reverse)" % b_end]
        end = "AGGTGCTG" + be["end"][1] + "TTAACCTTAGCAATACGTAACTGAACGAAGT"
        rate = rng.choice([0.0, 0.04, 0.08, 0.15])
        seq = body
        if kind < 0.75:
            seq = mutate(rng, start, rate)[rng.randint(0, 6):] + seq
        if kind > 0.15:
            e = mutate(rng, end, rate)
            seq = seq + e[:len(e) - rng.randint(0, 6)]
        if with_chimeras and rng.random() < 0.12 and length >= 2500:
            pos = rng.randint(1100, len(seq) - 1100)
            seq = seq[:pos] + mutate(rng, end, 0.03) + mutate(rng, start, 0.03) + seq[pos:]
        if rng.random() < 0.04:
            seq = seq[:rng.randint(20, 140)]                                # shorter than the end windows
        if rng.random() < 0.05:
            seq = seq.lower()
        if rng.random() < 0.03:
            seq = seq.replace("T", "U")                                     # RNA
        name = "read%04d" % i + (" runid=abc ch=%d" % rng.randint(1, 512) if rng.random() < 0.7 else "")
        reads.append((name, seq, _quals(rng, len(seq))))
    return reads


def rapid_reads(seed, n, barcodes=(2, 4, 9)):
    """Rapid barcoding (SQK-RBK004 style): upstream + forward barcode + rapid adapter | body."""
    rng = random.Random(seed)
    panel = _panel()
    reads = []
    for i in range(n):
        b = rng.choice(barcodes)
        bc = panel["Barcode %d (forward)" % b]["start"][1]
        start = "AATGTACTTCGTTCAGTTACG" + "GCTTGGGTGTTTAACC" + bc + "GTTTTCGCATTTATCGTGAAACGCTTTCGCGTTTTTCGTGCGCCGCTTCA"
        seq = _body(rng, rng.choice([600, 1500, 3000]))
        if rng.random() < 0.85:
            seq = mutate(rng, start, rng.choice([0.0, 0.05, 0.1]))[rng.randint(0, 10):] + seq
        reads.append(("rapid%04d" % i, seq, _quals(rng, len(seq))))
    return reads


def ligation_reads(seed, n):
    """Plain SQK-NSK007 ligation reads, 8 % chimeras (Y_Bottom + Y_Top junction), no barcodes."""
    rng = random.Random(seed)
    reads = []
    for i in range(n):
        seq = _body(rng, rng.choice([900, 3000, 6000]))
        if rng.random() < 0.9:
            seq = mutate(rng, Y_TOP, 0.1)[rng.randint(0, 8):] + seq
        if rng.random() < 0.5:
            e = mutate(rng, Y_BOTTOM, 0.1)
            seq += e[:len(e) - rng.randint(0, 8)]
        if rng.random() < 0.08 and len(seq) > 2600:
            for _ in range(rng.choice([1, 1, 2])):
                pos = rng.randint(1050, len(seq) - 1050)
                seq = seq[:pos] + mutate(rng, Y_BOTTOM, 0.04) + mutate(rng, Y_TOP, 0.04) + seq[pos:]
        reads.append(("lig%04d some description" % i, seq, _quals(rng, len(seq))))
    return reads


def fastq_text(reads):
    return "".join("@%s\n%s\n+\n%s\n" % r for r in reads)


def fasta_text(reads, width=80):
    out = []
    for name, seq, _ in reads:
        out.append(">" + name + "\n")
        for p in range(0, len(seq), width):
            out.append(seq[p:p + width] + "\n")
    return "".join(out)


def _gz(text):
    buf = io.BytesIO()
    with gzip.GzipFile(fileobj=buf, mode="wb", mtime=0) as g:
        g.write(text.encode())
    return buf.getvalue()


# dataset name -> builder(dir) -> input path (file or directory)
def build_dataset(name, root):
    os.makedirs(root, exist_ok=True)

    def put(rel, data):
        p = os.path.join(root, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "wb") as f:
            f.write(data if isinstance(data, bytes) else data.encode())
        return p

    if name == "native":
        return put("native.fastq", fastq_text(native_reads(11, 160)))
    if name == "native_fasta":
        return put("native.fasta", fasta_text(native_reads(12, 90)))
    if name == "native_gz":
        return put("native_gz.fastq.gz", _gz(fastq_text(native_reads(13, 90))))
    if name == "rapid":
        return put("rapid.fastq", fastq_text(rapid_reads(21, 120)))
    if name == "ligation":
        return put("ligation.fastq", fastq_text(ligation_reads(31, 140)))
    if name == "nothing":
        rng = random.Random(41)
        return put("nothing.fastq", fastq_text([("plain%d" % i, _body(rng, 700), "5" * 700) for i in range(12)]))
    if name == "edge":
        # boundary shapes: empty / 1-base / exactly end_size reads, adapter-only reads, reads that trim to
        # nothing, names with tabs and repeated spaces, short qualities, a read that is all N, CRLF-free
        rng = random.Random(61)
        panel = _panel()
        bc = panel["Barcode 5 (reverse)"]
        start = "AATGTACTTCGTTCAGTTACGTATTGCTAAGGTTAA" + bc["start"][1] + "CAGCACCT"
        end = "AGGTGCTG" + bc["end"][1] + "TTAACCTTAGCAATACGTAACTGAACGAAGT"
        rr = [("empty", "", ""), ("one", "A", "I"), ("two words\ttab  double", start + end, "5" * len(start + end)),
              ("only_start", start, "5" * len(start)), ("only_end", end, "5" * len(end)),
              ("exact150", start + _body(rng, 150 - len(start)), "5" * 150),
              ("exact151 x", start + _body(rng, 151 - len(start)), "5" * 151),
              ("alln", "N" * 400, "5" * 400), ("dashes", start + "-" * 30 + _body(rng, 300) + end, "5" * (330 + len(start + end))),
              ("shortq", start + _body(rng, 500) + end, "5" * 100)]
        for i in range(40):
            body = _body(rng, rng.choice([10, 60, 149, 150, 151, 299, 300, 301, 1000, 2300]))
            s5 = mutate(rng, start, 0.05) if rng.random() < 0.8 else ""
            e5 = mutate(rng, end, 0.05) if rng.random() < 0.8 else ""
            seq = s5 + body + e5
            if i % 9 == 0 and len(body) >= 1000:
                seq = s5 + body[:500] + end + start + body[500:] + e5          # junction close to both ends
            rr.append(("e%02d" % i, seq, _quals(rng, len(seq))))
        return put("edge.fastq", fastq_text(rr))
    if name == "bc_tie":
        # BC01 at one read's start, at the other read's end.  The two orientations tie on the
        # or-sums; the reference decides on the and-sums, which need the EXACT below-threshold identities
        rng = random.Random(71)
        bc = _panel()["Barcode 1 (forward)"]["start"][1]
        a, b = _body(rng, 900), _body(rng, 1100)
        rr = [("tie_start", bc + a, "5" * (len(bc) + 900)), ("tie_end", b + bc, "5" * (len(bc) + 1100))]
        return put("bc_tie.fastq", fastq_text(rr))
    if name == "albacore":
        # workspace/pass/barcodeXX/*.fastq + unclassified, as Albacore lays them out
        reads = native_reads(51, 150, barcodes=(1, 2, 3))
        d = os.path.join(root, "albacore")
        groups = {"pass/barcode01": reads[0:40], "pass/barcode02": reads[40:75], "pass/barcode03": reads[75:100],
                  "pass/unclassified": reads[100:130], "fail/barcode01": reads[130:150]}
        for sub, rr in groups.items():
            put(os.path.join("albacore", "workspace", sub, "fastq_runid_0.fastq"), fastq_text(rr))
        return d
    raise KeyError(name)


def _content(path):
    with open(path, "rb") as fh:
        data = fh.read()
    return gzip.decompress(data) if data[:3] == b"\x1f\x8b\x08" else data   # hash what is IN a gzip file, not its framing


def dataset_sha1(path):
    h = hashlib.sha1()
    if os.path.isdir(path):
        for d, _, fs in sorted(os.walk(path)):
            for f in sorted(fs):
                h.update(os.path.relpath(os.path.join(d, f), path).encode())
                h.update(_content(os.path.join(d, f)))
    else:
        h.update(_content(path))
    return h.hexdigest()


def output_md5s(target):
    """file name -> md5 of its (decompressed) content, for a -o file or a -b directory."""
    if os.path.isdir(target):
        return {f: hashlib.md5(_content(os.path.join(target, f))).hexdigest() for f in sorted(os.listdir(target))}
    return {os.path.basename(target): hashlib.md5(_content(target)).hexdigest()}


# (case name, dataset, mode, extra argv)   mode: "o:<filename>" = -o file, "b" = -b dir
RUNNER_CASES = [
    ("native_default", "native", "o:out.fastq", []),
    ("native_to_fasta", "native", "o:out.fasta", []),
    ("native_format_fasta", "native", "o:out.txt", ["--format", "fasta"]),
    ("native_gz_out", "native", "o:out.fastq.gz", []),
    ("native_no_split", "native", "o:out.fastq", ["--no_split"]),
    ("native_discard_middle", "native", "o:out.fastq", ["--discard_middle"]),
    ("native_split_sizes", "native", "o:out.fastq", ["--min_split_read_size", "300", "--extra_middle_trim_good_side", "3",
                                                     "--extra_middle_trim_bad_side", "40"]),
    ("native_end_opts", "native", "o:out.fastq", ["--end_size", "90", "--min_trim_size", "8", "--extra_end_trim", "5",
                                                  "--end_threshold", "85"]),
    ("native_check20", "native", "o:out.fastq", ["--check_reads", "20", "--adapter_threshold", "95"]),
    ("native_scheme", "native", "o:out.fastq", ["--scoring_scheme", "2,-3,-5,-2"]),
    ("native_linear_scheme", "native", "o:out.fastq", ["--scoring_scheme", "3,-6,-5,-5"]),
    ("native_bins", "native", "b", []),
    ("native_bins_two", "native", "b", ["--require_two_barcodes"]),
    ("native_bins_strict", "native", "b", ["--barcode_threshold", "85", "--barcode_diff", "12"]),
    ("native_bins_discard", "native", "b", ["--discard_unassigned"]),
    ("native_bins_untrimmed", "native", "b", ["--untrimmed"]),
    ("native_bins_fasta", "native", "b", ["--format", "fasta"]),
    ("native_bins_gz", "native", "b", ["--format", "fastq.gz"]),
    ("native_fasta_in", "native_fasta", "o:out.fasta", []),
    ("native_fasta_in_fastq_out", "native_fasta", "o:out.fastq", []),
    ("native_fasta_bins", "native_fasta", "b", []),
    ("native_gz_in_bins", "native_gz", "b", []),
    ("rapid_default", "rapid", "o:out.fastq", []),
    ("rapid_bins", "rapid", "b", []),
    ("ligation_default", "ligation", "o:out.fastq", []),
    ("ligation_mid80", "ligation", "o:out.fastq", ["--middle_threshold", "80"]),
    ("nothing_found", "nothing", "o:out.fastq", []),
    ("native_gap_scheme", "native", "o:out.fastq", ["--scoring_scheme", "3,-6,-2,-5"]),
    ("native_loose", "native", "b", ["--barcode_threshold", "0", "--barcode_diff", "0", "--end_threshold", "50",
                                     "--middle_threshold", "70", "--adapter_threshold", "60"]),
    ("native_tight", "native", "o:out.fastq", ["--end_threshold", "100", "--middle_threshold", "100", "--min_trim_size", "0",
                                               "--extra_end_trim", "0", "--min_split_read_size", "1"]),
    ("native_end20", "native", "o:out.fastq", ["--end_size", "20"]),
    ("native_check0", "native", "o:out.fastq", ["--check_reads", "0"]),
    ("edge_default", "edge", "o:out.fastq", []),
    ("edge_fasta", "edge", "o:out.fasta", []),
    ("edge_bins", "edge", "b", []),
    ("edge_bins_two_untrimmed", "edge", "b", ["--require_two_barcodes", "--untrimmed"]),
    ("edge_small_split", "edge", "o:out.fastq", ["--min_split_read_size", "0", "--extra_middle_trim_good_side", "0",
                                                 "--extra_middle_trim_bad_side", "0"]),
    ("albacore_bins", "albacore", "b", []),
    ("albacore_bins_check30", "albacore", "b", ["--check_reads", "30"]),
    ("albacore_file_out", "albacore", "o:out.fastq", []),
    ("bc_tie_bins", "bc_tie", "b", []),
    # schemes the packed 16-bit kernels do not take (the reference takes any four integers, porechop.py:145,196-202):
    # the default scheme x 100 (same alignments, magnitudes beyond 16 bits), a zero gap extension, a positive gap opening
    ("ligation_wide_scheme", "ligation", "o:out.fastq", ["--scoring_scheme", "300,-600,-500,-200"]),
    ("ligation_zero_extend_scheme", "ligation", "o:out.fastq", ["--scoring_scheme", "4,-5,-3,0"]),
    ("edge_posgap_scheme", "edge", "o:out.fastq", ["--scoring_scheme", "5,-7,1,-3"]),
]
