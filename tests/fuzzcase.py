"""One case of the differential fuzz campaign (tools/diff_fuzz.py) as a function of ONE seed: the input file or directory,
the output mode, the options, and how the runner is to be driven (middle scan behind the exact prefilter or the score bound;
whole file or a stream of small blocks).  The build container runs the unchanged reference CLI over the case and records
the md5 of every output file; the GPU box regenerates the very same case from the seed (tools/replay_fuzz.py,
tests/test_gpu_fuzz_replay.py) and must produce those files through the HIP library -- no reference needed there, no
input files in the repository.  `content_md5` pins the regenerated input (decompressed bytes, files in path order)."""
import gzip
import hashlib
import os
import random

from tests import readgen


def random_options(rng, barcodes):
    o = []

    def maybe(p, *args):
        if rng.random() < p:
            o.extend(args)
    maybe(0.3, "--end_size", str(rng.choice([30, 80, 120, 150, 200])))
    maybe(0.3, "--min_trim_size", str(rng.choice([0, 2, 4, 10])))
    maybe(0.3, "--extra_end_trim", str(rng.choice([0, 1, 2, 7])))
    maybe(0.3, "--end_threshold", str(rng.choice([60, 75, 90])))
    maybe(0.3, "--middle_threshold", str(rng.choice([75, 85, 90, 97])))
    maybe(0.2, "--adapter_threshold", str(rng.choice([80, 90, 97])))
    maybe(0.2, "--check_reads", str(rng.choice([5, 40, 10000])))
    maybe(0.2, "--min_split_read_size", str(rng.choice([1, 200, 1000, 3000])))
    maybe(0.2, "--extra_middle_trim_good_side", str(rng.choice([0, 10, 50])))
    maybe(0.2, "--extra_middle_trim_bad_side", str(rng.choice([0, 100, 300])))
    maybe(0.15, "--scoring_scheme", rng.choice(["3,-6,-5,-2", "2,-3,-5,-2", "3,-6,-2,-5", "4,-5,-6,-6"]))
    maybe(0.15, "--no_split")
    maybe(0.15, "--discard_middle")
    maybe(0.25, "--format", rng.choice(["fasta", "fastq", "fastq.gz", "auto"]))
    if barcodes:
        maybe(0.3, "--require_two_barcodes")
        maybe(0.3, "--barcode_threshold", str(rng.choice([60, 75, 85])))
        maybe(0.3, "--barcode_diff", str(rng.choice([0, 5, 15])))
        maybe(0.2, "--untrimmed")
        maybe(0.2, "--discard_unassigned")
    return o


def _plain(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:2] == b"\x1f\x8b":
        out, d = b"", data
        while d[:2] == b"\x1f\x8b":                      # members, possibly with zero padding between them
            z = __import__("zlib").decompressobj(31)
            out += z.decompress(d)
            d = z.unused_data.lstrip(b"\0")
        return out
    return data


def content_md5(inp):
    h = hashlib.md5()
    if os.path.isdir(inp):
        for root, dirs, files in sorted(os.walk(inp)):
            dirs.sort()
            for f in sorted(files):
                h.update(os.path.relpath(os.path.join(root, f), inp).encode())
                h.update(_plain(os.path.join(root, f)))
    else:
        h.update(_plain(inp))
    return h.hexdigest()


def make_case(cseed, work, sized_gzip=None):
    """-> dict(input, mode, argv, prefilter, blocks, kind).  sized_gzip(src, dst): writer of the sized-member layout
    (porechop_amd.io.gzip_file; given by the caller so that this module needs no library)."""
    rng = random.Random(cseed)
    kind = rng.choice(["native", "native", "rapid", "ligation", "edge"])
    seed = rng.randint(1, 10 ** 6)
    nreads = rng.choice([25, 60])
    reads = {"native": lambda: readgen.native_reads(seed, nreads, barcodes=tuple(rng.sample(range(1, 13), 3))),
             "rapid": lambda: readgen.rapid_reads(seed, nreads), "ligation": lambda: readgen.ligation_reads(seed, nreads),
             "edge": lambda: None}[kind]()
    os.makedirs(work, exist_ok=True)
    if reads is not None and rng.random() < 0.3:
        # odd reads among the ordinary ones: RNA (more U than T: aligned as T, written back with EVERY T as U,
        # nanopore_read.py:23-35,97-147), a few U's only, lower case, runs of N / '-', qualities shorter than the sequence,
        # names with tabs and repeated blanks, an empty read
        odd = []
        for name, seq, qual in reads:
            r = rng.random()
            if r < 0.08:
                seq = seq.replace("T", "U")
            elif r < 0.12:
                seq = "".join(("U" if c == "T" and rng.random() < 0.3 else c) for c in seq)
            elif r < 0.18:
                seq = seq.lower()
            elif r < 0.22 and len(seq) > 300:
                p0 = rng.randrange(len(seq) - 100)
                seq = seq[:p0] + rng.choice("N-n") * rng.randrange(1, 90) + seq[p0 + 60:]
                qual = (qual * 2)[:len(seq)]
            elif r < 0.25:
                qual = qual[:rng.randrange(len(qual) + 1)]
            elif r < 0.28:
                name = name + "\tx  y " + name
            elif r < 0.29:
                seq, qual = "", ""
            odd.append((name, seq, qual))
        reads = odd
    if reads is None:
        inp = readgen.build_dataset("edge", work)
    elif rng.random() < 0.15:
        # a directory the way Albacore / Guppy lay them out (porechop.py:232-259): fastq files found recursively, in path
        # order, the check reads spread over the files, the basecaller's barcode taken from a /barcodeNN/ or /unclassified/
        # path component; some files gzip-ed, a file that is not a fastq in between, an upper-case extension
        inp = os.path.join(work, "indir")
        subs = rng.sample(["", "pass/barcode01", "pass/barcode02", "pass/barcode11", "fail/unclassified", "x/barcode07/y", "misc"], rng.randrange(1, 5))
        nfiles = rng.randrange(1, 7)
        cuts = sorted(rng.randrange(len(reads) + 1) for _ in range(nfiles - 1))
        for j, (a, b) in enumerate(zip([0] + cuts, cuts + [len(reads)])):
            d = os.path.join(inp, rng.choice(subs))
            os.makedirs(d, exist_ok=True)
            text = readgen.fastq_text(reads[a:b]).encode()
            if not text and rng.random() < 0.7:
                continue                                               # (an empty .fastq file ends the reference with an error: rarely)
            ext = rng.choice([".fastq", ".fastq", ".fastq.gz", ".FASTQ"])
            with open(os.path.join(d, "part%d%s" % (rng.randrange(1000), ext)), "wb") as f:
                f.write(gzip.compress(text, 1) if ext.endswith(".gz") else text)
        os.makedirs(os.path.join(inp, "misc"), exist_ok=True)
        open(os.path.join(inp, "misc", "notes.txt"), "w").write("not reads\n")
    else:
        as_fasta = rng.random() < 0.2
        inp = os.path.join(work, "in.fasta" if as_fasta else "in.fastq")
        with open(inp, "w") as f:
            f.write(readgen.fasta_text(reads) if as_fasta else readgen.fastq_text(reads))
        # a third of the inputs gzip-ed, in the layouts the readers tell apart (the reference decides by magic bytes,
        # porechop/misc.py:60-81): one member, several members with zero padding between two of them, sized members
        layout = rng.choice(["", "", "one", "members", "sized"])
        if layout:
            text = open(inp, "rb").read()
            os.remove(inp)
            inp += ".gz"
            if layout == "one":
                blob = gzip.compress(text, rng.choice([1, 6, 9]))
            elif layout == "sized" and sized_gzip is not None:
                open(inp[:-3], "wb").write(text)
                sized_gzip(inp[:-3], inp)
                os.remove(inp[:-3])
                blob = None
            elif layout == "sized":
                blob = gzip.compress(text, 6)
            else:
                marks = sorted({0, len(text)} | {m for m in (text.find(b"\n>" if as_fasta else b"\n@", rng.randrange(max(1, len(text)))) + 1 for _ in range(3)) if m > 0})
                blob = b""
                for a, b in zip(marks, marks[1:]):
                    blob += gzip.compress(text[a:b], 1) + (b"\0" * rng.choice([0, 0, 13, 600]))
            if blob is not None:
                open(inp, "wb").write(blob)
    barcodes = kind in ("native", "rapid", "edge") and rng.random() < 0.5
    extra = random_options(rng, barcodes)
    if not barcodes and "--untrimmed" in extra:
        extra.remove("--untrimmed")
    mode = "b" if barcodes else "o:" + rng.choice(["out.fastq", "out.fasta", "out.fastq.gz", "out.txt"])
    # half of the runs take the middle scan behind the exact PREFILTER, the GPU library's default route (Pipeline.phase_c(prefilter
    # =True) -> _prefiltered_scan: survivors, set grouping, sparse records, the masked rounds); the other half behind the score bound
    prefilter = rng.random() < 0.5
    # a third of the runs as a STREAM of small blocks (run_streamed: plain and gzip FASTQ files above two blocks; parse k+1 ||
    # scan k || write k-1, phase A on the first block, gzip output appended member by member)
    blocks = str(rng.choice([3000, 20000, 150000])) if rng.random() < 0.33 else None
    return {"input": inp, "mode": mode, "argv": extra, "prefilter": prefilter, "blocks": blocks, "kind": kind}
