"""Host ingest (porechop_amd/csrc/pc_io.cpp) against the reference's own loaders
(porechop/misc.py load_fasta_or_fastq + NanoporeRead.__init__) on the reference's fixture files when
the checkout is present, and against a Python restatement of those semantics on generated
FASTA/FASTQ(.gz) files everywhere."""
import gzip
import os
import random
import sys

import pytest

from porechop_amd.io import ReadSet

REFERENCE = "/root/reference"


def python_semantics(path):
    """Restatement of misc.py:60-168 + nanopore_read.py:23-35 (test infrastructure)."""
    with open(path, "rb") as f:
        gz = f.read(3) == b"\x1f\x8b\x08"
    op = gzip.open if gz else open
    out = []
    with op(path, "rt") as f:
        first = f.read(1)
    with op(path, "rt") as f:
        if first == "@":
            for line in f:
                full = line.strip()[1:]
                seq = next(f).strip(); next(f); q = next(f).strip()
                out.append((full, seq, q))
        else:
            name, seq = "", ""
            for line in f:
                line = line.strip()
                if not line:
                    continue
                if line[0] == ">":
                    if name:
                        out.append((name, seq, None))
                        seq = ""
                    name = line[1:]
                else:
                    seq += line
            if name:
                out.append((name, seq, None))
    norm = []
    for name, seq, q in out:
        s = seq.upper()
        rna = s.count("U") > s.count("T")
        if rna:
            s = s.replace("U", "T")
        if q is not None and len(q) < len(seq):
            q += "+" * (len(seq) - len(q))
        norm.append((name, s, q, rna))
    return norm, first == "@"


def check(path, want, want_fastq):
    rs = ReadSet(path)
    assert rs.is_fastq == want_fastq and rs.count == len(want)
    for i, (name, seq, q, rna) in enumerate(want):
        assert rs.name(i) == name and rs.seq(i) == seq and rs.is_rna(i) == rna
        if want_fastq:
            assert rs.quals(i) == q
    assert rs.arena[-64:].tobytes() == b"N" * 64
    assert int(rs.lengths.sum()) + 64 == rs.arena.size
    rs.close()


def test_generated_files(tmp_path):
    rng = random.Random(3)
    recs = []
    for i in range(200):
        n = rng.choice([0, 1, 17, 150, 999, 4000])
        alpha = rng.choice(["ACGT", "acgtn", "ACGU", "ACGTU"])
        recs.append(("read_%d some description %d" % (i, i), "".join(rng.choice(alpha) for _ in range(n))))
    fq = tmp_path / "a.fastq"
    with open(fq, "w") as f:
        for k, (name, seq) in enumerate(recs):
            q = "5" * (len(seq) if k % 7 else max(0, len(seq) - 3))
            f.write("@%s\n%s\n+\n%s\n" % (name, seq, q))
    fa = tmp_path / "a.fasta"
    with open(fa, "w") as f:
        for name, seq in recs:
            if not name:
                continue
            f.write(">%s\r\n" % name)
            for i in range(0, len(seq), 60):
                f.write(seq[i:i + 60] + "\n")
            f.write("\n")
    for p in (fq, fa):
        gzp = str(p) + ".gz"
        with open(p, "rb") as src, gzip.open(gzp, "wb") as dst:
            dst.write(src.read())
        for path in (str(p), gzp):
            want, is_fq = python_semantics(path)
            check(path, want, is_fq)


def test_errors(tmp_path):
    bad = tmp_path / "bad.txt"
    bad.write_text("hello\n")
    with pytest.raises(ValueError):
        ReadSet(bad)
    with pytest.raises(ValueError):
        ReadSet(tmp_path / "missing.fastq")
    trunc = tmp_path / "trunc.fastq"
    trunc.write_text("@r1\nACGT\n+\n")
    with pytest.raises(ValueError):
        ReadSet(trunc)


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference checkout not present")
def test_against_reference_loaders_on_its_fixtures():
    sys.path.insert(0, REFERENCE)
    for m in [k for k in sys.modules if k == "porechop" or k.startswith("porechop.")]:
        del sys.modules[m]
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("pc_ref_misc", os.path.join(REFERENCE, "porechop", "misc.py"))
        misc = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(misc)
    finally:
        sys.path.remove(REFERENCE)
    files = ["test_one_adapter_set.fastq", "test_barcodes.fastq", "test_choose_barcodes_1.fasta", "test_format.fastq.gz",
             "test_format.fasta.gz", "test_format.fasta", "test_format_barcodes.fastq"]
    for fn in files:
        path = os.path.join(REFERENCE, "test", fn)
        recs, kind = misc.load_fasta_or_fastq(path)
        want = []
        for r in recs:
            if kind == "FASTQ":
                name, seq, q = r[4], r[1], r[3]
            else:
                name, seq, q = r[2], r[1], None
            s = seq.upper()
            rna = s.count("U") > s.count("T")
            if rna:
                s = s.replace("U", "T")
            if q is not None and len(q) < len(seq):
                q += "+" * (len(seq) - len(q))
            want.append((name, s, q, rna))
        check(path, want, kind == "FASTQ")


def test_parallel_fastq_parse_equals_the_semantics(tmp_path):
    """Files above 1 MB are parsed by several threads cut at record boundaries (a quality line may
    start with '@'); the result must be what the reference's line-by-line loader yields.  Also
    CRLF line ends and a file that is NOT regular 4-line FASTQ (falls back to the serial parser
    and its error)."""
    rng = random.Random(8)
    recs = []
    for i in range(1500):
        n = rng.choice([1, 40, 700, 5000])
        seq = "".join(rng.choice("ACGTacgtNU") for _ in range(n))
        q = "".join(rng.choice("@+5!~I") for _ in range(n if rng.random() < 0.9 else max(0, n - 3)))
        recs.append("@r%d desc %d\n%s\n+%s\n%s\n" % (i, i, seq, "" if i % 3 else "r%d" % i, q))
    for name, text in (("big.fastq", "".join(recs)), ("crlf.fastq", "".join(recs).replace("\n", "\r\n"))):
        path = tmp_path / name
        with open(path, "w", newline="") as f:
            f.write(text)
        assert os.path.getsize(path) > (1 << 20)
        want, is_fastq = python_semantics(str(path))
        for threads in ("1", "3", "16"):
            os.environ["PC_IO_THREADS"] = threads
            check(str(path), want, True)
    os.environ.pop("PC_IO_THREADS", None)
    bad = tmp_path / "bad.fastq"
    with open(bad, "w") as f:
        f.write("".join(recs) + "@dangling\nACGT\n")
    with pytest.raises(ValueError, match="could not be parsed"):
        ReadSet(str(bad))


def test_parallel_fasta_parse_equals_the_semantics(tmp_path):
    """FASTA files above 1 MB are parsed by several threads, cut where a line begins with '>' (misc.py:115-148 strips every
    line, skips blank ones, and takes a stripped line starting with '>' as a header): multi-line sequences, blank lines,
    headers with blanks in front of the '>', records without a name, CRLF -- the result is the line-by-line loader's."""
    rng = random.Random(9)
    recs = []
    for i in range(2500):
        n = rng.choice([0, 1, 40, 700, 5000])
        seq = "".join(rng.choice("ACGTacgtNU") for _ in range(n))
        width = rng.choice([60, 70, 1000000])
        lines = [seq[k:k + width] for k in range(0, len(seq), width)]
        if rng.random() < 0.1:
            lines.insert(rng.randrange(len(lines) + 1), "")                  # a blank line inside a record
        if rng.random() < 0.1:
            lines = ["  " + l + " " for l in lines]                           # blanks around sequence lines
        head = (">" if i % 97 != 3 else " >") + ("" if i % 211 == 5 else "r%d desc %d" % (i, i)) + (" " if i % 5 == 0 else "")
        recs.append(head + "\n" + "".join(l + "\n" for l in lines))
    for name, text in (("big.fasta", "".join(recs)), ("crlf.fasta", "".join(recs).replace("\n", "\r\n"))):
        path = tmp_path / name
        with open(path, "w", newline="") as f:
            f.write(text)
        assert os.path.getsize(path) > (1 << 20)
        want, is_fastq = python_semantics(str(path))
        assert not is_fastq and len(want) > 2000
        for threads in ("1", "3", "16"):
            os.environ["PC_IO_THREADS"] = threads
            check(str(path), want, False)
    # every other record nameless: its bases go in front of the next record's (the reference clears the running sequence only
    # under `if name:`), and a cut that falls behind one makes the parallel parser hand the file to the serial one
    path = tmp_path / "nameless.fasta"
    with open(path, "w") as f:
        for i in range(30000):
            f.write((">r%d\n" % i if i % 2 else ">\n") + "".join(rng.choice("ACGTU") for _ in range(40)) + "\n")
    want, _ = python_semantics(str(path))
    assert len(want) == 15000 and all(len(w[1]) == 80 for w in want[1:])
    for threads in ("1", "16"):
        os.environ["PC_IO_THREADS"] = threads
        check(str(path), want, False)
    os.environ.pop("PC_IO_THREADS", None)


def test_writer_large_outputs(tmp_path):
    """pc_readset_write formats big files with several threads writing in place: FASTQ and FASTA,
    numbered pieces, RNA, FASTA-sourced '+' qualities -- against plain Python formatting."""
    import numpy as np
    rng = random.Random(9)
    recs = []
    for i in range(3000):
        n = rng.choice([30, 800, 5000])
        alpha = "ACGU" if i % 11 == 0 else "ACGT"
        recs.append(("w%d%s" % (i, " extra words" if i % 2 else ""), "".join(rng.choice(alpha) for _ in range(n)),
                     "".join(rng.choice("!5I~") for _ in range(n))))
    fq = tmp_path / "w.fastq"
    with open(fq, "w") as f:
        f.write("".join("@%s\n%s\n+\n%s\n" % r for r in recs))
    fa = tmp_path / "w.fasta"
    with open(fa, "w") as f:
        f.write("".join(">%s\n%s\n" % (r[0], r[1]) for r in recs))

    def numbered(name, k):
        if k == 0:
            return name
        return name + "_%d" % k if " " not in name else name.replace(" ", "_%d " % k, 1)

    for src, from_fastq in ((fq, True), (fa, False)):
        rs = ReadSet(str(src))
        n = rs.count
        pr = np.repeat(np.arange(n), 2)
        st = np.tile(np.array([0, 7]), n).astype(np.int64)
        ln = np.maximum(np.repeat(rs.lengths, 2) - st - np.tile(np.array([0, 3]), n), 0)
        num = np.tile(np.array([0, 2]), n)
        for fastq in (True, False):
            out = tmp_path / ("o_%d_%d" % (from_fastq, fastq))
            for threads in ("1", "5"):
                os.environ["PC_IO_THREADS"] = threads
                rs.write(pr, st, ln, num, np.zeros(2 * n), [str(out)], fastq)
                want = []
                for k in range(2 * n):
                    name, seq, q = recs[pr[k]]
                    s = seq[st[k]:st[k] + ln[k]]           # RNA reads are stored with T and written back with U
                    qq = (q if from_fastq else "+" * len(seq))[st[k]:st[k] + ln[k]]
                    if fastq:
                        want.append("@%s\n%s\n+\n%s\n" % (numbered(name, num[k]), s, qq))
                    else:
                        body = "".join(s[p:p + 70] + "\n" for p in range(0, len(s), 70)) or "\n"
                        want.append(">%s\n%s" % (numbered(name, num[k]), body))
                assert open(out).read() == "".join(want), (from_fastq, fastq, threads)
        rs.close()
    os.environ.pop("PC_IO_THREADS", None)


def test_segments_and_continued_writes_equal_whole_file(tmp_path):
    """pc_readset_load_segment + pc_readset_write_at (the streamed runner's ingest and writer): the blocks of a
    FASTQ file, each written where the previous one stopped, give byte for byte what one load + one write
    give; FASTQ and FASTA output; qualities that start with '@' or '+'."""
    import os
    import subprocess
    import sys
    code = r'''
import hashlib, os, random, sys
sys.path.insert(0, ".")
import numpy as np
from porechop_amd.io import ReadSet
rng = random.Random(7)
src = sys.argv[1]
with open(src, "w") as f:
    for i in range(1500):
        n = rng.choice([1, 9, 80, 700, 3000])
        f.write("@r%d extra words\n%s\n+\n%s\n" % (i, "".join(rng.choice("ACGTacgtNU") for _ in range(n)),
                                               "".join(rng.choice("@+!5I") for _ in range(n))))
size = os.path.getsize(src)
whole = ReadSet(src)
for fastq in (True, False):
    def pieces(rs):
        # decided per read from its own length only, so that blocks and the whole file describe the same pieces:
        # one piece for every read, a second, numbered one for reads of even length; the file by length mod 3
        n = rs.count
        ln1 = rs.lengths.astype(np.int64)
        twice = np.nonzero(ln1 % 2 == 0)[0]
        pr = np.sort(np.concatenate([np.arange(n, dtype=np.int64), twice]), kind="stable")
        second = np.concatenate([[False], pr[1:] == pr[:-1]])
        ln = rs.lengths[pr]
        st = np.minimum(ln // 5, 3).astype(np.int32)
        return pr, st, (ln - st).astype(np.int32), np.where(second, 2, 0).astype(np.int32), (ln % 3 == 0).astype(np.int32)
    ref = [src + ".whole%d.%d" % (fastq, k) for k in range(2)]
    whole.write(*pieces(whole), ref, fastq)
    out = [src + ".blocks%d.%d" % (fastq, k) for k in range(2)]
    pos, fpos = 0, np.zeros(2, dtype=np.int64)
    nblocks = 0
    while pos < size:
        rs, pos = ReadSet.segment(src, pos, 20000)
        assert rs is not None
        rs.write_at(*pieces(rs), out, fastq, fpos)
        rs.close(); nblocks += 1
    assert nblocks > 20
    for a, b in zip(ref, out):
        assert open(a, "rb").read() == open(b, "rb").read(), (fastq, a)
print("OK")
'''
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code, str(tmp_path / "reads.fastq")], capture_output=True, text=True, cwd=repo, timeout=600)
    assert res.returncode == 0 and "OK" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]
