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
