#!/usr/bin/env python3
"""Build container only (needs /root/reference): random ODD input files through the reference's own loaders
(porechop/misc.py load_fasta_or_fastq + the normalisation of NanoporeRead.__init__, nanopore_read.py:23-35) and through
porechop_amd.io.ReadSet -- same reads, names, qualities, or both refuse.  Odd = blank lines, blanks around lines and in front
of headers, CRLF, no newline at the end, empty names, empty sequences, qualities shorter than the sequence, multi-line FASTA,
text before the first header, records cut short; plain, gzip-ed in one member, in several members with zero padding.
    python tools/fuzz_ingest_vs_reference.py [cases] [seed]"""
import gzip
import importlib.util
import io
import os
import random
import shutil
import sys
import tempfile
from contextlib import redirect_stderr, redirect_stdout

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from porechop_amd.io import ReadSet  # noqa: E402

REFERENCE = "/root/reference"
spec = importlib.util.spec_from_file_location("pc_ref_misc", os.path.join(REFERENCE, "porechop", "misc.py"))
misc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(misc)

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
tmp = tempfile.mkdtemp(prefix="pc_ingest_fuzz_")


def seq(n):
    return "".join(rng.choice("ACGTacgtNnUu-RY") for _ in range(n))


def odd_line(text):
    r = rng.random()
    if r < 0.08:
        return " " + text
    if r < 0.16:
        return text + " \t"
    return text


def fastq_text():
    out = []
    for i in range(rng.choice([1, 2, 5, 40, 400])):
        n = rng.choice([0, 1, 5, 60, 300])
        name = rng.choice(["r%d" % i, "r%d some words" % i, "", " ", "@r%d" % i, "r%d\tx" % i])
        q = "".join(rng.choice("!+@5I~") for _ in range(max(0, n - rng.choice([0, 0, 0, 2]))))
        lines = [odd_line("@" + name), odd_line(seq(n)), odd_line("+" + rng.choice(["", name])), odd_line(q)]
        if rng.random() < 0.03:
            lines.insert(rng.randrange(5), "")                      # a blank line somewhere in the record
        if rng.random() < 0.02:
            lines = lines[:rng.randrange(1, 4)]                      # a record cut short
        out += lines
    return out


def fasta_text():
    out = []
    if rng.random() < 0.05:
        out.append(rng.choice(["", "   "]))                          # blank lines before the first header are skipped by both? (first CHAR decides)
    for i in range(rng.choice([1, 2, 5, 40, 400])):
        n = rng.choice([0, 1, 5, 60, 300, 2000])
        name = rng.choice(["r%d" % i, "r%d some words" % i, "", " ", ">r%d" % i, " lead%d" % i])
        s = seq(n)
        w = rng.choice([60, 70, 10 ** 9])
        out.append(odd_line(">" + name))
        for k in range(0, len(s), w):
            out.append(odd_line(s[k:k + w]))
            if rng.random() < 0.02:
                out.append("")
    return out


def reference_reads(path):
    try:
        with redirect_stdout(io.StringIO()), redirect_stderr(io.StringIO()):
            recs, kind = misc.load_fasta_or_fastq(path)
    except SystemExit as e:
        return ("exit", str(e).strip()), None
    except Exception as e:                                              # the reference dies with a traceback
        return ("raise", type(e).__name__), None
    want = []
    for r in recs:
        name, s, q = (r[4], r[1], r[3]) if kind == "FASTQ" else (r[2], r[1], None)
        s = s.upper()
        rna = s.count("U") > s.count("T")
        if rna:
            s = s.replace("U", "T")
        if q is not None and len(q) < len(s):
            q += "+" * (len(s) - len(q))
        want.append((name, s, q, rna))
    return want, kind == "FASTQ"


bad = both_fail = 0
for k in range(cases):
    fastq = rng.random() < 0.6
    lines = fastq_text() if fastq else fasta_text()
    nl = rng.choice(["\n", "\n", "\r\n"])
    text = nl.join(lines) + (nl if rng.random() < 0.85 else "")
    if len(text) < 1 << 20 and rng.random() < 0.1:
        text = text * (1 + (1 << 20) // max(1, len(text)))              # above the parallel parsers' threshold
    data = text.encode()
    layout = rng.choice(["plain", "plain", "one", "members"])
    path = os.path.join(tmp, "case%d.%s%s" % (k, "fastq" if fastq else "fasta", "" if layout == "plain" else ".gz"))
    if layout == "one":
        data = gzip.compress(data, 1)
    elif layout == "members":
        cut = rng.randrange(len(data) + 1)
        data = gzip.compress(data[:cut], 1) + b"\0" * rng.choice([0, 7, 300]) + gzip.compress(data[cut:], 1) + b"\0" * rng.choice([0, 4])
    with open(path, "wb") as f:
        f.write(data)
    want, want_fastq = reference_reads(path)
    try:
        rs = ReadSet(path)
        got = [(rs.name(i), rs.seq(i), rs.quals(i) if rs.is_fastq else None, bool(rs.is_rna(i))) for i in range(rs.count)]
        got_fastq = rs.is_fastq
        rs.close()
    except ValueError as e:
        got, got_fastq = ("error", str(e)), None
    ref_failed = isinstance(want, tuple)
    our_failed = isinstance(got, tuple)
    ok = (ref_failed and our_failed) or (not ref_failed and not our_failed and got == want and got_fastq == want_fastq)
    both_fail += ref_failed and our_failed
    if not ok:
        bad += 1
        keep = os.path.join(tempfile.gettempdir(), "pc_ingest_bad_%d_%s" % (k, os.path.basename(path)))
        shutil.copy(path, keep)
        print("BAD case %d %s: reference %s | ours %s   (kept %s)" % (
            k, os.path.basename(path), want if ref_failed else "%d reads" % len(want), got if our_failed else "%d reads" % len(got), keep), flush=True)
shutil.rmtree(tmp, ignore_errors=True)
print("cases=%d mismatches=%d (refused by both: %d)" % (cases, bad, both_fail))
