"""ULTRA-LONG reads (65 535 bases and more, up to 1 000 000) through every route of the whole-read scan, against the
compiled reference (oracle/_ref; /root/reference/porechop/src/adapter_align.cpp:11-31 takes any char*) -- or the
oracle, which tests/test_oracle_ultralong.py pins to it on these very cases.

Routes ("what's missing" 1):
  (a) the run-time specialised score kernel, a whole read in ONE unit (PC_FORCE_CHUNKS=1): columns beyond 65 000 take the
      plain-int branch of its block-resolved maximum search (pc_jit_source.h), never compared with the reference before;
  (b) the same kernel chunked -- by under-fill (the default for a handful of reads) and by the length hint;
  (c) the generic score kernel (PC_DISABLE_JIT=1), one unit and chunked;
  (d) Pipeline.phase_c: mask-and-realign with two copies per read, with and without the exact prefilter;
  (e) runner.run() on a FASTQ holding such reads vs the reference CLI's output, byte for byte.
Planted copies: before / at / after column 65 535, across a chunk boundary, in the read's last columns, cut off by the read's end.
"""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# one process per route (the environment knobs are read once per process); the routes print what the GPU answered, the
# parent compares with the reference's answers, computed once per session
ROUTE_CODE = r"""
import os, sys
sys.path.insert(0, ".")
import porechop_amd
from tests.test_gpu_ultralong import ADS, route_cases
hint = int(os.environ.get("PC_TEST_HINT", "0"))
cs = route_cases(os.environ.get("PC_TEST_FULL", "1") == "1")
al = porechop_amd.Aligner(ADS)
if hint:
    al.set_length_hint(hint)
# in batches of one read length (a batch is one launch per adapter; its max_len decides the chunking)
by_len = {}
for k, c in enumerate(cs):
    by_len.setdefault(len(c[1]), []).append(k)
for n, ks in sorted(by_len.items()):
    recs = al.align_pairs([(cs[k][1], ADS.index(cs[k][2])) for k in ks], porechop_amd.MODE_TWO_PASS)
    for k, rec in zip(ks, recs):
        print("RESULT", k, porechop_amd.format_result(rec))
al.close()
# the reference-shaped per-call symbol, one long read
k = len(cs) // 2
print("PERCALL", k, porechop_amd.adapter_alignment(cs[k][1], cs[k][2], [3, -6, -5, -2]))
"""

from tests.longgen import MILLION, Y_BOTTOM, Y_TOP, cases, long_adapter, make_read, mutated

ADS = [Y_BOTTOM, Y_TOP, long_adapter()]


def route_cases(full):
    cs = cases(lengths=(65535, 65536, 70000, 131073) if full else (65536, 70000), adapters=ADS, chunk_cols=(16384, 32768, 65536))
    if full:
        rng = np.random.default_rng(5)
        for i, (ad, col) in enumerate(((Y_TOP, 999_990), (Y_BOTTOM, 65_536), (Y_TOP, MILLION), (ADS[2], 800_000))):
            cs.append(("million m=%d col=%d" % (len(ad), col),
                       make_read(MILLION, 40 + i, [(col, mutated(rng, ad))], n_runs=[(500_000, 1000)], dash_runs=[(700_000, 5000)]), ad))
    return cs


_WANT = {}


def wanted(full):
    """The compiled reference's answers (the oracle's where oracle/_ref is absent), once per session."""
    if full not in _WANT:
        from oracle.oracle import Oracle, Reference
        ref = Reference() if Reference.available() else Oracle()
        _WANT[full] = [ref.adapter_alignment(rd, ad) for _, rd, ad in route_cases(full)]
    return _WANT[full]


def run_route(env_extra, timeout=1500):
    full = env_extra.get("PC_TEST_FULL", "1") == "1"
    res = subprocess.run([sys.executable, "-c", ROUTE_CODE], capture_output=True, text=True, timeout=timeout,
                         env=dict(os.environ, **env_extra), cwd=REPO)
    want, cs = wanted(full), route_cases(full)
    got = {}
    for l in res.stdout.splitlines():
        if l.startswith("RESULT ") or l.startswith("PERCALL "):
            tag, k, val = l.split(" ", 2)
            got[(tag, int(k))] = val
    assert len(got) == len(cs) + 1, res.stdout[-2000:] + res.stderr[-3000:]
    bad = [(cs[k][0], tag, v, want[k]) for (tag, k), v in sorted(got.items()) if v != want[k]]
    assert not bad, (env_extra, len(bad), bad[:4])
    assert len(cs) > 50
    return res


def test_specialised_kernel_one_unit_per_read_beyond_65535_columns():
    """(a): PC_FORCE_CHUNKS=1 -> nmax > 65 000 in one unit: the plain-int branch of the block-resolved maximum search."""
    res = run_route({"PC_JIT_MIN_CELLS": "1", "PC_JIT_VERBOSE": "1", "PC_FORCE_CHUNKS": "1"})
    assert "specialised kernel R=" in res.stderr, res.stderr[-2000:]


def test_specialised_kernel_chunked_by_underfill_and_by_hint():
    """(b): the default launch plan of a few long reads (column chunks by under-fill), and chunks of about the hinted
    typical length (ragged batches); 64 chunks of 2 048 columns put boundaries at 65 536 and 32 768."""
    res = run_route({"PC_JIT_MIN_CELLS": "1", "PC_JIT_VERBOSE": "1"})
    assert "specialised kernel R=" in res.stderr, res.stderr[-2000:]
    run_route({"PC_JIT_MIN_CELLS": "1", "PC_TEST_HINT": "8000", "PC_TEST_FULL": "0"})
    run_route({"PC_JIT_MIN_CELLS": "1", "PC_FORCE_CHUNKS": "2", "PC_TEST_FULL": "0"})      # one boundary mid-read, chunks > 32 768 columns


def test_generic_score_kernel_beyond_65535_columns():
    """(c): PC_DISABLE_JIT=1, whole reads in one unit and chunked."""
    run_route({"PC_DISABLE_JIT": "1", "PC_FORCE_CHUNKS": "1", "PC_TEST_FULL": "0"})
    run_route({"PC_DISABLE_JIT": "1"})


def _long_reads():
    """Reads for the pipeline / runner routes: Y adapters at the ends, middle copies beyond column 65 535 (two in some
    reads: mask-and-realign round 2), one read of 300 kb, short reads between them."""
    from tests.longgen import Y_BOTTOM, Y_TOP, make_read, mutated
    rng = np.random.default_rng(77)
    junction = Y_BOTTOM + Y_TOP
    spec = [(70_000, [(66_000, junction)]), (9_000, []), (131_073, [(131_000, Y_TOP), (40_000, mutated(rng, Y_TOP, 1, 0, 0))]),
            (65_536, [(65_536 - 200, Y_BOTTOM)]), (300_000, [(65_535 + 14, Y_TOP), (250_000, junction)]), (5_000, [(2_500, junction)]),
            (65_535, []), (80_000, [(70_000, Y_TOP), (75_000, Y_TOP), (79_000, Y_BOTTOM)]), (12_000, []), (100_000, [(65_600, Y_BOTTOM)])]
    reads = []
    for i, (n, plants) in enumerate(spec):
        body = make_read(n, 900 + i, plants, n_runs=[(n // 3, 40)] if i % 3 == 0 else ())
        reads.append(Y_TOP + body + Y_BOTTOM if i % 2 == 0 else body)
    return reads


def check_phase_c(pl, device):
    """Pipeline.phase_c over _long_reads(), with and without the exact prefilter, against the reference's sequential
    per-read loop (nanopore_read.py:210-243) run over the compiled reference (shared with tests/test_ultralong_host_logic.py,
    which runs it over the oracle-backed stand-in on the CPU)."""
    import torch
    from oracle.oracle import Oracle, Reference
    from porechop_amd.pipeline import DeviceReads
    ref = Reference() if Reference.available() else Oracle()
    reads = _long_reads()
    p = pl.p
    blob = "".join(reads).encode() + b"N" * 64
    arena = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    lens = torch.tensor([len(r) for r in reads], dtype=torch.int32)
    off = (torch.cumsum(lens.to(torch.int64), 0) - lens.to(torch.int64)).to(device)
    dr = DeviceReads(arena, off, lens.to(device))
    st, et = pl.phase_b(dr, [0])[:2]
    want_hits = []
    for r, seq in enumerate(reads):
        masked = seq[int(st[r]):len(seq) - int(et[r])] if (int(st[r]) or int(et[r])) else seq
        for ai, ad in enumerate((Y_TOP, Y_BOTTOM)):
            while True:
                f = ref.adapter_alignment(masked, ad).split(",")
                if int(f[0]) == -1 or float(f[6]) < p.middle_threshold:
                    break
                rs, re_ = int(f[0]), int(f[1]) + 1
                want_hits.append((r, ai, rs, re_))
                masked = masked[:rs] + "-" * (re_ - rs) + masked[re_:]
    assert any(s > 65535 for _, _, s, _ in want_hits) and len(want_hits) >= 10
    for prefilter in (False, True):
        h = pl.phase_c(dr, st, et, [0], prefilter=prefilter)
        pl.aligner.sync()
        got = sorted(zip(h.read.cpu().tolist(), h.adapter.cpu().tolist(), h.start.cpu().tolist(), h.end.cpu().tolist()))
        assert got == sorted(want_hits), (prefilter, got, sorted(want_hits))
        assert h.rounds >= 2


def test_phase_c_two_hits_per_read_with_and_without_the_prefilter():
    """(d): the batch pipeline's middle scan over ultra-long reads: hits beyond column 65 535, a second hit found in the
    masked read (round 2), through the full scan and behind the exact prefilter."""
    from porechop_amd.pipeline import AdapterSet, Pipeline, ScanParams
    pl = Pipeline([AdapterSet("SQK-NSK007", ("SQK-NSK007_Y_Top", Y_TOP), ("SQK-NSK007_Y_Bottom", Y_BOTTOM))], ScanParams())
    check_phase_c(pl, "cuda")
    pl.close()


def write_long_fastq(path):
    reads = _long_reads()
    with open(path, "w") as f:
        for i, r in enumerate(reads):
            qual = "".join(chr(33 + (k * 7 + i) % 40) for k in range(64)) * (len(r) // 64) + "#" * (len(r) % 64)
            f.write("@read%d len=%d\n%s\n+\n%s\n" % (i, len(r), r, qual))
    return reads


def check_runner(tmp_path, **run_kw):
    """runner.run() on the FASTQ of _long_reads() vs the unchanged reference CLI (staged under oracle/_ref)."""
    from tests import ref_cli
    if not ref_cli.staged():
        pytest.skip("no staged reference CLI under oracle/_ref")
    from porechop_amd import runner
    inp = tmp_path / "long.fastq"
    reads = write_long_fastq(inp)
    want = tmp_path / "ref.fastq"
    res = subprocess.run([sys.executable, os.path.join(REPO, "tests", "ref_cli.py"), "--", "-i", str(inp), "-o", str(want),
                          "--threads", "8", "-v", "0"], capture_output=True, text=True, timeout=1200, cwd=REPO)
    assert res.returncode == 0, res.stderr[-2000:]
    got = tmp_path / "gpu.fastq"
    runner.run(str(inp), output=str(got), options=runner.Options(), **run_kw)
    md5 = lambda p: hashlib.md5(open(p, "rb").read()).hexdigest()
    assert os.path.getsize(want) > 500_000
    assert md5(got) == md5(want)
    # more pieces than reads: the middle copies beyond column 65 535 were found and the reads split there
    assert sum(1 for l in open(got) if l.startswith("@read")) > len(reads)


def test_runner_on_a_fastq_of_ultralong_reads_equals_the_reference_cli(tmp_path):
    """(e): file -> file.  The unchanged reference CLI and porechop_amd.runner on the same FASTQ: trimmed / split output
    byte-identical."""
    check_runner(tmp_path, device="cuda")
