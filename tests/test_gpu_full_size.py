"""BASELINE-size runs (the bench's own workload, 1 M x 8 kb reads with 1 % chimeras), checked through
properties that do not need the oracle to run the whole set:
  * a random sample of reads goes through the reference's sequential logic driven by the oracle and
    must agree on trims and middle hits;
  * determinism: a second pass over the same resident batch reproduces every output bit;
  * chunked vs whole-window score pass: the same reads scanned as a small batch (which triggers the
    column-chunked pass) and as part of the big batch (which does not) give identical records;
  * planted structure: ~90 % start trims, ~50 % end trims, middle-hit reads ~ the chimera fraction.
Also the end-trim-only shape of BASELINE configs[1] at 100 k reads."""
import random

import numpy as np
import pytest
import torch

from tests import ref_pipeline
from tests.golden_io import load_panel

pytestmark = pytest.mark.gpu


def panel_sets():
    from porechop_amd.pipeline import AdapterSet
    return [AdapterSet(a["name"], tuple(a["start"]) if a["start"] else None, tuple(a["end"]) if a["end"] else None)
            for a in load_panel()]


def run_all(pl, reads, n_check):
    check = torch.arange(n_check, device="cuda")
    bs, be = pl.phase_a(reads, check)
    matching = pl.matching_sets(bs, be)
    st, et = pl.phase_b(reads, matching)
    hits = pl.phase_c(reads, st, et, matching)
    pl.aligner.sync()
    return matching, st, et, hits


def host_seq(reads, r):
    o, n = int(reads.off[r]), int(reads.length[r])
    return reads.arena[o:o + n].cpu().numpy().tobytes().decode()


@pytest.mark.parametrize("n_reads,chimera", [(1_000_000, 0.01)])
def test_config4_full_size_properties(oracle, n_reads, chimera):
    from porechop_amd.pipeline import Pipeline, ScanParams, DeviceReads
    from porechop_amd.synth import make_reads
    p = ScanParams()
    pl = Pipeline(panel_sets(), p)
    reads = make_reads(n_reads, 8000, seed=3, start_frac=0.9, end_frac=0.5, chimera_frac=chimera)
    matching, st, et, hits = run_all(pl, reads, p.check_reads)
    names = [pl.sets[i].name for i in matching]
    assert "SQK-NSK007" in names

    # planted structure
    assert 0.85 < float((st > 0).float().mean()) < 0.95
    assert 0.42 < float((et > 0).float().mean()) < 0.58
    hit_reads = torch.unique(hits.read).numel()
    assert 0.6 * chimera * n_reads < hit_reads < 1.4 * chimera * n_reads

    # determinism (same resident inputs -> same bits)
    matching2, st2, et2, hits2 = run_all(pl, reads, p.check_reads)
    assert matching2 == matching and torch.equal(st, st2) and torch.equal(et, et2)
    for a, b in ((hits.read, hits2.read), (hits.adapter, hits2.adapter), (hits.start, hits2.start), (hits.end, hits2.end)):
        assert torch.equal(a, b)

    # a sample through the reference's sequential logic (oracle-driven), incl. reads with middle hits
    rng = random.Random(11)
    sample = rng.sample(range(n_reads), 24) + [int(x) for x in torch.unique(hits.read)[:24].cpu()]
    got = {}
    for r, a, s, e in zip(hits.read.cpu().tolist(), hits.adapter.cpu().tolist(), hits.start.cpu().tolist(), hits.end.cpu().tolist()):
        got.setdefault(r, []).append((a, s, e))
    stl, etl = st.cpu().tolist(), et.cpu().tolist()
    for r in sample:
        seq = host_seq(reads, r)
        assert (stl[r], etl[r]) == ref_pipeline.phase_b(oracle.adapter_alignment, seq, pl.sets, matching, p), r
        want = [(a, s, e) for a, s, e, _ in ref_pipeline.phase_c(oracle.adapter_alignment, seq, stl[r], etl[r], pl.middle_adapters, p)]
        assert got.get(r, []) == want, r

    # the same reads as a small batch: the score pass is cut into column chunks there
    sub = torch.tensor(sorted(set(sample)), device="cuda")
    small = DeviceReads(reads.arena, reads.off[sub], reads.length[sub])
    st_s, et_s = pl.phase_b(small, matching)
    hits_s = pl.phase_c(small, st_s, et_s, matching)
    pl.aligner.sync()
    assert torch.equal(st_s, st[sub]) and torch.equal(et_s, et[sub])
    got_s = {}
    for r, a, s, e in zip(hits_s.read.cpu().tolist(), hits_s.adapter.cpu().tolist(), hits_s.start.cpu().tolist(), hits_s.end.cpu().tolist()):
        got_s.setdefault(int(sub[r]), []).append((a, s, e))
    assert got_s == {r: v for r, v in got.items() if r in set(sub.cpu().tolist())}
    pl.close()


def test_config2_end_trim_only_100k(oracle):
    """BASELINE configs[1]: 100 k reads, end trim only (--no_split): phases A + B."""
    from porechop_amd.pipeline import Pipeline, ScanParams
    from porechop_amd.synth import make_reads
    p = ScanParams()
    pl = Pipeline(panel_sets(), p)
    reads = make_reads(100_000, 8000, seed=1, start_frac=0.9, end_frac=0.5, chimera_frac=0.0)
    bs, be = pl.phase_a(reads, torch.arange(p.check_reads, device="cuda"))
    matching = pl.matching_sets(bs, be)
    st, et = pl.phase_b(reads, matching)
    pl.aligner.sync()
    assert "SQK-NSK007" in [pl.sets[i].name for i in matching]
    stl, etl = st.cpu().tolist(), et.cpu().tolist()
    rng = random.Random(5)
    for r in rng.sample(range(100_000), 64):
        assert (stl[r], etl[r]) == ref_pipeline.phase_b(oracle.adapter_alignment, host_seq(reads, r), pl.sets, matching, p), r
    pl.close()


def test_specialised_score_pass_equals_generic_unchunked():
    """160 k x 8 kb reads -- 2 500 tiles, more than the 2 048 resident waves, so the specialised
    score pass runs as a balanced head of whole windows plus a column-chunked tail -- scanned
    against an adapter pair, once with the run-time specialised kernel (drifting coordinates,
    renormalised every ~1900 / ~200 columns, two skewed columns per wave) and once with the generic
    ahead-of-time kernel (PC_DISABLE_JIT=1): every output record must be bit-identical.  Two
    schemes; each variant in a fresh process."""
    import hashlib
    import os
    import subprocess
    import sys
    code = r'''
import hashlib, sys
sys.path.insert(0, ".")
import torch
import porechop_amd
from porechop_amd.synth import make_reads
ads = ["AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT"]
reads = make_reads(160_000, 8000, seed=11, start_frac=0.3, end_frac=0.3, chimera_frac=0.05)
n = 160_000
for scores in [(3, -6, -5, -2), (20, -30, -25, -12)]:
    al = porechop_amd.Aligner(ads, scores=scores)
    out = torch.zeros((2 * n, 8), dtype=torch.int32, device="cuda")
    al.scan_device(reads.arena, reads.off, reads.length, [0], [0, n], 8000, out, porechop_amd.MODE_TWO_PASS, job_adapter_b=[1])
    al.sync()
    print("DIGEST", scores, hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest(), int((out[:, 4] > 40).sum()))
'''
    outs = []
    for env_extra in ({"PC_JIT_MIN_CELLS": "1", "PC_JIT_VERBOSE": "1"}, {"PC_DISABLE_JIT": "1"}):
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900,
                             env=dict(os.environ, **env_extra),
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        lines = [l for l in res.stdout.splitlines() if l.startswith("DIGEST")]
        assert len(lines) == 2, res.stdout[-2000:] + res.stderr[-3000:]
        if "PC_JIT_VERBOSE" in env_extra:
            assert res.stderr.count("specialised kernel R=28") == 2, res.stderr[-2000:]
        outs.append(lines)
    assert outs[0] == outs[1]
    assert int(outs[0][0].split()[-1]) > 1000        # the planted adapters were found


def test_ragged_score_pass_chunked_and_drawn_from_the_counter_equals_plain():
    """120 k log-normal reads (mean 4 kb, longest > 40 kb), handed over longest first with the length hint set:
    the score pass cuts every window into typical-length chunks and a persistent grid draws the units from the
    work counter (pc_set_length_hint) -- with the run-time specialised kernel and with the generic one.  Both
    must produce, record for record, what the plain one-window-per-workgroup pass (no hint) produces, and a
    sample of the records must equal the oracle's.  Each variant in a fresh process."""
    import os
    import subprocess
    import sys
    code = r'''
import hashlib, os, sys
sys.path.insert(0, ".")
import torch
import porechop_amd
from porechop_amd.synth import make_ragged_reads
from oracle.oracle import Oracle
ads = ["AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT"]
reads = make_ragged_reads(120_000, mean_len=4000, sigma=0.7, min_len=20, seed=21, start_frac=0.3, end_frac=0.3, chimera_frac=0.05)
n = reads.n
order = torch.argsort(reads.length, descending=True, stable=True)
off, ln = reads.off[order].contiguous(), reads.length[order].contiguous()
assert int(ln[0]) > 40000
al = porechop_amd.Aligner(ads)
if os.environ.get("PC_TEST_HINT") == "1":
    al.set_length_hint(int(ln.sum().item()) // n)
out = torch.zeros((2 * n, 8), dtype=torch.int32, device="cuda")
al.scan_device(reads.arena, off, ln, [0], [0, n], int(ln[0]), out, porechop_amd.MODE_TWO_PASS, job_adapter_b=[1])
al.sync()
host = reads.arena.cpu().numpy().tobytes().decode()
ora = Oracle()
for k in list(range(0, 40)) + list(range(n // 2, n // 2 + 40)) + list(range(n - 40, n)):
    seq = host[int(off[k]):int(off[k]) + int(ln[k])]
    for a in range(2):
        assert porechop_amd.format_result(out[a * n + k].cpu().numpy()) == ora.adapter_alignment(seq, ads[a]), (k, a)
print("DIGEST", hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest(), int((out[:, 4] > 40).sum()))
'''
    outs = []
    for env_extra in ({"PC_JIT_MIN_CELLS": "1", "PC_JIT_VERBOSE": "1", "PC_TEST_HINT": "1"}, {"PC_DISABLE_JIT": "1", "PC_TEST_HINT": "1"},
                      {"PC_JIT_MIN_CELLS": "1", "PC_TEST_HINT": "0"}):
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900,
                             env=dict(os.environ, **env_extra),
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        lines = [l for l in res.stdout.splitlines() if l.startswith("DIGEST")]
        assert len(lines) == 1, res.stdout[-2000:] + res.stderr[-3000:]
        if "PC_JIT_VERBOSE" in env_extra:
            assert "specialised kernel R=28" in res.stderr, res.stderr[-2000:]
        outs.append(lines[0])
    assert outs[0] == outs[1] == outs[2]
    assert int(outs[0].split()[-1]) > 1000


def test_configs4_shape_barcodes_and_middle_scan():
    """The per-GPU shape of BASELINE configs[4] (bench.py step_configs4): 50 000 reads carrying one of the 96 PCR
    barcodes at both ends and 1 % chimeric junctions, full 119-set panel, demultiplexing run with the middle scan over
    every matching set's sequences (~196 adapters).  320 reads go through the reference's sequential per-read logic
    (phase B with barcode scores, determine_barcode, phase C) on the host cores through the compiled reference / the
    oracle; the exact-prefilter variant must reproduce trims, calls and hits of the full scan bit for bit."""
    import multiprocessing as mp
    from dataclasses import asdict
    import bench
    from porechop_amd.pipeline import Pipeline, ScanParams
    from porechop_amd.runner import Options
    from porechop_amd.synth import make_reads
    from tests.cpu_worker import run_chunk_demux_middle
    p, opts = ScanParams(), Options()
    pl = Pipeline(panel_sets(), p)
    pl.n_panel = len(pl.sets)
    fw = [a for a in load_panel() if a["name"].startswith("Barcode ") and "(forward)" in a["name"]]
    n = 50_000
    reads = make_reads(n, 8000, seed=4, start_frac=0.9, end_frac=0.5, chimera_frac=0.01,
                       barcodes_start=[a["start"][1] for a in fw], barcodes_end=[a["end"][1] for a in fw])
    matching, orientation, names, st, et, calls, hits = bench.step_configs4(pl, reads, p.check_reads, opts)
    pl.aligner.sync()
    m2, o2, n2, st2, et2, calls2, hits2 = bench.step_configs4(pl, reads, p.check_reads, opts, prefilter=True)
    pl.aligner.sync()
    assert orientation == "forward" and len(matching) >= 97 and len(pl.middle_adapter_list(matching)) >= 194
    assert (m2, o2, n2) == (matching, orientation, names)
    assert torch.equal(st, st2) and torch.equal(et, et2) and np.array_equal(calls, calls2)
    for f in ("read", "adapter", "start", "end", "identity"):
        assert torch.equal(getattr(hits, f), getattr(hits2, f)), f
    assert (hits.rounds, hits.alignments) == (hits2.rounds, hits2.alignments)
    hit_reads = torch.unique(hits.read).numel()
    assert 0.5 * 0.01 * n < hit_reads < 2.0 * 0.01 * n
    truth = reads.truth_barcode.cpu().numpy()
    want = np.array([names.index("BC%02d" % (b + 1)) if "BC%02d" % (b + 1) in names else -2 for b in range(len(fw))])[truth]
    assert float((calls == want).mean()) > 0.99

    # host check of a sample that contains reads WITH middle hits: the first 256 reads + 64 reads that had a hit
    with_hits = torch.unique(hits.read)[:64].cpu().tolist()
    sample = sorted(set(range(256)) | set(with_hits))
    seqs = [host_seq(reads, r) for r in sample]
    sets = [(s.name, s.start, s.end) for s in pl.sets]
    workers = bench.host_cores()
    per = max(1, (len(seqs) + workers - 1) // workers)
    chunks = [seqs[i:i + per] for i in range(0, len(seqs), per)]
    mk = lambda c: (c, sets, matching, asdict(p), True, orientation, opts.barcode_threshold, opts.barcode_diff, opts.require_two_barcodes)
    with mp.get_context("spawn").Pool(min(workers, len(chunks))) as pool:
        res = [x for r in pool.map(run_chunk_demux_middle, [mk(c) for c in chunks]) for x in r[2]]
    got = {}
    for r, a, s, e in zip(hits.read.cpu().tolist(), hits.adapter.cpu().tolist(), hits.start.cpu().tolist(), hits.end.cpu().tolist()):
        got.setdefault(r, []).append((a, s, e))
    stl, etl = st.cpu().tolist(), et.cpu().tolist()
    n_hits = 0
    for r, want_r in zip(sample, res):
        call = names[calls[r]] if calls[r] >= 0 else "none"
        assert (stl[r], etl[r], call, got.get(r, [])) == (want_r[0], want_r[1], want_r[2], list(want_r[3])), r
        n_hits += len(want_r[3])
    assert n_hits >= 64
    pl.close()
