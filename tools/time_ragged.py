#!/usr/bin/env python3
"""Where a step on log-normal read lengths spends its time, beside the same step on uniform lengths (GPU box).
    python tools/time_ragged.py [n_reads]"""
import sys, time
sys.path.insert(0, ".")
import torch
from bench import load_panel_sets, one_step
from porechop_amd.pipeline import Pipeline, ScanParams
from porechop_amd.synth import make_reads, make_ragged_reads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
p = ScanParams()
for name in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("uniform", "ragged")):
    pl = Pipeline(load_panel_sets(), p)
    if name == "uniform":
        reads = make_reads(n, 8000, seed=3, start_frac=0.9, end_frac=0.5, chimera_frac=0.01)
    elif name == "uniform-shuffled":      # the same reads handed over in random order: tiles of scattered streams
        reads = make_reads(n, 8000, seed=3, start_frac=0.9, end_frac=0.5, chimera_frac=0.01)
        perm = torch.randperm(n, device="cuda")
        reads.off = reads.off[perm].contiguous(); reads.length = reads.length[perm].contiguous()
    elif name == "ragged-laid-out-sorted":   # log-normal lengths, but lying in the arena longest first
        reads = make_ragged_reads(n, mean_len=8000, sigma=0.6, min_len=20, seed=5, start_frac=0.0, end_frac=0.0, chimera_frac=0.0)
        ln = torch.sort(reads.length, descending=True).values
        reads.length = ln.contiguous(); reads.off = (torch.cumsum(ln.to(torch.int64), 0) - ln.to(torch.int64)).contiguous()
    else:
        reads = make_ragged_reads(n, mean_len=8000, sigma=0.6, min_len=20, seed=5, start_frac=0.9, end_frac=0.5, chimera_frac=0.01)
    def T():
        pl.aligner.sync(); torch.cuda.synchronize(); return time.perf_counter()
    for _ in range(2):
        one_step(pl, reads, p.check_reads, 1)
    pl.aligner.set_timing(True); pl.aligner.get_timing()
    for rep in range(2):
        t0 = T()
        bs, be = pl.phase_a(reads, torch.arange(p.check_reads, device="cuda"))
        matching = pl.matching_sets(bs, be)
        t1 = T(); ka = pl.aligner.get_timing()
        st, et = pl.phase_b(reads, matching)
        t2 = T(); kb = pl.aligner.get_timing()
        hits = pl.phase_c(reads, st, et, matching)
        t3 = T(); kc = pl.aligner.get_timing()
        print("%s rep %d: A %.1f ms (kernels %.1f) | B %.1f (%.1f) | C %.1f (score_spec %.1f in %d launches, score %.1f, plan %.1f, trace %.1f) | total %.1f ms, %d hits, %d rounds"
              % (name, rep, (t1 - t0) * 1e3, sum(v[0] for v in ka.values()), (t2 - t1) * 1e3, sum(v[0] for v in kb.values()),
                 (t3 - t2) * 1e3, kc["score_spec"][0], kc["score_spec"][1], kc["score"][0], kc["plan"][0], kc["trace"][0], (t3 - t0) * 1e3,
                 int(hits.read.numel()), hits.rounds))
    pl.close(); del reads, pl; torch.cuda.empty_cache()
