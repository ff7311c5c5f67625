#!/usr/bin/env python3
"""GPU box: the headline step (phases A + B + C, every record computed), ms per step and the score kernel's share.
   python tools/time_headline.py [reads]       (PC_NO_FUSE=1: every middle adapter scanned alone)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from porechop_amd.panel import load_panel
from porechop_amd.pipeline import Pipeline, ScanParams
from porechop_amd.synth import make_reads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
p = ScanParams()
pl = Pipeline(load_panel(), p)
reads = make_reads(n, 8000, seed=3, start_frac=0.9, end_frac=0.5, chimera_frac=0.01)
check = torch.arange(p.check_reads, device="cuda")
def step():
    bs, be = pl.phase_a(reads, check)
    m = pl.matching_sets(bs, be)
    st, et = pl.phase_b(reads, m)
    return pl.phase_c(reads, st, et, m)
for _ in range(2):
    h = step()
pl.aligner.sync(); torch.cuda.synchronize()
pl.aligner.set_timing(True); pl.aligner.get_timing()
t0 = time.perf_counter()
N = int(os.environ.get('PC_LOOP', '5'))
for _ in range(N):
    h = step()
pl.aligner.sync(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N * 1e3
tm = pl.aligner.get_timing()
print("NO_FUSE=%s  %.2f ms/step  hits %d  kernels %s" % (os.environ.get("PC_NO_FUSE", "0"), dt, int(h.read.numel()), {k: (round(v[0] / N, 2), v[1] // N) for k, v in tm.items() if v[1]}))
