#!/usr/bin/env python3
"""Host-side profile of one benchmark step (GPU box): where the Python/torch time between kernel
launches goes.   python tools/host_profile.py"""
import cProfile, pstats, sys, io
sys.path.insert(0, ".")
import torch
from porechop_amd.panel import load_panel
from porechop_amd.pipeline import Pipeline, ScanParams
from porechop_amd.synth import make_reads
p = ScanParams()
pl = Pipeline(load_panel(), p)
reads = make_reads(1_000_000, 8000, seed=3, start_frac=0.9, end_frac=0.5, chimera_frac=0.01)
check = torch.arange(p.check_reads, device="cuda")

def step():
    bs, be = pl.phase_a(reads, check)
    matching = pl.matching_sets(bs, be)
    st, et = pl.phase_b(reads, matching)
    hits = pl.phase_c(reads, st, et, matching)
    pl.aligner.sync()
    return hits

for _ in range(2):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:4500])
