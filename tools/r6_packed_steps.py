#!/usr/bin/env python3
"""GPU box: per-step wall time of the prefilter step over reads resident at 2 bits per base, from the first step on
(is there a slow step after DeviceReads.packed_only is built?)   python tools/r6_packed_steps.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from porechop_amd.io import pack_reads
from porechop_amd.pipeline import DeviceReads, Pipeline, ScanParams
from porechop_amd.synth import make_reads
n = 1_000_000
p = ScanParams()
pl = Pipeline(bench.load_panel_sets(), p)
reads = make_reads(n, 8000, seed=3, start_frac=0.9, end_frac=0.5, chimera_frac=0.01)
def sync():
    pl.aligner.sync(); torch.cuda.synchronize()
for k in range(4):
    sync(); t0 = time.perf_counter(); bench.one_step(pl, reads, p.check_reads, 1, prefilter=True); sync()
    print("bytes step %d: %.2f ms" % (k, (time.perf_counter() - t0) * 1e3))
host = reads.arena[:n * 8000].cpu().numpy()
pk, exc = pack_reads(host, n * 8000)
packed = DeviceReads.packed_only(pl.aligner, torch.from_numpy(pk).cuda(), n * 8000, torch.from_numpy(exc).cuda() if exc.size else None, reads.off, reads.length)
for k in range(8):
    if k == 2:
        pl.aligner.set_timing(True); pl.aligner.get_timing()
    sync(); t0 = time.perf_counter(); bench.one_step(pl, packed, p.check_reads, 1, prefilter=True); sync()
    print("packed step %d: %.2f ms  mem alloc %.2f GB reserved %.2f GB" % (k, (time.perf_counter() - t0) * 1e3, torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9))
