#!/usr/bin/env python3
"""GPU box: the headline step behind the exact prefilter, three ways -- bytes resident, packed-only resident (the plane
scanned, survivors unpacked), and with a bitmap per seed length (PC_PF_MULTI_Q=1, the previous behaviour) -- plus a host
profile of the step (where the time between kernels goes).   python tools/time_prefilter_leg.py [reads]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from porechop_amd.io import pack_reads
from porechop_amd.panel import load_panel
from porechop_amd.pipeline import DeviceReads, Pipeline, ScanParams
from porechop_amd.synth import make_reads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
p = ScanParams()
pl = Pipeline(load_panel(), p)
reads = make_reads(n, 8000, seed=3, start_frac=0.9, end_frac=0.5, chimera_frac=0.01)
check = torch.arange(p.check_reads, device="cuda")


def step(rd):
    bs, be = pl.phase_a(rd, check)
    matching = pl.matching_sets(bs, be)
    st, et = pl.phase_b(rd, matching)
    hits = pl.phase_c(rd, st, et, matching, prefilter=True)
    return st, et, hits


def timed(rd, steps=10):
    for _ in range(2):
        step(rd)
    pl.aligner.sync(); torch.cuda.synchronize()
    pl.aligner.set_timing(True); pl.aligner.get_timing()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step(rd)
    pl.aligner.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    tm = pl.aligner.get_timing(); pl.aligner.set_timing(False)
    return out, dt, {k: round(v[0] / steps, 3) for k, v in tm.items() if v[1]}

(st, et, hits), dt, tm = timed(reads)
print("bytes resident   : %.2f ms/step  kernels %s" % (dt, tm))
host = reads.arena[:n * 8000].cpu().numpy()
pk, exc = pack_reads(host, n * 8000)
t0 = time.perf_counter()
packed = DeviceReads.packed_only(pl.aligner, torch.from_numpy(pk).cuda(), n * 8000, torch.from_numpy(exc).cuda() if exc.size else None, reads.off, reads.length)
pl.aligner.sync(); torch.cuda.synchronize()
print("packed_only build (upload of %.2f GB + end windows): %.1f ms" % (pk.size / 1e9, (time.perf_counter() - t0) * 1e3))
(st2, et2, hits2), dt2, tm2 = timed(packed)
same = torch.equal(st, st2) and torch.equal(et, et2) and torch.equal(hits.read, hits2.read) and torch.equal(hits.start, hits2.start)
print("packed resident  : %.2f ms/step  kernels %s  same=%s  unpacked/step %.3f GB" % (dt2, tm2, same, pl.stats.get("bases_unpacked_after_prefilter", 0) / 12 / 1e9))
if len(sys.argv) > 2:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5):
        step(packed)
    pl.aligner.sync(); torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25); print(s.getvalue()[:5000])
