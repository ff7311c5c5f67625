#!/usr/bin/env python3
"""GPU box, round 6: where a headline step behind the exact prefilter spends its time -- phases A / B / C timed apart (a sync
after each: upper bounds), phase B traced in full vs exactly pruned, and a host profile of the whole step.
    python tools/r6_step_times.py [reads] [profile]"""
import cProfile, io, os, pstats, sys, time
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


def sync():
    pl.aligner.sync(); torch.cuda.synchronize()


def clock(fn, reps=10):
    fn(); fn(); sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    sync()
    return out, (time.perf_counter() - t0) / reps * 1e3


(bs, be), ta = clock(lambda: pl.phase_a(reads, check))
matching = pl.matching_sets(bs, be)
(st, et), tb = clock(lambda: pl.phase_b(reads, matching, prune=False))
(st2, et2), tbp = clock(lambda: pl.phase_b(reads, matching, prune=True))
hits, tc = clock(lambda: pl.phase_c(reads, st, et, matching, prefilter=True))
hits_f, tcf = clock(lambda: pl.phase_c(reads, st, et, matching), reps=3)
print("phase A %.2f ms | phase B traced in full %.2f ms, pruned %.2f ms (same %s) | phase C behind the prefilter %.2f ms, in full %.2f ms"
      % (ta, tb, tbp, bool(torch.equal(st, st2) and torch.equal(et, et2)), tc, tcf))


def step():
    bs, be = pl.phase_a(reads, check)
    m = pl.matching_sets(bs, be)
    a, b = pl.phase_b(reads, m)
    return pl.phase_c(reads, a, b, m, prefilter=True)


_, ts = clock(step)
pl.aligner.set_timing(True); pl.aligner.get_timing()
_, ts2 = clock(step)
tm = pl.aligner.get_timing(); pl.aligner.set_timing(False)
print("whole step %.2f ms (timed kernels per step: %s)" % (ts, {k: round(v[0] / 12, 3) for k, v in tm.items() if v[1]}))
if len(sys.argv) > 2:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5):
        step()
    sync()
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30); print(s.getvalue()[:6000])
