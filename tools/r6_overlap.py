#!/usr/bin/env python3
"""GPU box, round 6: do the seed scan of the exact prefilter (HBM / LDS-probe bound) and phase B's traced launches (VALU bound)
overlap when they are enqueued on two streams?  Phase B alone, the prefilter over the UNTRIMMED reads alone, both one after
the other, both side by side.   python tools/r6_overlap.py [reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from porechop_amd.panel import load_panel
from porechop_amd.pipeline import Pipeline, ScanParams
from porechop_amd.synth import make_reads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
p = ScanParams()
pl = Pipeline(load_panel(), p)
al = pl.aligner
reads = make_reads(n, 8000, seed=3, start_frac=0.9, end_frac=0.5, chimera_frac=0.01)
check = torch.arange(p.check_reads, device="cuda")
bs, be = pl.phase_a(reads, check)
m = pl.matching_sets(bs, be)
ads = [a for a, _ in pl._middle_adapters_with_sets(m)]
aidx = [pl.seq_index[a[1]] for a in ads]
ks = [al.max_edits(len(pl.seqs[ai]), p.middle_threshold) for ai in aidx]
off, ln = reads.off.contiguous(), reads.length.contiguous()
s2 = torch.cuda.Stream()


def pf(stream=None):
    al.prefilter_defer_count(True)
    try:
        return al.prefilter_mask(reads.arena, off, ln, 8000, aidx, ks, stream=stream)
    finally:
        al.prefilter_defer_count(False)


def clock(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def both_serial():
    pl.phase_b(reads, m); pf()


def both_side_by_side():
    s2.wait_stream(torch.cuda.current_stream())
    pf(stream=s2.cuda_stream)
    pl.phase_b(reads, m)
    torch.cuda.current_stream().wait_stream(s2)


tb = clock(lambda: pl.phase_b(reads, m))
tp = clock(pf)
ts = clock(both_serial)
tc = clock(both_side_by_side)
print("phase B %.2f ms | prefilter over the untrimmed reads %.2f ms | one after the other %.2f ms | side by side %.2f ms" % (tb, tp, ts, tc))
