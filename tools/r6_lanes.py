#!/usr/bin/env python3
"""GPU box, round 6: does a step behind the exact prefilter get shorter when phases B + C of the batch's two halves run in
two host threads, each with a library context and a stream of its own (the host round trips of one half then overlap the
other half's kernels)?   python tools/r6_lanes.py [reads] [lanes] [prefilter 0/1]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from porechop_amd.panel import load_panel
from porechop_amd.pipeline import Pipeline, ScanParams, DeviceReads
from porechop_amd.synth import make_reads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pref = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
p = ScanParams()
pls = [Pipeline(load_panel(), p) for _ in range(lanes)]
reads = make_reads(n, 8000, seed=3, start_frac=0.9, end_frac=0.5, chimera_frac=0.01)
check = torch.arange(p.check_reads, device="cuda")
streams = [torch.cuda.Stream() for _ in range(lanes)]
cut = [n * k // lanes for k in range(lanes + 1)]
subs = [DeviceReads(reads.arena, reads.off[cut[k]:cut[k + 1]], reads.length[cut[k]:cut[k + 1]]) for k in range(lanes)]


def single():
    pl = pls[0]
    bs, be = pl.phase_a(reads, check)
    m = pl.matching_sets(bs, be)
    a, b = pl.phase_b(reads, m)
    return a, b, pl.phase_c(reads, a, b, m, prefilter=pref)


def laned():
    pl = pls[0]
    bs, be = pl.phase_a(reads, check)
    m = pl.matching_sets(bs, be)
    ev = torch.cuda.Event(); ev.record()
    out = [None] * lanes

    def work(k):
        with torch.cuda.stream(streams[k]):
            streams[k].wait_event(ev)
            a, b = pls[k].phase_b(subs[k], m)
            h = pls[k].phase_c(subs[k], a, b, m, prefilter=pref)
            out[k] = (a, b, h)
            streams[k].synchronize()
    th = [threading.Thread(target=work, args=(k,)) for k in range(lanes)]
    for t in th: t.start()
    for t in th: t.join()
    st = torch.cat([o[0] for o in out]); et = torch.cat([o[1] for o in out])
    hr = torch.cat([o[2].read + cut[k] for k, o in enumerate(out)])
    return st, et, hr, [o[2] for o in out]


def clock(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        o = fn()
    torch.cuda.synchronize()
    return o, (time.perf_counter() - t0) / reps * 1e3


(s_st, s_et, s_h), t1 = clock(single)
(l_st, l_et, l_hr, l_h), t2 = clock(laned)
def canon(read, h_list):
    """hits as rows (read, start, end, adapter, identity), sorted: a run lists them round by round, not read by read"""
    rows = torch.stack([read.double(), torch.cat([h.start for h in h_list]).double(), torch.cat([h.end for h in h_list]).double(),
                        torch.cat([h.adapter for h in h_list]).double(), torch.cat([h.identity for h in h_list])], 1)
    key = rows[:, 0] * 1e6 + rows[:, 1]
    return rows[torch.argsort(key)]


trims = bool(torch.equal(s_st, l_st) and torch.equal(s_et, l_et))
hits = bool(torch.equal(canon(s_h.read, [s_h]), canon(l_hr, l_h)))
same = "trims %s hits %s (%d)" % (trims, hits, int(l_hr.numel()))
print("reads %d prefilter %s: one lane %.2f ms | %d lanes %.2f ms | same %s" % (n, pref, t1, lanes, t2, same))
