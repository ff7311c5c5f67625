#!/usr/bin/env python3
"""GPU box, round 6: which source lines of the host side issue the torch ops of a step behind the exact prefilter
(TorchDispatchMode: every ATen call, attributed to the innermost frame inside porechop_amd/).   python tools/r6_opcount.py [reads]"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
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
    a, b = pl.phase_b(reads, m)
    return pl.phase_c(reads, a, b, m, prefilter=True)


step(); step(); torch.cuda.synchronize()
by_line, by_op = collections.Counter(), collections.Counter()


class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        where = "?"
        for fr in reversed(traceback.extract_stack(limit=14)):
            if "porechop_amd" in fr.filename:
                where = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
                break
        by_line[where] += 1
        by_op[str(func)] += 1
        return func(*args, **(kwargs or {}))


with Count():
    step()
torch.cuda.synchronize()
print("ATen calls in one step: %d" % sum(by_line.values()))
for k, v in by_line.most_common(70):
    print("  %4d  %s" % (v, k))
print("by op:")
for k, v in by_op.most_common(30):
    print("  %4d  %s" % (v, k))
