#!/usr/bin/env python3
"""GPU box, round 6: every torch call of one prefilter-route step that SYNCHRONISES the host with the device
(torch.cuda.set_sync_debug_mode: .cpu() / .item() / nonzero / boolean-mask indexing / pageable uploads), by issuing source line.
The library's own round trips (pc_prefilter_device's count, ...) are not torch's and are not listed.   python tools/r6_syncs.py [reads] [headline]"""
import collections, os, sys, traceback, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from porechop_amd.panel import load_panel
from porechop_amd.pipeline import Pipeline, ScanParams
from porechop_amd.synth import make_reads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pref = not (len(sys.argv) > 2 and sys.argv[2] == "headline")
p = ScanParams()
pl = Pipeline(load_panel(), p)
reads = make_reads(n, 8000, seed=3, start_frac=0.9, end_frac=0.5, chimera_frac=0.01)
check = torch.arange(p.check_reads, device="cuda")


def step():
    bs, be = pl.phase_a(reads, check)
    m = pl.matching_sets(bs, be)
    a, b = pl.phase_b(reads, m)
    return pl.phase_c(reads, a, b, m, prefilter=pref)


step(); step(); torch.cuda.synchronize()
seen = collections.Counter()
orig = warnings.showwarning


def show(message, category, filename, lineno, file=None, line=None):
    where = "?"
    for fr in reversed(traceback.extract_stack(limit=60)):
        if "porechop_amd" in fr.filename:
            where = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
            break
    if where == "?":                      # no frame of the package: say where it was, whatever it is
        where = "? " + " <- ".join("%s:%d" % (os.path.basename(fr.filename), fr.lineno) for fr in reversed(traceback.extract_stack(limit=8)[:-1]))
    seen[where] += 1


warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode(1)
step()
torch.cuda.set_sync_debug_mode(0)
warnings.showwarning = orig
print("synchronising torch calls in one step: %d" % sum(seen.values()))
for k, v in sorted(seen.items(), key=lambda kv: kv[0]):
    print("  %3d  %s" % (v, k))
