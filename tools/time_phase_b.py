"""Phase B of a barcoded batch, full vs exactly pruned (diagnostic, GPU):  python tools/time_phase_b.py [reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from porechop_amd.pipeline import Pipeline, ScanParams
from porechop_amd.runner import Options
from porechop_amd.synth import make_reads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
p, opts = ScanParams(), Options()
pl = Pipeline(bench.load_panel_sets(), p)
fw = [a for a in bench.load_panel_json() if a["name"].startswith("Barcode ") and "(forward)" in a["name"]]
reads = make_reads(n, 8000, seed=2, start_frac=0.9, end_frac=0.5, barcodes_start=[a["start"][1] for a in fw], barcodes_end=[a["end"][1] for a in fw])
for prune in (False, True, True):
    pl.stats["pairs_end"] = pl.stats["pairs_end_traced_after_pruning"] = 0
    pl.aligner.set_timing(True); pl.aligner.get_timing()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = bench.step_demux(pl, reads, p.check_reads, opts, prune=prune)
    pl.aligner.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    k = pl.aligner.get_timing()
    print("prune", prune, "%.1f ms" % dt, {a: (round(b[0], 1), b[1]) for a, b in k.items() if b[1]}, "traced fraction %.3f" % (pl.stats["pairs_end_traced_after_pruning"] / max(1, pl.stats["pairs_end"])), flush=True)
    if prune:
        assert torch.equal(out[3], ref[3]) and torch.equal(out[4], ref[4]) and np.array_equal(out[5], ref[5])
    else:
        ref = out
