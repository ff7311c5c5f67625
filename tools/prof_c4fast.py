"""Host-side profile of the configs[4]-shape step with prefilter + pruned phase B (diagnostic, GPU)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from porechop_amd.pipeline import Pipeline, ScanParams
from porechop_amd.runner import Options
from porechop_amd.synth import make_reads
dev = torch.device("cuda", 0)
p, opts = ScanParams(), Options()
pl = Pipeline(bench.load_panel_sets(), p, device=dev)
pl.n_panel = len(pl.sets)
fw = [a for a in bench.load_panel_json() if a["name"].startswith("Barcode ") and "(forward)" in a["name"]]
bc = dict(barcodes_start=[a["start"][1] for a in fw], barcodes_end=[a["end"][1] for a in fw])
reads = make_reads(1250000, 8000, seed=4, start_frac=0.9, end_frac=0.5, chimera_frac=0.01, device=dev, **bc)
step = lambda: bench.step_configs4(pl, reads, p.check_reads, opts, prefilter=True, prune_b=True)
for _ in range(2):
    step()
torch.cuda.synchronize()
# where the wall time goes with every call synchronised (per-function device time shows up in its caller)
os.environ["PC_PROF_SYNC"] = "1"
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr.disable()
print("ms/step %.1f" % ((time.perf_counter() - t0) / 3 * 1e3))
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
