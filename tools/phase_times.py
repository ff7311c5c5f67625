"""Per-phase wall times of the batched pipeline on the GPU (diagnostic)."""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import load_panel_sets
from porechop_amd.pipeline import Pipeline, ScanParams
from porechop_amd.synth import make_reads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
pl = Pipeline(load_panel_sets(), ScanParams())
reads = make_reads(n, 8000, seed=3, start_frac=0.9, end_frac=0.5, chimera_frac=0.01)
def sync(): torch.cuda.synchronize()
for it in range(3):
    pl.aligner.set_timing(True)
    sync(); t0 = time.perf_counter()
    bs, be = pl.phase_a(reads, torch.arange(10000, device="cuda")); sync(); t1 = time.perf_counter()
    ka = pl.aligner.get_timing()
    m = pl.matching_sets(bs, be)
    st, et = pl.phase_b(reads, m); sync(); t2 = time.perf_counter()
    kb = pl.aligner.get_timing()
    hits = pl.phase_c(reads, st, et, m); sync(); t3 = time.perf_counter()
    kc = pl.aligner.get_timing()
    print("iter", it, "A %.1f ms (kernels %.1f)  B %.1f ms (kernels %.1f)  C %.1f ms (score %.1f trace %.1f) rounds %d" % (
        (t1-t0)*1e3, sum(v[0] for v in ka.values()), (t2-t1)*1e3, sum(v[0] for v in kb.values()),
        (t3-t2)*1e3, kc["score"][0], kc["trace"][0], hits.rounds))
