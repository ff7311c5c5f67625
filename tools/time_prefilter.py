"""Times of the exact bit-parallel prefilter and of phase C with / without it (diagnostic, GPU).
usage: time_prefilter.py [reads] [barcodes]     barcodes = number of forward barcodes planted / matched (0 = none)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import load_panel_sets, load_panel_json
from porechop_amd.pipeline import Pipeline, ScanParams
from porechop_amd.synth import make_reads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
nbc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
p = ScanParams()
pl = Pipeline(load_panel_sets(), p)
kw = {}
if nbc:
    fw = [a for a in load_panel_json() if a["name"].startswith("Barcode ") and "(forward)" in a["name"]][:nbc]
    kw = dict(barcodes_start=[a["start"][1] for a in fw], barcodes_end=[a["end"][1] for a in fw])
reads = make_reads(n, 8000, seed=4 if nbc else 3, start_frac=0.9, end_frac=0.5, chimera_frac=0.01, **kw)
sync = torch.cuda.synchronize
bs, be = pl.phase_a(reads, torch.arange(10000, device="cuda"))
m = pl.matching_sets(bs, be)
st, et = pl.phase_b(reads, m)
sync()
print("matching sets:", len(m), "middle adapters:", len(pl.middle_adapter_list(m)))
out = {}
for name, kwargs in (("prefilter", dict(prefilter=True)), ("prove", dict(prove=True)), ("full", {})):
    if name != "prefilter" and nbc > 12 and n > 200000:
        continue
    for it in range(3):
        pl.aligner.set_timing(True); pl.aligner.get_timing()
        sync(); t0 = time.perf_counter()
        h = pl.phase_c(reads, st, et, m, **kwargs); pl.aligner.sync(); sync()
        dt = (time.perf_counter() - t0) * 1e3
        k = pl.aligner.get_timing()
        print(name, it, "%.1f ms" % dt, {a: (round(b[0], 2), b[1]) for a, b in k.items() if b[1]}, "hits", int(h.read.numel()), "rounds", h.rounds, flush=True)
    out[name] = {"ms": dt, "kernels": {a: b[0] for a, b in k.items()}, "hits": int(h.read.numel())}
lib = pl.aligner.lib
import ctypes
c, d = ctypes.c_int64(), ctypes.c_int64()
lib.pc_jit_stats(ctypes.byref(c), ctypes.byref(d))
out["jit"] = {"compiled_now": c.value, "from_disk": d.value}
A = len(pl.middle_adapter_list(m))
out["prefilter_pair_columns_per_s"] = n * A * 8000 / (out["prefilter"]["kernels"]["prefilter"] / 1e3)
print(json.dumps(out))
