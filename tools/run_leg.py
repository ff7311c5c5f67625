#!/usr/bin/env python3
"""One leg of bench.py, alone, for the profiler: every launch in the process then belongs to that leg, so rocprofv3's
per-kernel numbers can be attributed to it.  `steps` identical steps, nothing else (no CPU legs, no cross-checks).
usage: run_leg.py headline|prefilter|prefilter_packed|ultralong|ultralong_prefilter|configs1|configs2|configs2_pruned|configs4|configs4_prefilter [steps]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from porechop_amd.pipeline import Pipeline, ScanParams  # noqa: E402
from porechop_amd.runner import Options  # noqa: E402
from porechop_amd.synth import make_reads  # noqa: E402

leg = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
p, opts = ScanParams(), Options()
pl = Pipeline(bench.load_panel_sets(), p, device=dev)
pl.n_panel = len(pl.sets)
fw = [a for a in bench.load_panel_json() if a["name"].startswith("Barcode ") and "(forward)" in a["name"]]
bc = dict(barcodes_start=[a["start"][1] for a in fw], barcodes_end=[a["end"][1] for a in fw])
if leg in ("headline", "prefilter", "prefilter_packed"):
    reads = make_reads(1_000_000, 8000, seed=3, start_frac=0.9, end_frac=0.5, chimera_frac=0.01, device=dev)
    if leg == "prefilter_packed":             # the reads resident at 2 bits per base: the plane scanned, survivors unpacked
        from porechop_amd.io import pack_reads
        from porechop_amd.pipeline import DeviceReads
        nb = reads.n * 8000
        pk, exc = pack_reads(reads.arena[:nb].cpu().numpy(), nb)
        reads = DeviceReads.packed_only(pl.aligner, torch.from_numpy(pk).to(dev), nb, torch.from_numpy(exc).to(dev) if exc.size else None,
                                        reads.off, reads.length, end_size=p.end_size)
        torch.cuda.empty_cache()
    step = lambda: bench.one_step(pl, reads, p.check_reads, 1, prefilter=(leg != "headline"))
elif leg in ("ultralong", "ultralong_prefilter"):
    from porechop_amd.synth import make_ragged_reads
    reads = make_ragged_reads(40_000, mean_len=20000, sigma=1.2, min_len=20, seed=9, start_frac=0.9, end_frac=0.5, chimera_frac=0.05, device=dev)
    step = lambda: bench.one_step(pl, reads, p.check_reads, 1, prefilter=(leg == "ultralong_prefilter"))
elif leg == "configs1":
    reads = make_reads(100_000, 8000, seed=1, start_frac=0.9, end_frac=0.5, chimera_frac=0.0, device=dev)
    step = lambda: bench.step_end_trim(pl, reads, p.check_reads)
elif leg in ("configs2", "configs2_pruned"):
    reads = make_reads(1_000_000, 8000, seed=2, start_frac=0.9, end_frac=0.5, chimera_frac=0.0, device=dev, **bc)
    step = lambda: bench.step_demux(pl, reads, p.check_reads, opts, prune=(leg == "configs2_pruned"))
elif leg in ("configs4", "configs4_prefilter"):
    reads = make_reads(int(os.environ.get("PC_LEG_READS4", "1250000")), 8000, seed=4, start_frac=0.9, end_frac=0.5, chimera_frac=0.01, device=dev, **bc)
    step = lambda: bench.step_configs4(pl, reads, p.check_reads, opts, prefilter=(leg == "configs4_prefilter"), prune_b=(leg == "configs4_prefilter"))
else:
    raise SystemExit("unknown leg " + leg)
if os.environ.get("PC_LEG_SYNCS", "0") not in ("", "0"):
    # instead of timing: the torch calls of ONE step of this leg that synchronise host and device, by source line (tools/r6_syncs.py)
    import collections, traceback, warnings
    step(); step(); torch.cuda.synchronize()
    seen = collections.Counter()

    def show(message, category, filename, lineno, file=None, line=None):
        where = "(not the package)"
        for fr in reversed(traceback.extract_stack(limit=60)):
            if "porechop_amd" in fr.filename or fr.filename.endswith("bench.py"):
                where = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
                break
        seen[where] += 1
    warnings.showwarning = show
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode(1)
    step()
    torch.cuda.set_sync_debug_mode(0)
    print("%s: synchronising torch calls in one step: %d" % (leg, sum(seen.values())))
    for k, v in sorted(seen.items()):
        print("  %3d  %s" % (v, k))
    sys.exit(0)
import time  # noqa: E402
for _ in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    print("%s: %.1f ms / step, %.3f M reads/s" % (leg, 1e3 * (time.perf_counter() - t0), reads.n / (time.perf_counter() - t0) / 1e6), file=sys.stderr)
    pl.aligner.sync()
torch.cuda.synchronize()
print("LEG", leg, "steps", steps, "library_sha1", bench.library_fingerprint())
pl.close()
