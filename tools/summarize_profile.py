#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (tools/profile_round.sh) -> profiles/<round>_*.csv + <round>_summary.json.

The summary holds, per kernel of this library, calls / average and max duration from
`rocprofv3 --kernel-trace --stats`, and the per-launch FETCH_SIZE / WRITE_SIZE (KB, as rocprofv3
reports them) of the largest launch from the two separate `--pmc` passes."""
import collections
import csv
import json
import os
import shutil
import sys

tag, rnd, reads = sys.argv[1], sys.argv[2], int(sys.argv[3])
src = os.path.join("gpurun_out", "prof_" + tag)
out = {"command": "python bench.py --reads %d --steps 2 --warmup 1 --cpu-seconds 0" % reads, "reads_per_gpu": reads,
       "kernels": {}}
for r in csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))):
    name = r["Name"]
    if "pck::" in name or "pc_spec" in name:
        out["kernels"][name] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                "max_ms": int(r["MaxNs"]) / 1e6, "total_ms": int(r["TotalDurationNs"]) / 1e6}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(src, "pmc_%s.csv" % counter))):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if k in out["kernels"]:
            out["kernels"][k][counter + "_KB_largest_launch"] = max(v)
            out["kernels"][k][counter + "_KB_mean_launch"] = sum(v) / len(v)
            out["kernels"][k][counter + "_launches"] = len(v)
for f in ("kernel_stats.csv", "kernel_trace_scan.csv", "pmc_FETCH_SIZE.csv", "pmc_WRITE_SIZE.csv"):
    shutil.copy(os.path.join(src, f), os.path.join("profiles", "%s_%s" % (rnd, f)))
with open(os.path.join("profiles", rnd + "_summary.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps({k: v for k, v in out["kernels"].items() if v.get("max_ms", 0) > 5}, indent=1))
