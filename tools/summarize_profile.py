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
out = {"command": "python bench.py --reads %d --steps 2 --warmup 1 --cpu-seconds 0 --no-extra" % reads, "reads_per_gpu": reads,
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
# VALU activity pass (headline legs only): per kernel, over its launches -- wave64 VALU instructions, the cycles the
# VALU was busy with them (SQ_ACTIVE_INST_VALU counts quad-cycles: x4), and the effective clock
# GRBM_GUI_ACTIVE / 8 XCDs / launch duration (the chip clocks to its power budget, MI355X_MICROARCH.md "DVFS give-back")
valu_csv = os.path.join(src, "pmc_VALU.csv")
if os.path.isfile(valu_csv):
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(valu_csv)):
        k = (r["Dispatch_Id"], r["Kernel_Name"])
        disp[k][r["Counter_Name"]] = float(r["Counter_Value"])
        disp[k]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for (_, name), v in disp.items():
        if name in out["kernels"] and v.get("GRBM_GUI_ACTIVE", 0) > 0:
            for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "ns"):
                per[name][c] += v.get(c, 0.0)
            per[name]["launches"] += 1
    for name, v in per.items():
        gui = v["GRBM_GUI_ACTIVE"] / 8.0                       # cycles of one XCD's clock
        out["kernels"][name]["valu_pass"] = {
            "launches": int(v["launches"]), "total_ms": v["ns"] / 1e6,
            "effective_clock_ghz": gui / v["ns"],
            "valu_wave_instructions": v["SQ_INSTS_VALU"],
            "valu_instr_per_s_per_simd": v["SQ_INSTS_VALU"] / 1024.0 / (v["ns"] / 1e9),
            "valu_busy_frac": v["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / gui,
            "resident_waves_per_simd": v["SQ_WAVE_CYCLES"] * 4.0 / 1024.0 / gui}
    shutil.copy(valu_csv, os.path.join("profiles", "%s_pmc_VALU.csv" % rnd))
for f in ("kernel_stats.csv", "kernel_trace_scan.csv", "pmc_FETCH_SIZE.csv", "pmc_WRITE_SIZE.csv"):
    shutil.copy(os.path.join(src, f), os.path.join("profiles", "%s_%s" % (rnd, f)))
# the full default run (all legs): per-kernel stats only
legs = os.path.join(src, "kernel_stats_all_legs.csv")
if os.path.isfile(legs):
    out["all_legs"] = {"command": "python bench.py --steps 2 --warmup 1 --cpu-seconds 0", "kernels": {}}
    for r in csv.DictReader(open(legs)):
        if "pck::" in r["Name"] or "pc_spec" in r["Name"]:
            out["all_legs"]["kernels"][r["Name"]] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                                     "max_ms": int(r["MaxNs"]) / 1e6, "total_ms": int(r["TotalDurationNs"]) / 1e6}
    shutil.copy(legs, os.path.join("profiles", "%s_kernel_stats_all_legs.csv" % rnd))
with open(os.path.join("profiles", rnd + "_summary.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps({k: v for k, v in out["kernels"].items() if v.get("max_ms", 0) > 5}, indent=1))
