#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/<leg>/ (tools/profile_round6.sh) -> profiles/<round>_summary.json + the CSVs it was made from.

Per leg and kernel FAMILY (every instantiation of a template counts as one kernel: trace16_kernel<R>, prefilter_kernel<P>,
scan_kernel<...>), per STEP of the leg: launches, summed duration, FETCH_SIZE and WRITE_SIZE in bytes (rocprofv3 reports
KB), and -- where the VALU pass ran -- issued wave64 VALU instructions, VALU busy fraction and the effective clock.
The summary records the sha1 of the library the profiled processes loaded; bench.py only uses the traffic figures when
that equals the library it is running (a kernel change without a re-profile reports null, never stale counters)."""
import collections
import csv
import json
import os
import re
import shutil
import sys

# MI355X_MICROARCH.md, "HBM": on gfx950 rocprofv3's FETCH_SIZE reports exactly half of the bytes of a wide (16 B per lane)
# streaming read -- "double it before comparing with a byte count".  That is the access pattern of the two prefilter kernels
# (aligned dwordx4 per lane, consecutive lanes consecutive addresses).  The DP kernels GATHER -- every lane reads 16 bytes of
# its own window, 64-128 different cache lines per load -- an access pattern the guide gives no calibration for: as reported.
FETCH_CORRECTION = {"seed_scan_kernel": 2.0, "prefilter_kernel": 2.0, "seed_scan_packed_kernel": 2.0}

tag, rnd = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", "prof_" + tag)


def family(name):
    m = re.search(r"(pc_spec_score|trace16_kernel|seed_scan_packed_kernel|seed_verify_packed_kernel|unpack_windows_kernel|unpack_windows_exceptions_kernel|seed_scan_kernel|seed_verify_kernel|(?<![a-z_])scan_kernel<[^,>]*,[^,>]*, *(?:true|false)>|(?<![a-z_])scan_kernel|prefilter_kernel|plan_kernel|reduce_kernel|"
                  r"expand_tiles_kernel|copy_windows_kernel|unit_real_kernel|unit_scan_kernel|bucket_count_kernel|bucket_prefix_kernel|bucket_scatter_kernel|select_kernel|gather_records_kernel|gather_kernel|scatter_kernel|unpack_kernel|unpack_exceptions_kernel|slow_kernel)", name)
    if not m:
        return None
    f = m.group(1)
    if f in ("select_kernel", "gather_kernel", "scatter_kernel", "unpack_kernel", "reduce_kernel") and "pck::" not in name:
        return None                                # (torch has kernels of these names too)
    if f.startswith("scan_kernel<"):
        f = "scan_kernel<traced>" if f.endswith("true>") else "scan_kernel<score>"
    return f


out = {"legs": {}, "library_sha1": None, "how": "tools/profile_round6.sh %s; per-step figures; FETCH_SIZE / WRITE_SIZE as rocprofv3 reports "
       "them (KB -> bytes), each counter in its own --pmc pass; FETCH_SIZE of the kernels that stream 16 B per lane doubled as "
       "MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE_correction)" % tag}
for leg in sorted(os.listdir(src)):
    d = os.path.join(src, leg)
    if not os.path.isfile(os.path.join(d, "kernel_stats.csv")):
        continue
    steps = int(open(os.path.join(d, "steps")).read().strip())
    for f in os.listdir(d):
        if f.endswith(".leg"):
            for line in open(os.path.join(d, f)):
                m = re.search(r"library_sha1 (\w+)", line)
                if m:
                    # (legs taken in two calls may have met two builds that differ in the host I/O code only -- pc_io.cpp is not
                    # part of the device fingerprint the summary is bound by, tools/device_fingerprint.py --stamp; both are recorded)
                    out.setdefault("library_sha1_by_leg", {})[leg] = m.group(1)
                    out["library_sha1"] = m.group(1)
    ks = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))):
        fam = family(r["Name"])
        if fam:
            ks[fam]["launches_per_step"] += int(r["Calls"]) / steps
            ks[fam]["ms_per_step"] += int(r["TotalDurationNs"]) / 1e6 / steps
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        pth = os.path.join(d, "pmc_%s.csv" % counter)
        if not os.path.isfile(pth):
            continue
        n = collections.defaultdict(int)
        for r in csv.DictReader(open(pth)):
            fam = family(r["Kernel_Name"])
            if fam:
                corr = FETCH_CORRECTION.get(fam, 1.0) if counter == "FETCH_SIZE" else 1.0
                ks[fam][counter + "_bytes_per_step"] += float(r["Counter_Value"]) * 1024.0 * corr / steps
                if corr != 1.0:
                    ks[fam]["FETCH_SIZE_correction"] = corr
                n[fam] += 1
        shutil.copy(pth, os.path.join("profiles", "%s_%s_pmc_%s.csv" % (rnd, leg, counter)))
    pth = os.path.join(d, "pmc_VALU.csv")
    if os.path.isfile(pth):
        disp = collections.defaultdict(dict)
        for r in csv.DictReader(open(pth)):
            k = (r["Dispatch_Id"], family(r["Kernel_Name"]))
            disp[k][r["Counter_Name"]] = float(r["Counter_Value"])
            disp[k]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for (_, fam), v in disp.items():
            if fam and v.get("GRBM_GUI_ACTIVE", 0) > 0:
                for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "ns"):
                    per[fam][c] += v.get(c, 0.0)
        for fam, v in per.items():
            gui = v["GRBM_GUI_ACTIVE"] / 8.0                       # cycles of one XCD's clock
            ks[fam]["valu_pass"] = {"effective_clock_ghz": gui / v["ns"],
                                    "valu_wave_instructions_per_step": v["SQ_INSTS_VALU"] / steps,
                                    "valu_instr_per_s_per_simd": v["SQ_INSTS_VALU"] / 1024.0 / (v["ns"] / 1e9),
                                    "valu_busy_frac": v["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / gui,
                                    "resident_waves_per_simd": v["SQ_WAVE_CYCLES"] * 4.0 / 1024.0 / gui}
        shutil.copy(pth, os.path.join("profiles", "%s_%s_pmc_VALU.csv" % (rnd, leg)))
    shutil.copy(os.path.join(d, "kernel_stats.csv"), os.path.join("profiles", "%s_%s_kernel_stats.csv" % (rnd, leg)))
    single = os.path.join(d, "kernel_stats_single_stream.csv")
    if os.path.isfile(single):
        # every launch on one stream (PC_NO_TRACE_FORK=1 PC_NO_SCORE_FORK=1): per-kernel durations that add up to the step
        for r in csv.DictReader(open(single)):
            fam = family(r["Name"])
            if fam:
                ks[fam]["single_stream_launches_per_step"] += int(r["Calls"]) / steps
                ks[fam]["single_stream_ms_per_step"] += int(r["TotalDurationNs"]) / 1e6 / steps
        shutil.copy(single, os.path.join("profiles", "%s_%s_kernel_stats_single_stream.csv" % (rnd, leg)))
    out["legs"][leg] = {"steps_profiled": steps, "command": "python tools/run_leg.py %s %d" % (leg, steps),
                        "kernels": {k: dict(v) for k, v in ks.items()}}
# the sources the profiled library was built from, when that library is the one in the tree now (tools/device_fingerprint.py:
# bench.py then keeps using these counters after a change to the host I/O code alone)
try:
    import hashlib
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import device_fingerprint
    lib = os.path.join("porechop_amd", "libporechop_amd.so")
    if out["library_sha1"] and hashlib.sha1(open(lib, "rb").read()).hexdigest() == out["library_sha1"]:
        out["device_sources_sha1"] = device_fingerprint.fingerprint()
except Exception:
    pass
with open(os.path.join("profiles", rnd + "_summary.json"), "w") as f:
    json.dump(out, f, indent=1)
for leg, v in out["legs"].items():
    for k, kv in v["kernels"].items():
        if kv.get("ms_per_step", 0) > 0.5:
            print("%-20s %-22s %8.2f ms/step %6.1f launches  fetch %7.2f GB  write %7.2f GB  %s" % (
                leg, k, kv["ms_per_step"], kv["launches_per_step"], kv.get("FETCH_SIZE_bytes_per_step", 0) / 1e9,
                kv.get("WRITE_SIZE_bytes_per_step", 0) / 1e9,
                ("busy %.2f clk %.2f" % (kv["valu_pass"]["valu_busy_frac"], kv["valu_pass"]["effective_clock_ghz"])) if "valu_pass" in kv else ""))
print("library_sha1", out["library_sha1"])
