#!/bin/bash
# GPU box, round 6: VALU counters + clock of the traced end-window kernel (tools/time_trace_parts.py) per library variant
#   tools/r6_pmc.sh "<variant>:<PC_DEBUG_TRACE>" ...      (variant "" = the real library)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for SPEC in "$@"; do
V=${SPEC%%:*}; D=${SPEC##*:}
L=$ROOT/porechop_amd/libporechop_amd${V:+_$V}.so
echo "=== variant=${V:-base} debug=$D"
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/pmc_tp
  PC_LIBRARY=$L PC_DEBUG_TRACE=$D PC_LOOP=2 timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/pmc_tp -o pmc -- python $ROOT/tools/time_trace_parts.py > /tmp/pmc_tp.log 2>&1
  grep TCUPS /tmp/pmc_tp.log | head -2 || tail -5 /tmp/pmc_tp.log
  python - /tmp/pmc_tp <<'PY'
import csv, sys, collections, glob
cc = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
kt = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r.get("Dispatch_Id")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
if not cc:
    print("no counter file"); sys.exit()
rows = list(csv.DictReader(open(cc[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if "trace16" not in r["Kernel_Name"]:
        continue
    agg[(r["Kernel_Name"][:34], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in list(agg.items())[-2:]:
    gui = v.get("GRBM_GUI_ACTIVE", 0) or 1
    ns = dur.get(k[1])
    s = "%s disp %s" % k
    if ns: s += "  %.3f ms  clock %.2f GHz" % (ns / 1e6, gui / ns)
    print(s, {c: "%.4g" % x for c, x in v.items()})
    if "SQ_ACTIVE_INST_VALU" in v:
        print("   valu busy %.3f  waves/simd %.2f  valu insts/s/SIMD %s" % (v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / gui, v["SQ_WAVE_CYCLES"] * 4 / 1024 / gui,
              ("%.3f G" % (v["SQ_INSTS_VALU"] / 1024 / ns) if ns else "?")))
PY
done
done
