#!/bin/bash
# GPU box: VALU counters of the traced end-window kernel alone (tools/time_trace_parts.py), per launch
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for D in 0; do
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc_tp
  PC_DEBUG_TRACE=$D rocprofv3 --pmc $SET --output-format csv -d /tmp/pmc_tp -o pmc -- python $ROOT/tools/time_trace_parts.py > /tmp/pmc_tp.log 2>&1
  grep TCUPS /tmp/pmc_tp.log || tail -5 /tmp/pmc_tp.log
  python - $(find /tmp/pmc_tp -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if "trace16" not in r["Kernel_Name"]:
        continue
    key = r["Kernel_Name"][:40] + " disp " + r["Dispatch_Id"]
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in list(agg.items())[-2:]:
    gui = v.get("GRBM_GUI_ACTIVE", 0) or 1
    print(k, {c: "%.3g" % x for c, x in v.items()})
    print("   valu busy %.2f  waves/simd %.2f  valu instr/wave-cycle... insts %.3g salu %.3g wait_any frac %.2f" % (
        v.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / gui, v["SQ_WAVE_CYCLES"] * 4 / 1024 / gui, v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_SALU", 0),
        v.get("SQ_WAIT_INST_ANY", 0) / max(1.0, v["SQ_WAVE_CYCLES"])))
PY
done
done
