#!/bin/bash
# rocprofv3 evidence for one round, leg by leg (run through gpurun):  bash tools/profile_round6.sh <tag> [legs...]
# Per leg (tools/run_leg.py: 3 identical steps of that leg and nothing else):
#   --kernel-trace --stats                      -> per-kernel calls / average / total duration
#   --pmc FETCH_SIZE, --pmc WRITE_SIZE          -> HBM-side traffic, each in its own pass (never with trace domains)
#   --pmc SQ_* GRBM_GUI_ACTIVE (VALU activity)  -> for the legs named in VALU_LEGS
#   --kernel-trace --stats again with PC_NO_TRACE_FORK=1 PC_NO_SCORE_FORK=1 -> the single-stream durations
# Only the filtered summaries are kept under gpurun_out/prof_<tag>/<leg>/ ; tools/summarize_profile6.py turns them into
# profiles/<round>_summary.json (+ the CSVs it was made from).
TAG=${1:-r06}; shift
LEGS=${@:-headline prefilter configs1 configs2 configs2_pruned configs4_prefilter}
VALU_LEGS=${VALU_LEGS:-headline prefilter configs2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for LEG in $LEGS; do
  STEPS=3; case $LEG in configs4) STEPS=1;; configs4_prefilter) STEPS=2;; esac
  CMD="python $ROOT/tools/run_leg.py $LEG $STEPS"
  W=/tmp/prof_work_$LEG; rm -rf $W; mkdir -p $W $OUT/$LEG
  echo $STEPS > $OUT/$LEG/steps
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/trace -o trace -- $CMD > $OUT/$LEG/trace.log 2>&1
  find $W/trace -name "*kernel_stats.csv" -exec cp {} $OUT/$LEG/kernel_stats.csv \;
  # the same leg with every launch on ONE stream (no forked row classes, no alternating score launches): rocprofv3's
  # AverageNs x Calls then adds up to the step, and is what the roofline's per-launch duration is checked against
  PC_NO_TRACE_FORK=1 PC_NO_SCORE_FORK=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/trace1 -o trace -- $CMD > $OUT/$LEG/trace1.log 2>&1
  find $W/trace1 -name "*kernel_stats.csv" -exec cp {} $OUT/$LEG/kernel_stats_single_stream.csv \;
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $C --output-format csv -d $W/$C -o pmc -- $CMD > $OUT/$LEG/$C.log 2>&1
    CC=$(find $W/$C -name "*counter_collection.csv" | head -1)
    if [ -n "$CC" ]; then head -1 $CC > $OUT/$LEG/pmc_$C.csv; grep -E "pck::|pc_spec" $CC >> $OUT/$LEG/pmc_$C.csv; fi
  done
  if echo " $VALU_LEGS " | grep -q " $LEG "; then
    timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $W/VALU -o pmc -- $CMD > $OUT/$LEG/VALU.log 2>&1
    CC=$(find $W/VALU -name "*counter_collection.csv" | head -1)
    if [ -n "$CC" ]; then head -1 $CC > $OUT/$LEG/pmc_VALU.csv; grep -E "pck::|pc_spec" $CC >> $OUT/$LEG/pmc_VALU.csv; fi
  fi
  for f in $OUT/$LEG/*.log; do grep "^LEG" $f > $f.leg; rm $f; done
  rm -rf $W
  echo "== $LEG"; head -4 $OUT/$LEG/kernel_stats.csv | cut -c1-160
done
du -sh $OUT
