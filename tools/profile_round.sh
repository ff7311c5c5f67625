#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   1. --kernel-trace --stats of the HEADLINE bench (--no-extra: every launch of the dominant kernel in the process is then a
#      launch of the workload bench.py's roofline object is about, so rocprofv3's average duration can be compared with it)
#   2. FETCH_SIZE and WRITE_SIZE of the same command in their own passes (never together with trace domains)
#   3. VALU activity counters (busy / issued VALU cycles, wave cycles, GPU clock ticks), own pass
#   4. --kernel-trace --stats of the full default run (all legs): the traced kernels of configs[1] / configs[2]
# Only the summaries are kept under gpurun_out/prof_<tag>/ ; tools/summarize_profile.py copies them into profiles/.
TAG=${1:-r02}
READS=${2:-1000000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --reads $READS --steps 2 --warmup 1 --cpu-seconds 0 --no-extra"
W=/tmp/prof_work; rm -rf $W; mkdir -p $W
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/trace -o trace -- $CMD > $OUT/bench_trace.log 2>&1
find $W/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# per-dispatch durations of our kernels only (start/end timestamps -> ns)
KT=$(find $W/trace -name "*kernel_trace.csv" | head -1)
if [ -n "$KT" ]; then head -1 $KT > $OUT/kernel_trace_scan.csv; grep -E "pck::|pc_spec" $KT >> $OUT/kernel_trace_scan.csv; fi
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $W/$C -o pmc -- $CMD > $OUT/bench_$C.log 2>&1
  CC=$(find $W/$C -name "*counter_collection.csv" | head -1)
  if [ -n "$CC" ]; then head -1 $CC > $OUT/pmc_$C.csv; grep -E "pck::|pc_spec" $CC >> $OUT/pmc_$C.csv; fi
done
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $W/VALU -o pmc -- $CMD > $OUT/bench_VALU.log 2>&1
CC=$(find $W/VALU -name "*counter_collection.csv" | head -1)
if [ -n "$CC" ]; then head -1 $CC > $OUT/pmc_VALU.csv; grep -E "pck::|pc_spec" $CC >> $OUT/pmc_VALU.csv; fi
if [ "${3:-legs}" = legs ]; then
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $W/legs -o trace -- python $ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/bench_legs.log 2>&1
  find $W/legs -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_all_legs.csv \;
fi
for f in $OUT/bench_*.log; do tail -1 $f > $f.json; grep -v "^W2\|^I2\|^E2" $f | tail -5 > $f.tail; rm $f; done
ls -la $OUT; du -sh $OUT
head -6 $OUT/kernel_stats.csv | cut -c1-200
