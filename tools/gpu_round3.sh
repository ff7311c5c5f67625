#!/bin/bash
# One GPU-box call of round 3 (through gpurun):  bash tools/gpu_round3.sh <tag> [tests] [fuzz] [bench] [profile:<legs,comma separated>]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for WHAT in "$@"; do
  case $WHAT in
    tests)
      timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $OUT/${TAG}_gpu_tests.log
      tail -6 $OUT/${TAG}_gpu_tests.log
      timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 ;;
    fuzz)
      PC_CHECK_RANGE=1 PC_JIT_CHECK_RANGE=1 PC_JIT_MIN_CELLS=1 PC_JIT_CACHE_DIR=off timeout 900 python tools/fuzz_parity.py 40 7 > $OUT/${TAG}_fuzz_check_range.txt 2>&1
      tail -4 $OUT/${TAG}_fuzz_check_range.txt ;;
    bench)
      timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"
      tail -c 600 $OUT/${TAG}_bench.err; wc -c $OUT/${TAG}_bench.json ;;
    profile:*)
      LEGS=$(echo ${WHAT#profile:} | tr ',' ' ')
      timeout 2400 bash tools/profile_round3.sh $TAG $LEGS 2>&1 | tail -40 ;;
    *) echo "unknown step $WHAT" ;;
  esac
done
