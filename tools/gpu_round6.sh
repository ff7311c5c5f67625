#!/bin/bash
# One GPU-box call of round 6 (through gpurun):  bash tools/gpu_round6.sh <tag> [dropin] [tests] [bench[:args]] [profile:<legs>] [cmd:<shell>]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for WHAT in "$@"; do
  case $WHAT in
    dropin)
      timeout 1500 python -m pytest tests/test_gpu_dropin.py -m gpu -x -q > $OUT/${TAG}_gpu_dropin.log 2>&1; echo "dropin tests rc=$?" | tee -a $OUT/${TAG}_gpu_dropin.log
      tail -8 $OUT/${TAG}_gpu_dropin.log ;;
    tests)
      timeout 2400 python -m pytest tests -m gpu -q > $OUT/${TAG}_gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $OUT/${TAG}_gpu_tests.log
      tail -6 $OUT/${TAG}_gpu_tests.log
      timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 ;;
    bench*)
      ARGS=$(echo ${WHAT#bench} | tr ':' ' ')
      ( time timeout 1800 python bench.py $ARGS --full-json $OUT/${TAG}_bench_full.json > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err ) 2>&1 | grep real; echo "bench rc=$?"
      tail -c 600 $OUT/${TAG}_bench.err; wc -c $OUT/${TAG}_bench.json ;;
    profile:*)
      LEGS=$(echo ${WHAT#profile:} | tr ',' ' ')
      timeout 2400 bash tools/profile_round6.sh $TAG $LEGS 2>&1 | tail -40 ;;
    cmd:*)
      bash -c "${WHAT#cmd:}" 2>&1 | tail -60 ;;
    *) echo "unknown step $WHAT" ;;
  esac
done
