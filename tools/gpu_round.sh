#!/bin/bash
# One GPU-box call of a round: GPU tests, smoke, micro-benchmarks, the default bench line, the rocprofv3 evidence.
# usage (through gpurun): bash tools/gpu_round.sh <tag> [tests|notests] [profile|noprofile]
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
if [ "${2:-tests}" = tests ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $OUT/${TAG}_gpu_tests.log
  tail -5 $OUT/${TAG}_gpu_tests.log
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
fi
[ -x tools/ubench_valu ] && timeout 120 tools/ubench_valu > $OUT/${TAG}_ubench_valu.txt 2>&1
[ -x tools/ubench_row ] && timeout 120 tools/ubench_row > $OUT/${TAG}_ubench_row.txt 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"
tail -c 3000 $OUT/${TAG}_bench.json
if [ "${3:-profile}" = profile ]; then
  timeout 1500 bash tools/profile_round.sh $TAG 400000 2>&1 | tail -30
fi
