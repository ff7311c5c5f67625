#!/bin/bash
# GPU box, round 6: the two-pass end scan on / off (PC_NO_TWO_PASS_ENDS=1): traced end-window kernel alone, and the step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for OFF in 0 1; do
  echo "=== PC_NO_TWO_PASS_ENDS=$OFF"
  for N in 1000000 100000; do
    PC_NO_TWO_PASS_ENDS=$OFF timeout 300 python tools/time_trace_parts.py $N 2>&1 | grep TCUPS
  done
  PC_NO_TWO_PASS_ENDS=$OFF timeout 600 python tools/r6_step_times.py 1000000 2>&1 | grep -v amdgpu.ids | head -3
done
