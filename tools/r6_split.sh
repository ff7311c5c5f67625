#!/bin/bash
# GPU box, round 6: tracebacks as launches of their own (PC_SPLIT_WALK=1) against the in-kernel walk
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for ON in 0 1 0 1; do
  echo "=== PC_SPLIT_WALK=$ON"
  for N in 1000000 100000; do
    PC_SPLIT_WALK=$ON timeout 300 python tools/time_trace_parts.py $N 2>&1 | grep TCUPS
  done
done
for ON in 0 1; do
  echo "=== PC_SPLIT_WALK=$ON"
  PC_SPLIT_WALK=$ON timeout 600 python tools/r6_step_times.py 1000000 2>&1 | grep -v amdgpu.ids | head -3
  PC_SPLIT_WALK=$ON timeout 600 python tools/time_phase_b.py 2>&1 | tail -2
done
