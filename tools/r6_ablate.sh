#!/bin/bash
# GPU box, round 6: the traced end-window kernel under ablations (library variants of tools/ab_kernels.sh), full and small launches
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for V in "" nostore notable nostoretable w3; do
  L=$ROOT/porechop_amd/libporechop_amd${V:+_$V}.so
  [ -f $L ] || continue
  for D in 0 1; do
    for N in 1000000 100000; do
      echo "== variant=${V:-base} n=$N"
      PC_LIBRARY=$L PC_DEBUG_TRACE=$D timeout 300 python tools/time_trace_parts.py $N 2>&1 | grep TCUPS
    done
  done
done
