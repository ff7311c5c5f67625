#!/bin/bash
# GPU box, round 6: A/B of library variants on ONE box:  tools/r6_ab.sh <variant> ...   ("base" = the real library)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2; do
for V in "$@"; do
  L=$ROOT/porechop_amd/libporechop_amd_$V.so
  [ "$V" = base ] && L=$ROOT/porechop_amd/libporechop_amd.so
  [ -f $L ] || continue
  for N in 1000000 100000; do
    echo "== variant=$V n=$N rep=$rep"
    PC_LIBRARY=$L timeout 300 python tools/time_trace_parts.py $N 2>&1 | grep TCUPS
  done
done
done
