#!/bin/bash
# GPU box: same-box A/B of two builds of the library on whole legs:  tools/r6_ab_api.sh <variant> [legs...]   (PC_LIBRARY switches)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
V=$1; shift
LEGS=${@:-prefilter prefilter_packed configs1 headline}
for rep in 1 2 3; do
  for L in base $V; do
    LIB=$ROOT/porechop_amd/libporechop_amd.so; [ "$L" = base ] || LIB=$ROOT/porechop_amd/libporechop_amd_$L.so
    for LEG in $LEGS; do
      echo -n "rep=$rep lib=$L "; PC_LIBRARY=$LIB timeout 300 python tools/run_leg.py $LEG 6 2>&1 | grep "ms / step" | tail -3 | awk '{printf "%s %s | ", $1, $2} END {print ""}'
    done
  done
done
