#!/bin/bash
# GPU box: rocprofv3 kernel trace + stats of one leg (tools/run_leg.py) at two step counts; the per-step cost of every
# kernel is the difference (read generation and warm-up cancel).  Summary to gpurun_out/<tag>_<leg>_per_step.txt
# usage: tools/trace_leg.sh <tag> <leg> [env assignments...]
TAG=$1; LEG=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for S in 3 13; do
  rm -rf /tmp/prof_${LEG}_$S
  env "$@" PC_NO_TRACE_FORK=1 PC_NO_SCORE_FORK=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${LEG}_$S -o leg -- python $ROOT/tools/run_leg.py $LEG $S > /tmp/prof_${LEG}_$S.log 2>&1
  grep "ms / step" /tmp/prof_${LEG}_$S.log | tail -2
done
python - /tmp/prof_${LEG}_3 /tmp/prof_${LEG}_13 <<'PY' | tee $OUT/${TAG}_${LEG}_per_step.txt
import csv, glob, sys
def load(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k, (c, t) in b.items():
    c0, t0 = a.get(k, (0, 0.0))
    rows.append(((t - t0) / 10 / 1e6, (c - c0) / 10, k))
rows.sort(reverse=True)
print("per step: %.3f ms in kernels, %.0f launches" % (sum(r[0] for r in rows), sum(r[1] for r in rows)))
for ms, calls, k in rows[:28]:
    print("%8.3f ms  %6.1f launches  %s" % (ms, calls, k[:110]))
PY
