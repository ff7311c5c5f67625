#!/bin/bash
# GPU box: board power and shader clock while ONE kernel family runs in a loop -- is the traced kernel held by the power
# cap like the score kernel?   tools/power_probe.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
probe() {
  NAME=$1; shift
  "$@" > /tmp/pp_$NAME.log 2>&1 &
  PID=$!
  sleep ${WARM:-14}
  for i in 1 2 3 4 5 6; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|Average Graphics Package Power|sclk|mclk|fclk" | tr -s ' ' | tr '\n' ';'
    echo
    sleep 0.5
  done
  wait $PID
  grep -E "TCUPS|ms/step|NO_FUSE" /tmp/pp_$NAME.log | tail -3
}
echo "== idle"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr -s ' ' | tr '\n' ';'; echo
echo "== traced end windows (trace16_kernel), looped"
PC_LOOP=400 probe trace python tools/time_trace_parts.py
echo "== headline step (pc_spec_score 80 % of it), looped"
PC_LOOP=60 probe spec python tools/time_headline.py
