for n in 0 1 2 3; do
  echo "== PC_FORK_STREAMS=$n"
  PC_FORK_STREAMS=$n timeout 500 python tools/r6_h2d.py 2>&1 | tail -1
  PC_FORK_STREAMS=$n timeout 300 python tools/r6_step_times.py 1000000 2>&1 | grep "phase A\|whole step"
done
