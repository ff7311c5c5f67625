#!/bin/bash
# timing experiment (GPU box): where the traced end-window kernel's time goes
for d in 0 1 2 3; do echo "PC_DEBUG_TRACE=$d (1 = no traceback, 2 = trace words all to column 0, 3 = both)"; PC_DEBUG_TRACE=$d python tools/time_trace.py ${1:-1000000} 2>&1 | grep "one traced"; done
