#!/bin/bash
# Developer tool (GPU box): combinations of the trace scheduler thresholds on two scenes.  usage: tools/sweep8c.sh
cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; env "$@" MCPT_TIME_KERNELS=8 timeout -k 10 200 python3 tools/perf_probe.py 512 cornell-box 2 2>/dev/null | grep -E "best" | sed 's/.*spp=[0-9]*//'; env "$@" MCPT_TIME_KERNELS=8 timeout -k 10 200 python3 tools/perf_probe.py 64 bathroom:64 2 2>/dev/null | grep -E "best" | sed 's/.*spp=[0-9]*//'; }
run MCPT_WF_REFILL=28
run MCPT_WF_INNER=32
run MCPT_WF_INNER=32 MCPT_WF_REFILL=32
run MCPT_WF_INNER=32 MCPT_WF_REFILL=32 MCPT_WF_PEND=56
run MCPT_WF_INNER=32 MCPT_WF_PEND=56
run MCPT_WF_INNER=28 MCPT_WF_REFILL=32 MCPT_WF_PEND=56
run MCPT_WF_INNER=32 MCPT_WF_REFILL=32 MCPT_WF_PEND=56 MCPT_WF_LEAF=12
run MCPT_WF_INNER=36 MCPT_WF_REFILL=32 MCPT_WF_PEND=64
run MCPT_WF_REFILL=28
