#!/bin/bash
# Developer tool (GPU box): scheduler thresholds of the 8-wide trace kernel on S-veach (cheap rays: the refill block is 40 % of a wave's time there).
cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; env "$@" MCPT_TIME_KERNELS=8 timeout -k 10 200 python3 tools/perf_probe.py 256 veach-mis 2 2>/dev/null | grep -E "best" | sed 's/.*spp=[0-9]*//'; }
run MCPT_WF_REFILL=28
for v in 16 20 24 32 36 44; do run MCPT_WF_REFILL=$v; done
for v in 16 20 28 32; do run MCPT_WF_INNER=$v; done
for v in 8 12 20 24; do run MCPT_WF_LEAF=$v; done
for v in 32 56 64; do run MCPT_WF_PEND=$v; done
run MCPT_WF_REFILL=28
