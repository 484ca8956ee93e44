#!/bin/bash
# Developer tool (GPU box): scheduler / grid sweep of the 8-wide trace kernel on the bench workload.  usage: tools/sweep8.sh [spp]
cd "$GRAFT_REPO_ROOT"
SPP="${1:-512}"
run() { echo "== $*"; env "$@" MCPT_TIME_KERNELS=8 timeout -k 10 200 python3 tools/perf_probe.py "$SPP" cornell-box 2 2>/dev/null | grep -E "best|trace" | sed 's/.*spp=[0-9]*//'; }
run MCPT_WF_REFILL=28
for v in 20 24 32 36 40; do run MCPT_WF_REFILL=$v; done
for v in 16 20 28 32 36; do run MCPT_WF_INNER=$v; done
for v in 8 12 20 24 28; do run MCPT_WF_LEAF=$v; done
for v in 0 32 40 56 64; do run MCPT_WF_PEND=$v; done
for v in 192 208 240 256; do run MCPT_WF_GRID=$v; done
run MCPT_WF_REFILL=28
