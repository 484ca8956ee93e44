#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r04_pytest_e.log 2>&1; tail -5 gpurun_out/r04_pytest_e.log
timeout -k 10 300 tools/frame_loop.sh 300 trace > gpurun_out/r04_frame_loop.log 2>&1; cat gpurun_out/r04_frame_loop.log
tools/ab.sh 512 cornell-box base default base default > gpurun_out/r04_ab_extq.log 2>&1
tools/ab.sh 64 cornell-box base default >> gpurun_out/r04_ab_extq.log 2>&1
MCPT_DEPTH=16 tools/ab.sh 32 bathroom:420 base default >> gpurun_out/r04_ab_extq.log 2>&1
tools/ab.sh 64 bathroom:160 base default >> gpurun_out/r04_ab_extq.log 2>&1
grep -v amdgpu.ids gpurun_out/r04_ab_extq.log
timeout -k 10 120 python3 tools/frame_mode_probe.py --calls 30 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_frame_mode.txt; cat gpurun_out/r04_frame_mode.txt
