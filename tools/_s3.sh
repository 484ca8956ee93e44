#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 240 tools/build/uarch_probe2 valu > gpurun_out/r04_uarch_probe3.txt 2>&1 || echo "uarch probe3 failed"
tail -3 gpurun_out/r04_uarch_probe3.txt
tools/ab.sh 512 cornell-box base default base default > gpurun_out/r04_ab_sel.log 2>&1
MCPT_DEPTH=16 tools/ab.sh 32 bathroom:420 base default >> gpurun_out/r04_ab_sel.log 2>&1
tools/ab.sh 64 bathroom:160 base default >> gpurun_out/r04_ab_sel.log 2>&1
tools/ab.sh 256 veach-mis base default >> gpurun_out/r04_ab_sel.log 2>&1
grep -v amdgpu.ids gpurun_out/r04_ab_sel.log
for S in 1 64; do MCPT_LIB_PATH=$GRAFT_REPO_ROOT/monte-carlo-path-tracer_amd/csrc/build/libmcpt_hip_stats.so timeout -k 10 120 python3 tools/sched_stats.py $S c2; done > gpurun_out/r04_sched_1spp.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04_sched_1spp.txt
for G in 256 128 64; do echo "grid $G"; MCPT_WF_GRID=$G timeout -k 10 120 python3 tools/frame_mode_probe.py --calls 30 2>&1 | grep -v amdgpu.ids | head -3; done > gpurun_out/r04_frame_grid.txt 2>&1
cat gpurun_out/r04_frame_grid.txt
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/r04_pytest_c.log 2>&1; tail -5 gpurun_out/r04_pytest_c.log
