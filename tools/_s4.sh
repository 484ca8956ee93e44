#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 240 tools/build/uarch_probe2 valu > gpurun_out/r04_uarch_probe3.txt 2>&1 || echo "uarch probe3 failed"
grep -A4 "s_and_b64 vcc\|v_cmp_le_f32 vcc, %0, %14$" gpurun_out/r04_uarch_probe3.txt | tail -24
tools/ab.sh 512 cornell-box base default base default > gpurun_out/r04_ab_wfsel.log 2>&1
MCPT_DEPTH=16 tools/ab.sh 32 bathroom:420 base default >> gpurun_out/r04_ab_wfsel.log 2>&1
tools/ab.sh 64 bathroom:160 base default >> gpurun_out/r04_ab_wfsel.log 2>&1
tools/ab.sh 256 veach-mis base default >> gpurun_out/r04_ab_wfsel.log 2>&1
grep -v amdgpu.ids gpurun_out/r04_ab_wfsel.log
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r04_pytest_d.log 2>&1; tail -5 gpurun_out/r04_pytest_d.log
