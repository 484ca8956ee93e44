#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 200 tools/build/uarch_probe > gpurun_out/r04_uarch_probe.txt 2>&1 || echo "uarch probe failed"
tail -5 gpurun_out/r04_uarch_probe.txt
tools/ab.sh 512 cornell-box default dv16 dv48 dm1 dm2 dl2 default > gpurun_out/r04_regime_c2.log 2>&1; cat gpurun_out/r04_regime_c2.log
MCPT_DEPTH=16 tools/ab.sh 32 bathroom:420 default dv48 dm2 dl2 > gpurun_out/r04_regime_c5.log 2>&1; cat gpurun_out/r04_regime_c5.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r04_pytest_a.log 2>&1; tail -5 gpurun_out/r04_pytest_a.log
