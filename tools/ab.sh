#!/usr/bin/env bash
# Developer tool (GPU box): A/B library builds on the bench workload.  usage: tools/ab.sh <spp> <scene> <lib1> <lib2> ...  ("default" = csrc/libmcpt_hip.so)
set -uo pipefail
SPP="$1"; SCENE="$2"; shift 2
cd "$GRAFT_REPO_ROOT"
for L in "$@"; do
  if [ "$L" = default ]; then unset MCPT_LIB_PATH; else export MCPT_LIB_PATH="$GRAFT_REPO_ROOT/monte-carlo-path-tracer_amd/csrc/build/libmcpt_hip_$L.so"; fi
  MCPT_TIME_KERNELS=8 timeout -k 10 300 python3 tools/perf_probe.py "$SPP" "$SCENE" 3 || exit 1
done
