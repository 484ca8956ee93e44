#!/bin/bash
# Build and run tools/wide_bvh_probe.cpp (host only).  usage: tools/wide_bvh_probe.sh scene.obj [rays]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; C="$ROOT/monte-carlo-path-tracer_amd/csrc"
OUT="${TMPDIR:-/tmp}/mcpt_wide_bvh_probe"
g++ -std=c++17 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I"$ROOT/monte-carlo-path-tracer_amd/host" "$ROOT/tools/wide_bvh_probe.cpp" \
    "$C/libmcpt_host.a" -L"$C" -lmcpt_hip -L/opt/rocm/lib -lamdhip64 -lz -lpthread -Wl,-rpath,"$C" -Wl,-rpath,/opt/rocm/lib -o "$OUT"   # (build_host_scene comes from libmcpt_hip.so)
"$OUT" "$@"
