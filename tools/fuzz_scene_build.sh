#!/bin/bash
# ASan + UBSan run of the host scene builder on hostile scene descriptions (CPU only).  Usage: tools/fuzz_scene_build.sh [cases] [seed]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${TMPDIR:-/tmp}/mcpt_fuzz_scene_build"
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include \
    "$ROOT/tools/fuzz_scene_build.cpp" "$ROOT/monte-carlo-path-tracer_amd/csrc/scene_build.cpp" -lpthread -o "$OUT"
"$OUT" "${1:-400}" "${2:-1}"
