#!/bin/bash
# The reference's display loop through the host classes, S-cornell 800x800 (tools/frame_loop.cpp).  usage (GPU box): tools/frame_loop.sh [frames] [trace]
set -e
cd "$(dirname "$0")/.."
FRAMES="${1:-300}"
DIR="${TMPDIR:-/tmp}/mcpt_frame_loop"
mkdir -p gpurun_out
python3 - "$DIR" <<'PY'
import sys
sys.path.insert(0, ".")
import __graft_entry__ as ge
pkg = ge.load_package()
print(pkg.scenes.cornell_box(800, 800).write(sys.argv[1]))
PY
EXE=monte-carlo-path-tracer_amd/csrc/frame_loop
for D in 8 0; do "$EXE" "$DIR/cornell-box.obj" "$FRAMES" "$D"; done
if [ "${2:-}" = trace ]; then
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/frame_trace" -- "$OLDPWD/$EXE" "$DIR/cornell-box.obj" 12 8 4 ) > gpurun_out/frame_trace.log 2>&1 || tail -5 gpurun_out/frame_trace.log
  find gpurun_out/frame_trace -name "*kernel_trace.csv" | head -1
fi
