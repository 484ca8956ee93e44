#!/bin/bash
# Drop-in per-frame mode through the C++ host path: the reference's loop (one sample per pixel per call), S-cornell 800x800 depth 8.
# usage (GPU box): tools/frame_mode_cli.sh [frames]
set -e
cd "$(dirname "$0")/.."
FRAMES="${1:-200}"
DIR="${TMPDIR:-/tmp}/mcpt_frame_mode"
python3 - "$DIR" <<'PY'
import sys
sys.path.insert(0, ".")
import __graft_entry__ as ge
pkg = ge.load_package()
print(pkg.scenes.cornell_box(800, 800).write(sys.argv[1]))
PY
CLI=monte-carlo-path-tracer_amd/csrc/mcpt_cli
for B in 1 4 16; do
  "$CLI" "$DIR/cornell-box.obj" --spp "$FRAMES" --batch "$B" --depth 8 --out "$DIR/img" > "$DIR/log_$B.txt"
  awk -v b="$B" '/frame cost/ { n++; if (n > 3) { s += $NF + 0; k++ } } END { printf "batch %2d: %.3f ms per call, %.3f ms per sample (%d calls after 3 warm-up)\n", b, 1e3 * s / k, 1e3 * s / k / b, k }' "$DIR/log_$B.txt"
  tail -2 "$DIR/log_$B.txt" | head -1
done
