#!/usr/bin/env bash
# Developer tool (GPU box): the evidence bundle of a round -- GPU tests, bench line, rocprofv3 kernel stats of the same bench command,
# PMC passes (SQ / TCC / HBM traffic) at the bench's 1024 spp.  usage: tools/final_profile.sh <tag>   -> gpurun_out/<tag>/
set -uo pipefail
TAG="${1:-final}"; OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1 || { tail -20 "$OUT/pytest_gpu.log"; exit 1; }
tail -1 "$OUT/pytest_gpu.log"
timeout -k 10 600 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
cat "$OUT/bench.json"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err" ) || { tail -20 "$OUT/rocprof.err"; exit 1; }
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/stats" -name "*kernel_trace.csv" -delete      # large
head -5 "$OUT/kernel_stats.csv"
bash tools/pmc_passes.sh "$TAG/pmc" 1024 "" "1 2 3 4 5" > "$OUT/pmc.log" 2>&1 || { tail -20 "$OUT/pmc.log"; exit 1; }
tail -3 "$OUT/pmc.log"
