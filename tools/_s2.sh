#!/usr/bin/env bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 240 tools/build/uarch_probe2 > gpurun_out/r04_uarch_probe2.txt 2>&1 || echo "uarch probe2 failed"
tail -3 gpurun_out/r04_uarch_probe2.txt
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/calib" -- "$GRAFT_REPO_ROOT/tools/build/uarch_probe2" calib ) > gpurun_out/r04_calib.log 2>&1 || echo calib failed
tail -2 gpurun_out/r04_calib.log
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/calib/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:40], r["Counter_Name"])] += float(r["Counter_Value"])
    for k, v in sorted(acc.items()): print("calib", k, v)
PY
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "miss_lanes or facade or frame_by_frame or sample_split or stream_ordered or tonemap or clone or random_walk or two_threads" > gpurun_out/r04_pytest_b.log 2>&1; tail -5 gpurun_out/r04_pytest_b.log
timeout -k 10 300 tools/frame_loop.sh 300 trace > gpurun_out/r04_frame_loop.log 2>&1; cat gpurun_out/r04_frame_loop.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04_pytest_c.log 2>&1; tail -5 gpurun_out/r04_pytest_c.log
tools/ab.sh 512 cornell-box default > gpurun_out/r04_ab_sel.log 2>&1; MCPT_DEPTH=16 tools/ab.sh 32 bathroom:420 default >> gpurun_out/r04_ab_sel.log 2>&1; tools/ab.sh 64 bathroom:160 default >> gpurun_out/r04_ab_sel.log 2>&1; cat gpurun_out/r04_ab_sel.log
