#!/usr/bin/env bash
# Developer tool (GPU box): the evidence behind DESIGN section 6's analysis of the reference's own loop (render(scene); getPixelsColor(); per frame) -> gpurun_out/r04_frame_loop.txt
set -uo pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
OUT=gpurun_out/r04_frame_loop.txt
{
echo "# 1. tools/frame_loop.cpp through the classes of host/ (S-cornell 800x800; depth 8, then the reference's unbounded depth)"
tools/frame_loop.sh 300 trace 2>&1 | grep -v "amdgpu.ids\|^\[Model\]\|^/tmp"
echo
echo "# 2. tools/frame_mode_probe.py: mcpt_render calls of 1 .. 64 samples (render + sync per call)"
timeout -k 10 120 python3 tools/frame_mode_probe.py --calls 60 2>&1 | grep -v amdgpu.ids
echo
echo "# 3. tools/frame_pipeline_probe.py: K contexts in rotation on one GPU, nobody waits between calls (K frames in flight)"
timeout -k 10 200 python3 tools/frame_pipeline_probe.py 2>&1 | grep -v amdgpu.ids
echo
echo "# 4. tools/sched_stats.py (-DWF_SCHED_STATS build): a one-sample job, then a 64-spp job of the same scene"
for S in 1 64; do MCPT_LIB_PATH=$GRAFT_REPO_ROOT/monte-carlo-path-tracer_amd/csrc/build/libmcpt_hip_stats.so timeout -k 10 120 python3 tools/sched_stats.py $S c2 2>&1 | grep -v amdgpu.ids; done
echo
echo "# 5. kernel trace of the last frames of (1) (rocprofv3 --kernel-trace; start and duration in us, queue = sub-pipeline stream)"
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/frame_trace/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-62:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-28s q=%s start %9.1f dur %7.1f" % (r["Kernel_Name"][:28], r["Queue_Id"], s / 1e3, (e - s) / 1e3))
PY
} > $OUT 2>&1
head -12 $OUT | cut -c1-300
