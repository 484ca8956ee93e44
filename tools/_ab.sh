set -e
for lib in "" mw4; do
 if [ -n "$lib" ]; then export MCPT_LIB_PATH=monte-carlo-path-tracer_amd/csrc/build/libmcpt_hip_$lib.so; else unset MCPT_LIB_PATH; fi
 for g in 192 256 320; do MCPT_WF_GRID=$g timeout -k 10 200 python tools/perf_probe.py 1024 cornell-box 2 2>&1 | grep -E "Mray" | cut -c100-200 | sed "s/^/lib=$lib grid=$g /"; done
done
