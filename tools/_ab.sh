set -e
for lib in "" top512 top1024 top1536; do
 if [ -n "$lib" ]; then export MCPT_LIB_PATH=monte-carlo-path-tracer_amd/csrc/build/libmcpt_hip_$lib.so; else unset MCPT_LIB_PATH; fi
 echo "lib=$lib"
 timeout -k 10 200 python tools/perf_probe.py 1024 cornell-box 3 2>&1 | grep -E "Mray" | cut -c40-200
 timeout -k 10 200 python tools/perf_probe.py 64 bathroom:64 2 2>&1 | grep -E "Mray" | cut -c40-200
done
