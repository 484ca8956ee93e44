set -e
for fl in 0 8; do
 MCPT_FLAGS=$fl timeout -k 10 200 python tools/perf_probe.py 1024 cornell-box 2 2>&1 | grep Mray | cut -c40-220
 MCPT_FLAGS=$((fl+4)) timeout -k 10 200 python tools/perf_probe.py 64 cornell-box 1 2>&1 | grep box/ray
 MCPT_FLAGS=$fl timeout -k 10 200 python tools/perf_probe.py 32 bathroom:160 2 2>&1 | grep Mray | cut -c40-220
 MCPT_FLAGS=$((fl+4)) timeout -k 10 200 python tools/perf_probe.py 8 bathroom:160 1 2>&1 | grep box/ray
done
MCPT_FLAGS=8 timeout -k 10 400 python tools/big_config_probe.py 16 2>&1 | grep -v amdgpu
