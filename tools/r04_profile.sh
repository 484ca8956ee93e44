#!/usr/bin/env bash
# Developer tool (GPU box): the evidence bundle for one BASELINE.json configuration -- bench line, rocprofv3 kernel stats of the same bench
# command, PMC passes (SQ / TCC / HBM traffic; counter-only runs, one counter group per run, as the pool requires).
# usage: tools/r04_profile.sh <c2|c5|...> <pmc_spp> [steps]     -> gpurun_out/r04_prof_<cfg>/ ; then tools/update_profiles_r04.py <cfg>
set -uo pipefail
CFG="$1"; SPP="$2"; STEPS="${3:-3}"
OUT="$GRAFT_REPO_ROOT/gpurun_out/r04_prof_$CFG"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python3 bench.py --config "$CFG" --steps "$STEPS" > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
cat "$OUT/bench.json" | cut -c1-400
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --config "$CFG" --steps "$STEPS" --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err" ) || { tail -20 "$OUT/rocprof.err"; exit 1; }
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/stats" -name "*kernel_trace.csv" -delete      # large
head -4 "$OUT/kernel_stats.csv"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"
P2="SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT"
P3="TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
i=0; mkdir -p "$OUT/pmc"
cd /tmp; export TMPDIR=/tmp
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d "$OUT/pmc/pass$i" -- python3 "$GRAFT_REPO_ROOT/tools/pmc_workload.py" "$CFG" "$SPP" "$OUT/pmc/workload$i.json" > "$OUT/pmc/pass$i.log" 2>&1 || { echo "pmc pass $i failed"; tail -5 "$OUT/pmc/pass$i.log"; exit 1; }
done
python3 "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$OUT/pmc" | tee "$OUT/pmc/summary.txt" | grep -E "====|derived|wait_any|lane util|L2 hit|FETCH|WRITE"
