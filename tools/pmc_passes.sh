#!/usr/bin/env bash
# Developer tool (GPU box): collect rocprofv3 PMC counters for the render kernel in separate passes
# (counter-only runs, no --kernel-trace / --stats mixed in, as the pool requires).
# usage: tools/pmc_passes.sh <outdir-under-gpurun_out> [spp] [lib.so]
set -uo pipefail
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; SPP="${2:-64}"; LIB="${3:-}"; PASSES="${4:-1 2 3 4 5 6}"
[ -n "$LIB" ] && export MCPT_LIB_PATH="$GRAFT_REPO_ROOT/$LIB"
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"
P2="SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT"
P3="TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
P6="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  i=$((i+1))
  case " $PASSES " in *" $i "*) ;; *) continue;; esac
  timeout -k 10 240 rocprofv3 --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 "$GRAFT_REPO_ROOT/tools/perf_probe.py" "$SPP" cornell-box 1 > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/pass$i.log"; exit 1; }
done
python3 "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$OUT" | tee "$OUT/summary.txt"
