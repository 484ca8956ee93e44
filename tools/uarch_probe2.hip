// Developer tool (GPU box), second edition of tools/uarch_probe.hip: what one wave64 instruction of each kind the trace / shade kernels use
// costs a SIMD, by WALL time (hipEvents) with W waves per SIMD all running the same stream of independent instructions, relative to v_mov_b32;
// and the latency / throughput of per-lane 16-B gathers (the trace kernel's node and triangle fetches) with no integer division in the loop.
//   hipcc --offload-arch=gfx950 -O3 tools/uarch_probe2.hip -o tools/build/uarch_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// eight independent chains on v[a0..a7]; every register an instruction WRITES is an in/out operand (%8 u, %9 s, %10 m64, %11 w64, %12 d0, %13 d2):
// the first edition declared them inputs, the compiler kept threadIdx.x in w64's register pair, v_mad_u64_u32 overwrote it and the final store faulted
#define REP8(T) T(0) "\n" T(1) "\n" T(2) "\n" T(3) "\n" T(4) "\n" T(5) "\n" T(6) "\n" T(7)
#define KINDS(X) \
    X(MOV,       "v_mov_b32 %0, %14",                      "v_mov_b32 %1, %14", "v_mov_b32 %2, %14", "v_mov_b32 %3, %14", "v_mov_b32 %4, %14", "v_mov_b32 %5, %14", "v_mov_b32 %6, %14", "v_mov_b32 %7, %14") \
    X(ADD_F32,   "v_add_f32 %0, %0, %14",                  "v_add_f32 %1, %1, %14", "v_add_f32 %2, %2, %14", "v_add_f32 %3, %3, %14", "v_add_f32 %4, %4, %14", "v_add_f32 %5, %5, %14", "v_add_f32 %6, %6, %14", "v_add_f32 %7, %7, %14") \
    X(MUL_F32,   "v_mul_f32 %0, %0, %14",                  "v_mul_f32 %1, %1, %14", "v_mul_f32 %2, %2, %14", "v_mul_f32 %3, %3, %14", "v_mul_f32 %4, %4, %14", "v_mul_f32 %5, %5, %14", "v_mul_f32 %6, %6, %14", "v_mul_f32 %7, %7, %14") \
    X(FMAC_F32,  "v_fmac_f32 %0, %14, %15",                 "v_fmac_f32 %1, %14, %15", "v_fmac_f32 %2, %14, %15", "v_fmac_f32 %3, %14, %15", "v_fmac_f32 %4, %14, %15", "v_fmac_f32 %5, %14, %15", "v_fmac_f32 %6, %14, %15", "v_fmac_f32 %7, %14, %15") \
    X(FMA_F32,   "v_fma_f32 %0, %0, %14, %15",              "v_fma_f32 %1, %1, %14, %15", "v_fma_f32 %2, %2, %14, %15", "v_fma_f32 %3, %3, %14, %15", "v_fma_f32 %4, %4, %14, %15", "v_fma_f32 %5, %5, %14, %15", "v_fma_f32 %6, %6, %14, %15", "v_fma_f32 %7, %7, %14, %15") \
    X(FMA_MIX,   "v_fma_mix_f32 %0, %8, %0, %15 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %1, %8, %1, %15 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %2, %8, %2, %15 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %3, %8, %3, %15 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %4, %8, %4, %15 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %5, %8, %5, %15 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %6, %8, %6, %15 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %7, %8, %7, %15 op_sel_hi:[1,0,0]") \
    X(MAX_F32,   "v_max_f32 %0, %0, %14",                  "v_max_f32 %1, %1, %14", "v_max_f32 %2, %2, %14", "v_max_f32 %3, %3, %14", "v_max_f32 %4, %4, %14", "v_max_f32 %5, %5, %14", "v_max_f32 %6, %6, %14", "v_max_f32 %7, %7, %14") \
    X(MAX3_F32,  "v_max3_f32 %0, %0, %14, %15",             "v_max3_f32 %1, %1, %14, %15", "v_max3_f32 %2, %2, %14, %15", "v_max3_f32 %3, %3, %14, %15", "v_max3_f32 %4, %4, %14, %15", "v_max3_f32 %5, %5, %14, %15", "v_max3_f32 %6, %6, %14, %15", "v_max3_f32 %7, %7, %14, %15") \
    X(PERM,      "v_perm_b32 %0, %8, %0, %9",           "v_perm_b32 %1, %8, %1, %9", "v_perm_b32 %2, %8, %2, %9", "v_perm_b32 %3, %8, %3, %9", "v_perm_b32 %4, %8, %4, %9", "v_perm_b32 %5, %8, %5, %9", "v_perm_b32 %6, %8, %6, %9", "v_perm_b32 %7, %8, %7, %9") \
    X(CVT_UB,    "v_cvt_f32_ubyte1 %0, %8",              "v_cvt_f32_ubyte1 %1, %8", "v_cvt_f32_ubyte1 %2, %8", "v_cvt_f32_ubyte1 %3, %8", "v_cvt_f32_ubyte1 %4, %8", "v_cvt_f32_ubyte1 %5, %8", "v_cvt_f32_ubyte1 %6, %8", "v_cvt_f32_ubyte1 %7, %8") \
    X(CND_VCC,   "v_cndmask_b32 %0, %0, %14, vcc",         "v_cndmask_b32 %1, %1, %14, vcc", "v_cndmask_b32 %2, %2, %14, vcc", "v_cndmask_b32 %3, %3, %14, vcc", "v_cndmask_b32 %4, %4, %14, vcc", "v_cndmask_b32 %5, %5, %14, vcc", "v_cndmask_b32 %6, %6, %14, vcc", "v_cndmask_b32 %7, %7, %14, vcc") \
    X(CND_SGPR,  "v_cndmask_b32 %0, %0, %14, %10",         "v_cndmask_b32 %1, %1, %14, %10", "v_cndmask_b32 %2, %2, %14, %10", "v_cndmask_b32 %3, %3, %14, %10", "v_cndmask_b32 %4, %4, %14, %10", "v_cndmask_b32 %5, %5, %14, %10", "v_cndmask_b32 %6, %6, %14, %10", "v_cndmask_b32 %7, %7, %14, %10") \
    X(CND_FRESH, "v_cndmask_b32 %0, %14, %15, vcc",         "v_cndmask_b32 %1, %14, %15, vcc", "v_cndmask_b32 %2, %14, %15, vcc", "v_cndmask_b32 %3, %14, %15, vcc", "v_cndmask_b32 %4, %14, %15, vcc", "v_cndmask_b32 %5, %14, %15, vcc", "v_cndmask_b32 %6, %14, %15, vcc", "v_cndmask_b32 %7, %14, %15, vcc") \
    X(BFI,       "v_bfi_b32 %0, %8, %0, %14",             "v_bfi_b32 %1, %8, %1, %14", "v_bfi_b32 %2, %8, %2, %14", "v_bfi_b32 %3, %8, %3, %14", "v_bfi_b32 %4, %8, %4, %14", "v_bfi_b32 %5, %8, %5, %14", "v_bfi_b32 %6, %8, %6, %14", "v_bfi_b32 %7, %8, %7, %14") \
    X(CMP_E32,   "v_cmp_le_f32 vcc, %0, %14",              "v_cmp_le_f32 vcc, %1, %14", "v_cmp_le_f32 vcc, %2, %14", "v_cmp_le_f32 vcc, %3, %14", "v_cmp_le_f32 vcc, %4, %14", "v_cmp_le_f32 vcc, %5, %14", "v_cmp_le_f32 vcc, %6, %14", "v_cmp_le_f32 vcc, %7, %14") \
    X(CMP_E64,   "v_cmp_le_f32 %10, %0, %14",              "v_cmp_le_f32 %10, %1, %14", "v_cmp_le_f32 %10, %2, %14", "v_cmp_le_f32 %10, %3, %14", "v_cmp_le_f32 %10, %4, %14", "v_cmp_le_f32 %10, %5, %14", "v_cmp_le_f32 %10, %6, %14", "v_cmp_le_f32 %10, %7, %14") \
    X(ADDC,      "v_addc_co_u32 %0, vcc, %0, %0, vcc",    "v_addc_co_u32 %1, vcc, %1, %1, vcc", "v_addc_co_u32 %2, vcc, %2, %2, vcc", "v_addc_co_u32 %3, vcc, %3, %3, vcc", "v_addc_co_u32 %4, vcc, %4, %4, vcc", "v_addc_co_u32 %5, vcc, %5, %5, vcc", "v_addc_co_u32 %6, vcc, %6, %6, vcc", "v_addc_co_u32 %7, vcc, %7, %7, vcc") \
    X(ADD_U32,   "v_add_u32 %0, %0, %8",                 "v_add_u32 %1, %1, %8", "v_add_u32 %2, %2, %8", "v_add_u32 %3, %3, %8", "v_add_u32 %4, %4, %8", "v_add_u32 %5, %5, %8", "v_add_u32 %6, %6, %8", "v_add_u32 %7, %7, %8") \
    X(AND,       "v_and_b32 %0, %0, %8",                 "v_and_b32 %1, %1, %8", "v_and_b32 %2, %2, %8", "v_and_b32 %3, %3, %8", "v_and_b32 %4, %4, %8", "v_and_b32 %5, %5, %8", "v_and_b32 %6, %6, %8", "v_and_b32 %7, %7, %8") \
    X(LSHL,      "v_lshlrev_b32 %0, 3, %0",               "v_lshlrev_b32 %1, 3, %1", "v_lshlrev_b32 %2, 3, %2", "v_lshlrev_b32 %3, 3, %3", "v_lshlrev_b32 %4, 3, %4", "v_lshlrev_b32 %5, 3, %5", "v_lshlrev_b32 %6, 3, %6", "v_lshlrev_b32 %7, 3, %7") \
    X(AND_OR,    "v_and_or_b32 %0, %0, %8, %14",          "v_and_or_b32 %1, %1, %8, %14", "v_and_or_b32 %2, %2, %8, %14", "v_and_or_b32 %3, %3, %8, %14", "v_and_or_b32 %4, %4, %8, %14", "v_and_or_b32 %5, %5, %8, %14", "v_and_or_b32 %6, %6, %8, %14", "v_and_or_b32 %7, %7, %8, %14") \
    X(LSHL_OR,   "v_lshl_or_b32 %0, %0, 3, %8",          "v_lshl_or_b32 %1, %1, 3, %8", "v_lshl_or_b32 %2, %2, 3, %8", "v_lshl_or_b32 %3, %3, 3, %8", "v_lshl_or_b32 %4, %4, 3, %8", "v_lshl_or_b32 %5, %5, 3, %8", "v_lshl_or_b32 %6, %6, 3, %8", "v_lshl_or_b32 %7, %7, 3, %8") \
    X(BFE,       "v_bfe_u32 %0, %0, 3, 8",                "v_bfe_u32 %1, %1, 3, 8", "v_bfe_u32 %2, %2, 3, 8", "v_bfe_u32 %3, %3, 3, 8", "v_bfe_u32 %4, %4, 3, 8", "v_bfe_u32 %5, %5, 3, 8", "v_bfe_u32 %6, %6, 3, 8", "v_bfe_u32 %7, %7, 3, 8") \
    X(BCNT,      "v_bcnt_u32_b32 %0, %0, %8",            "v_bcnt_u32_b32 %1, %1, %8", "v_bcnt_u32_b32 %2, %2, %8", "v_bcnt_u32_b32 %3, %3, %8", "v_bcnt_u32_b32 %4, %4, %8", "v_bcnt_u32_b32 %5, %5, %8", "v_bcnt_u32_b32 %6, %6, %8", "v_bcnt_u32_b32 %7, %7, %8") \
    X(FFBL,      "v_ffbl_b32 %0, %0",                     "v_ffbl_b32 %1, %1", "v_ffbl_b32 %2, %2", "v_ffbl_b32 %3, %3", "v_ffbl_b32 %4, %4", "v_ffbl_b32 %5, %5", "v_ffbl_b32 %6, %6", "v_ffbl_b32 %7, %7") \
    X(MAD_U64,   "v_mad_u64_u32 %11, vcc, %8, %8, %11", "v_mad_u64_u32 %11, vcc, %8, %8, %11", "v_mad_u64_u32 %11, vcc, %8, %8, %11", "v_mad_u64_u32 %11, vcc, %8, %8, %11", "v_mad_u64_u32 %11, vcc, %8, %8, %11", "v_mad_u64_u32 %11, vcc, %8, %8, %11", "v_mad_u64_u32 %11, vcc, %8, %8, %11", "v_mad_u64_u32 %11, vcc, %8, %8, %11") \
    X(PK_FMA,    "v_pk_fma_f32 %12, %12, %16, %16",       "v_pk_fma_f32 %13, %13, %16, %16", "v_pk_fma_f32 %12, %12, %16, %16", "v_pk_fma_f32 %13, %13, %16, %16", "v_pk_fma_f32 %12, %12, %16, %16", "v_pk_fma_f32 %13, %13, %16, %16", "v_pk_fma_f32 %12, %12, %16, %16", "v_pk_fma_f32 %13, %13, %16, %16") \
    X(FMA_F64,   "v_fma_f64 %12, %12, %16, %16",          "v_fma_f64 %13, %13, %16, %16", "v_fma_f64 %12, %12, %16, %16", "v_fma_f64 %13, %13, %16, %16", "v_fma_f64 %12, %12, %16, %16", "v_fma_f64 %13, %13, %16, %16", "v_fma_f64 %12, %12, %16, %16", "v_fma_f64 %13, %13, %16, %16") \
    X(RCP,       "v_rcp_f32 %0, %0",                      "v_rcp_f32 %1, %1", "v_rcp_f32 %2, %2", "v_rcp_f32 %3, %3", "v_rcp_f32 %4, %4", "v_rcp_f32 %5, %5", "v_rcp_f32 %6, %6", "v_rcp_f32 %7, %7") \
    X(MBCNT,     "v_mbcnt_lo_u32_b32 %0, %8, %0",        "v_mbcnt_lo_u32_b32 %1, %8, %1", "v_mbcnt_lo_u32_b32 %2, %8, %2", "v_mbcnt_lo_u32_b32 %3, %8, %3", "v_mbcnt_lo_u32_b32 %4, %8, %4", "v_mbcnt_lo_u32_b32 %5, %8, %5", "v_mbcnt_lo_u32_b32 %6, %8, %6", "v_mbcnt_lo_u32_b32 %7, %8, %7") \
    X(CND_E64VCC,"v_cndmask_b32_e64 %0, %0, %14, vcc",     "v_cndmask_b32_e64 %1, %1, %14, vcc", "v_cndmask_b32_e64 %2, %2, %14, vcc", "v_cndmask_b32_e64 %3, %3, %14, vcc", "v_cndmask_b32_e64 %4, %4, %14, vcc", "v_cndmask_b32_e64 %5, %5, %14, vcc", "v_cndmask_b32_e64 %6, %6, %14, vcc", "v_cndmask_b32_e64 %7, %7, %14, vcc") \
    X(CMP_CND,   "v_cmp_le_f32 vcc, %0, %14\n v_cndmask_b32 %0, %0, %15, vcc", "v_cmp_le_f32 vcc, %1, %14\n v_cndmask_b32 %1, %1, %15, vcc", "v_cmp_le_f32 vcc, %2, %14\n v_cndmask_b32 %2, %2, %15, vcc", "v_cmp_le_f32 vcc, %3, %14\n v_cndmask_b32 %3, %3, %15, vcc", "v_cmp_le_f32 vcc, %4, %14\n v_cndmask_b32 %4, %4, %15, vcc", "v_cmp_le_f32 vcc, %5, %14\n v_cndmask_b32 %5, %5, %15, vcc", "v_cmp_le_f32 vcc, %6, %14\n v_cndmask_b32 %6, %6, %15, vcc", "v_cmp_le_f32 vcc, %7, %14\n v_cndmask_b32 %7, %7, %15, vcc") \
    X(CMP_CND64, "v_cmp_le_f32 %10, %0, %14\n v_cndmask_b32 %0, %0, %15, %10", "v_cmp_le_f32 %10, %1, %14\n v_cndmask_b32 %1, %1, %15, %10", "v_cmp_le_f32 %10, %2, %14\n v_cndmask_b32 %2, %2, %15, %10", "v_cmp_le_f32 %10, %3, %14\n v_cndmask_b32 %3, %3, %15, %10", "v_cmp_le_f32 %10, %4, %14\n v_cndmask_b32 %4, %4, %15, %10", "v_cmp_le_f32 %10, %5, %14\n v_cndmask_b32 %5, %5, %15, %10", "v_cmp_le_f32 %10, %6, %14\n v_cndmask_b32 %6, %6, %15, %10", "v_cmp_le_f32 %10, %7, %14\n v_cndmask_b32 %7, %7, %15, %10") \
    X(SUB_F32,   "v_sub_f32 %0, %0, %14",                  "v_sub_f32 %1, %1, %14", "v_sub_f32 %2, %2, %14", "v_sub_f32 %3, %3, %14", "v_sub_f32 %4, %4, %14", "v_sub_f32 %5, %5, %14", "v_sub_f32 %6, %6, %14", "v_sub_f32 %7, %7, %14") \
    X(MIN_F32,   "v_min_f32 %0, %0, %14",                  "v_min_f32 %1, %1, %14", "v_min_f32 %2, %2, %14", "v_min_f32 %3, %3, %14", "v_min_f32 %4, %4, %14", "v_min_f32 %5, %5, %14", "v_min_f32 %6, %6, %14", "v_min_f32 %7, %7, %14") \
    X(MED3_F32,  "v_med3_f32 %0, %0, %14, %15",            "v_med3_f32 %1, %1, %14, %15", "v_med3_f32 %2, %2, %14, %15", "v_med3_f32 %3, %3, %14, %15", "v_med3_f32 %4, %4, %14, %15", "v_med3_f32 %5, %5, %14, %15", "v_med3_f32 %6, %6, %14, %15", "v_med3_f32 %7, %7, %14, %15") \
    X(ALIGNBIT,  "v_alignbit_b32 %0, %0, %8, 31",          "v_alignbit_b32 %1, %1, %8, 31", "v_alignbit_b32 %2, %2, %8, 31", "v_alignbit_b32 %3, %3, %8, 31", "v_alignbit_b32 %4, %4, %8, 31", "v_alignbit_b32 %5, %5, %8, 31", "v_alignbit_b32 %6, %6, %8, 31", "v_alignbit_b32 %7, %7, %8, 31") \
    X(XOR,       "v_xor_b32 %0, %0, %8",                   "v_xor_b32 %1, %1, %8", "v_xor_b32 %2, %2, %8", "v_xor_b32 %3, %3, %8", "v_xor_b32 %4, %4, %8", "v_xor_b32 %5, %5, %8", "v_xor_b32 %6, %6, %8", "v_xor_b32 %7, %7, %8") \
    X(OR,        "v_or_b32 %0, %0, %8",                    "v_or_b32 %1, %1, %8", "v_or_b32 %2, %2, %8", "v_or_b32 %3, %3, %8", "v_or_b32 %4, %4, %8", "v_or_b32 %5, %5, %8", "v_or_b32 %6, %6, %8", "v_or_b32 %7, %7, %8") \
    X(SUB_U32,   "v_sub_u32 %0, %0, %8",                   "v_sub_u32 %1, %1, %8", "v_sub_u32 %2, %2, %8", "v_sub_u32 %3, %3, %8", "v_sub_u32 %4, %4, %8", "v_sub_u32 %5, %5, %8", "v_sub_u32 %6, %6, %8", "v_sub_u32 %7, %7, %8") \
    X(LSHRREV,   "v_lshrrev_b32 %0, 3, %0",                "v_lshrrev_b32 %1, 3, %1", "v_lshrrev_b32 %2, 3, %2", "v_lshrrev_b32 %3, 3, %3", "v_lshrrev_b32 %4, 3, %4", "v_lshrrev_b32 %5, 3, %5", "v_lshrrev_b32 %6, 3, %6", "v_lshrrev_b32 %7, 3, %7") \
    X(LSHL_ADD,  "v_lshl_add_u32 %0, %0, 3, %8",           "v_lshl_add_u32 %1, %1, 3, %8", "v_lshl_add_u32 %2, %2, 3, %8", "v_lshl_add_u32 %3, %3, 3, %8", "v_lshl_add_u32 %4, %4, 3, %8", "v_lshl_add_u32 %5, %5, 3, %8", "v_lshl_add_u32 %6, %6, 3, %8", "v_lshl_add_u32 %7, %7, 3, %8") \
    X(ADD3,      "v_add3_u32 %0, %0, %8, %8",              "v_add3_u32 %1, %1, %8, %8", "v_add3_u32 %2, %2, %8, %8", "v_add3_u32 %3, %3, %8, %8", "v_add3_u32 %4, %4, %8, %8", "v_add3_u32 %5, %5, %8, %8", "v_add3_u32 %6, %6, %8, %8", "v_add3_u32 %7, %7, %8, %8") \
    X(MUL_LO,    "v_mul_lo_u32 %0, %0, %8",                "v_mul_lo_u32 %1, %1, %8", "v_mul_lo_u32 %2, %2, %8", "v_mul_lo_u32 %3, %3, %8", "v_mul_lo_u32 %4, %4, %8", "v_mul_lo_u32 %5, %5, %8", "v_mul_lo_u32 %6, %6, %8", "v_mul_lo_u32 %7, %7, %8") \
    X(MAD_U24,   "v_mad_u32_u24 %0, %0, %8, %8",           "v_mad_u32_u24 %1, %1, %8, %8", "v_mad_u32_u24 %2, %2, %8, %8", "v_mad_u32_u24 %3, %3, %8, %8", "v_mad_u32_u24 %4, %4, %8, %8", "v_mad_u32_u24 %5, %5, %8, %8", "v_mad_u32_u24 %6, %6, %8, %8", "v_mad_u32_u24 %7, %7, %8, %8") \
    X(CVT_F32_U, "v_cvt_f32_u32 %0, %8",                   "v_cvt_f32_u32 %1, %8", "v_cvt_f32_u32 %2, %8", "v_cvt_f32_u32 %3, %8", "v_cvt_f32_u32 %4, %8", "v_cvt_f32_u32 %5, %8", "v_cvt_f32_u32 %6, %8", "v_cvt_f32_u32 %7, %8") \
    X(PK_MUL,    "v_pk_mul_f32 %12, %12, %16",             "v_pk_mul_f32 %13, %13, %16", "v_pk_mul_f32 %12, %12, %16", "v_pk_mul_f32 %13, %13, %16", "v_pk_mul_f32 %12, %12, %16", "v_pk_mul_f32 %13, %13, %16", "v_pk_mul_f32 %12, %12, %16", "v_pk_mul_f32 %13, %13, %16") \
    X(EXP,       "v_exp_f32 %0, %0",                       "v_exp_f32 %1, %1", "v_exp_f32 %2, %2", "v_exp_f32 %3, %3", "v_exp_f32 %4, %4", "v_exp_f32 %5, %5", "v_exp_f32 %6, %6", "v_exp_f32 %7, %7") \
    X(SQRT,      "v_sqrt_f32 %0, %0",                      "v_sqrt_f32 %1, %1", "v_sqrt_f32 %2, %2", "v_sqrt_f32 %3, %3", "v_sqrt_f32 %4, %4", "v_sqrt_f32 %5, %5", "v_sqrt_f32 %6, %6", "v_sqrt_f32 %7, %7") \
    X(ADD_F64,   "v_add_f64 %12, %12, %16",                "v_add_f64 %13, %13, %16", "v_add_f64 %12, %12, %16", "v_add_f64 %13, %13, %16", "v_add_f64 %12, %12, %16", "v_add_f64 %13, %13, %16", "v_add_f64 %12, %12, %16", "v_add_f64 %13, %13, %16") \
    X(RCP_F64,   "v_rcp_f64 %12, %12",                     "v_rcp_f64 %13, %13", "v_rcp_f64 %12, %12", "v_rcp_f64 %13, %13", "v_rcp_f64 %12, %12", "v_rcp_f64 %13, %13", "v_rcp_f64 %12, %12", "v_rcp_f64 %13, %13") \
    X(SAND_CND4, "s_and_b64 vcc, %10, exec\n v_cndmask_b32 %0, %0, %14, vcc\n v_cndmask_b32 %1, %1, %14, vcc\n v_cndmask_b32 %2, %2, %14, vcc\n v_cndmask_b32 %3, %3, %14, vcc", "s_and_b64 vcc, %10, exec\n v_cndmask_b32 %4, %4, %14, vcc\n v_cndmask_b32 %5, %5, %14, vcc\n v_cndmask_b32 %6, %6, %14, vcc\n v_cndmask_b32 %7, %7, %14, vcc", "s_nop 0", "s_nop 0", "s_nop 0", "s_nop 0", "s_nop 0", "s_nop 0") \
    X(CMP_CND4,  "v_cmp_le_f32 vcc, %0, %14\n v_cndmask_b32 %0, %0, %14, vcc\n v_cndmask_b32 %1, %1, %14, vcc\n v_cndmask_b32 %2, %2, %14, vcc\n v_cndmask_b32 %3, %3, %14, vcc", "v_cmp_le_f32 vcc, %4, %14\n v_cndmask_b32 %4, %4, %14, vcc\n v_cndmask_b32 %5, %5, %14, vcc\n v_cndmask_b32 %6, %6, %14, vcc\n v_cndmask_b32 %7, %7, %14, vcc", "s_nop 0", "s_nop 0", "s_nop 0", "s_nop 0", "s_nop 0", "s_nop 0") \
    X(SNOP,      "s_nop 0",                                "s_nop 0", "s_nop 0", "s_nop 0", "s_nop 0", "s_nop 0", "s_nop 0", "s_nop 0") \
    X(PERM_VSEL, "v_perm_b32 %0, %8, %0, %14",           "v_perm_b32 %1, %8, %1, %14", "v_perm_b32 %2, %8, %2, %14", "v_perm_b32 %3, %8, %3, %14", "v_perm_b32 %4, %8, %4, %14", "v_perm_b32 %5, %8, %5, %14", "v_perm_b32 %6, %8, %6, %14", "v_perm_b32 %7, %8, %7, %14") \
    X(FMA_MIX3V, "v_fma_mix_f32 %0, %8, %14, %0 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %1, %8, %14, %1 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %2, %8, %14, %2 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %3, %8, %14, %3 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %4, %8, %14, %4 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %5, %8, %14, %5 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %6, %8, %14, %6 op_sel_hi:[1,0,0]", "v_fma_mix_f32 %7, %8, %14, %7 op_sel_hi:[1,0,0]") \
    X(MIN3_F32,  "v_min3_f32 %0, %0, %14, %15",          "v_min3_f32 %1, %1, %14, %15", "v_min3_f32 %2, %2, %14, %15", "v_min3_f32 %3, %3, %14, %15", "v_min3_f32 %4, %4, %14, %15", "v_min3_f32 %5, %5, %14, %15", "v_min3_f32 %6, %6, %14, %15", "v_min3_f32 %7, %7, %14, %15") \
    X(S_ADD,     "s_add_u32 %9, %9, 3",                 "s_add_u32 %9, %9, 3", "s_add_u32 %9, %9, 3", "s_add_u32 %9, %9, 3", "s_add_u32 %9, %9, 3", "s_add_u32 %9, %9, 3", "s_add_u32 %9, %9, 3", "s_add_u32 %9, %9, 3") \
    X(S_AND64,   "s_and_b64 %10, %10, exec",              "s_and_b64 %10, %10, exec", "s_and_b64 %10, %10, exec", "s_and_b64 %10, %10, exec", "s_and_b64 %10, %10, exec", "s_and_b64 %10, %10, exec", "s_and_b64 %10, %10, exec", "s_and_b64 %10, %10, exec")

enum {
#define X(N, ...) K_##N,
    KINDS(X)
#undef X
    K_COUNT
};
static const char* kind_name[] = {
#define X(N, A, ...) A,
    KINDS(X)
#undef X
};

template <int KIND>
__global__ void __launch_bounds__(256) valu_kernel(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    uint32_t u = threadIdx.x * 2654435761u, s = 0x04010400u;
    unsigned long long m64 = 0x5555555555555555ull, w64 = threadIdx.x;
    double d0 = a0, d1 = 1.0000001, d2 = a2;
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a0), "v"(100.0f) : "vcc");
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            switch (KIND) {
#define X(N, A, B, C, D, E, F, G, H) case K_##N: asm volatile(A "\n" B "\n" C "\n" D "\n" E "\n" F "\n" G "\n" H \
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(u), "+s"(s), "+s"(m64), "+v"(w64), "+v"(d0), "+v"(d2) : "v"(b), "v"(c), "v"(d1) : "vcc", "scc"); break;
                KINDS(X)
#undef X
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d2) + (float)w64 + (float)s + (float)m64;
}
typedef void (*valu_fn)(float*, int);
static valu_fn valu_table[] = {
#define X(N, ...) valu_kernel<K_##N>,
    KINDS(X)
#undef X
};

// ---------------------------------------------------------------------------------------------------------------- gathers
// Every lane walks a pseudo-random chain through a table of 2^k 80-B records: index = next bits of an LCG + a data dependence (so that step
// i + 1 cannot start before step i's loads have returned).  LOADS = 16-B loads per step from the lane's record (1..6; 6 = wraps to 5 + 1).
template <int LOADS, bool LDS, bool UNIFORM>
__global__ void __launch_bounds__(1024) gather_kernel(const float4* __restrict__ table, uint32_t mask, int iters, float* out) {
    __shared__ float4 s_tab[5 * 256];
    if (LDS) { for (uint32_t i = threadIdx.x; i < 5 * 256; i += blockDim.x) s_tab[i] = table[i]; __syncthreads(); }
    uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        h = h * 1664525u + 1013904223u;
        uint32_t r = (h >> 9) & mask;
        if (UNIFORM) r = __builtin_amdgcn_readfirstlane(r);
        float4 v[6];
        if (LDS) {
#pragma unroll
            for (int k = 0; k < LOADS; k++) v[k] = s_tab[(k % 5) * 256 + (r & 255u)];
        } else {
            const float4* p = (const float4*)((const char*)table + r * 80u);
#pragma unroll
            for (int k = 0; k < LOADS; k++) v[k] = p[k % 5 + (k >= 5 ? 5 : 0)];
        }
#pragma unroll
        for (int k = 0; k < LOADS; k++) acc += v[k].x;
        h += (uint32_t)(acc == 12345.678f);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
typedef void (*gather_fn)(const float4*, uint32_t, int, float*);

// FETCH_SIZE calibration (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern"): `uarch_probe2 calib` runs (a) a float4 streaming read of 1 GiB and (b) 52 M random 80-B record gathers from a 335-MB table (beyond
// the 256-MiB Infinity Cache) -- each record lies in exactly two 64-B sectors (128 B) and in 1.5 128-B lines on average (192 B) -- as two kernels
// whose names a `rocprofv3 --pmc FETCH_SIZE` pass lists separately.
__global__ void __launch_bounds__(256) calib_stream_kernel(const float4* __restrict__ src, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) acc += src[i].x;
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void __launch_bounds__(1024) calib_gather_kernel(const float4* __restrict__ table, uint32_t mask, int iters, float* out) {
    uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        h = h * 1664525u + 1013904223u;
        const float4* p = (const float4*)((const char*)table + ((h >> 9) & mask) * 80u);
        acc += p[0].x + p[1].x + p[2].x + p[3].x + p[4].x;
        h += (uint32_t)(acc == 12345.678f);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "calib") {
        float* out; CHECK(hipMalloc(&out, sizeof(float) * 1024 * 1024));
        const size_t bytes = size_t(1) << 30;
        float4* big; CHECK(hipMalloc(&big, bytes)); CHECK(hipMemset(big, 0, bytes));
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(calib_stream_kernel, dim3(256 * 8), dim3(256), 0, 0, big, bytes / 16, out);
        CHECK(hipDeviceSynchronize());
        const int iters = 200, grid = 256;
        hipLaunchKernelGGL(calib_gather_kernel, dim3(grid), dim3(1024), 0, 0, big, (1u << 22) - 1u, iters, out);
        CHECK(hipDeviceSynchronize());
        printf("calib: stream kernel read %zu bytes; gather kernel fetched %zu records of 80 B (= %zu B in 64-B sectors, %zu B in 128-B lines on average)\n",
               bytes, size_t(grid) * 1024 * iters, size_t(grid) * 1024 * iters * 128, size_t(grid) * 1024 * iters * 192);
        return 0;
    }
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    float* out; CHECK(hipMalloc(&out, sizeof(float) * 1024 * 1024 * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto time_valu = [&](int k, int w, int iters) -> double {
        const int grid = n_cu * w;
        hipLaunchKernelGGL(valu_table[k], dim3(grid), dim3(256), 0, 0, out, 50);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(valu_table[k], dim3(grid), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        return ms;
    };
    printf("== cost of one wave64 instruction to a SIMD, by wall time: ns per instruction per SIMD (all SIMDs of the chip busy with the same stream), and relative to v_mov_b32\n");
    printf("   (256-thread blocks = one wave per SIMD; W blocks per CU; 4000 x 32 instructions per wave)\n");
    printf("%-58s %9s %9s %9s | rel(W=4) rel(W=8)\n", "instruction", "W=2", "W=4", "W=8");
    double base4 = 0, base8 = 0;
    for (int k = 0; k < K_COUNT; k++) {
        const int iters = 4000;
        double t[3]; int ws[3] = {2, 4, 8};
        for (int j = 0; j < 3; j++) t[j] = time_valu(k, ws[j], iters) * 1e6 / (double(iters) * 32 * ws[j]);     // ns per instruction per SIMD
        if (k == 0) { base4 = t[1]; base8 = t[2]; }
        printf("%-58s %9.3f %9.3f %9.3f | %7.2f %7.2f\n", kind_name[k], t[0], t[1], t[2], t[1] / base4, t[2] / base8);
    }
    if (argc > 1 && std::string(argv[1]) == "valu") return 0;
    // ---- gathers
    std::vector<float> host(size_t(1) << 22);
    for (size_t i = 0; i < host.size(); i++) host[i] = float(i % 977) * 1e-3f;
    float4* table; CHECK(hipMalloc(&table, host.size() * 4 + 4096)); CHECK(hipMemcpy(table, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    static gather_fn g_tab[] = {gather_kernel<1, false, false>, gather_kernel<2, false, false>, gather_kernel<3, false, false>, gather_kernel<4, false, false>, gather_kernel<5, false, false>,
                                gather_kernel<6, false, false>, gather_kernel<5, true, false>, gather_kernel<1, true, false>, gather_kernel<5, false, true>};
    static const char* g_name[] = {"1 x 16 B per lane", "2 x 16 B", "3 x 16 B", "4 x 16 B", "5 x 16 B (one 80-B record)", "6 x 16 B", "LDS 5 x 16 B", "LDS 1 x 16 B", "5 x 16 B, wave-uniform record"};
    static const int g_loads[] = {1, 2, 3, 4, 5, 6, 5, 1, 5};
    printf("\n== dependent gather steps: ns per step (= latency of the step incl. ~12 VALU), lane-loads per ns per CU; 1024-thread blocks\n");
    const uint32_t masks[] = {127, 4095, 65535};            // 10 KB (L1), 320 KB (L2), 5 MB (all L2s)
    for (int bpc = 1; bpc <= 2; bpc++)
        for (uint32_t mask : masks) {
            printf("-- %u records (%.0f KB), %d waves per CU\n", mask + 1, (mask + 1) * 80 / 1024.0, 16 * bpc);
            for (int g = 0; g < 9; g++) {
                const int iters = 3000, grid = n_cu * bpc;
                hipLaunchKernelGGL(g_tab[g], dim3(grid), dim3(1024), 0, 0, table, mask, 30, out);
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(g_tab[g], dim3(grid), dim3(1024), 0, 0, table, mask, iters, out);
                (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
                float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
                printf("   %-34s %8.1f ns per step   %7.3f lane-loads/ns/CU\n", g_name[g], ms * 1e6 / iters, double(g_loads[g]) * 1024 * bpc * iters / (ms * 1e6));
            }
        }
    return 0;
}
