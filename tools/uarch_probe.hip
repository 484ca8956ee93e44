// Developer tool (GPU box): micro-measurements behind DESIGN.md's "what binds the pair" section.
//   hipcc --offload-arch=gfx950 -O3 tools/uarch_probe.hip -o gpurun_out/uarch_probe && gpurun_out/uarch_probe
// (1) VALU issue price: independent instructions of one kind, W waves per SIMD -> cycles per wave64 instruction per SIMD.
// (2) Vector-memory address rate: per-lane 16-B gathers from an L1- / L2-resident table of 80-B records (the trace kernel's node fetch)
//     against the same reads from LDS -> lane-loads per cycle per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

enum { K_FMA, K_FMA_MIX, K_PERM, K_MAX3, K_CNDMASK, K_CMP_ADDC, K_PK_FMA, K_FMA64, K_RCP, K_CVT_UB, K_MOV, K_ADD_U32, K_SALU, K_COUNT };
static const char* kind_name[K_COUNT] = {"v_fma_f32", "v_fma_mix_f32", "v_perm_b32", "v_max3_f32", "v_cndmask_b32", "v_cmp_le+v_addc", "v_pk_fma_f32 (2 fma)", "v_fma_f64", "v_rcp_f32",
                                          "v_cvt_f32_ubyte1", "v_mov_b32", "v_add_u32", "s_add_u32"};

template <int KIND>
__global__ void __launch_bounds__(256) valu_kernel(float* out, int iters, unsigned long long* cyc) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    uint32_t u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7, m = 0;
    uint32_t s0 = blockIdx.x, s1 = 3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {      // 32 instructions of the kind per inner iteration
            if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            if (KIND == K_FMA_MIX) asm volatile("v_fma_mix_f32 %0, %8, %0, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %1, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %8, %2, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, %3, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %8, %4, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, %5, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %8, %6, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, %7, %9 op_sel_hi:[1,0,0]"
                                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(u0), "v"(c));
            if (KIND == K_PERM) asm volatile("v_perm_b32 %0, %4, %0, %5\n v_perm_b32 %1, %4, %1, %5\n v_perm_b32 %2, %4, %2, %5\n v_perm_b32 %3, %4, %3, %5\n v_perm_b32 %0, %4, %0, %5\n v_perm_b32 %1, %4, %1, %5\n v_perm_b32 %2, %4, %2, %5\n v_perm_b32 %3, %4, %3, %5"
                                            : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(m), "s"(0x04010400u));
            if (KIND == K_MAX3) asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9"
                                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
            if (KIND == K_CMP_ADDC) asm volatile("v_cmp_le_f32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_cmp_le_f32 vcc, %2, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_cmp_le_f32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_cmp_le_f32 vcc, %2, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc"
                                            : "+v"(m) : "v"(a0), "v"(b) : "vcc");
            if (KIND == K_PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                                            : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
            if (KIND == K_FMA64) asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                                            : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)b), "v"((double)c));
            if (KIND == K_RCP) asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            if (KIND == K_CVT_UB) asm volatile("v_cvt_f32_ubyte1 %0, %8\n v_cvt_f32_ubyte1 %1, %8\n v_cvt_f32_ubyte1 %2, %8\n v_cvt_f32_ubyte1 %3, %8\n v_cvt_f32_ubyte1 %4, %8\n v_cvt_f32_ubyte1 %5, %8\n v_cvt_f32_ubyte1 %6, %8\n v_cvt_f32_ubyte1 %7, %8"
                                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(u0));
            if (KIND == K_MOV) asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8"
                                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            if (KIND == K_ADD_U32) asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                                            : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(m));
            if (KIND == K_SALU) asm volatile("s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %1"
                                            : "+s"(s0) : "s"(s1) : "scc");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(u0 + u1 + u2 + u3 + m + s0);
}

typedef void (*valu_fn)(float*, int, unsigned long long*);
static valu_fn valu_table[K_COUNT] = {valu_kernel<K_FMA>, valu_kernel<K_FMA_MIX>, valu_kernel<K_PERM>, valu_kernel<K_MAX3>, valu_kernel<K_CNDMASK>, valu_kernel<K_CMP_ADDC>, valu_kernel<K_PK_FMA>,
                                      valu_kernel<K_FMA64>, valu_kernel<K_RCP>, valu_kernel<K_CVT_UB>, valu_kernel<K_MOV>, valu_kernel<K_ADD_U32>, valu_kernel<K_SALU>};

// ---------------------------------------------------------------------------------------------------------------- gathers
// MODE 0: every lane its own 80-B record (five 16-B loads); 1: wave-uniform record; 2: one 16-B load per lane (divergent); 3: one 4-B load per lane (divergent);
// 4: records in LDS (five ds_read_b128, divergent); 5: lane-consecutive 16-B loads (coalesced); 6: two lanes share a record (pairs);
// 7: five 16-B loads of a 128-B-aligned record (stride 128)
template <int MODE>
__global__ void __launch_bounds__(1024) gather_kernel(const float4* __restrict__ table, uint32_t n_rec, int iters, float* out, unsigned long long* cyc) {
    __shared__ float4 s_tab[5 * 512];
    if (MODE == 4) { for (uint32_t i = threadIdx.x; i < 5 * 512; i += blockDim.x) s_tab[i] = table[i % (5 * n_rec)]; __syncthreads(); }
    uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        h = h * 1664525u + 1013904223u;
        uint32_t r = (h >> 8) % n_rec;
        if (MODE == 1) r = __builtin_amdgcn_readfirstlane(r);
        if (MODE == 6) r = (uint32_t)__shfl((int)r, (int)(threadIdx.x & 62u), 64);
        if (MODE == 0 || MODE == 1 || MODE == 6) {
            const float4* p = (const float4*)((const char*)table + r * 80u);
            const float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
            acc += a.x + b.y + c.z + d.w + e.x;
        } else if (MODE == 7) {
            const float4* p = (const float4*)((const char*)table + (r % (n_rec * 5 / 8)) * 128u);
            const float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
            acc += a.x + b.y + c.z + d.w + e.x;
        } else if (MODE == 2) {
            acc += table[(r * 5u) % (n_rec * 5u)].x;
        } else if (MODE == 3) {
            acc += ((const float*)table)[(r * 20u) % (n_rec * 20u)];
        } else if (MODE == 4) {
            const uint32_t q = r & 511u;
            const float4 a = s_tab[q], b = s_tab[512 + q], c = s_tab[1024 + q], d = s_tab[1536 + q], e = s_tab[2048 + q];
            acc += a.x + b.y + c.z + d.w + e.x;
        } else if (MODE == 5) {
            const uint32_t base = ((h >> 8) % (n_rec * 5u / 64u)) * 64u;
            acc += table[__builtin_amdgcn_readfirstlane(base) + (threadIdx.x & 63u)].x;
        }
        // the next index depends on the data: one dependent round trip per iteration, like a traversal step
        h += (uint32_t)(acc != 12345.678f ? 0 : 1);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
typedef void (*gather_fn)(const float4*, uint32_t, int, float*, unsigned long long*);
static gather_fn gather_table[8] = {gather_kernel<0>, gather_kernel<1>, gather_kernel<2>, gather_kernel<3>, gather_kernel<4>, gather_kernel<5>, gather_kernel<6>, gather_kernel<7>};
static const char* gather_name[8] = {"80-B record per lane (5 x dwordx4)", "wave-uniform record (5 x dwordx4)", "one dwordx4 per lane, divergent", "one dword per lane, divergent",
                                     "LDS: 5 x ds_read_b128 per lane, divergent", "one dwordx4 per lane, coalesced", "80-B record per lane PAIR", "128-B-aligned record per lane (5 x dwordx4)"};
static const int gather_loads[8] = {5, 5, 1, 1, 5, 1, 5, 5};

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, n_cu, prop.clockRate);
    float* out; CHECK(hipMalloc(&out, sizeof(float) * 1024 * 1024 * 8));
    unsigned long long* cyc; CHECK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("\n== VALU issue: cycles per wave64 instruction per SIMD (s_memtime cycles of wave 0 / instructions of ALL waves on its SIMD); wall = from hipEvents at the measured clock\n");
    printf("%-24s", "kind \\ waves per SIMD");
    const int wps[] = {1, 2, 3, 4, 6, 8};
    for (int w : wps) printf("   w=%d cyc(wall)", w);
    printf("\n");
    for (int k = 0; k < K_COUNT; k++) {
        printf("%-24s", kind_name[k]);
        for (int w : wps) {
            const int iters = 4000, grid = n_cu * w;          // 256-thread blocks: one wave per SIMD each; w blocks per CU resident together
            hipLaunchKernelGGL(valu_table[k], dim3(grid), dim3(256), 0, 0, out, 10, cyc);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(valu_table[k], dim3(grid), dim3(256), 0, 0, out, iters, cyc);
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            const double n_inst = double(iters) * 32 * w;       // instructions issued on one SIMD
            const double mhz = double(c) / (ms * 1e3);          // shader clock implied by s_memtime vs wall (if s_memtime ticks at the shader clock)
            printf("   %5.2f (%4.0f MHz)", double(c) / n_inst, mhz);
        }
        printf("\n");
    }
    // ---- gathers
    printf("\n== vector-memory address rate: 1024-thread blocks, B blocks per CU; table of N 80-B records; lane-loads per s_memtime cycle per CU | ns per dependent step\n");
    const uint32_t sizes[] = {160, 4096, 65536, 4194304};     // 12.8 KB (L1), 320 KB (L2), 5 MB (L2 of all XCDs / MALL), 335 MB (HBM)
    std::vector<float> host(size_t(4194304) * 20);
    for (size_t i = 0; i < host.size(); i++) host[i] = float(i % 977) * 1e-3f;
    float4* table; CHECK(hipMalloc(&table, host.size() * 4)); CHECK(hipMemcpy(table, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    for (int bpc = 1; bpc <= 2; bpc++) {
        for (uint32_t n : sizes) {
            printf("-- N = %u records (%.1f KB), %d block(s) = %d waves per CU\n", n, n * 80 / 1024.0, bpc, 16 * bpc);
            for (int mode = 0; mode < 8; mode++) {
                if (mode == 4 && n != 160) continue;
                const int iters = 2000, grid = n_cu * bpc;
                hipLaunchKernelGGL(gather_table[mode], dim3(grid), dim3(1024), 0, 0, table, n, 20, out, cyc);
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(gather_table[mode], dim3(grid), dim3(1024), 0, 0, table, n, iters, out, cyc);
                CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                unsigned long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
                const double lane_loads = double(iters) * gather_loads[mode] * 1024 * bpc;   // per CU
                printf("   %-46s %6.3f lane-loads/cyc/CU   %7.1f cycles per step   %6.3f ms  -> %7.1f G lane-loads/s chip\n", gather_name[mode], lane_loads / double(c), double(c) / iters, ms,
                       lane_loads * n_cu / (ms * 1e-3) / 1e9);
            }
        }
    }
    return 0;
}
