// Wavefront formulation of Render::render -> cast_Ray -> ray_tracing (Render.cpp:56-175) for gfx950.
//
// Why not one megakernel (kernels.hip keeps one for cross-checking): measured on MI355X the per-thread bounce loop keeps
// the VALUs ~78 % busy at 14 % lane utilisation -- rays of one wave need very different numbers of BVH steps, only ~40 %
// of the lanes own a shadow ray, and shading code inflates the traversal loop to 169 VGPRs (2 waves/SIMD).  Here the
// path state lives in HBM (PathPool, 16-B records, one slot per lane => coalesced) and each iteration runs two kernels:
//
//   wf_shade_kernel  branch-sorted: a 256-thread block loads its 256 slots' state into LDS, classifies the slots (finished / ends here /
//                    diffuse / Phong / mirror), sorts the slot indices by class, and lane j processes slot perm[j] in LDS: consumes the
//                    hit of the slot's extend ray (emitter MIS, Russian roulette, termination, film write, regeneration of the next camera
//                    ray) and produces the next extend ray + at most one shadow ray (atomic-free queue of complete ray records: block b
//                    owns entries [256 b, 256 b + n)); the slots' own lanes then store the changed records coalesced.  ~100 VGPRs.
//   wf_trace8_kernel persistent waves over the ray list [P extend slots | per-block shadow queues]; closest-hit and any-hit rays share
//                    one traversal loop over the 8-wide compressed tree (80-B nodes, octant-ordered children, group stack entries).  Each
//                    wave schedules itself with __ballot/__popcll: it runs the inner-node block while most lanes sit at inner nodes, the
//                    leaf block once enough lanes cannot go on without their (parked) leaf, and the refill block (write results back, pull
//                    fresh rays from a wave-private chunk of the list) once enough lanes are idle -- so the expensive blocks execute with
//                    well-packed lanes.  71 VGPRs; 1024-thread blocks with 63 KB of LDS (top tree levels + per-lane stack).
// DESIGN.md §5 has the measurements behind each of these choices.
#include "pt_device.h"
#include "wavefront.h"

// Path-pool traffic is pure streaming (every record is read once and written once per iteration, ~1 GB per iteration): it is
// issued NON-TEMPORAL so it does not evict the few MB of scene data (BVH, triangles) that every ray and every shaded hit
// gathers from out of L2 / Infinity Cache.
typedef float wf_v4f __attribute__((ext_vector_type(4)));
typedef unsigned int wf_v4u __attribute__((ext_vector_type(4)));
typedef double wf_v2d __attribute__((ext_vector_type(2)));
#ifndef MCPT_NO_NT
__device__ __forceinline__ float4 ld_s(const float4* p) { const wf_v4f v = __builtin_nontemporal_load(reinterpret_cast<const wf_v4f*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st_s(float4* p, float4 v) { wf_v4f t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; __builtin_nontemporal_store(t, reinterpret_cast<wf_v4f*>(p)); }
__device__ __forceinline__ uint4 ld_s(const uint4* p) { const wf_v4u v = __builtin_nontemporal_load(reinterpret_cast<const wf_v4u*>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st_s(uint4* p, uint4 v) { wf_v4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; __builtin_nontemporal_store(t, reinterpret_cast<wf_v4u*>(p)); }
__device__ __forceinline__ double4 ld_s(const double4* p) {
    const wf_v2d a = __builtin_nontemporal_load(reinterpret_cast<const wf_v2d*>(p)), b = __builtin_nontemporal_load(reinterpret_cast<const wf_v2d*>(p) + 1);
    return make_double4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ void st_s(double4* p, double4 v) {
    wf_v2d a, b; a.x = v.x; a.y = v.y; b.x = v.z; b.y = v.w;
    __builtin_nontemporal_store(a, reinterpret_cast<wf_v2d*>(p)); __builtin_nontemporal_store(b, reinterpret_cast<wf_v2d*>(p) + 1);
}
typedef unsigned int wf_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 ld_s(const uint2* p) { const wf_v2u v = __builtin_nontemporal_load(reinterpret_cast<const wf_v2u*>(p)); return make_uint2(v.x, v.y); }
__device__ __forceinline__ void st_s(uint2* p, uint2 v) { wf_v2u t; t.x = v.x; t.y = v.y; __builtin_nontemporal_store(t, reinterpret_cast<wf_v2u*>(p)); }
__device__ __forceinline__ uint32_t ld_s(const uint32_t* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_s(uint32_t* p, uint32_t v) { __builtin_nontemporal_store(v, p); }
#else
template <class T> __device__ __forceinline__ T ld_s(const T* p) { return *p; }
template <class T> __device__ __forceinline__ void st_s(T* p, T v) { *p = v; }
#endif
__device__ __forceinline__ f3 xyz(const float4 v) { return mk3(v.x, v.y, v.z); }
__device__ __forceinline__ float4 mk4(f3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
__device__ __forceinline__ f3 wf_scrub_nan(f3 c) {        // Scene::set_Pixel (Scene.cpp:16-18)
    if (c.x != c.x) c.x = 0.f;
    if (c.y != c.y) c.y = 0.f;
    if (c.z != c.z) c.z = 0.f;
    return c;
}
__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {            // number of set bits below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t wave_first(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// A wave-uniform word written by an EARLIER launch, through the scalar cache (the compiler only does this by itself where it can prove that no
// store of the kernel aliases it; a vector load + readfirstlane costs a vmcnt(0) round trip instead).
__device__ __forceinline__ uint32_t s_load_u32(const uint32_t* p) { uint32_t v; asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory"); return v; }

// A per-lane select spelled as v_cndmask_b32_e64 with its lane mask as an explicit 64-bit scalar operand.  The compiler prefers the 32-bit VOP2
// encoding, whose mask is implicitly VCC (`s_and_b64 vcc, ...` then a run of `v_cndmask_b32_e32 ..., vcc`) -- and on MI355X a run of those costs a
// SIMD 9.7 ns per wave-instruction against 1.8 ns for the SAME select in the e64 encoding (mask in an SGPR pair or in VCC alike), 1.0 - 1.3 ns for
// a v_mov / v_add / v_fma_f32 (tools/uarch_probe2.hip, profiles/r04_uarch_probe.txt; only an e32 select right behind the v_cmp that wrote VCC is
// as cheap): five of them after a triangle test were a third of the leaf block's VALU time.
__device__ __forceinline__ uint64_t wf_mask(bool c) { return __builtin_amdgcn_ballot_w64(c); }
__device__ __forceinline__ uint32_t wf_sel(uint64_t mask, uint32_t if_set, uint32_t if_clear) {
    uint32_t r; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask)); return r;
}
__device__ __forceinline__ float wf_sel(uint64_t mask, float if_set, float if_clear) { return __uint_as_float(wf_sel(mask, __float_as_uint(if_set), __float_as_uint(if_clear))); }
__device__ __forceinline__ int wf_sel(uint64_t mask, int if_set, int if_clear) { return (int)wf_sel(mask, (uint32_t)if_set, (uint32_t)if_clear); }

// ====================================================================================================== shade
// Branch-sorted shading.  The reference assembles a different lobe set per material (BSDF.cpp:95-107) and a path may end at
// every vertex (miss, Russian roulette, depth), so lanes that simply own "their" slot diverge: measured 0.48 VALU lane utilisation
// with one lane per slot in slot order.  Here a 256-thread block first CLASSIFIES its 256 slots from three coalesced 16-B records
// (state word + throughput, hit record, ids), sorts the slot indices by class through LDS (stable counting sort: ballots + one
// 20-entry scan), and only then does lane j pick up slot perm[j] -- so whole waves run the same branch:
//
//   K_END     nothing to shade: dead / draining slot, or the extend ray missed            -> finish the sample, regenerate
//   K_EMIT    the path ends at this vertex (Russian roulette or depth) -- both are functions of the slot's throughput, bounce
//             and RNG key, so they are decided during classification -- but the reference still adds the emitter-MIS term
//             of this hit first (Render.cpp:146-162 precedes :164-170)                    -> hit record + emission, regenerate
//   K_DIFF    full vertex, Diffuse lobe only  (BSDF.cpp:105)
//   K_PHONG   full vertex, Blinn-Phong + Diffuse (BSDF.cpp:99-105)
//   K_MIRROR  full vertex, perfect mirror + Diffuse (BSDF.cpp:97-98)
//
// The lobe class of a hit comes with the hit: the trace kernel ORs the two class bits stored in the triangle's intersection
// record (tri_isect[].v0.w) into the triangle index it writes back -- no extra fetch.  All slot traffic of the permuted lanes stays
// inside the block's own 256-slot window of each pool array (the same cache lines the block would read in slot order).
#ifndef MCPT_SHADE_MIN_WAVES
#define MCPT_SHADE_MIN_WAVES 5      // => 96 VGPRs without scratch (100 unconstrained): five waves per SIMD where no trace block shares the CU
#endif
#define K_END 0u
#define K_EMIT 1u
#define K_DIFF 2u      // K_DIFF + lobe class (HIT_CLASS_*) = K_PHONG, K_MIRROR
#define K_COUNT 5u
// x / d by multiply + shift (RenderParams::div_*): exact for x < MCPT_FASTDIV_MAX
__device__ __forceinline__ uint32_t wf_fastdiv(uint32_t x, uint32_t m, uint32_t s) { return (uint32_t)(((unsigned long long)x * m) >> s); }
template <bool COUNT>
__global__ void __launch_bounds__(WF_SHADE_BLOCK, MCPT_SHADE_MIN_WAVES) wf_shade_kernel(DevScene sc, RenderParams p, PathPool pool, IterCtl* ctl, uint32_t it, uint32_t n_items,
                                                              float4* __restrict__ accum, DevCounters* gcnt) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));   // wave index: scalar
    const uint32_t base = blockIdx.x * WF_SHADE_BLOCK;                             // pool.P is a multiple of WF_SHADE_BLOCK
#ifdef WF_SHADE_STATS      // diagnostic build (tools/shade_stats.py): shader-clock cycles of every wave per section of the kernel
    unsigned long long sh_t[7] = {0, 0, 0, 0, 0, 0, 0}, sh_mark = __builtin_amdgcn_s_memtime();
#define SH_TICK(I) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); sh_t[I] += t_now - sh_mark; sh_mark = t_now; }
#else
#define SH_TICK(I)
#endif
    if (base + tid == 0) { const uint32_t n = (it + 1) & 3; ctl->trace_head[n] = 0; ctl->any_active[n] = 0; }
    // The host notices the end of a job a few iterations late (it polls control-block snapshots and runs up to 8 iterations ahead): once an
    // iteration has left no live slot -- a dead slot takes the next work item in the same call, so that also means no item is left -- every
    // later launch of the job ends here, before it streams the pool.  (An empty iteration used to cost 0.5 - 0.65 ms in this kernel and
    // 0.2 - 0.4 ms in the trace kernel: ~10 % of a 128-spp share of a strong-scaled job.)
    // After a drain compaction (wf_compact_*) the live slots sit at the front of the pool and only they are swept.
    { const uint32_t p_act = ctl->pad[WF_CTL_P_ACTIVE]; if ((it != 0u && ctl->any_active[(it - 1u) & 3u] == 0u) || (p_act != 0u && base >= p_act)) return; }
    __shared__ uint32_t s_wave_cnt[WF_SHADE_BLOCK / 64], s_shadow_cnt[WF_SHADE_BLOCK / 64], s_live_cnt[WF_SHADE_BLOCK / 64];
    __shared__ uint32_t s_base, s_sel, s_scan, s_priv_base, s_priv_take, s_priv_next, s_priv_end;
    // the whole slot state of the block's 256-slot window, fetched coalesced in ONE batch by the slots' own lanes and handed to the
    // lanes that will process them through LDS: no dependent second round of (gathering) global loads after the sort ...
    __shared__ float4 s_beta[WF_SHADE_BLOCK], s_hit[WF_SHADE_BLOCK], s_L[WF_SHADE_BLOCK], s_rd[WF_SHADE_BLOCK], s_nee[WF_SHADE_BLOCK], s_ro[WF_SHADE_BLOCK];
    __shared__ uint2 s_ids[WF_SHADE_BLOCK];                                        // pixel, sample index: the half of the ids record every iteration needs
    uint2* const id_ps = reinterpret_cast<uint2*>(pool.ids);                       // pool.ids = [P x {pixel, sample}] [P x {next sample, end sample}]: the second half
    uint2* const id_ne = id_ps + pool.P;                                           // is touched by multi-sample items only (one-sample items are exhausted by definition)
    // ... and the way back: the processing lane leaves the slot's new records in LDS (each output record reuses the LDS cell of an
    // input record of the SAME slot, which only this lane read: no hazard), and after one barrier the slots' own lanes store them
    // coalesced.  Permuted lanes storing straight to the pool write partial 128-B lines from several waves: measured 1.5x (class
    // sorted) to 2.3x (interleaved) longer launches.  Cells: beta, L, ray_d, ray_o, nee, ids in place; sh_d in s_hit; the "which
    // records changed" flags travel in bits 1.. of ray_d.w (bit 0 = an extend ray is pending, what the trace kernel looks at).
    // 28 KB of cells + 3 KB of tables: three shade blocks and one 64-KB trace block fit a CU's 160 KB together.
#define OUT_RAY_O 2u
#define OUT_SHADOW 4u
#define OUT_IDS 8u
    __shared__ uint32_t s_perm[WF_SHADE_BLOCK];
    __shared__ uint32_t s_kcnt[K_COUNT * (WF_SHADE_BLOCK / 64)];                   // [key][wave]: count, then exclusive prefix
    // Small scene tables staged in LDS once per block: every scattered global load costs vector-memory issue time whether its
    // lanes hit 2 distinct lines or 64.  Lights (record + fp64 corners) and materials are tiny in typical scenes; larger tables
    // fall back to global memory.
    __shared__ float4 s_mats[WF_LDS_MATS * 4];
    __shared__ float4 s_lights[WF_LDS_LIGHTS * 4];
    __shared__ double s_light_pos[WF_LDS_LIGHTS * 9];
    const bool mats_lds = sc.n_mats <= WF_LDS_MATS, lights_lds = sc.n_lights <= WF_LDS_LIGHTS;

    // ---- classification of the lane's OWN slot (coalesced): which branch will this slot take?
    {
        const float4 bt0 = ld_s(&pool.beta[base + tid]), h0 = ld_s(&pool.hit[base + tid]);
        const uint2 id0 = ld_s(&id_ps[base + tid]);
        const float4 L0 = ld_s(&pool.L[base + tid]), rd0 = ld_s(&pool.ray_d[base + tid]), nee0 = ld_s(&pool.nee[base + tid]), ro0 = ld_s(&pool.ray_o[base + tid]);
        // the block's private work-item range: its cursor rides in this batch of loads too (thread 0) and waits in LDS until the item pull
        uint2 priv = make_uint2(0u, 0u);
        if (tid == 0 && p.priv_items) {
            if (it == 0) { priv.x = 0u; priv.y = p.priv_items; }
            else priv = pool.block_items[blockIdx.x];
        }
        // the tables ride in the same batch of loads, behind the slot records (the light corners are a contiguous copy: fetching them
        // through lights[i].tri was a second, dependent round trip in front of the block's first barrier)
        if (mats_lds) for (uint32_t i = tid; i < (uint32_t)sc.n_mats * 4; i += WF_SHADE_BLOCK) s_mats[i] = reinterpret_cast<const float4*>(sc.mats)[i];
        if (lights_lds) {
            for (uint32_t i = tid; i < (uint32_t)sc.n_lights * 4; i += WF_SHADE_BLOCK) s_lights[i] = reinterpret_cast<const float4*>(sc.lights)[i];
            for (uint32_t i = tid; i < (uint32_t)sc.n_lights * 9; i += WF_SHADE_BLOCK) s_light_pos[i] = sc.light_pos64[i];
        }
        s_L[tid] = L0; s_rd[tid] = rd0; s_nee[tid] = nee0; s_ro[tid] = ro0;
        if (tid == 0) { s_priv_next = priv.x; s_priv_end = priv.y; }
        s_beta[tid] = bt0; s_hit[tid] = h0; s_ids[tid] = id0;
        const uint32_t st0 = __float_as_uint(bt0.w);
        const int bounce0 = (int)(st0 >> 8), tri0 = __float_as_int(h0.x);
        uint32_t key = K_END;
        if ((st0 & 3u) == SLOT_ALIVE && tri0 >= 0) {
            bool ends = p.max_depth != 0 && (uint32_t)bounce0 >= p.max_depth;                       // `bounces < max_depth` (Render.cpp:116)
            if (bounce0 - 1 > 3) {                                                                 // Render.cpp:164-168
                const float q = fminf(max3(xyz(bt0)), 0.95f);
                const Rng4 r = rng_block(id0.x, id0.y, 2u + 2u * (uint32_t)(bounce0 - 1), p.seed_lo, p.seed_hi);
                ends = ends || r.v[2] > q;
            }
            key = ends ? K_EMIT : K_DIFF + ((uint32_t)tri0 >> HIT_CLASS_SHIFT);
        }
#ifdef MCPT_SHADE_NOSORT                                                                           // A/B build: slot order, same code otherwise
        const uint32_t skey = 0;
#else
        const uint32_t skey = key;
#endif
        uint32_t my_rank = 0;
#pragma unroll
        for (uint32_t k = 0; k < K_COUNT; k++) {
            const uint64_t m = __ballot(skey == k);
            if (skey == k) my_rank = lane_rank(m);
            if (lane == k) s_kcnt[k * (WF_SHADE_BLOCK / 64) + wv] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        // exclusive prefix of the slot's (key, wave) counter over the [key][wave] table, summed by every thread for itself (20 broadcast LDS
        // reads) -- a serial scan by one thread between two barriers cost a barrier
        const uint32_t my_cell = skey * (WF_SHADE_BLOCK / 64) + wv;
#ifdef MCPT_SHADE_SERIAL_PREFIX      // A/B: every thread sums the table itself (20 broadcast reads, 60 VALU issues per wave)
        uint32_t before_me = 0;
#pragma unroll
        for (uint32_t i = 0; i < K_COUNT * (WF_SHADE_BLOCK / 64); i++) before_me += i < my_cell ? s_kcnt[i] : 0u;
#else
        // ... as a wave scan: lane i < 20 holds cell i, five DPP adds make the inclusive prefix over lanes 0..31 (row_shr 1/2/4/8 inside
        // each row of 16 lanes, row_bcast:15 carries row 0's total into row 1), and every lane fetches the entry of its own cell with one
        // ds_bpermute -- ~15 instead of ~60 VALU issues per wave, in a kernel half of whose issues are this prologue and the epilogue
        static_assert(K_COUNT * (WF_SHADE_BLOCK / 64) <= 32, "the scan covers two DPP rows");
        const uint32_t cell_cnt = lane < K_COUNT * (WF_SHADE_BLOCK / 64) ? s_kcnt[lane] : 0u;
        int inc = (int)cell_cnt;
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);      // row_shr:1  (lanes shifted in from outside the row read 0)
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);      // row_shr:2
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);      // row_shr:4
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);      // row_shr:8
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3 (other rows add the `old` operand, 0)
        const uint32_t before_me = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(my_cell << 2), inc - (int)cell_cnt);   // exclusive prefix of my cell
#endif
        s_perm[(before_me + my_rank) & (WF_SHADE_BLOCK - 1u)] = tid | (key << 16);
        __syncthreads();
    }
#ifdef MCPT_SHADE_PERM_TEST     // diagnostic: a class-blind interleave -- every wave touches every line of the window, no sorting benefit
    const uint32_t pk = s_perm[((tid & 63u) << 2) | (tid >> 6)];
#else
    const uint32_t pk = s_perm[tid];
#endif
    const uint32_t key = pk >> 16, src = pk & 0xffffu, slot = base + src;
    SH_TICK(0)                                                                                    // tables + slot records + classification sort

    // From here on the slot's LDS cells are the HOME of its state: values are read where a phase needs them and parked again when it is
    // done (the cells belong to this lane alone), so that almost nothing but indices stays in registers across the register-hungry
    // fp64 light-sampling phase.  WF_PHASE() is a compiler-only fence: it keeps the scheduler from hoisting the next phase's LDS reads
    // (and their destination registers) up into the previous one.
#define WF_PHASE() asm volatile("" ::: "memory")
    const uint32_t st = __float_as_uint(s_beta[src].w);
    uint32_t state = st & 3u;
    bool prev_mirror = (st & 4u) != 0;
    int bounce = (int)(st >> 8);
    float4 sm = make_float4(0.f, 0.f, 0.f, 0.f); bool sm_loaded = false;    // item accumulator: fetched only when a path ends
    const float nl = (float)sc.n_lights;
    const bool correct_t2 = (p.flags & MCPT_FLAG_CORRECT_SHADOW_T2) != 0;

    bool terminated = false, emit_extend = false, emit_shadow = false, sum_dirty = false, id_dirty = false;
    bool c_prim = false, c_cont = false, c_self_t = false, c_self_h = false, c_shaded = false;
    uint32_t c_texel = 0;
    uint32_t out_flags = 0;

    if (state != SLOT_DEAD && (st & 8u) != 0) {                                        // Render.cpp:125-130: the previous vertex's light sample
        const float4 n4 = s_nee[src];                                                  // (.w != 0: the trace kernel found the shadow ray blocked)
        if (__float_as_uint(n4.w) == 0u) { float4 L4 = s_L[src]; L4.x += n4.x; L4.y += n4.y; L4.z += n4.z; s_L[src] = L4; }
    }
    if (key != K_END) do {                                                             // ALIVE and the extend ray hit something
        int tri;
        double4 plane64;                                                               // the hit triangle's fp64 plane: requested here, beside the shading record,
        f3 kd_early = mk3(0.f, 0.f, 0.f);
        {   // ---- phase 1: the hit record (Triangle.cpp:68-76), emitter MIS (Render.cpp:146-162), roulette rescale, first-hit emission
            const float4 h = s_hit[src];
            tri = __float_as_int(h.x) & HIT_TRI_MASK;
            plane64 = *reinterpret_cast<const double4*>(sc.tri_shade + MCPT_TRI_SHADE_F4 * (size_t)tri + 4);   // the second half of the hit's 128-B record (the same cache line as load_hit_shade's half); used a phase later
            const f3 d = xyz(s_rd[src]);
            const HitShade hs = load_hit_shade(sc, tri, h.y, h.z, d);                            // h.y, h.z: fp32 barycentrics of the traversal
            const float4 m1 = mats_lds ? s_mats[4 * hs.mat + 1] : reinterpret_cast<const float4*>(sc.mats)[4 * hs.mat + 1];   // radiance | flags
            const uint32_t mflags = __float_as_uint(m1.w);
            c_shaded = true;
            if (bounce > 0 && (mflags & MAT_EMISSIVE) && hs.front) {                            // Render.cpp:146-162
                const f3 rad = xyz(m1), beta = xyz(s_beta[src]);
                float4 L4 = s_L[src];
                f3 add;
                if (prev_mirror) add = beta * rad;
                else {
                    // Render.cpp:150-152 measures |prev - p| and the cosine along normalize(prev - p): the traced ray IS that segment
                    // (unit direction d, hit distance h.w), so neither the previous vertex nor a square root is needed
                    const float cosine = -dot(d, hs.n);
                    float light_pdf = 0.f;
                    if (cosine != 0.f) light_pdf = h.w * h.w * rcp(cosine * nl * tri_area(sc, tri));
                    add = beta * rad * power_heuristic(L4.w, light_pdf);                        // L4.w = pdf of the BSDF sample that got here
                }
                L4.x += add.x; L4.y += add.y; L4.z += add.z;
                s_L[src] = L4;
            }
            if (key == K_EMIT) { terminated = true; break; }                                    // Render.cpp:164-170 / `bounces < max_depth`
            if (bounce - 1 > 3) {                                                               // survived the roulette (decided above)
                float4 b4 = s_beta[src];
                const float iq = rcp(fminf(fmaxf(fmaxf(b4.x, b4.y), b4.z), 0.95f));
                b4.x *= iq; b4.y *= iq; b4.z *= iq;
                s_beta[src] = b4;
            }
            if (bounce == 0 && (mflags & MAT_EMIT_0)) { float4 L4 = s_L[src]; L4.x += m1.x; L4.y += m1.y; L4.z += m1.z; s_L[src] = L4; }   // :121-122
            // park what the BSDF phase needs of the hit record in the two cells this slot no longer needs (hit, NEE payload)
            s_hit[src] = mk4(hs.n, hs.tu); s_nee[src] = make_float4(hs.tv, __int_as_float(hs.mat), 0.f, 0.f);
            {   // the diffuse colour's texel is requested here and used two phases later (three registers across the light sample: S-bath
                // -0.4 ... -1 % of the step; nothing to fetch for a constant-colour material)
                DevMaterial mat;
                if (mats_lds) { float4* m4 = reinterpret_cast<float4*>(&mat); m4[0] = s_mats[4 * hs.mat]; m4[1] = s_mats[4 * hs.mat + 1]; m4[2] = s_mats[4 * hs.mat + 2]; m4[3] = s_mats[4 * hs.mat + 3]; }
                else mat = sc.mats[hs.mat];
                kd_early = tex_color(sc, mat, hs.tu, hs.tv, c_texel);
            }
        }
        WF_PHASE();
        SH_TICK(1)                                                                                // phase 1: hit record gather, emitter MIS
        // ---- phase 2: light sample (Render.cpp:124, :202-223) with the fp64 self-hit predicate of SURVEY A-9
        // The hit point in fp64: ray (fp32 origin the trace kernel used, fp32 direction) x the triangle's fp64 plane.  Like the
        // reference's point (Triangle.cpp:35-38) it lies on that plane to ~1e-16 with full fp64 noise in its low bits -- the two
        // properties the self-occlusion statistics of SURVEY A-9 rest on; where exactly on the plane moves by ~1e-8 with the origin's
        // rounding, which no statistic sees (r01 carried a 32-B fp64 origin per slot for this: 64 B of pool traffic per bounce).
        LightSample ls; f3 p32; float xi_lobe;
        {
            const uint2 id = s_ids[src];
            const Rng4 ra = rng_block(id.x, id.y, 1u + 2u * (uint32_t)bounce, p.seed_lo, p.seed_hi);
            xi_lobe = ra.v[3];
            const d3 p64 = hit_point64_plane(plane64, to_d3(xyz(s_ro[src])), xyz(s_rd[src]));
            p32 = to_f3(p64);
            LightData ld;
            if (lights_lds) {
                const int cnt = sc.n_lights;
                int li = (int)(ra.v[0] * (float)cnt); li = li < cnt - 1 ? li : cnt - 1;              // Render.cpp:204-205
                ld.a = s_lights[4 * li]; ld.b = s_lights[4 * li + 1]; ld.c = s_lights[4 * li + 2]; ld.e = s_lights[4 * li + 3];
                const double* LP = s_light_pos + 9 * li;
                ld.v0 = mkd(LP[0], LP[1], LP[2]); ld.v1 = mkd(LP[3], LP[4], LP[5]); ld.v2 = mkd(LP[6], LP[7], LP[8]);
            } else ld = light_fetch(sc, ra.v[0]);
            ls = sample_light(ld, p64, ra.v[1], ra.v[2], true, mkd(sc.centre[0], sc.centre[1], sc.centre[2]));
        }
        WF_PHASE();
        SH_TICK(2)                                                                                // phase 2: fp64 hit point + light sample
        // ---- phase 3: BSDF (BSDF.cpp:87-110), NEE with MIS (Render.cpp:125-130), BSDF sample (Render.cpp:133-140)
        Bsdf bsdf;
        {
            const float4 hn = s_hit[src], ht = s_nee[src];
            const int mi = __float_as_int(ht.y);
            DevMaterial mat;
            if (mats_lds) { float4* m4 = reinterpret_cast<float4*>(&mat); m4[0] = s_mats[4 * mi]; m4[1] = s_mats[4 * mi + 1]; m4[2] = s_mats[4 * mi + 2]; m4[3] = s_mats[4 * mi + 3]; }
            else mat = sc.mats[mi];
            const f3 kd = kd_early;                                                             // (Texture::get_color: fetched in phase 1)
            bsdf = make_bsdf(mat, kd, xyz(hn), -xyz(s_rd[src]));
        }
        int sh_skip = -1;
        if (ls.pdf != 0.f) {
            c_self_t = true; c_self_h = ls.self_hit;
            if (correct_t2 || !ls.self_hit) {
                f3 fx; float bpdf;
                bsdf_eval(bsdf, ls.wo, fx, bpdf);
                const float cos_theta = fabsf(dot(bsdf.w, ls.wo));
                const float weight = power_heuristic(ls.pdf * rcp(nl), bpdf);
                const f3 nee = weight * xyz(s_beta[src]) * ls.rad * fx * (cos_theta * rcp(ls.pdf) * nl);   // Render.cpp:127-129
                s_hit[src] = mk4(ls.wo, ls.t2); s_nee[src] = mk4(nee, 0.f); out_flags |= OUT_SHADOW;   // the shadow ray's direction | t2, NEE payload
                sh_skip = ls.tri; emit_shadow = true;
            }
        }
        s_ro[src] = mk4(p32, __int_as_float(sh_skip)); out_flags |= OUT_RAY_O;                  // both new rays start at the hit point
        Scatter s;
        {
            const uint2 id = s_ids[src];
            const Rng4 rb = rng_block(id.x, id.y, 2u + 2u * (uint32_t)bounce, p.seed_lo, p.seed_hi);
            s = bsdf_sample(bsdf, xi_lobe, rb.v[0], rb.v[1]);                                   // Render.cpp:133-134
        }
        if (s.pdf == 0.f) {                                                                     // Render.cpp:135-136: path ends, but its last
            state = SLOT_DRAIN;                                                                 // shadow ray is still in flight -> finalise next call
            break;
        }
        {
            const float sc_ = fabsf(dot(bsdf.w, s.wo)) * rcp(s.pdf);
            float4 b4 = s_beta[src];
            b4.x *= s.f.x * sc_; b4.y *= s.f.y * sc_; b4.z *= s.f.z * sc_;                      // Render.cpp:140
            s_beta[src] = b4;
            float4 L4 = s_L[src]; L4.w = s.pdf; s_L[src] = L4;
        }
        prev_mirror = s.mirror;
        s_rd[src] = mk4(s.wo, 0.f); bounce++;
        emit_extend = true; c_cont = true;
    } while (0);
    else if (state != SLOT_DEAD) terminated = true;                                             // miss (Render.cpp:118-119,144-145) or DRAIN
    WF_PHASE();
    SH_TICK(3)                                                                                    // phase 3: BSDF, NEE, BSDF sample
    uint4 id;                                                                                   // pixel, sample, s_next, s_end
    { const uint2 ps = s_ids[src]; id.x = ps.x; id.y = ps.y; id.z = id.w = 0u; }
    const bool multi = p.samples_per_item != 1u && p.probe_n == 0u;                             // (a probe item is one sample)
    if (multi && (terminated || state == SLOT_DEAD)) { const uint2 ne = ld_s(&id_ne[slot]); id.z = ne.x; id.w = ne.y; }   // fetched only when a path ends, like `sum`

    // one-sample items (the default) flush every finished sample straight to the film: their accumulator record is always zero, so
    // it is neither fetched (a dependent round trip in front of the regeneration) nor written back
    if (p.samples_per_item != 1u && (terminated || (state == SLOT_DEAD && id.z == id.w))) { sm = ld_s(&pool.sum[slot]); sm_loaded = true; }
    if (terminated) {                                                                           // Scene::set_Pixel, per sample
        const f3 c = wf_scrub_nan(xyz(s_L[src]));
        sm.x += c.x; sm.y += c.y; sm.z += c.z; sm.w += 1.f; sum_dirty = true;
        state = SLOT_DEAD;
    }
    bool want_item = false;
    if (state == SLOT_DEAD && id.z == id.w) {                                                   // item exhausted (or never had one)
        if (sm.w > 0.f) {                                                                       // film: sum + count (Scene.cpp:19-20)
            float* a = reinterpret_cast<float*>(accum + id.x);
            if (p.atomic_accum) { atomicAdd(a + 0, sm.x); atomicAdd(a + 1, sm.y); atomicAdd(a + 2, sm.z); atomicAdd(a + 3, sm.w); }
            else { float4 cur = accum[id.x]; cur.x += sm.x; cur.y += sm.y; cur.z += sm.z; cur.w += sm.w; accum[id.x] = cur; }
            sm = make_float4(0.f, 0.f, 0.f, 0.f); sum_dirty = true;
        }
        want_item = true;
    }
    // ---- pull new work items: one atomic per BLOCK on one of WF_ITEM_SHARDS cursors (own shard first, then a few others);
    //      a plain load screens out exhausted shards so the end-of-render tail costs no atomics at all
    {
        const uint64_t m = __ballot(want_item);
        if (lane == 0) s_wave_cnt[wv] = (uint32_t)__popcll(m);
        __syncthreads();
        if (tid == 0) {
            uint32_t tot = 0; for (uint32_t k = 0; k < WF_SHADE_BLOCK / 64; k++) tot += s_wave_cnt[k];
            uint32_t b0 = 0xffffffffu, sel = 0;
            const uint32_t priv_next = s_priv_next, priv_end = s_priv_end;
            const uint32_t take = min(tot, priv_end - priv_next);                                 // from the private range: no atomic
            s_priv_base = priv_next; s_priv_take = take;
            if (p.priv_items && (take || it == 0)) pool.block_items[blockIdx.x] = make_uint2(priv_next + take, priv_end);
            const uint32_t rest = tot - take;
            if (rest) {
                for (uint32_t probe = 0; probe < 4; probe++) {
                    const uint32_t k = (blockIdx.x + (it * 4u + probe) * 17u) & (WF_ITEM_SHARDS - 1);   // (rotates with the iteration: a small grid still visits every shard)
                    const uint32_t cap = wf_shard_capacity(n_items, k);
                    if (__hip_atomic_load(&ctl->item_cursor[k].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < cap) {
                        const uint32_t b = atomicAdd(&ctl->item_cursor[k].v, rest);
                        if (b < cap) { b0 = b; sel = k; break; }
                    }
                }
            }
            s_scan = (rest && b0 == 0xffffffffu) ? 1u : 0u;
            s_base = b0; s_sel = sel;
        }
        __syncthreads();
        // Slots stay empty although items may be left in shards this call did not probe: say so, or the launches after this one would take
        // "no live slot" for "job finished" (the early return at the top of both kernels).  Only reached at the end of a job; wave 0 looks
        // at all 64 cursors at once.
        static_assert(WF_ITEM_SHARDS == 64, "one lane of wave 0 per work-item cursor");
        if (wv == 0 && s_scan) {
            const bool left = __hip_atomic_load(&ctl->item_cursor[lane].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wf_shard_capacity(n_items, lane);
            if (__ballot(left) != 0 && lane == 0) ctl->any_active[it & 3u] = 1u;
        }
        if (want_item) {
            id.z = id.w = 0; id_dirty = true;
            uint32_t before = 0;
            for (uint32_t k = 0; k < wv; k++) before += s_wave_cnt[k];
            const uint32_t j = before + lane_rank(m);                       // this lane's rank among the block's requests
            const bool mine = j < s_priv_take;
            if (mine || s_base != 0xffffffffu) {
                const uint32_t l = s_base + (j - s_priv_take);
                const uint32_t shared_item = ((l / WF_SHADE_BLOCK) * WF_ITEM_SHARDS + s_sel) * WF_SHADE_BLOCK + (l % WF_SHADE_BLOCK);
                // private item jl of block b = tile-sized unit (jl / 64) * n_blocks + b: every block's range is spread over the whole image,
                // so all blocks see the same mix of path lengths and use their ranges up at the same pace (contiguous ranges did not:
                // +2 % iterations at the end of a call, which ate what the missing atomic had saved)
                const uint32_t jl = s_priv_base + j;
                const uint32_t item = mine ? ((jl >> 6) * gridDim.x + blockIdx.x) * 64u + (jl & 63u) : p.shared_base + shared_item;
                if (p.probe_n) {                                            // probe: item = film entry, one sample
                    if (item < p.probe_n) { id.x = item; id.z = p.first_sample; id.w = p.first_sample + 1u; }
                } else if (mine || shared_item < n_items) {
                    const uint32_t n_tiles = p.n_owned;
                    const uint32_t iw = item >> 6, il = item & 63u;
                    const uint32_t chunk = wf_fastdiv(iw, p.div_owned_m, p.div_owned_s), tile = p.tile_rem + (iw - chunk * n_tiles) * p.tile_mod;   // iw / n_tiles
                    const uint32_t ty = wf_fastdiv(tile, p.div_tiles_x_m, p.div_tiles_x_s);                                                    // tile / tiles_x
                    const uint32_t px = (tile - ty * p.tiles_x) * 8 + (il & 7), py = ty * 8 + (il >> 3);
                    if (px < (uint32_t)sc.cam.width && py < (uint32_t)sc.cam.height) {
                        id.x = py * (uint32_t)sc.cam.width + px;
                        id.z = p.first_sample + chunk * p.samples_per_item;
                        id.w = p.first_sample + min(p.spp, (chunk + 1) * p.samples_per_item);
                    }
                }
            }
        }
    }
    if (state == SLOT_DEAD && id.z < id.w) {                                                    // next sample of the item: camera ray
        id.y = id.z++; id_dirty = true;
        d3 eye64; f3 no, nd;                                                                    // (the fp32 origin `no` is what travels)
        if (p.probe_n) {                                                                        // mcpt_probe_paths: the caller's ray
            eye64 = mkd(p.probe_o[3 * id.x], p.probe_o[3 * id.x + 1], p.probe_o[3 * id.x + 2]); no = to_f3(eye64);
            nd = mk3((float)p.probe_d[3 * id.x], (float)p.probe_d[3 * id.x + 1], (float)p.probe_d[3 * id.x + 2]);
        } else {
            const Rng4 r = rng_block(id.x, id.y, 0u, p.seed_lo, p.seed_hi);
            const int py = (int)wf_fastdiv(id.x, p.div_width_m, p.div_width_s), px = (int)(id.x - (uint32_t)py * (uint32_t)sc.cam.width);   // id.x / width, id.x % width
            cast_ray(sc.cam, px, py, r.v[0], r.v[1], eye64, no, nd);                            // Render.cpp:64
        }
        s_ro[src] = mk4(no, __int_as_float(-1)); out_flags |= OUT_RAY_O;
        s_rd[src] = mk4(nd, 0.f);
        s_beta[src] = make_float4(1.f, 1.f, 1.f, 0.f); s_L[src] = make_float4(0.f, 0.f, 0.f, 0.f);
        bounce = 0; prev_mirror = false;
        state = SLOT_ALIVE; emit_extend = true; c_prim = true;
    }

    SH_TICK(4)                                                                                    // film write, item pull, camera ray
    // ---- finish the slot's cells: state word, "which records changed" flags
    {
        float4 b4 = s_beta[src];
        b4.w = __uint_as_float(state | (prev_mirror ? 4u : 0u) | (emit_shadow ? 8u : 0u) | ((uint32_t)bounce << 8));
        s_beta[src] = b4;
        if (sum_dirty && sm_loaded) st_s(&pool.sum[slot], sm);                   // multi-sample items only (not the default): stored directly
        if (id_dirty) { s_ids[src] = make_uint2(id.x, id.y); out_flags |= OUT_IDS; if (multi) st_s(&id_ne[slot], make_uint2(id.z, id.w)); }
        float4 r4 = s_rd[src];
        r4.w = __uint_as_float(out_flags | (emit_extend ? 1u : 0u));
        s_rd[src] = r4;
    }

    // ---- shadow queue append, atomic-free: ranks inside the block through LDS; the block owns queue entries [256 b, 256 b + n) and
    //      publishes n with a plain store.  (A returning atomic per block on a shared cursor -- even sharded 8 ways -- held every
    //      block's four waves at the barrier for its round trip: 0.58 vs 0.24 ms per launch.)
    const uint32_t cur = it & 3;
    const uint64_t ms = __ballot(emit_shadow);
    const uint32_t n_live = (uint32_t)__popcll(__ballot(state != SLOT_DEAD));
    if (lane == 0) { s_shadow_cnt[wv] = (uint32_t)__popcll(ms); s_live_cnt[wv] = n_live; }   // (cells of their own: the item pull's s_wave_cnt may still be read by slower waves)
    __syncthreads();
    if (tid == 0) {
        uint32_t tot = 0, live = 0;
        for (uint32_t k = 0; k < WF_SHADE_BLOCK / 64; k++) { tot += s_shadow_cnt[k]; live += s_live_cnt[k]; }
        st_s(&pool.shadow_count[blockIdx.x], tot); st_s(&pool.live_cnt[blockIdx.x], live);
    }
    if (emit_shadow) {
        uint32_t before = 0;
        for (uint32_t k = 0; k < wv; k++) before += s_shadow_cnt[k];
        // the shadow ray goes to the queue as a complete record (origin | triangle to skip, direction | t2, slot): the trace kernel reads a
        // queued ray in ONE coalesced round trip instead of chasing queue -> slot -> ray records through two
        const uint32_t q = base + before + lane_rank(ms);
        st_s(&pool.shadow_queue[q], slot); st_s(&pool.sq_o[q], s_ro[src]); st_s(&pool.sq_d[q], s_hit[src]);
    }
    {   // the barrier above also published every slot's output cells: store the OWN slot's records, coalesced
        const uint32_t own = base + tid;
        const float4 rdo = s_rd[tid];
        const uint32_t fl = __float_as_uint(rdo.w);
        st_s(&pool.beta[own], s_beta[tid]); st_s(&pool.L[own], s_L[tid]); st_s(&pool.ray_d[own], rdo);
        if (fl & OUT_RAY_O) st_s(&pool.ray_o[own], s_ro[tid]);
        if (fl & OUT_SHADOW) st_s(&pool.nee[own], s_nee[tid]);
        if (fl & OUT_IDS) st_s(&id_ps[own], s_ids[tid]);
    }
    SH_TICK(5)                                                                                    // shadow-queue append + coalesced stores
    // ---- bookkeeping: liveness flag (plain store) and ray counters (replicated per block => uncontended atomics)
    const uint64_t ma = __ballot(state != SLOT_DEAD);
    const uint64_t m_term = __ballot(terminated), m_prim = __ballot(c_prim), m_cont = __ballot(c_cont);
    const uint64_t m_st = __ballot(c_self_t), m_sh = __ballot(c_self_h), m_shaded = __ballot(c_shaded);
    unsigned long long texels = 0;
    if (COUNT) { texels = c_texel; for (int off = 32; off > 0; off >>= 1) texels += __shfl_xor(texels, off, 64); }
    if (lane == 0) {
        DevCounters* g = gcnt + (blockIdx.x & (WF_COUNTER_REPLICAS - 1));
        if (ma) ctl->any_active[cur] = 1u;
#ifndef WF_SCHED_STATS
        if (m_term) atomicAdd(&g->paths, (unsigned long long)__popcll(m_term));
#endif
        if (m_prim) atomicAdd(&g->rays_primary, (unsigned long long)__popcll(m_prim));
        if (m_cont) atomicAdd(&g->rays_continuation, (unsigned long long)__popcll(m_cont));
        if (ms) atomicAdd(&g->rays_shadow, (unsigned long long)__popcll(ms));
#ifndef WF_SCHED_STATS
        if (m_st) atomicAdd(&g->self_shadow_tests, (unsigned long long)__popcll(m_st));
        if (m_sh) atomicAdd(&g->self_shadow_hits, (unsigned long long)__popcll(m_sh));
#endif
#ifndef WF_SCHED_STATS
        if (COUNT) { if (m_shaded) atomicAdd(&g->shaded_hits, (unsigned long long)__popcll(m_shaded)); if (texels) atomicAdd(&g->texel_fetches, texels); }
#endif
#ifdef WF_SHADE_STATS
        SH_TICK(6)
        atomicAdd(&g->debug[0], sh_t[0]); atomicAdd(&g->debug[1], sh_t[1]); atomicAdd(&g->debug[2], sh_t[2]); atomicAdd(&g->debug[3], sh_t[3]);
        atomicAdd(&g->box_tests, sh_t[4]); atomicAdd(&g->tri_tests, sh_t[5]); atomicAdd(&g->stack_spills, sh_t[6]); atomicAdd(&g->texel_fetches, 1ull);   // texel_fetches: waves
#endif
    }
}
#undef SH_TICK

// ====================================================================================================== trace
#ifndef WF_TRACE_BLOCK
#define WF_TRACE_BLOCK 1024
#endif
#ifndef WF_CHUNK_BATCH
#define WF_CHUNK_BATCH 4
#endif
// BVH_node::hit / has_hit (BVH.cpp:95-136), AABB::Intersection (AABB.cpp:25-36), Triangle::hit / isIntersect (Triangle.cpp:48-106) for the
// whole ray list of one iteration, over the 8-wide compressed tree (device_scene.h: nodes8; Ylitie, Karras & Laine 2017, re-laid for
// gfx950).  Acceptance rules: see bvh_traverse / tri_test in pt_device.h.  Block = WF_TRACE_BLOCK (1024) threads = 16 waves sharing one LDS
// image of the top MCPT_TOP_NODES8 records + a WF8_LDS_STACK-entry per-lane stack; the host launches one block per CU (mcpt_api.cpp).
// Persistent waves with a three-block scheduler (refill / leaf / inner, below); the unit of traversal:
//   * one inner step = one 80-B record = EIGHT child boxes (8-bit offsets in the node's frame, two FMAs per plane after a v_cvt_f32_ubyteN),
//     the result an 8-bit hit mask -- no entry distances, no sorting network: the children sit in octant slots (scene_build.cpp), a ray
//     visits them in the order of slot ^ octant, which is front to back up to ties;
//   * the traversal state is a GROUP: G = {child base, imask << 8 | pending hit mask (octant-permuted: lowest bit first)} names all pending
//     inner children of one node in 8 B, T = {triangle base, leaf hits << 24 | count planes << 8} all its pending leaf children.  One stack
//     entry per node instead of one per child: the stack is an 8-B-per-level LDS array, WF8_LDS_STACK deep (deeper levels spill to a global
//     overflow area), with at most two pushes and one pop per step;
//   * speculative traversal (Aila & Laine 2009): a lane parks the leaf group of a node in T and goes on with the node's inner children -- the
//     leaf block tests the parked leaves of all lanes at once; the leaf group of a later node found while T is still occupied is pushed UNDER
//     that node's inner group and parked when it is popped.
// LDS budget (r04, profiles/r04_lds_budget.txt): 8 levels x 1024 lanes x 8 B = 64 KB of stack + 12.5 KB of top records + 2 KB = 78.5 KB of static
// LDS per block (gfx950 addresses 160 KB per workgroup; the 64-KB habit of earlier rounds is not a limit here), beside two shade blocks of
// 29.2 KB on the CU.  6 -> 8 levels: -3.4 % on the 4 M-triangle scene (depth-15 tree), -2.3 % at 93 k triangles, -0 ... -1 % on S-cornell;
// 9 levels and more (>= 86 KB), or a larger record image (320 / 420 / 512 records), are SLOWER on every scene (+2 ... +8 %).
#ifndef WF8_LDS_STACK
#define WF8_LDS_STACK 8
#endif
#ifndef MCPT_TOP_NODES8
#define MCPT_TOP_NODES8 160         // records numbered breadth-first by the builder; 160 x 80 B = 12.5 KB of LDS
#endif
// One child box.  The six plane bytes reach the fma as the low bytes of fp16 values 1024 + q (0x64qq: one v_perm_b32 makes two of them) and
// v_fma_mix_f32 converts on the way in -- a perm per two planes + one fma per plane instead of a v_cvt_f32_ubyte + an fma per plane
// (-3 of 16 VALU issues per child).  The 1024 lives in the STORED frame origin (the builders subtract 1024 steps and quantise against the
// rounded result): t = (1024 + q) a + (R0 - o) / d; the offset's rounding (<= 2^-24 x 1024 a = 6e-5 quantisation steps more than a
// plain q a + b would carry) is covered by the builders, which keep every quantised plane >= MCPT_Q_MARGIN = 2^-10 steps outside its box.
typedef _Float16 wf_h2 __attribute__((ext_vector_type(2)));
#define WF8_H(V, K) ((float)(V)[(K) & 1])
#define WF8_CHILD(K, NX, FX, NY, FY, NZ, FZ)                                                                                         \
        {                                                                                                                    \
            const float t0x = fmaf(WF8_H(NX, K), ax, bx), t1x = fmaf(WF8_H(FX, K), ax, bx);                                  \
            const float t0y = fmaf(WF8_H(NY, K), ay, by), t1y = fmaf(WF8_H(FY, K), ay, by);                                  \
            const float t0z = fmaf(WF8_H(NZ, K), az, bz), t1z = fmaf(WF8_H(FZ, K), az, bz);                                  \
            const float tn = fmaxf(fmaxf(t0x, t0y), fmaxf(t0z, 1e-4f));                                                      \
            const float tf = fminf(fminf(t1x, t1y), fminf(t1z, tmax));                                                       \
            WF8_MASK_IN(tn, tf)                                                                                              \
        }
/* m = 2 m + (tn <= tf): the compare's carry shifted in by ONE add-with-carry (slot 7 first, so slot s ends up in bit s).  (Round 4 A/B: the sign of
   tf - tn shifted in by v_sub_f32 + v_alignbit_b32 -- a cheaper pair by the instruction prices of profiles/r04_uarch_probe.txt -- changed nothing.) */
#define WF8_MASK_IN(TN, TF) asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(TN), "v"(TF) : "vcc");
template <int N> struct WfKind { static constexpr int value = N; };
template <bool COUNT>
__global__ void __launch_bounds__(WF_TRACE_BLOCK) wf_trace8_kernel(DevScene sc, PathPool pool, IterCtl* ctl, uint32_t it, WaveTuning tune, DevCounters* gcnt,
                                                                   int* __restrict__ stack_overflow) {
    if (ctl->any_active[it & 3u] == 0u) return;                       // no live slot after this iteration's shade call: no ray to trace (see wf_shade_kernel)
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    typedef float v4f __attribute__((ext_vector_type(4)));
    __shared__ v2u s_stack[WF8_LDS_STACK * WF_TRACE_BLOCK];
    __shared__ float4 s_top[5 * MCPT_TOP_NODES8 + 1];                  // [record][node]  (+1: MCPT_TOP_NODES8 = 0 is a legal A/B setting)
    // explicit address spaces: with generic pointers hipcc folds `lds ? : global` into ONE flat_load (select of pointers), which is slower than
    // either path and hides the LDS traffic from the LDS pipe -- typed pointers keep ds_read / global_load apart
    typedef __attribute__((address_space(3))) v2u lds_u2;
    typedef __attribute__((address_space(3))) v4f lds_f4;
    typedef __attribute__((address_space(1))) v2u glb_u2;
    typedef __attribute__((address_space(1))) const v4f glb_cf4;
    lds_u2* stk = (lds_u2*)s_stack + threadIdx.x;
    lds_f4* top = (lds_f4*)s_top;
    const uint32_t ovf_stride = gridDim.x * WF_TRACE_BLOCK;
#define OVF8(LEVEL) (((glb_u2*)stack_overflow)[(uint32_t)(LEVEL) * ovf_stride + (blockIdx.x * WF_TRACE_BLOCK + threadIdx.x)])
    glb_cf4* gnodes = (glb_cf4*)sc.nodes8;
    const int n_top = sc.n_nodes8 < MCPT_TOP_NODES8 ? sc.n_nodes8 : MCPT_TOP_NODES8;
    for (int i = threadIdx.x; i < 5 * n_top; i += WF_TRACE_BLOCK) s_top[(i % 5) * MCPT_TOP_NODES8 + (i / 5)] = sc.nodes8[i];
    // the octant permutation of a hit mask (bit j <- slot j ^ oct) is one LDS byte read: three conditional swaps in registers cost 11 VALU
    // issues per step and 1.2 % of the step time
    __shared__ unsigned char s_perm[8 * 256];
    for (int i = threadIdx.x; i < 8 * 256; i += WF_TRACE_BLOCK) { const uint32_t oc = (uint32_t)i >> 8, mm = (uint32_t)i & 255u; uint32_t rr = 0; for (uint32_t jj = 0; jj < 8; jj++) rr |= ((mm >> (jj ^ oc)) & 1u) << jj; s_perm[i] = (unsigned char)rr; }
    __syncthreads();
#ifdef WF_SCHED_STATS
    const unsigned long long t_start = wall_clock64();
#endif
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t P = ctl->pad[WF_CTL_P_ACTIVE] ? ctl->pad[WF_CTL_P_ACTIVE] : pool.P;      // (after a drain compaction: the live front of the pool)
    // Ray list = 2 * P / 256 chunks: chunk c < P/256 holds the extend rays of slots [256 c, 256 c + 256); chunk P/256 + b holds the
    // shadow rays shade block b queued (shadow_count[b] of them).  A wave owns one chunk at a time: the first statically (no atomic),
    // later ones from `head`.  A list with fewer chunks than waves (a small call: one sample per pixel of a small film) is handed out in
    // quarter chunks of 64 rays, so that every wave of the grid gets rays and no lane works four of them one after the other.
    const uint32_t n_ext_chunks = P / WF_SHADE_BLOCK;
    const uint32_t n_waves = gridDim.x * (WF_TRACE_BLOCK / 64);
    const uint32_t sub_sh = 2 * n_ext_chunks < n_waves ? 2u : 0u;
    const uint32_t n_chunks = (2 * n_ext_chunks) << sub_sh;
    uint32_t* head = &ctl->trace_head[it & 3];
    uint32_t w_next = 0, w_end = 0, q_base = 0;
    bool chunk_shadow = false, exhausted = false;
    auto take_chunk = [&](uint32_t c) {              // (straight-line on purpose: with early returns the compiler kept w_next / w_end in scratch)
        const uint32_t cc = c >> sub_sh, part = c & ((1u << sub_sh) - 1u), span = (uint32_t)WF_SHADE_BLOCK >> sub_sh;
        const bool none = c >= n_chunks, ext = cc < n_ext_chunks, shadow = !none && !ext;
        const uint32_t b = shadow ? cc - n_ext_chunks : 0u;
        uint32_t cnt = ext ? (uint32_t)WF_SHADE_BLOCK : 0u;
        if (shadow) cnt = s_load_u32(&pool.shadow_count[b]);            // (uniform address, written by the shade launch before this one)
        const uint32_t lo = min(part * span, cnt), hi = min(lo + span, cnt), base = ext ? cc * WF_SHADE_BLOCK : 0u;
        exhausted = exhausted || none;
        chunk_shadow = shadow;
        q_base = b * WF_SHADE_BLOCK;
        w_next = base + lo;
        w_end = base + hi;
    };
    // chunks are reserved `batch` at a time: WF_CHUNK_BATCH when the list is long (one atomic on `head` per ~1-2 k rays per wave).  A short
    // list (a one-sample-per-pixel call: 5 000 chunks for 3 584 waves) would leave most waves -- whole CUs, with block-major wave numbers --
    // without a first chunk: there every wave takes n_chunks / n_waves (at least one) and wave numbers interleave across the blocks.
    const bool short_list = n_chunks < n_waves * WF_CHUNK_BATCH;
    const uint32_t batch = short_list ? max(1u, n_chunks / n_waves) : (uint32_t)WF_CHUNK_BATCH;
    const uint32_t wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar: the chunk bookkeeping stays in SGPRs
    const uint32_t wave_id = short_list ? wave_in_block * gridDim.x + blockIdx.x : blockIdx.x * (WF_TRACE_BLOCK / 64) + wave_in_block;
    uint32_t c_next = wave_id * batch, c_end = c_next + batch;
    take_chunk(c_next++);

    bool have = false, any = false;
    uint32_t blocked = 0u;                                             // any-hit rays: an occluder was found (0 / 1; updated by selects on SGPR masks, see wf_sel)
    uint32_t slot = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1);
    float idx = 0, idy = 0, idz = 0, tmax = 0;
    float nox = 0, noy = 0, noz = 0;                                   // -o * (1 / d): the plane offsets of a node then cost one fma each (the register file has room up to 80)
    // cur = the group on top of the lane's stack, held in registers: .y & 0xff != 0 -> inner group G; .y != 0 otherwise -> a leaf group
    // waiting for the parking place; .y == 0 -> bottom of the stack (the ray is finished once T is empty too).  T = the parked leaf group.
    uint32_t cur_x = 0, cur_y = 0, t_x = 0, t_y = 0, oct = 0;
    uint32_t snx = 0, sfx = 0, sny = 0, sfy = 0, snz = 0, sfz = 0;     // v_perm_b32 selectors of the ray's entry / exit plane bytes per axis (inner_consume)
    int sp = 1;
    int htri = -1; float hu = 0, hv = 0;
    uint32_t hrank = 0;                                                // tie rank of the best hit so far (closest-hit rays)
    uint32_t n_box = 0, n_tri = 0, n_spill = 0;
#ifdef WF_SCHED_STATS
    uint32_t x_inner = 0, x_leaf = 0, x_refill = 0, l_refill = 0;
    unsigned long long t_inner = 0, t_leaf = 0, t_refill = 0, t_mark = __builtin_amdgcn_s_memtime();
#define WF_TICK(acc) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); acc += t_now - t_mark; t_mark = t_now; }
#else
#define WF_TICK(acc)
#endif
#define WF8_POP() { sp--; v2u e_; if (sp < WF8_LDS_STACK) e_ = stk[sp * WF_TRACE_BLOCK]; else e_ = OVF8(sp - WF8_LDS_STACK); cur_x = e_.x; cur_y = e_.y; }
    // a leaf group on top of the stack moves to the parking place as soon as that is free, and the lane goes on with what lies below
#define WF8_PARK() if (t_y == 0u && cur_y != 0u && (cur_y & 0xffu) == 0u) { t_x = cur_x; t_y = cur_y; WF8_POP() }
    const bool speculate = tune.pend_cap != 0u;      // MCPT_WF_PEND=0 (developer knob): a lane with a parked leaf group waits for the leaf block
    const bool greedy = tune.policy == 1;

    // One inner step in two halves: `issue` picks the next child of the group on top of the stack and requests its record, `consume` tests the
    // eight boxes and updates the stack.
    v4f R0, R1, R2, R3, R4;
#if defined(WF_DUMMY_VMEM)
    v4f DM[WF_DUMMY_VMEM];
#elif defined(WF_DUMMY_LDS)
    v4f DM[WF_DUMMY_LDS];
#endif
    bool order_matters = true;
    uint32_t k64 = 0x64646464u; asm volatile("" : "+v"(k64));      // the fp16 exponent byte of WF8_CHILD's plane values, pinned in a VGPR (v_perm_b32 has one constant-bus operand: the selector)
    auto inner_issue = [&]() __attribute__((always_inline)) {
        // next child of the group on top: lowest pending bit j = slot j ^ oct; its record = base + rank among the inner slots
        const uint32_t j = (uint32_t)__builtin_ctz(cur_y);
        const uint32_t s = j ^ oct;
        const uint32_t node = cur_x + (uint32_t)__popc((cur_y >> 8) & ((1u << s) - 1u));
        cur_y &= cur_y - 1u;
        if (node < MCPT_TOP_NODES8) { R0 = top[node]; R1 = top[MCPT_TOP_NODES8 + node]; R2 = top[2 * MCPT_TOP_NODES8 + node]; R3 = top[3 * MCPT_TOP_NODES8 + node]; R4 = top[4 * MCPT_TOP_NODES8 + node]; }
        else { glb_cf4* n = (glb_cf4*)((const __attribute__((address_space(1))) char*)gnodes + node * 80u); R0 = n[0]; R1 = n[1]; R2 = n[2]; R3 = n[3]; R4 = n[4]; }   // (32-bit byte offset from a uniform base: saddr + voffset addressing)
#ifdef WF_DUMMY_VMEM    /* regime probe: N extra 16-B gathers per inner step from the record just requested (L1 hits: pure vector-memory address work) */
        { const volatile __attribute__((address_space(1))) v4f* n = (const volatile __attribute__((address_space(1))) v4f*)((const __attribute__((address_space(1))) char*)gnodes + node * 80u);
#pragma unroll
          for (int k = 0; k < WF_DUMMY_VMEM; k++) DM[k] = n[k % 5]; }
#endif
#ifdef WF_DUMMY_LDS     /* regime probe: N extra 16-B LDS reads per inner step */
        { const volatile lds_f4* n = (const volatile lds_f4*)top;
#pragma unroll
          for (int k = 0; k < WF_DUMMY_LDS; k++) DM[k] = n[(k * MCPT_TOP_NODES8 + node) % (5 * MCPT_TOP_NODES8)]; }
#endif
    };
    auto inner_consume = [&]() __attribute__((always_inline)) {
        const uint32_t sxy = __float_as_uint(R0.w), masks = __float_as_uint(R1.w);
        const float ax = __uint_as_float(sxy & 0xffff0000u) * idx, ay = __uint_as_float(sxy << 16) * idy, az = R1.z * idz;   // (R1.z: the builders leave its low half zero)
        const float bx = fmaf(R0.x, idx, nox), by = fmaf(R0.y, idy, noy), bz = fmaf(R0.z, idz, noz);   // R0.xyz = the frame origin LESS 1024 steps: plane q lies at R0 + (1024 + q) step (WF8_CHILD)
        // Word j of an axis record = the planes of slots 2j, 2j + 1 as bytes { lo, lo, hi, hi } (device_scene.h).  One v_perm_b32 turns the pair's
        // ENTRY planes into the fp16 pair {1024 + q, 1024 + q} and another its EXIT planes; which bytes are "entry" is the ray's business: the
        // selectors sn* / sf* (low bytes for an axis the ray travels along positively, high bytes otherwise) are made once per ray at the refill.
        // (r03 selected whole plane words with 12 v_cndmask per node.)
        uint32_t m = 0u;
        auto pair16 = [&](float w, uint32_t sel) __attribute__((always_inline)) { const uint32_t r = __builtin_amdgcn_perm(k64, __float_as_uint(w), sel); wf_h2 h; __builtin_memcpy(&h, &r, 4); return h; };
        { const wf_h2 NX = pair16(R2.w, snx), FX = pair16(R2.w, sfx), NY = pair16(R3.w, sny), FY = pair16(R3.w, sfy), NZ = pair16(R4.w, snz), FZ = pair16(R4.w, sfz); WF8_CHILD(7, NX, FX, NY, FY, NZ, FZ) WF8_CHILD(6, NX, FX, NY, FY, NZ, FZ) }
        { const wf_h2 NX = pair16(R2.z, snx), FX = pair16(R2.z, sfx), NY = pair16(R3.z, sny), FY = pair16(R3.z, sfy), NZ = pair16(R4.z, snz), FZ = pair16(R4.z, sfz); WF8_CHILD(5, NX, FX, NY, FY, NZ, FZ) WF8_CHILD(4, NX, FX, NY, FY, NZ, FZ) }
        { const wf_h2 NX = pair16(R2.y, snx), FX = pair16(R2.y, sfx), NY = pair16(R3.y, sny), FY = pair16(R3.y, sfy), NZ = pair16(R4.y, snz), FZ = pair16(R4.y, sfz); WF8_CHILD(3, NX, FX, NY, FY, NZ, FZ) WF8_CHILD(2, NX, FX, NY, FY, NZ, FZ) }
        { const wf_h2 NX = pair16(R2.x, snx), FX = pair16(R2.x, sfx), NY = pair16(R3.x, sny), FY = pair16(R3.x, sfy), NZ = pair16(R4.x, snz), FZ = pair16(R4.x, sfz); WF8_CHILD(1, NX, FX, NY, FY, NZ, FZ) WF8_CHILD(0, NX, FX, NY, FY, NZ, FZ) }
        const uint32_t leaf_slots = masks >> 24;                 // = p0 | p1, stored by the builder
#ifndef WF_SCHED_STATS
        if (COUNT) n_box += (uint32_t)__popc((masks & 0xffu) | leaf_slots);
#endif
        uint32_t mi = m & masks & 0xffu;                         // hit inner children, slot order
        const uint32_t ml = m & leaf_slots;                      // hit leaf children (an empty slot's inverted box cannot be hit; the mask keeps that exact)
        if (order_matters) mi = s_perm[(oct << 8) | mi];         // bit j <- slot j ^ oct (wave-uniform branch; identity for any-hit rays, whose oct is 0)
        const uint32_t gn_y = (masks << 8) | mi;                 // (the count planes ride along in bits 16-31; every use masks them off)
        const uint32_t tn_y = (masks & 0x00ffff00u) | (ml << 24);    // the node's leaf group -- meaningful only if ml != 0 (no select: a select on VCC is the dearest VALU instruction here, see wf_sel)
        const bool keep_cur = (cur_y & 0xffu) != 0u;             // siblings of the child just taken are still pending
        const bool t_park = ml != 0u && t_y == 0u, t_push = ml != 0u && t_y != 0u;
        v2u e_cur, e_tn; e_cur.x = cur_x; e_cur.y = cur_y; e_tn.x = __float_as_uint(R1.y); e_tn.y = tn_y;
        if (sp + 2 <= WF8_LDS_STACK) {
            // common case, branch-free: store both candidates, advance the stack pointer only past the real ones
            stk[sp * WF_TRACE_BLOCK] = e_cur; sp += keep_cur ? 1 : 0;
            stk[sp * WF_TRACE_BLOCK] = e_tn;  sp += t_push ? 1 : 0;
            const v2u below_top = stk[(sp - 1) * WF_TRACE_BLOCK];
            if (mi) { cur_x = __float_as_uint(R1.x); cur_y = gn_y; } else { cur_x = below_top.x; cur_y = below_top.y; sp--; }
        } else {                                                   // rare: near the LDS limit -> entries may go to the overflow area
            if (keep_cur) { if (sp < WF8_LDS_STACK) stk[sp * WF_TRACE_BLOCK] = e_cur; else { OVF8(sp - WF8_LDS_STACK) = e_cur; if (COUNT) n_spill++; } sp++; }
            if (t_push) { if (sp < WF8_LDS_STACK) stk[sp * WF_TRACE_BLOCK] = e_tn; else { OVF8(sp - WF8_LDS_STACK) = e_tn; if (COUNT) n_spill++; } sp++; }
            if (mi) { cur_x = __float_as_uint(R1.x); cur_y = gn_y; } else WF8_POP()
        }
        if (t_park) { t_x = e_tn.x; t_y = tn_y; }
        WF8_PARK()
#if defined(WF_DUMMY_VMEM) || defined(WF_DUMMY_LDS)    /* (the probes' results are only kept alive until here) */
#pragma unroll
        for (int k = 0; k < (int)(sizeof(DM) / sizeof(DM[0])); k++) asm volatile("" :: "v"(DM[k]));
#endif
#ifdef WF_DUMMY_VNOP    /* regime probe: N v_nop per inner step -- VALU issue slots without operands, registers or a dependence chain */
        {
#pragma unroll
          for (int k = 0; k < WF_DUMMY_VNOP; k++) asm volatile("v_nop"); }
#endif
#ifdef WF_DUMMY_VALU    /* regime probe: N extra DEPENDENT v_fma_f32 per inner step (one chain: adds issue slots AND ~N x the FMA latency to the wave's critical path) */
        { float dz = idx;
#pragma unroll
          for (int k = 0; k < WF_DUMMY_VALU; k++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(dz));
          asm volatile("" :: "v"(dz)); }
#endif
    };

    uint32_t watchdog = 0;
    for (;;) {
        if (++watchdog > (1u << 24)) { if (lane == 0) ctl->pad[0] = 1u; break; }
        const bool at_leaf = have && t_y != 0u;                                          // carries a parked leaf group
        const bool at_inner = have && (cur_y & 0xffu) != 0u && (speculate || !at_leaf);   // can take an inner step
        const int n_inner = __popcll(__ballot(at_inner)), n_pend = __popcll(__ballot(at_leaf));
        const int n_leaf = __popcll(__ballot(at_leaf && !at_inner));                     // ... and cannot go on without the leaf block
        const int n_idle = 64 - n_inner - n_leaf;

        const int most = n_inner > n_leaf ? n_inner : n_leaf;
        // Which block runs this round -- as an opaque scalar, so that the three blocks stay three `if` regions IN SEQUENCE: written as
        // if / continue the compiler gave the loop one latch with a three-way merge of the ten state registers two blocks modify, and copied
        // all ten into the merge registers at the end of every block and back out of them at the latch (20 v_mov per scheduler round).
        uint32_t sel = ((n_inner + n_leaf == 0) || (!exhausted && (greedy ? n_idle >= most : n_idle >= (int)tune.refill_at))) ? 0u
                     : (n_leaf >= (int)tune.leaf_at || n_inner == 0 || (speculate && n_pend >= (int)tune.pend_cap)) ? 1u : 2u;
        asm volatile("" : "+s"(sel));
        if (sel == 0u) {
            // ------------------------------------------------------------------ refill block
#ifdef WF_SCHED_STATS
            x_refill++; l_refill += (uint32_t)n_idle;
#endif
            if (have && cur_y == 0u && t_y == 0u) {                      // finished: write the result back
                if (any) {
                    if (blocked != 0u) st_s(reinterpret_cast<uint32_t*>(&pool.nee[slot]) + 3, 1u);
                } else {
                    st_s(&pool.hit[slot], make_float4(__int_as_float(htri), hu, hv, tmax));
                }
                have = false;
            }
            // (Tried in round 3 and dropped: an inner step of the busy lanes INSIDE this block's memory round trip -- node records requested
            // before the ray records, boxes tested while the rays are in flight, `s_waitcnt vmcnt(3)`.  The wait worked; the step did not pay:
            // it runs with the ~25 lanes that are busy at a refill, i.e. it ADDS low-occupancy box-test issues to a pipeline that is bound by
            // VALU issue, and the node + ray records in flight together cost 30 registers: 234 vs 218 ms per 512 spp.)
            if (!exhausted) {
                const uint64_t m_idle = __ballot(!have);
                const uint32_t rank = lane_rank(m_idle);
                uint32_t remaining = (uint32_t)__popcll(m_idle), assigned = 0;
                bool got = false, my_shadow = false; uint32_t my_w = 0, my_q = 0;
                for (int pass = 0; pass < 4 && remaining > 0; pass++) {
                    if (w_next == w_end) {
                        if (c_next == c_end) {
                            uint32_t c = 0;
                            if (lane == 0) c = atomicAdd(head, batch);
                            c_next = wave_first(c) + n_waves * batch; c_end = c_next + batch;
                        }
                        take_chunk(c_next++);
                        if (exhausted) break;
                    }
                    const uint32_t take = min(w_end - w_next, remaining);
                    if (!have && !got && rank >= assigned && rank < assigned + take) { got = true; my_w = w_next + (rank - assigned); my_shadow = chunk_shadow; my_q = q_base; }
                    w_next += take; assigned += take; remaining -= take;
                }
                if (got) {
                    bool valid;
                    if (!my_shadow) {
                        slot = my_w;
                        const float4 rd = ld_s(&pool.ray_d[my_w]);
                        valid = (__float_as_uint(rd.w) & 1u) != 0u;
                        const float4 ro = ld_s(&pool.ray_o[my_w]);
                        o = xyz(ro); d = xyz(rd); tmax = 3.0e38f; any = false; htri = -1; hrank = 0u;
                    } else {
                        slot = ld_s(&pool.shadow_queue[my_q + my_w]);
                        const float4 ro = ld_s(&pool.sq_o[my_q + my_w]), sd = ld_s(&pool.sq_d[my_q + my_w]);
                        o = xyz(ro); d = xyz(sd); tmax = sd.w; any = true; htri = __float_as_int(ro.w); valid = true;
                    }
                    if (valid) {
                        const float tiny = 1e-30f;
                        // v_rcp_f32 (1 ulp) instead of the IEEE divide sequence (~10 VALU issues each, executed for the ~19 lanes a refill feeds):
                        // the child boxes carry ~16 ulp of padding for exactly this kind of rounding in the slab arithmetic
                        idx = __builtin_amdgcn_rcpf(wf_sel(wf_mask(fabsf(d.x) > tiny), d.x, copysignf(tiny, d.x)));
                        idy = __builtin_amdgcn_rcpf(wf_sel(wf_mask(fabsf(d.y) > tiny), d.y, copysignf(tiny, d.y)));
                        idz = __builtin_amdgcn_rcpf(wf_sel(wf_mask(fabsf(d.z) > tiny), d.z, copysignf(tiny, d.z)));
                        nox = -o.x * idx; noy = -o.y * idy; noz = -o.z * idz;
                        // bytes {0, 1} of a plane word are a slot pair's low planes, {2, 3} its high planes; 0x64 = the fp16 exponent byte from k64
                        constexpr uint32_t SEL_LO = 0x04010400u, SEL_HI = 0x04030402u;
                        const uint64_t mnx = wf_mask(idx < 0.0f), mny = wf_mask(idy < 0.0f), mnz = wf_mask(idz < 0.0f);
                        snx = wf_sel(mnx, SEL_HI, SEL_LO); sfx = wf_sel(mnx, SEL_LO, SEL_HI);
                        sny = wf_sel(mny, SEL_HI, SEL_LO); sfy = wf_sel(mny, SEL_LO, SEL_HI);
                        snz = wf_sel(mnz, SEL_HI, SEL_LO); sfz = wf_sel(mnz, SEL_LO, SEL_HI);
                        // the visiting order only matters to closest-hit rays (Render.cpp:125 asks whether the light is visible at all):
                        // any-hit rays keep octant 0, i.e. slot order -- and a wave that carries no closest-hit ray skips the permutation
                        oct = wf_sel(wf_mask(any), 0u, wf_sel(mnx, 1u, 0u) | wf_sel(mny, 2u, 0u) | wf_sel(mnz, 4u, 0u));
                        v2u bottom; bottom.x = 0u; bottom.y = 0u;
                        stk[0] = bottom; sp = 1;
                        cur_x = 0u; cur_y = 1u;                          // "child 0 of base 0, no inner siblings": the root
                        t_x = 0u; t_y = 0u;
                        hu = 0.f; hv = 0.f; blocked = 0u;
                        have = true;
                    }
                }
            }
            WF_TICK(t_refill)
            if (exhausted && __ballot(have) == 0) break;
        }

        if (sel == 1u) {
            // ------------------------------------------------------------------ leaf block: every lane with a parked leaf group tests ONE of its leaves
#ifdef WF_SCHED_STATS
            x_leaf++; if (at_leaf) n_tri++;
#endif
            if (at_leaf) {
                const uint32_t s = (uint32_t)__builtin_ctz(t_y >> 24);                   // leaves in slot order (any order gives the same result)
                const uint32_t below = (1u << s) - 1u;
                const uint32_t first = t_x + (uint32_t)__popc((t_y >> 8) & below) + 2u * (uint32_t)__popc((t_y >> 16) & below);
                const uint32_t cnt = ((t_y >> (8u + s)) & 1u) + 2u * ((t_y >> (16u + s)) & 1u);
                t_y &= ~(1u << (24u + s));
                if ((t_y >> 24) == 0u) t_y = 0u;
                uint64_t done = 0ull;                                    // lane mask: any-hit rays that found their occluder in this leaf
                // Both acceptance rules (Triangle::hit / Triangle::isIntersect, see tri_accept_* in pt_device.h) as ONE predicate and the hit-state
                // update as four selects: spelled with short-circuit `&&` and `if` the compiler built a branch per term and copied the hit state
                // (tmax, triangle, u, v) at every join -- ~12 of ~64 VALU issues per triangle.  `u <= 1` of the any-hit rule is implied by
                // v >= 0 and fl(u + v) <= 1 (rounding is monotone) and is not tested separately.
                // KIND: 2 = every lane of this execution carries an any-hit ray (no closest-hit rule, none of its selects), 0 = any mix -- the two forms
                // that are instantiated.  (1 = "all closest-hit" is spelled out below but NOT instantiated: measured in round 3, a third form took the
                // kernel from 74 to 86 registers and the closest-only + mixed pair alone was 2.5 % slower than the mixed form doing both jobs.)
                // The verdicts live as LANE MASKS in SGPR pairs (accept, update, occluded) and every per-lane consequence is a select on such a mask.
                auto leaf_test = [&](auto kind_c, const float4 v0, const float4 e1, const float4 e2, const int ti, const bool use, const uint64_t skip_lanes) __attribute__((always_inline)) {
                    constexpr int kind = decltype(kind_c)::value;
                    const bool is_any = kind == 2 ? true : (kind == 1 ? false : any);
                    const TriTest r = tri_test(v0, e1, e2, o, d);
                    const bool a_ok = fabsf(r.a) >= (is_any ? 1e-6f : 1e-5f);
                    const bool uv_ok = (r.u >= 0.0f) & (r.v >= 0.0f) & (is_any ? (r.u + r.v <= 1.0f) : ((1.0f - r.u - r.v) >= 0.0f));
                    // a hit at EXACTLY the distance of the best one so far wins iff its tie rank (low 28 bits of v0.w: the triangle's place in the
                    // leaf order, or in the reference's own triangle order -- MCPT_FLAG_REFERENCE_TIE_ORDER) is lower; hrank = the best hit's rank
                    const uint32_t rank = __float_as_uint(v0.w) & HIT_TRI_MASK;
                    const bool tie_win = (r.t == tmax) & (rank < hrank);
                    const bool t_ok = (r.t >= 1e-4f) & (is_any ? (r.t <= tmax) : ((r.t < tmax) | tie_win));
                    const uint64_t acc = wf_mask(use & a_ok & uv_ok & t_ok) & ~skip_lanes;
                    const uint64_t m_any = kind == 2 ? ~0ull : (kind == 1 ? 0ull : wf_mask(any));
                    const uint64_t upd = acc & ~m_any, occ = acc & m_any;
                    if (kind != 2) {
                        tmax = wf_sel(upd, r.t, tmax); hu = wf_sel(upd, r.u, hu); hv = wf_sel(upd, r.v, hv);
                        htri = wf_sel(upd, ti | (__float_as_int(v0.w) & ~HIT_TRI_MASK), htri);   // v0.w = lobe class << 28 | tie rank
                        hrank = wf_sel(upd, rank, hrank);
                    }
                    blocked = wf_sel(occ, 1u, blocked);
                    return occ;
                };
                constexpr bool one_pair = MCPT_LEAF_MAX <= 2;            // a leaf holds at most MCPT_LEAF_MAX triangles: with two the pair loop is one pass
                uint32_t i = 0;
#pragma unroll 1
                do {
                    const int ta = (int)(first + i), tb = ta + 1;
                    const bool use_a = !(any && ta == htri), use_b = i + 1 < cnt && !(any && tb == htri);   // any-hit rays keep their `skip` triangle in htri
                    const float4* T = (const float4*)((const char*)sc.tri_isect + (uint32_t)ta * 48u);   // (n_tris < 2^28 x 48 B would overflow 32 bits: checked at launch)
                    // triangle a is fetched whether or not it is the ray's `skip` triangle (its test is skipped, not its load: no zero-initialised
                    // merge registers, -1 ... -2 % step time); fetching b unconditionally as well is slower even where the registers are there
                    // (68 without the SLP vectoriser: 204.2 vs 201.4 ms -- half of the leaves have one triangle and the load is not free)
                    const float4 v0a = T[0], e1a = T[1], e2a = T[2];
                    const float4* Tb = use_b ? T + 3 : T;                // (a lane without b reads a's record again: the same cache lines, no merge registers; r04 A/B: all such lanes reading ONE fixed record instead: +-0)
                    const float4 v0b = Tb[0], e1b = Tb[1], e2b = Tb[2];
#ifndef WF_SCHED_STATS
                    if (COUNT) n_tri += (use_a ? 1u : 0u) + (use_b ? 1u : 0u);
#endif
                    const uint64_t m_kind = __ballot(any), m_here = __ballot(true);
                    if (m_kind == m_here) {
                        done |= leaf_test(WfKind<2>{}, v0a, e1a, e2a, ta, use_a, done);
                        if (__ballot(use_b) != 0) done |= leaf_test(WfKind<2>{}, v0b, e1b, e2b, tb, use_b, done);
                    } else {
                        done |= leaf_test(WfKind<0>{}, v0a, e1a, e2a, ta, use_a, done);
                        if (__ballot(use_b) != 0) done |= leaf_test(WfKind<0>{}, v0b, e1b, e2b, tb, use_b, done);
                    }
                    i += 2;
                } while (!one_pair && i < cnt && ((done >> lane) & 1ull) == 0ull);
                cur_y = wf_sel(done, 0u, cur_y); t_y = wf_sel(done, 0u, t_y);   // any-hit: stop at the first occluder
                WF8_PARK()                                               // the group is worked off and another one was waiting on top of the stack (never true for a lane that just stopped: its cur_y is 0)
            }
            WF_TICK(t_leaf)
        }

        if (sel != 2u) continue;
        // ---------------------------------------------------------------------- inner-node block: one 80-B record = eight child boxes
        int keep = (int)tune.inner_keep;
        order_matters = __ballot(have && !any) != 0;
        do {
#ifdef WF_SCHED_STATS
            x_inner++; if (at_inner) n_box++;
#endif
            const bool step = have && (cur_y & 0xffu) != 0u && (speculate || t_y == 0u);
            if (step) { inner_issue(); inner_consume(); }
            const bool still = have && (cur_y & 0xffu) != 0u && (speculate || t_y == 0u);
            if (greedy) {
                const int cl = __popcll(__ballot(have && t_y != 0u && !still)), ci = __popcll(__ballot(still));
                const int cf = exhausted ? 0 : 64 - cl - ci;
                keep = (cl > cf ? cl : cf) + 1;
            }
            if (__popcll(__ballot(still)) < keep) break;
        } while (true);
        WF_TICK(t_inner)
    }

    if (COUNT) {
        unsigned long long b = n_box, t = n_tri, sx = n_spill;
        for (int off = 32; off > 0; off >>= 1) { b += __shfl_xor(b, off, 64); t += __shfl_xor(t, off, 64); sx += __shfl_xor(sx, off, 64); }
        if (lane == 0) {
            DevCounters* g = gcnt + (blockIdx.x & (WF_COUNTER_REPLICAS - 1)); atomicAdd(&g->box_tests, b); atomicAdd(&g->tri_tests, t);
            if (sx) atomicAdd(&g->stack_spills, sx);
#ifdef WF_SCHED_STATS   // tools/sched_stats.py: the shade-side counters are re-purposed in this diagnostic build
            atomicAdd(&g->paths, wall_clock64() - t_start);
            atomicAdd(&g->shaded_hits, (unsigned long long)x_inner); atomicAdd(&g->texel_fetches, (unsigned long long)x_leaf);
            atomicAdd(&g->self_shadow_tests, (unsigned long long)x_refill); atomicAdd(&g->self_shadow_hits, (unsigned long long)l_refill);
            atomicAdd(&g->debug[0], t_inner); atomicAdd(&g->debug[1], t_leaf); atomicAdd(&g->debug[2], t_refill);
#endif
        }
    }
}
#undef OVF8
#undef WF8_CHILD
#undef WF8_MASK_IN
#undef WF8_H
#undef WF8_POP
#undef WF8_PARK
#undef WF_TICK
// ====================================================================================================== drain compaction (wavefront.h: CompactBufs)
// Three small launches behind a trace launch, enqueued by the host once it has seen the shared work-item cursors run out; the first decides on the
// device whether anything happens at all.
//   plan  (one block)      every work item handed out (shared cursors AND every block's private range) and at most half of the swept slots alive?
//                          -> exclusive prefix of the blocks' live counts = where each block's live slots go; the new sweep length (a multiple of 4096)
//   move  (block per 256)  live slots -> scratch, in slot order
//   back  (block per 256)  scratch -> the front of the pool; the slots between the last live one and the new sweep length are marked DEAD
__global__ void __launch_bounds__(1024) wf_compact_plan_kernel(PathPool pool, CompactBufs cb, IterCtl* ctl, uint32_t it, uint32_t n_shared, uint32_t priv_items) {
    const uint32_t tid = threadIdx.x;
    __shared__ uint32_t s_sum[1024];
    __shared__ uint32_t s_veto;
    if (tid == 0) { s_veto = 0u; ctl->pad[WF_CTL_DO_COMPACT] = 0u; }
    __syncthreads();
    const uint32_t p_act = ctl->pad[WF_CTL_P_ACTIVE] ? ctl->pad[WF_CTL_P_ACTIVE] : pool.P;
    if (ctl->any_active[it & 3u] == 0u || p_act < WF_COMPACT_MIN_SLOTS || it == 0u) return;      // job over / pool small (uniform: no barrier is skipped by part of the block)
    const uint32_t nb = p_act / WF_SHADE_BLOCK, per = (nb + 1023u) / 1024u;
    bool veto = false;
    if (tid < WF_ITEM_SHARDS) veto = __hip_atomic_load(&ctl->item_cursor[tid].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wf_shard_capacity(n_shared, tid);
    uint32_t mine = 0;
    for (uint32_t b = tid * per; b < min(nb, (tid + 1u) * per); b++) {
        mine += pool.live_cnt[b];
        if (priv_items) { const uint2 r = pool.block_items[b]; veto = veto || r.x != r.y; }          // a block still holds private work items
    }
    if (veto) s_veto = 1u;
    s_sum[tid] = mine;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {                                                // inclusive scan (Hillis-Steele; 10 rounds, once per drain iteration)
        const uint32_t v = tid >= off ? s_sum[tid - off] : 0u;
        __syncthreads();
        s_sum[tid] += v;
        __syncthreads();
    }
    const uint32_t total = s_sum[1023];
    if (s_veto != 0u || (unsigned long long)total * 8ull > (unsigned long long)p_act * cb.eighths || total > cb.capacity) return;
    uint32_t run = s_sum[tid] - mine;
    for (uint32_t b = tid * per; b < min(nb, (tid + 1u) * per); b++) { cb.dst_off[b] = run; run += pool.live_cnt[b]; }
    if (tid == 0) {
        ctl->pad[WF_CTL_LIVE] = total;
        ctl->pad[WF_CTL_P_NEXT] = max((total + 4095u) & ~4095u, 4096u);
        ctl->pad[WF_CTL_DO_COMPACT] = 1u;
    }
}
__global__ void __launch_bounds__(WF_SHADE_BLOCK) wf_compact_move_kernel(PathPool pool, CompactBufs cb, IterCtl* ctl) {
    if (ctl->pad[WF_CTL_DO_COMPACT] == 0u) return;
    const uint32_t p_act = ctl->pad[WF_CTL_P_ACTIVE] ? ctl->pad[WF_CTL_P_ACTIVE] : pool.P;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, slot = blockIdx.x * WF_SHADE_BLOCK + tid;
    if (blockIdx.x * WF_SHADE_BLOCK >= p_act) return;
    __shared__ uint32_t s_cnt[WF_SHADE_BLOCK / 64];
    const float4 bt = ld_s(&pool.beta[slot]);
    const bool live = (__float_as_uint(bt.w) & 3u) != SLOT_DEAD;
    const uint64_t m = __ballot(live);
    if (lane == 0) s_cnt[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    if (!live) return;
    uint32_t before = 0;
    for (uint32_t k = 0; k < wv; k++) before += s_cnt[k];
    const uint32_t dst = cb.dst_off[blockIdx.x] + before + lane_rank(m);
    const uint2* id_ps = reinterpret_cast<const uint2*>(pool.ids);
    st_s(&cb.beta[dst], bt); st_s(&cb.L[dst], ld_s(&pool.L[slot])); st_s(&cb.ray_d[dst], ld_s(&pool.ray_d[slot])); st_s(&cb.ray_o[dst], ld_s(&pool.ray_o[slot]));
    st_s(&cb.hit[dst], ld_s(&pool.hit[slot])); st_s(&cb.nee[dst], ld_s(&pool.nee[slot])); st_s(&cb.ids[dst], ld_s(&id_ps[slot]));
}
__global__ void __launch_bounds__(WF_SHADE_BLOCK) wf_compact_back_kernel(PathPool pool, CompactBufs cb, IterCtl* ctl) {
    if (ctl->pad[WF_CTL_DO_COMPACT] == 0u) return;
    const uint32_t total = ctl->pad[WF_CTL_LIVE], p_next = ctl->pad[WF_CTL_P_NEXT];
    const uint32_t i = blockIdx.x * WF_SHADE_BLOCK + threadIdx.x;
    uint2* id_ps = reinterpret_cast<uint2*>(pool.ids);
    if (i < total) {
        st_s(&pool.beta[i], ld_s(&cb.beta[i])); st_s(&pool.L[i], ld_s(&cb.L[i])); st_s(&pool.ray_d[i], ld_s(&cb.ray_d[i])); st_s(&pool.ray_o[i], ld_s(&cb.ray_o[i]));
        st_s(&pool.hit[i], ld_s(&cb.hit[i])); st_s(&pool.nee[i], ld_s(&cb.nee[i])); st_s(&id_ps[i], ld_s(&cb.ids[i]));
    } else if (i < p_next) {                                                                        // padding up to the new sweep length: dead, no ray pending
        st_s(&pool.beta[i], make_float4(0.f, 0.f, 0.f, __uint_as_float(SLOT_DEAD))); st_s(&pool.ray_d[i], make_float4(0.f, 0.f, 1.f, 0.f));
    }
    if (i == 0) { ctl->pad[WF_CTL_P_ACTIVE] = p_next; ctl->pad[WF_CTL_COMPACTIONS] += 1u; }          // (no other thread of this launch reads it)
}
hipError_t launch_wf_compact(const PathPool& pool, const CompactBufs& cb, IterCtl* ctl, uint32_t iteration, uint32_t n_shared, uint32_t priv_items, hipStream_t stream) {
    hipLaunchKernelGGL(wf_compact_plan_kernel, dim3(1), dim3(1024), 0, stream, pool, cb, ctl, iteration, n_shared, priv_items);
    hipLaunchKernelGGL(wf_compact_move_kernel, dim3(pool.P / WF_SHADE_BLOCK), dim3(WF_SHADE_BLOCK), 0, stream, pool, cb, ctl);
    hipLaunchKernelGGL(wf_compact_back_kernel, dim3((cb.capacity + WF_SHADE_BLOCK - 1) / WF_SHADE_BLOCK), dim3(WF_SHADE_BLOCK), 0, stream, pool, cb, ctl);
    return hipGetLastError();
}
// Start of a job: every slot DEAD (beta.w = state 0), no ids, no partial sums, a zeroed control block -- ONE launch per sub-pipeline where four
// hipMemsetAsync stood in line (a one-sample frame is a chain of ~25 dependent launches: four fewer per sub-pipeline).
__global__ void __launch_bounds__(256) wf_pool_reset_kernel(PathPool pool, IterCtl* ctl) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < pool.P) { st_s(&pool.beta[i], z); st_s(&pool.sum[i], z); st_s(&pool.ids[i], make_uint4(0u, 0u, 0u, 0u)); }
    if (i < sizeof(IterCtl) / 4) reinterpret_cast<uint32_t*>(ctl)[i] = 0u;
}
hipError_t launch_wf_pool_reset(const PathPool& pool, IterCtl* ctl, hipStream_t stream) {
    static_assert(sizeof(IterCtl) % 4 == 0 && sizeof(IterCtl) / 4 <= 256 * 16, "the control block is cleared by the first blocks of the reset grid");
    const uint32_t n = pool.P > uint32_t(sizeof(IterCtl) / 4) ? pool.P : uint32_t(sizeof(IterCtl) / 4);
    hipLaunchKernelGGL(wf_pool_reset_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, pool, ctl);
    return hipGetLastError();
}
// ====================================================================================================== launchers
// RenderParams::div_*: multiplier and shift of an exact x / d for x < 2^30 (d = 0 is treated as 1: such a call has no work items)
void wf_make_fastdiv(uint32_t d, uint32_t& m, uint32_t& s) {
    if (d == 0u) d = 1u;
    uint32_t L = 0; while ((1ull << L) < d) L++;                          // ceil(log2 d)
    s = 30u + L;
    m = uint32_t(((1ull << s) / d) + 1ull);                               // < 2^31 + 2
}
hipError_t launch_wf_shade(const DevScene& sc, const RenderParams& p, const PathPool& pool, IterCtl* ctl, uint32_t iteration, uint32_t n_items,
                           float4* accum, DevCounters* cnt, hipStream_t stream) {
    const dim3 grid(pool.P / WF_SHADE_BLOCK), block(WF_SHADE_BLOCK);
    RenderParams q = p;
    wf_make_fastdiv(p.n_owned, q.div_owned_m, q.div_owned_s);
    wf_make_fastdiv(p.tiles_x, q.div_tiles_x_m, q.div_tiles_x_s);
    wf_make_fastdiv((uint32_t)sc.cam.width, q.div_width_m, q.div_width_s);
    if (p.flags & MCPT_FLAG_COUNT_TRAVERSAL) hipLaunchKernelGGL(wf_shade_kernel<true>, grid, block, 0, stream, sc, q, pool, ctl, iteration, n_items, accum, cnt);
    else hipLaunchKernelGGL(wf_shade_kernel<false>, grid, block, 0, stream, sc, q, pool, ctl, iteration, n_items, accum, cnt);
    return hipGetLastError();
}
hipError_t launch_wf_trace(const DevScene& sc, const PathPool& pool, IterCtl* ctl, uint32_t iteration, const WaveTuning& tune, bool count,
                           DevCounters* cnt, uint32_t grid_blocks, int* stack_overflow, hipStream_t stream) {
    if (count) hipLaunchKernelGGL(wf_trace8_kernel<true>, dim3(grid_blocks), dim3(WF_TRACE_BLOCK), 0, stream, sc, pool, ctl, iteration, tune, cnt, stack_overflow);
    else hipLaunchKernelGGL(wf_trace8_kernel<false>, dim3(grid_blocks), dim3(WF_TRACE_BLOCK), 0, stream, sc, pool, ctl, iteration, tune, cnt, stack_overflow);
    return hipGetLastError();
}
int wf_trace_blocks_per_cu(bool count) {
    int n = 0;
    const hipError_t e = count ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wf_trace8_kernel<true>, WF_TRACE_BLOCK, 0)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wf_trace8_kernel<false>, WF_TRACE_BLOCK, 0);
    if (e != hipSuccess || n <= 0) n = 1;
    return n;
}
uint32_t wf_trace_block_threads() { return WF_TRACE_BLOCK; }
// Bytes of the global stack-overflow area per trace lane: one 8-B group per level for the pending siblings plus one for a leaf group found
// while another is parked.
size_t wf_trace_overflow_bytes_per_lane(uint32_t depth) {
    const uint32_t need = 2 * depth + 3;
    return size_t(need > WF8_LDS_STACK ? need - WF8_LDS_STACK : 1) * 8;
}
