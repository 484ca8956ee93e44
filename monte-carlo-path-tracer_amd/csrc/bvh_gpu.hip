// GPU construction of the binary BVH: the MI355X replacement for BVH::BVH / BVH::build (BVH.cpp:6-54) when the
// scene is large enough for the host's binned-SAH builder to matter (2.2 s for 4 M triangles).  Linear BVH:
//
//   1. morton_kernel     63-bit Morton code (21 bits per axis) of every triangle's box centre inside the centroid bounds
//   2. rocprim radix sort of (code, triangle) pairs -- the sorted order IS the leaf order of the tree
//   3. hierarchy_kernel  Karras 2012: every internal node finds its key range and split from the common-prefix lengths of its
//                        neighbours (equal codes are told apart by their position), all n-1 nodes in parallel
//   4. refit_kernel      bottom-up: one thread per triangle climbs towards the root; the second thread to arrive at a node
//                        (atomic counter) owns it, merges the children's boxes and depths, and continues
//   5. emit_kernel       Karras nodes covering <= MCPT_LEAF_MAX triangles become leaves; the others are compacted (exclusive scan)
//                        and written in the host builder's 64-B node format (both child boxes in the parent, child codes)
//
// Everything downstream (breadth-first renumbering, collapse to the 8-wide quantised tree, triangle streams in leaf order) is the
// same host code that follows the SAH builder.  Traversal results do not depend on the tree (closest hit = min t), only its cost
// does: LBVH trees cost ~1.3x the SAH tree's node visits (S-bath 0.59 M: +17 % render time), so MCPT_FLAG_GPU_BVH_BUILD uses the
// SAH-costed PLOC builder further down (gpu_build_ploc: +4 %) and this one is kept behind MCPT_GPU_BVH=lbvh.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "bvh_gpu.h"
#include "device_scene.h"

namespace {

struct DBuf {
    void* p = nullptr;
    ~DBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 4); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

__device__ __forceinline__ uint64_t expand21(uint64_t v) {   // spread 21 bits to every third bit
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

__global__ void morton_kernel(const float* __restrict__ boxes, uint32_t n, float3 lo, float3 inv_ext, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* b = boxes + 6 * (size_t)i;
    const float cx = 0.5f * (b[0] + b[3]), cy = 0.5f * (b[1] + b[4]), cz = 0.5f * (b[2] + b[5]);
    const float s = 2097152.0f;   // 2^21
    const uint64_t x = (uint64_t)fminf(fmaxf((cx - lo.x) * inv_ext.x * s, 0.f), s - 1.f);
    const uint64_t y = (uint64_t)fminf(fmaxf((cy - lo.y) * inv_ext.y * s, 0.f), s - 1.f);
    const uint64_t z = (uint64_t)fminf(fmaxf((cz - lo.z) * inv_ext.z * s, 0.f), s - 1.f);
    keys[i] = (expand21(x) << 2) | (expand21(y) << 1) | expand21(z);
    vals[i] = i;
}

// common-prefix length of the keys at sorted positions i and j; equal keys are distinguished by their positions (Karras 2012, §4)
__device__ __forceinline__ int delta(const uint64_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)(a ^ b));
}

#define LEAF_BIT 0x80000000u   // child reference: bit 31 set = triangle position in sorted order, else internal node index

__global__ void hierarchy_kernel(const uint64_t* __restrict__ keys, int n, uint32_t* __restrict__ left, uint32_t* __restrict__ right,
                                 uint32_t* __restrict__ parent_of_internal, uint32_t* __restrict__ parent_of_leaf,
                                 uint32_t* __restrict__ first, uint32_t* __restrict__ last) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    first[i] = (uint32_t)lo; last[i] = (uint32_t)hi;
    if (lo == gamma) { left[i] = LEAF_BIT | (uint32_t)gamma; parent_of_leaf[gamma] = (uint32_t)i; }
    else { left[i] = (uint32_t)gamma; parent_of_internal[gamma] = (uint32_t)i; }
    if (hi == gamma + 1) { right[i] = LEAF_BIT | (uint32_t)(gamma + 1); parent_of_leaf[gamma + 1] = (uint32_t)i; }
    else { right[i] = (uint32_t)(gamma + 1); parent_of_internal[gamma + 1] = (uint32_t)i; }
}

struct Box6 { float lx, ly, lz, hx, hy, hz; };
__device__ __forceinline__ Box6 load_tri_box(const float* __restrict__ boxes, const uint32_t* __restrict__ order, uint32_t pos) {
    const float* b = boxes + 6 * (size_t)order[pos];
    return Box6{b[0], b[1], b[2], b[3], b[4], b[5]};
}
__device__ __forceinline__ Box6 merge(const Box6& a, const Box6& b) {
    return Box6{fminf(a.lx, b.lx), fminf(a.ly, b.ly), fminf(a.lz, b.lz), fmaxf(a.hx, b.hx), fmaxf(a.hy, b.hy), fmaxf(a.hz, b.hz)};
}

// device-scope loads: a neighbouring record of the same cache line may sit stale in this CU's L1
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ Box6 load_node_box(const Box6* p) {
    const uint32_t* u = reinterpret_cast<const uint32_t*>(p);
    return Box6{__uint_as_float(ld_agent(u)), __uint_as_float(ld_agent(u + 1)), __uint_as_float(ld_agent(u + 2)),
                __uint_as_float(ld_agent(u + 3)), __uint_as_float(ld_agent(u + 4)), __uint_as_float(ld_agent(u + 5))};
}
// node_box / node_depth are written by the thread that arrives second at a node and read by whoever arrives second at its parent:
// the __threadfence() pairs around the counter make those writes visible (same pattern as Karras 2012 §5).
__global__ void refit_kernel(const float* __restrict__ boxes, const uint32_t* __restrict__ order, int n, const uint32_t* __restrict__ left,
                             const uint32_t* __restrict__ right, const uint32_t* __restrict__ parent_of_internal,
                             const uint32_t* __restrict__ parent_of_leaf, const uint32_t* __restrict__ first, const uint32_t* __restrict__ last,
                             uint32_t* __restrict__ arrivals, Box6* node_box, uint32_t* node_depth, uint32_t leaf_max) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t cur = parent_of_leaf[k];
    for (;;) {
        __threadfence();
        if (atomicAdd(&arrivals[cur], 1u) == 0u) return;            // the sibling subtree is not finished yet: its thread will continue
        __threadfence();
        const uint32_t l = left[cur], r = right[cur];
        const Box6 bl = (l & LEAF_BIT) ? load_tri_box(boxes, order, l & ~LEAF_BIT) : load_node_box(node_box + l);
        const Box6 br = (r & LEAF_BIT) ? load_tri_box(boxes, order, r & ~LEAF_BIT) : load_node_box(node_box + r);
        const uint32_t dl = (l & LEAF_BIT) ? 0u : ld_agent(node_depth + l), dr = (r & LEAF_BIT) ? 0u : ld_agent(node_depth + r);
        node_box[cur] = merge(bl, br);
        // depth in OUTPUT inner nodes: a Karras node spanning <= leaf_max triangles becomes a leaf (depth 0)
        node_depth[cur] = (last[cur] - first[cur] + 1u > leaf_max) ? 1u + (dl > dr ? dl : dr) : 0u;
        if (cur == 0u) return;
        cur = parent_of_internal[cur];
    }
}

__global__ void flag_kernel(const uint32_t* __restrict__ first, const uint32_t* __restrict__ last, int n_internal, uint32_t leaf_max, uint32_t* __restrict__ is_inner) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_internal) is_inner[i] = (last[i] - first[i] + 1u > leaf_max) ? 1u : 0u;
}

__device__ __forceinline__ int leaf_code_dev(uint32_t first, uint32_t count) { return ~(int)((first << 3) | count); }

__global__ void emit_kernel(const float* __restrict__ boxes, const uint32_t* __restrict__ order, int n_internal, const uint32_t* __restrict__ left,
                            const uint32_t* __restrict__ right, const uint32_t* __restrict__ first, const uint32_t* __restrict__ last,
                            const uint32_t* __restrict__ is_inner, const uint32_t* __restrict__ new_id, const Box6* __restrict__ node_box,
                            uint32_t leaf_max, float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_internal || !is_inner[i]) return;
    Box6 b[2]; int code[2];
    const uint32_t ch[2] = {left[i], right[i]};
    for (int k = 0; k < 2; k++) {
        const uint32_t c = ch[k];
        if (c & LEAF_BIT) { b[k] = load_tri_box(boxes, order, c & ~LEAF_BIT); code[k] = leaf_code_dev(c & ~LEAF_BIT, 1u); }
        else {
            b[k] = node_box[c];
            code[k] = is_inner[c] ? (int)new_id[c] : leaf_code_dev(first[c], last[c] - first[c] + 1u);
        }
        // same padding rule as the host builder (scene_build.cpp write_node): ~16 ulp of the largest coordinate
        const float m = fmaxf(fmaxf(fmaxf(fabsf(b[k].lx), fabsf(b[k].hx)), fmaxf(fabsf(b[k].ly), fabsf(b[k].hy))), fmaxf(fabsf(b[k].lz), fabsf(b[k].hz)));
        const float pad = m * 1e-6f + 1e-30f;
        b[k].lx -= pad; b[k].ly -= pad; b[k].lz -= pad; b[k].hx += pad; b[k].hy += pad; b[k].hz += pad;
    }
    float4* o = out + 4 * (size_t)new_id[i];
    o[0] = make_float4(b[0].lx, b[0].hx, b[0].ly, b[0].hy);
    o[1] = make_float4(b[1].lx, b[1].hx, b[1].ly, b[1].hy);
    o[2] = make_float4(b[0].lz, b[0].hz, b[1].lz, b[1].hz);
    o[3] = make_float4(__int_as_float(code[0]), __int_as_float(code[1]), 0.f, 0.f);
}

// ---------------------------------------------------------------------------------------------- PLOC (SAH-costed agglomeration)
// Parallel locally-ordered clustering (Meister & Bittner 2018) over the Morton-sorted triangles: every cluster looks PLOC_RADIUS
// positions to either side for the partner that minimises the SURFACE AREA of the merged box (the SAH's cost term), mutual nearest
// neighbours merge, the array is compacted, repeat until one cluster is left.  Unlike the Karras tree above, whose splits are the
// bits of a space-filling curve, every merge here is chosen by area -- the tree quality of a sweep-SAH build at Morton-sort speed.
#ifndef PLOC_RADIUS
#define PLOC_RADIUS 16
#endif
__device__ __forceinline__ float half_area(const Box6& b) { const float x = b.hx - b.lx, y = b.hy - b.ly, z = b.hz - b.lz; return x * y + y * z + z * x; }

__global__ void ploc_init_kernel(const float* __restrict__ boxes, const uint32_t* __restrict__ order, uint32_t n, Box6* __restrict__ cbox, uint32_t* __restrict__ cref) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    cbox[i] = load_tri_box(boxes, order, i); cref[i] = LEAF_BIT | i;
}
// nearest neighbour within +-PLOC_RADIUS positions; the block's window of boxes is staged in LDS
__global__ void __launch_bounds__(256) ploc_nn_kernel(const Box6* __restrict__ cbox, uint32_t nc, uint32_t* __restrict__ nn) {
    __shared__ Box6 tile[256 + 2 * PLOC_RADIUS];
    const int base = (int)(blockIdx.x * 256u) - PLOC_RADIUS;
    for (int t = threadIdx.x; t < 256 + 2 * PLOC_RADIUS; t += 256) { const int g = base + t; if (g >= 0 && g < (int)nc) tile[t] = cbox[g]; }
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nc) return;
    const Box6 me = tile[threadIdx.x + PLOC_RADIUS];
    float best = 3.4e38f; uint32_t bj = i;
    for (int o = -PLOC_RADIUS; o <= PLOC_RADIUS; o++) {
        const int g = (int)i + o;
        if (o == 0 || g < 0 || g >= (int)nc) continue;
        const float a = half_area(merge(me, tile[threadIdx.x + PLOC_RADIUS + o]));
        if (a < best) { best = a; bj = (uint32_t)g; }                 // ties: the lower index, on both sides of a pair -> mutual
    }
    nn[i] = bj;
}
// mutual pairs merge into a new inner node (claimed from an atomic counter) at the lower position; the upper position dies
__global__ void ploc_merge_kernel(const Box6* __restrict__ cbox, const uint32_t* __restrict__ cref, const uint32_t* __restrict__ nn, uint32_t nc,
                                  uint32_t* __restrict__ node_count, uint32_t* __restrict__ left, uint32_t* __restrict__ right, Box6* __restrict__ nbox,
                                  Box6* __restrict__ obox, uint32_t* __restrict__ oref, uint32_t* __restrict__ valid) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    const uint32_t j = nn[i];
    if (j != i && nn[j] == i) {
        if (i < j) {
            const uint32_t k = atomicAdd(node_count, 1u);
            const Box6 b = merge(cbox[i], cbox[j]);
            left[k] = cref[i]; right[k] = cref[j]; nbox[k] = b;
            obox[i] = b; oref[i] = k; valid[i] = 1u;
        } else valid[i] = 0u;
    } else { obox[i] = cbox[i]; oref[i] = cref[i]; valid[i] = 1u; }
}
__global__ void ploc_compact_kernel(const Box6* __restrict__ obox, const uint32_t* __restrict__ oref, const uint32_t* __restrict__ valid,
                                    const uint32_t* __restrict__ pos, uint32_t nc, Box6* __restrict__ cbox, uint32_t* __restrict__ cref, uint32_t* __restrict__ next_nc) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    if (i == nc - 1) *next_nc = pos[i] + valid[i];                       // clusters left after this round: the ONE word the host reads back per round
    if (!valid[i]) return;
    cbox[pos[i]] = obox[i]; cref[pos[i]] = oref[i];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return false; } } while (0)

}  // namespace

bool gpu_build_ploc(const float* tri_boxes, uint32_t n, GpuBvh& out, std::string& err) {
    if (n <= (uint32_t)MCPT_LEAF_MAX) { err = "gpu_build_ploc: needs more than MCPT_LEAF_MAX triangles"; return false; }
    const auto t0 = std::chrono::steady_clock::now();
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            const float c = 0.5f * (tri_boxes[6 * (size_t)i + a] + tri_boxes[6 * (size_t)i + 3 + a]);
            if (!(c == c) || std::fabs(c) > 3.0e38f) { err = "gpu_build_ploc: non-finite triangle bounds"; return false; }
            lo[a] = std::fmin(lo[a], c); hi[a] = std::fmax(hi[a], c);
        }
    const float3 dlo = make_float3(lo[0], lo[1], lo[2]);
    const float3 inv = make_float3(hi[0] > lo[0] ? 1.f / (hi[0] - lo[0]) : 0.f, hi[1] > lo[1] ? 1.f / (hi[1] - lo[1]) : 0.f, hi[2] > lo[2] ? 1.f / (hi[2] - lo[2]) : 0.f);
    const uint32_t ni = n - 1;
    DBuf d_boxes, d_k0, d_k1, d_v0, d_v1, d_left, d_right, d_nbox, d_cb0, d_cb1, d_cr0, d_cr1, d_nn, d_valid, d_pos, d_cnt, d_next, d_tmp;
    CK(d_boxes.alloc(sizeof(float) * 6 * (size_t)n));
    CK(d_k0.alloc(8 * (size_t)n)); CK(d_k1.alloc(8 * (size_t)n)); CK(d_v0.alloc(4 * (size_t)n)); CK(d_v1.alloc(4 * (size_t)n));
    CK(d_left.alloc(4 * (size_t)ni)); CK(d_right.alloc(4 * (size_t)ni)); CK(d_nbox.alloc(sizeof(Box6) * (size_t)ni));
    CK(d_cb0.alloc(sizeof(Box6) * (size_t)n)); CK(d_cb1.alloc(sizeof(Box6) * (size_t)n)); CK(d_cr0.alloc(4 * (size_t)n)); CK(d_cr1.alloc(4 * (size_t)n));
    CK(d_nn.alloc(4 * (size_t)n)); CK(d_valid.alloc(4 * (size_t)n)); CK(d_pos.alloc(4 * (size_t)n)); CK(d_cnt.alloc(4)); CK(d_next.alloc(4));
    CK(hipMemcpy(d_boxes.p, tri_boxes, sizeof(float) * 6 * (size_t)n, hipMemcpyHostToDevice));
    CK(hipMemset(d_cnt.p, 0, 4));
    const int B = 256;
    hipLaunchKernelGGL(morton_kernel, dim3((n + B - 1) / B), dim3(B), 0, 0, d_boxes.as<float>(), n, dlo, inv, d_k0.as<uint64_t>(), d_v0.as<uint32_t>());
    CK(hipGetLastError());
    size_t tmp_bytes = 0, scan_bytes = 0;
    CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_k0.as<uint64_t>(), d_k1.as<uint64_t>(), d_v0.as<uint32_t>(), d_v1.as<uint32_t>(), (size_t)n, 0u, 63u));
    CK(rocprim::exclusive_scan(nullptr, scan_bytes, d_valid.as<uint32_t>(), d_pos.as<uint32_t>(), 0u, (size_t)n, rocprim::plus<uint32_t>()));
    CK(d_tmp.alloc(tmp_bytes > scan_bytes ? tmp_bytes : scan_bytes));
    CK(rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, d_k0.as<uint64_t>(), d_k1.as<uint64_t>(), d_v0.as<uint32_t>(), d_v1.as<uint32_t>(), (size_t)n, 0u, 63u));
    const uint32_t* order = d_v1.as<uint32_t>();
    hipLaunchKernelGGL(ploc_init_kernel, dim3((n + B - 1) / B), dim3(B), 0, 0, d_boxes.as<float>(), order, n, d_cb0.as<Box6>(), d_cr0.as<uint32_t>());
    CK(hipGetLastError());
    uint32_t nc = n, iterations = 0;
    while (nc > 1) {
        const dim3 g((nc + B - 1) / B);
        hipLaunchKernelGGL(ploc_nn_kernel, g, dim3(B), 0, 0, d_cb0.as<Box6>(), nc, d_nn.as<uint32_t>());
        hipLaunchKernelGGL(ploc_merge_kernel, g, dim3(B), 0, 0, d_cb0.as<Box6>(), d_cr0.as<uint32_t>(), d_nn.as<uint32_t>(), nc, d_cnt.as<uint32_t>(), d_left.as<uint32_t>(),
                           d_right.as<uint32_t>(), d_nbox.as<Box6>(), d_cb1.as<Box6>(), d_cr1.as<uint32_t>(), d_valid.as<uint32_t>());
        CK(hipGetLastError());
        CK(rocprim::exclusive_scan(d_tmp.p, scan_bytes, d_valid.as<uint32_t>(), d_pos.as<uint32_t>(), 0u, (size_t)nc, rocprim::plus<uint32_t>()));
        hipLaunchKernelGGL(ploc_compact_kernel, g, dim3(B), 0, 0, d_cb1.as<Box6>(), d_cr1.as<uint32_t>(), d_valid.as<uint32_t>(), d_pos.as<uint32_t>(), nc, d_cb0.as<Box6>(),
                           d_cr0.as<uint32_t>(), d_next.as<uint32_t>());
        CK(hipGetLastError());
        uint32_t next = 0;
        CK(hipMemcpy(&next, d_next.p, 4, hipMemcpyDeviceToHost));
        if (next >= nc) { err = "gpu_build_ploc: no pair merged (internal error)"; return false; }
        nc = next;
        if (++iterations > 4096) { err = "gpu_build_ploc: did not converge"; return false; }
    }
    uint32_t made = 0, root_ref = 0;
    CK(hipMemcpy(&made, d_cnt.p, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&root_ref, d_cr0.p, 4, hipMemcpyDeviceToHost));
    if (made != ni || (root_ref & LEAF_BIT)) { err = "gpu_build_ploc: inconsistent node count"; return false; }
    // ---- host: depth-first emission.  Subtrees of <= MCPT_LEAF_MAX triangles become leaves, leaf order = depth-first order (so every
    //      leaf's triangles are contiguous), inner nodes are written in the host builder's 64-B format (child boxes in the parent).
    std::vector<uint32_t> left(ni), right(ni), sorted(n); std::vector<Box6> nbox(ni); std::vector<float> hboxes;
    CK(hipMemcpy(left.data(), d_left.p, 4 * (size_t)ni, hipMemcpyDeviceToHost)); CK(hipMemcpy(right.data(), d_right.p, 4 * (size_t)ni, hipMemcpyDeviceToHost));
    CK(hipMemcpy(nbox.data(), d_nbox.p, sizeof(Box6) * (size_t)ni, hipMemcpyDeviceToHost)); CK(hipMemcpy(sorted.data(), order, 4 * (size_t)n, hipMemcpyDeviceToHost));
    std::vector<uint32_t> size(ni, 0);
    {   // nodes are created children-first (a merge refers only to existing clusters): one forward pass gives the subtree sizes
        for (uint32_t k = 0; k < ni; k++) {
            const uint32_t l = left[k], r = right[k];
            size[k] = ((l & LEAF_BIT) ? 1u : size[l]) + ((r & LEAF_BIT) ? 1u : size[r]);
        }
    }
    auto tri_box = [&](uint32_t pos) { const float* b = tri_boxes + 6 * (size_t)sorted[pos]; return Box6{b[0], b[1], b[2], b[3], b[4], b[5]}; };
    auto ref_size = [&](uint32_t ref) { return (ref & LEAF_BIT) ? 1u : size[ref]; };
    auto ref_box = [&](uint32_t ref) { return (ref & LEAF_BIT) ? tri_box(ref & ~LEAF_BIT) : nbox[ref]; };
    out.nodes.clear(); out.nodes.reserve(4 * (size_t)ni); out.order.clear(); out.order.reserve(n);
    uint32_t depth = 0, max_leaf = 0;
    struct Item { uint32_t ref; int parent_out; int which; uint32_t depth; };
    std::vector<Item> stack; stack.push_back({root_ref, -1, 0, 1});
    std::vector<uint32_t> leaf_tris;
    auto pad_box = [](Box6 b) {
        const float m = std::fmax(std::fmax(std::fmax(std::fabs(b.lx), std::fabs(b.hx)), std::fmax(std::fabs(b.ly), std::fabs(b.hy))), std::fmax(std::fabs(b.lz), std::fabs(b.hz)));
        const float pad = m * 1e-6f + 1e-30f;
        b.lx -= pad; b.ly -= pad; b.lz -= pad; b.hx += pad; b.hy += pad; b.hz += pad; return b;
    };
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        int code;
        if (ref_size(it.ref) <= (uint32_t)MCPT_LEAF_MAX) {                       // leaf: collect its triangles in order
            leaf_tris.clear();
            uint32_t st[2 * MCPT_LEAF_MAX + 2]; int sp = 0; st[sp++] = it.ref;      // a subtree of <= MCPT_LEAF_MAX triangles: tiny fixed stack
            while (sp > 0) { const uint32_t r = st[--sp]; if (r & LEAF_BIT) leaf_tris.push_back(sorted[r & ~LEAF_BIT]); else { st[sp++] = right[r]; st[sp++] = left[r]; } }
            code = ~(int)(((uint32_t)out.order.size() << 3) | (uint32_t)leaf_tris.size());
            for (uint32_t t : leaf_tris) out.order.push_back(t);
            max_leaf = std::max<uint32_t>(max_leaf, (uint32_t)leaf_tris.size());
        } else {
            code = (int)(out.nodes.size() / 4);
            depth = std::max(depth, it.depth);
            const Box6 b0 = pad_box(ref_box(left[it.ref])), b1 = pad_box(ref_box(right[it.ref]));
            out.nodes.push_back({b0.lx, b0.hx, b0.ly, b0.hy}); out.nodes.push_back({b1.lx, b1.hx, b1.ly, b1.hy});
            out.nodes.push_back({b0.lz, b0.hz, b1.lz, b1.hz}); out.nodes.push_back({0.f, 0.f, 0.f, 0.f});
            stack.push_back({right[it.ref], code, 1, it.depth + 1});              // left is popped (emitted) first
            stack.push_back({left[it.ref], code, 0, it.depth + 1});
        }
        if (it.parent_out >= 0) { f4h& c = out.nodes[4 * (size_t)it.parent_out + 3]; float f; std::memcpy(&f, &code, 4); if (it.which == 0) c.x = f; else c.y = f; }
    }
    if (out.order.size() != n || out.nodes.empty()) { err = "gpu_build_ploc: emission lost triangles"; return false; }
    out.depth = depth; out.max_leaf = max_leaf;
    out.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return true;
}

bool gpu_build_bvh2(const float* tri_boxes, uint32_t n, GpuBvh& out, std::string& err) {
    if (n <= (uint32_t)MCPT_LEAF_MAX) { err = "gpu_build_bvh2: needs more than MCPT_LEAF_MAX triangles"; return false; }
    const auto t0 = std::chrono::steady_clock::now();
    // centroid bounds on the host: one pass over data the host has just produced (the boxes); everything O(n log n) runs on the device
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            const float c = 0.5f * (tri_boxes[6 * (size_t)i + a] + tri_boxes[6 * (size_t)i + 3 + a]);
            if (!(c == c) || std::fabs(c) > 3.0e38f) { err = "gpu_build_bvh2: non-finite triangle bounds"; return false; }
            lo[a] = std::fmin(lo[a], c); hi[a] = std::fmax(hi[a], c);
        }
    float3 dlo = make_float3(lo[0], lo[1], lo[2]);
    float3 inv = make_float3(hi[0] > lo[0] ? 1.f / (hi[0] - lo[0]) : 0.f, hi[1] > lo[1] ? 1.f / (hi[1] - lo[1]) : 0.f, hi[2] > lo[2] ? 1.f / (hi[2] - lo[2]) : 0.f);

    const int ni = (int)n - 1;
    DBuf d_boxes, d_k0, d_k1, d_v0, d_v1, d_left, d_right, d_pi, d_pl, d_first, d_last, d_arr, d_nbox, d_depth, d_inner, d_newid, d_tmp, d_out;
    CK(d_boxes.alloc(sizeof(float) * 6 * (size_t)n));
    CK(d_k0.alloc(8 * (size_t)n)); CK(d_k1.alloc(8 * (size_t)n)); CK(d_v0.alloc(4 * (size_t)n)); CK(d_v1.alloc(4 * (size_t)n));
    CK(d_left.alloc(4 * (size_t)ni)); CK(d_right.alloc(4 * (size_t)ni)); CK(d_pi.alloc(4 * (size_t)ni)); CK(d_pl.alloc(4 * (size_t)n));
    CK(d_first.alloc(4 * (size_t)ni)); CK(d_last.alloc(4 * (size_t)ni)); CK(d_arr.alloc(4 * (size_t)ni)); CK(d_nbox.alloc(sizeof(Box6) * (size_t)ni));
    CK(d_depth.alloc(4 * (size_t)ni)); CK(d_inner.alloc(4 * (size_t)ni)); CK(d_newid.alloc(4 * (size_t)ni));
    CK(hipMemcpy(d_boxes.p, tri_boxes, sizeof(float) * 6 * (size_t)n, hipMemcpyHostToDevice));
    CK(hipMemset(d_arr.p, 0, 4 * (size_t)ni));
    CK(hipMemset(d_pi.p, 0, 4 * (size_t)ni));

    const int B = 256;
    hipLaunchKernelGGL(morton_kernel, dim3((n + B - 1) / B), dim3(B), 0, 0, d_boxes.as<float>(), n, dlo, inv, d_k0.as<uint64_t>(), d_v0.as<uint32_t>());
    CK(hipGetLastError());
    size_t tmp_bytes = 0, scan_bytes = 0;
    CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_k0.as<uint64_t>(), d_k1.as<uint64_t>(), d_v0.as<uint32_t>(), d_v1.as<uint32_t>(), (size_t)n, 0u, 63u));
    CK(rocprim::exclusive_scan(nullptr, scan_bytes, d_inner.as<uint32_t>(), d_newid.as<uint32_t>(), 0u, (size_t)ni, rocprim::plus<uint32_t>()));
    CK(d_tmp.alloc(tmp_bytes > scan_bytes ? tmp_bytes : scan_bytes));
    CK(rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, d_k0.as<uint64_t>(), d_k1.as<uint64_t>(), d_v0.as<uint32_t>(), d_v1.as<uint32_t>(), (size_t)n, 0u, 63u));
    const uint64_t* keys = d_k1.as<uint64_t>(); const uint32_t* order = d_v1.as<uint32_t>();
    hipLaunchKernelGGL(hierarchy_kernel, dim3((ni + B - 1) / B), dim3(B), 0, 0, keys, (int)n, d_left.as<uint32_t>(), d_right.as<uint32_t>(), d_pi.as<uint32_t>(),
                       d_pl.as<uint32_t>(), d_first.as<uint32_t>(), d_last.as<uint32_t>());
    CK(hipGetLastError());
    hipLaunchKernelGGL(refit_kernel, dim3((n + B - 1) / B), dim3(B), 0, 0, d_boxes.as<float>(), order, (int)n, d_left.as<uint32_t>(), d_right.as<uint32_t>(),
                       d_pi.as<uint32_t>(), d_pl.as<uint32_t>(), d_first.as<uint32_t>(), d_last.as<uint32_t>(), d_arr.as<uint32_t>(), d_nbox.as<Box6>(),
                       d_depth.as<uint32_t>(), (uint32_t)MCPT_LEAF_MAX);
    CK(hipGetLastError());
    hipLaunchKernelGGL(flag_kernel, dim3((ni + B - 1) / B), dim3(B), 0, 0, d_first.as<uint32_t>(), d_last.as<uint32_t>(), ni, (uint32_t)MCPT_LEAF_MAX, d_inner.as<uint32_t>());
    CK(hipGetLastError());
    CK(rocprim::exclusive_scan(d_tmp.p, scan_bytes, d_inner.as<uint32_t>(), d_newid.as<uint32_t>(), 0u, (size_t)ni, rocprim::plus<uint32_t>()));
    uint32_t last_id = 0, last_flag = 0, depth = 0;
    CK(hipMemcpy(&last_id, d_newid.as<uint32_t>() + (ni - 1), 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&last_flag, d_inner.as<uint32_t>() + (ni - 1), 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&depth, d_depth.p, 4, hipMemcpyDeviceToHost));                 // root = Karras node 0
    const uint32_t n_out = last_id + last_flag;
    CK(d_out.alloc(64 * (size_t)n_out));
    hipLaunchKernelGGL(emit_kernel, dim3((ni + B - 1) / B), dim3(B), 0, 0, d_boxes.as<float>(), order, ni, d_left.as<uint32_t>(), d_right.as<uint32_t>(),
                       d_first.as<uint32_t>(), d_last.as<uint32_t>(), d_inner.as<uint32_t>(), d_newid.as<uint32_t>(), d_nbox.as<Box6>(), (uint32_t)MCPT_LEAF_MAX,
                       d_out.as<float4>());
    CK(hipGetLastError());
    out.nodes.resize(4 * (size_t)n_out);
    out.order.resize(n);
    CK(hipMemcpy(out.nodes.data(), d_out.p, 64 * (size_t)n_out, hipMemcpyDeviceToHost));
    CK(hipMemcpy(out.order.data(), order, 4 * (size_t)n, hipMemcpyDeviceToHost));
    out.depth = depth; out.max_leaf = (uint32_t)MCPT_LEAF_MAX;
    out.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return true;
}


// ---------------------------------------------------------------------------------------------- helpers of the wide collapse
namespace {
struct KidD { int code; float lo[3], hi[3]; };
__device__ __forceinline__ KidD kid_of(const float4* __restrict__ n2, int n, int k) {
    const float4 a = n2[4 * (size_t)n + k], z = n2[4 * (size_t)n + 2], c = n2[4 * (size_t)n + 3];
    KidD r;
    r.code = __float_as_int(k == 0 ? c.x : c.y);
    r.lo[0] = a.x; r.hi[0] = a.y; r.lo[1] = a.z; r.hi[1] = a.w; r.lo[2] = k == 0 ? z.x : z.z; r.hi[2] = k == 0 ? z.y : z.w;
    return r;
}
}  // namespace


// ---------------------------------------------------------------------------------------------- 8-wide collapse (round 3)
// The device version of build_bvh8 (scene_build.cpp), statement for statement: the same dynamic programme (double arithmetic, contraction
// off, so that the plan -- and with it every record and the leaf order -- is the host's bit for bit; tests/test_gpu_bvh_build.py compares the
// two), the same breadth-first emission one level at a time.
//   1. parent_kernel     every inner child learns its parent; a node's arrival counter starts at its number of leaf children
//   2. plan_kernel       bottom-up: a thread starts at every node whose children are both leaves and climbs; the SECOND arrival at a node
//                        (atomic counter) computes it -- cost[n][1..8] and split[n][1..8] of plan_collapse
//   3. per level:        emit_kernel (children from the plan, octant slots, quantised planes -> the record without its two bases + the
//                        inner / leaf children in slot order), two exclusive scans (inner children, leaf triangles), number_kernel (child
//                        base, triangle base, the next level's work list, the triangles' new positions) -- the serial sweep of the host code
//   4. remap_kernel      the binary tree's leaf codes follow the new leaf order
namespace {
struct C8Item { int node2, rec; };
struct C8Emit { int inner2[8]; uint32_t leaf_first[8]; unsigned char leaf_cnt[8]; uint32_t n_inner, n_leaf, leaf_tris; };
constexpr int K8 = 8;

__device__ __forceinline__ int c8_child(const float4* __restrict__ n2, int n, int k) { const float4 c = n2[4 * (size_t)n + 3]; return __float_as_int(k == 0 ? c.x : c.y); }
__device__ __forceinline__ double c8_area(const float* lo, const float* hi) {
#pragma clang fp contract(off)
    const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
    return 2.0 * (x * y + y * z + z * x);
}
__device__ __forceinline__ double c8_own_area(const float4* __restrict__ n2, int n) {
    const KidD a = kid_of(n2, n, 0), b = kid_of(n2, n, 1);
    float lo[3], hi[3];
    for (int x = 0; x < 3; x++) { lo[x] = fminf(a.lo[x], b.lo[x]); hi[x] = fmaxf(a.hi[x], b.hi[x]); }
    return c8_area(lo, hi);
}

__global__ void c8_parent_kernel(const float4* __restrict__ n2, uint32_t n, int* __restrict__ parent, uint32_t* __restrict__ arrive) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t leaves = 0;
    for (int k = 0; k < 2; k++) { const int c = c8_child(n2, (int)i, k); if (c >= 0) parent[c] = (int)i; else leaves++; }
    arrive[i] = leaves;
    if (i == 0) parent[0] = -1;
}

__device__ void c8_plan_node(const float4* __restrict__ n2, int n, double inv_root_area_unused, double root_area, double* __restrict__ cost, unsigned char* __restrict__ split) {
#pragma clang fp contract(off)
    const int l = c8_child(n2, n, 0), r = c8_child(n2, n, 1);
    double cl[K8 + 1], cr[K8 + 1];
    // a child's costs were written by another thread, possibly on another CU: agent-scope loads (the per-CU vector L1 is not refreshed by other
    // CUs' stores, and a 128-B line holds more than one node's costs), paired with the fences around the arrival counter in c8_plan_kernel
    auto ld = [](const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    for (int i = 1; i <= K8; i++) { cl[i] = l < 0 ? 0.0 : ld(&cost[(size_t)l * (K8 + 1) + i]); cr[i] = r < 0 ? 0.0 : ld(&cost[(size_t)r * (K8 + 1) + i]); }
    double* cn = cost + (size_t)n * (K8 + 1); unsigned char* sn = split + (size_t)n * (K8 + 1);
    auto distribute = [&](int j, int& best_a) { double best = 1e300; for (int a = 1; a < j; a++) { const double c = cl[a] + cr[j - a]; if (c < best) { best = c; best_a = a; } } return best; };
    int a = 1;
    cn[1] = c8_own_area(n2, n) / root_area + distribute(K8, a); sn[1] = (unsigned char)a;
    for (int i = 2; i <= K8; i++) {
        const double d = distribute(i, a), keep = cn[i - 1];
        if (d < keep) { cn[i] = d; sn[i] = (unsigned char)a; } else { cn[i] = keep; sn[i] = 0; }
    }
}   // (cn[] is re-read above through registers only: `keep` comes from the value just stored by this thread)
__global__ void c8_plan_kernel(const float4* __restrict__ n2, uint32_t n, const int* __restrict__ parent, uint32_t* __restrict__ arrive, double root_area,
                               double* __restrict__ cost, unsigned char* __restrict__ split) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (c8_child(n2, (int)i, 0) >= 0 || c8_child(n2, (int)i, 1) >= 0) return;          // start at the nodes whose children are both leaves
    int cur = (int)i;
    for (uint32_t guard = 0; guard < 4096; guard++) {                                  // (a path to the root is at most the tree's depth long)
        c8_plan_node(n2, cur, 0.0, root_area, cost, split);
        const int p = parent[cur];
        if (p < 0) break;
        __threadfence();                                                               // this node's costs before the arrival is seen
        if (atomicAdd(&arrive[p], 1u) + 1u < 2u) break;                                // the other child is still on its way: it will compute p
        __threadfence();
        cur = p;
    }
}

__global__ void c8_emit_kernel(const float4* __restrict__ n2, const unsigned char* __restrict__ split, const C8Item* __restrict__ items, uint32_t n_items,
                               float4* __restrict__ n8, C8Emit* __restrict__ emit, uint32_t* __restrict__ cnt_inner, uint32_t* __restrict__ cnt_tris) {
#pragma clang fp contract(off)
    const uint32_t wi = blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= n_items) return;
    const C8Item w = items[wi];
    // the roots of the forest below w.node2 that fills eight slots (planned_kids of scene_build.cpp, same visiting order)
    KidD kids[K8]; int nk = 0;
    struct Todo { int parent, k, slots; } todo[2 * K8 + 2]; int nt = 0;
    { const int a = split[(size_t)w.node2 * (K8 + 1) + 1]; todo[nt++] = {w.node2, 1, K8 - a}; todo[nt++] = {w.node2, 0, a}; }
    while (nt > 0) {
        const Todo it = todo[--nt];
        const int c = c8_child(n2, it.parent, it.k);
        int i = it.slots;
        if (c >= 0) while (i > 1 && split[(size_t)c * (K8 + 1) + i] == 0) i--;
        if (c < 0 || i == 1) { if (nk < K8) kids[nk++] = kid_of(n2, it.parent, it.k); continue; }
        const int a = split[(size_t)c * (K8 + 1) + i];
        todo[nt++] = {c, 1, i - a}; todo[nt++] = {c, 0, a};
    }
    { int m = 0; for (int k = 0; k < nk; k++) if (!(kids[k].code < 0 && (((uint32_t)~kids[k].code) & 7u) == 0u)) { if (m != k) kids[m] = kids[k]; m++; } nk = m; }
    float lo[3], hi[3];
    for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; }
    for (int k = 0; k < nk; k++) for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], kids[k].lo[a]); hi[a] = fmaxf(hi[a], kids[k].hi[a]); }
    if (nk == 0) for (int a = 0; a < 3; a++) lo[a] = hi[a] = 0.f;
    int kid_in[8]; for (int sl = 0; sl < 8; sl++) kid_in[sl] = -1;
    {
        double cost[8][8]; bool ck[8] = {false, false, false, false, false, false, false, false}, cs[8] = {false, false, false, false, false, false, false, false};
        for (int k = 0; k < nk; k++) {
            double cc[3];
            for (int a = 0; a < 3; a++) cc[a] = 0.5 * ((double)kids[k].lo[a] + kids[k].hi[a]) - 0.5 * ((double)lo[a] + hi[a]);
            for (int sl = 0; sl < 8; sl++) { double v = 0; for (int a = 0; a < 3; a++) v += ((sl >> a) & 1) ? cc[a] : -cc[a]; cost[k][sl] = v; }
        }
        for (int r = 0; r < nk; r++) {
            int bk = -1, bs = -1; double bv = -INFINITY;
            for (int k = 0; k < nk; k++) if (!ck[k]) for (int sl = 0; sl < 8; sl++) if (!cs[sl] && (bk < 0 || cost[k][sl] > bv)) { bv = cost[k][sl]; bk = k; bs = sl; }
            ck[bk] = cs[bs] = true; kid_in[bs] = bk;
        }
    }
    int ebits[3]; double scale[3];
    for (int a = 0; a < 3; a++) {                                   // the frame: exactly as build_bvh8 (scene_build.cpp) forms it -- stored origin 1024 + 2 margins steps below the lowest bound
        const double ext = (double)hi[a] - (double)lo[a];
        int e = ext > 0 ? (int)ceil(log2(ext / 255.0)) : -100;
        e = e < -126 ? -126 : (e > 127 ? 127 : e);
        float org = lo[a];
        for (;; e++) {
            const double sc = ldexp(1.0, e), of = (double)lo[a] - (1024.0 + 2.0 * MCPT_Q_MARGIN) * sc;
            org = (float)of; if ((double)org > of) org = nextafterf(org, -INFINITY);
            if (!(ext > 0) || e >= 127 || ((double)hi[a] - (double)org) / sc - 1024.0 + MCPT_Q_MARGIN <= 255.0) break;
        }
        lo[a] = org; ebits[a] = e; scale[a] = ldexp(1.0, e);
    }
    uint32_t q[3][4];                                                                // [axis][slot pair]: bytes lo, lo, hi, hi (device_scene.h: MCPT_N8_*)
    for (int a = 0; a < 3; a++) for (int j = 0; j < 4; j++) q[a][j] = MCPT_N8_EMPTY_WORD;
    uint32_t imask = 0, p0 = 0, p1 = 0;
    C8Emit em; em.n_inner = em.n_leaf = em.leaf_tris = 0;
    for (int sl = 0; sl < 8; sl++) {
        const int k = kid_in[sl]; if (k < 0) continue;
        for (int a = 0; a < 3; a++) {
            double ql = floor(((double)kids[k].lo[a] - (double)lo[a]) / scale[a] - 1024.0 - MCPT_Q_MARGIN);
            double qh = ceil(((double)kids[k].hi[a] - (double)lo[a]) / scale[a] - 1024.0 + MCPT_Q_MARGIN);
            ql = fmin(255.0, fmax(0.0, ql)); qh = fmin(255.0, fmax(0.0, qh));
            const int wj = MCPT_N8_WORD(sl);
            q[a][wj] = (q[a][wj] & ~(0xffu << MCPT_N8_LO_SHIFT(sl)) & ~(0xffu << MCPT_N8_HI_SHIFT(sl))) | ((uint32_t)ql << MCPT_N8_LO_SHIFT(sl)) | ((uint32_t)qh << MCPT_N8_HI_SHIFT(sl));
        }
        if (kids[k].code >= 0) { imask |= 1u << sl; em.inner2[em.n_inner++] = kids[k].code; }
        else {
            const uint32_t leaf = (uint32_t)~kids[k].code, cnt = leaf & 7u;
            p0 |= (cnt & 1u) << sl; p1 |= ((cnt >> 1) & 1u) << sl;
            em.leaf_first[em.n_leaf] = leaf >> 3; em.leaf_cnt[em.n_leaf++] = (unsigned char)cnt; em.leaf_tris += cnt;
        }
    }
    auto bf16 = [](int e) { return (uint32_t)(e + 127) << 7; };
    float4* r = n8 + 5 * (size_t)w.rec;
    r[0] = make_float4(lo[0], lo[1], lo[2], __uint_as_float((bf16(ebits[0]) << 16) | bf16(ebits[1])));
    r[1] = make_float4(0.f, 0.f, __uint_as_float(bf16(ebits[2]) << 16), __uint_as_float(imask | (p0 << 8) | (p1 << 16) | ((p0 | p1) << 24)));
    r[2] = make_float4(__uint_as_float(q[0][0]), __uint_as_float(q[0][1]), __uint_as_float(q[0][2]), __uint_as_float(q[0][3]));
    r[3] = make_float4(__uint_as_float(q[1][0]), __uint_as_float(q[1][1]), __uint_as_float(q[1][2]), __uint_as_float(q[1][3]));
    r[4] = make_float4(__uint_as_float(q[2][0]), __uint_as_float(q[2][1]), __uint_as_float(q[2][2]), __uint_as_float(q[2][3]));
    emit[wi] = em; cnt_inner[wi] = em.n_inner; cnt_tris[wi] = em.leaf_tris;
}

__global__ void c8_number_kernel(const C8Item* __restrict__ items, uint32_t n_items, const C8Emit* __restrict__ emit, const uint32_t* __restrict__ pos_inner,
                                 const uint32_t* __restrict__ pos_tris, uint32_t rec_base, uint32_t tri_base, float4* __restrict__ n8, C8Item* __restrict__ next,
                                 const int* __restrict__ order, int* __restrict__ new_order, int* __restrict__ new_first, uint32_t* __restrict__ totals) {
    const uint32_t wi = blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= n_items) return;
    const C8Emit em = emit[wi];
    const uint32_t cb = rec_base + pos_inner[wi]; uint32_t tb = tri_base + pos_tris[wi];
    float4* r = n8 + 5 * (size_t)items[wi].rec;
    float4 r1 = r[1]; r1.x = __uint_as_float(cb); r1.y = __uint_as_float(tb); r[1] = r1;
    for (uint32_t i = 0; i < em.n_inner; i++) next[pos_inner[wi] + i] = C8Item{em.inner2[i], (int)(cb + i)};
    for (uint32_t i = 0; i < em.n_leaf; i++) {
        new_first[em.leaf_first[i]] = (int)tb;
        for (uint32_t t = 0; t < em.leaf_cnt[i]; t++) new_order[tb++] = order[em.leaf_first[i] + t];
    }
    if (wi == n_items - 1) { totals[0] = pos_inner[wi] + em.n_inner; totals[1] = pos_tris[wi] + em.leaf_tris; }
}

__global__ void c8_remap_kernel(float4* __restrict__ n2, uint32_t n, const int* __restrict__ new_first) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 c = n2[4 * (size_t)i + 3];
    for (int k = 0; k < 2; k++) {
        const int code = __float_as_int(k == 0 ? c.x : c.y);
        if (code >= 0) continue;
        const uint32_t leaf = (uint32_t)~code, first = leaf >> 3, cnt = leaf & 7u;
        if (cnt == 0) continue;
        const float v = __int_as_float(~(int)(((uint32_t)new_first[first] << 3) | cnt));
        if (k == 0) c.x = v; else c.y = v;
    }
    n2[4 * (size_t)i + 3] = c;
}
}  // namespace

bool gpu_collapse_bvh8(std::vector<f4h>& nodes2, std::vector<int>& order, std::vector<f4h>& nodes8, uint32_t& depth8, std::string& err) {
    const uint32_t n2 = (uint32_t)(nodes2.size() / 4), nf = (uint32_t)order.size();
    if (n2 == 0) { err = "gpu_collapse_bvh8: empty binary tree"; return false; }
    double root_area;
    {   // own_area(0) of plan_collapse, on the host (same arithmetic)
        float lo[3], hi[3];
        const f4h a = nodes2[0], b = nodes2[1], z = nodes2[2];
        const float l0[3] = {a.x, a.z, z.x}, h0[3] = {a.y, a.w, z.y}, l1[3] = {b.x, b.z, z.z}, h1[3] = {b.y, b.w, z.w};
        for (int x = 0; x < 3; x++) { lo[x] = std::fmin(l0[x], l1[x]); hi[x] = std::fmax(h0[x], h1[x]); }
        const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], zz = (double)hi[2] - lo[2];
        root_area = 2.0 * (x * y + y * zz + zz * x);
        if (!(root_area > 1e-300)) root_area = 1e-300;
    }
    DBuf d_n2, d_n8, d_parent, d_arrive, d_cost, d_split, d_q0, d_q1, d_emit, d_ci, d_ct, d_pi, d_pt, d_order, d_norder, d_nfirst, d_tot, d_tmp;
    CK(d_n2.alloc(64 * (size_t)n2)); CK(d_n8.alloc(80 * (size_t)n2));            // an 8-wide tree never has more nodes than its binary source
    CK(d_parent.alloc(4 * (size_t)n2)); CK(d_arrive.alloc(4 * (size_t)n2));
    CK(d_cost.alloc(sizeof(double) * (size_t)n2 * (K8 + 1))); CK(d_split.alloc((size_t)n2 * (K8 + 1)));
    CK(d_q0.alloc(sizeof(C8Item) * (size_t)n2)); CK(d_q1.alloc(sizeof(C8Item) * (size_t)n2)); CK(d_emit.alloc(sizeof(C8Emit) * (size_t)n2));
    CK(d_ci.alloc(4 * (size_t)n2)); CK(d_ct.alloc(4 * (size_t)n2)); CK(d_pi.alloc(4 * (size_t)n2)); CK(d_pt.alloc(4 * (size_t)n2));
    CK(d_order.alloc(4 * (size_t)nf)); CK(d_norder.alloc(4 * (size_t)nf)); CK(d_nfirst.alloc(4 * ((size_t)nf + 1))); CK(d_tot.alloc(8));
    CK(hipMemcpy(d_n2.p, nodes2.data(), 64 * (size_t)n2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_order.p, order.data(), 4 * (size_t)nf, hipMemcpyHostToDevice));
    CK(hipMemset(d_split.p, 0, (size_t)n2 * (K8 + 1))); CK(hipMemset(d_cost.p, 0, sizeof(double) * (size_t)n2 * (K8 + 1))); CK(hipMemset(d_n8.p, 0, 80));
    CK(hipMemset(d_nfirst.p, 0xff, 4 * ((size_t)nf + 1)));
    const int B = 128;
    hipLaunchKernelGGL(c8_parent_kernel, dim3((n2 + B - 1) / B), dim3(B), 0, 0, d_n2.as<float4>(), n2, d_parent.as<int>(), d_arrive.as<uint32_t>());
    hipLaunchKernelGGL(c8_plan_kernel, dim3((n2 + B - 1) / B), dim3(B), 0, 0, d_n2.as<float4>(), n2, d_parent.as<int>(), d_arrive.as<uint32_t>(), root_area, d_cost.as<double>(),
                       d_split.as<unsigned char>());
    CK(hipGetLastError());
    size_t scan_bytes = 0;
    CK(rocprim::exclusive_scan(nullptr, scan_bytes, d_ci.as<uint32_t>(), d_pi.as<uint32_t>(), 0u, (size_t)n2, rocprim::plus<uint32_t>()));
    CK(d_tmp.alloc(scan_bytes));
    const C8Item root{0, 0};
    CK(hipMemcpy(d_q0.p, &root, sizeof root, hipMemcpyHostToDevice));
    uint32_t n_items = 1, n_rec = 1, n_tri = 0; depth8 = 0;
    C8Item* cur = d_q0.as<C8Item>(); C8Item* nxt = d_q1.as<C8Item>();
    while (n_items > 0) {
        if (++depth8 > 256) { err = "gpu_collapse_bvh8: tree too deep"; return false; }
        const dim3 g((n_items + B - 1) / B);
        hipLaunchKernelGGL(c8_emit_kernel, g, dim3(B), 0, 0, d_n2.as<float4>(), d_split.as<unsigned char>(), cur, n_items, d_n8.as<float4>(), d_emit.as<C8Emit>(), d_ci.as<uint32_t>(),
                           d_ct.as<uint32_t>());
        CK(hipGetLastError());
        CK(rocprim::exclusive_scan(d_tmp.p, scan_bytes, d_ci.as<uint32_t>(), d_pi.as<uint32_t>(), 0u, (size_t)n_items, rocprim::plus<uint32_t>()));
        CK(rocprim::exclusive_scan(d_tmp.p, scan_bytes, d_ct.as<uint32_t>(), d_pt.as<uint32_t>(), 0u, (size_t)n_items, rocprim::plus<uint32_t>()));
        hipLaunchKernelGGL(c8_number_kernel, g, dim3(B), 0, 0, cur, n_items, d_emit.as<C8Emit>(), d_pi.as<uint32_t>(), d_pt.as<uint32_t>(), n_rec, n_tri, d_n8.as<float4>(), nxt,
                           d_order.as<int>(), d_norder.as<int>(), d_nfirst.as<int>(), d_tot.as<uint32_t>());
        CK(hipGetLastError());
        uint32_t tot[2];
        CK(hipMemcpy(tot, d_tot.p, 8, hipMemcpyDeviceToHost));
        if ((size_t)n_rec + tot[0] > n2 || (size_t)n_tri + tot[1] > nf) { err = "gpu_collapse_bvh8: inconsistent counts (internal error)"; return false; }
        n_rec += tot[0]; n_tri += tot[1]; n_items = tot[0];
        C8Item* t = cur; cur = nxt; nxt = t;
    }
    if (n_tri != nf) { err = "gpu_collapse_bvh8: not every triangle was placed (internal error)"; return false; }
    hipLaunchKernelGGL(c8_remap_kernel, dim3((n2 + B - 1) / B), dim3(B), 0, 0, d_n2.as<float4>(), n2, d_nfirst.as<int>());
    CK(hipGetLastError());
    nodes8.resize(5 * (size_t)n_rec);
    CK(hipMemcpy(nodes8.data(), d_n8.p, 80 * (size_t)n_rec, hipMemcpyDeviceToHost));
    CK(hipMemcpy(nodes2.data(), d_n2.p, 64 * (size_t)n2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(order.data(), d_norder.p, 4 * (size_t)nf, hipMemcpyDeviceToHost));
    return true;
}
