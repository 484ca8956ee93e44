// C ABI of libmcpt_hip.so (include/mcpt.h): context management, HBM upload, kernel launches.
// No CPU path exists in this file: every compute entry point launches a gfx950 kernel or fails.
#include "../../include/mcpt.h"
#include "kernels.h"
#include "scene_build.h"
#include "bvh_gpu.h"
#include "wavefront.h"

#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    hipError_t alloc(size_t n) { bytes = n; return hipMalloc(&p, n ? n : 16); }
    void free_() { if (p) (void)hipFree(p); p = nullptr; }
};

}  // namespace

struct mcpt_ctx {
    int device = 0;
    mcpt_opts opts{};
    DevScene dev{};
    DevBuf nodes, nodes8, tri_isect, tri_shade, tri_pos64, tri_face, mats, lights, light_pos64, texels, accum_own, counters;
    float4* accum = nullptr;           // bound accumulator (own or external)
    hipStream_t own_stream = nullptr, stream = nullptr;
    // HIP-event brackets of the render calls whose duration has not been read yet: a ring, so that a call does not have to wait for the one
    // before it (the reference's loop issues a call per sample); resolve_timing() reads the finished ones, oldest first
    static constexpr uint32_t TIMED = 16;
    hipEvent_t ev0[TIMED] = {}, ev1[TIMED] = {};
    uint32_t timed_head = 0, timed_tail = 0;      // calls [timed_tail, timed_head) are outstanding
    double last_kernel_ms = 0.0, total_kernel_ms = 0.0;
    uint64_t launches = 0;
    mcpt_scene_info info{};
    int width = 0, height = 0;
    // ---- wavefront pipeline (the default for MCPT_INTEGRATOR_MIS)
    bool use_wavefront = true;
    // Sub-pipelines ("lanes"): each owns a path pool, a control block and a stream and runs its own [shade, trace] loop on its
    // share of the sample range.  Two of them in flight let the issue-bound shade kernel of one overlap the memory-bound trace
    // kernel of the other on the same CUs (measured +15 % on MI355X).
    struct WfLane {
        PathPool pool{};
        CompactBufs compact{};             // scratch of the end-of-job drain compaction (wavefront.h); capacity 0 = none (small pools)
        std::vector<DevBuf> pool_bufs;
        DevBuf ctl_buf, ovf_buf;
        IterCtl* h_ctl = nullptr;          // pinned ring of control-block snapshots (termination check)
        std::vector<hipEvent_t> chk_ev;
        std::vector<hipEvent_t> k_ev;      // per-kernel event chain (only with detailed timing)
        hipStream_t stream = nullptr;
        hipEvent_t done_ev = nullptr;
        uint64_t last_iterations = 0, last_timed = 0;
        // Known-length jobs (every item has its own slot and one sample, depth-limited: render_wavefront) are enqueued whole and NOT waited for:
        // the control-block snapshot taken after their last iteration is looked at later -- by the next call that drains the stream, or when the
        // ring of snapshots is full -- so consecutive one-sample calls (the reference's loop, main.cpp:26-33) cost the host only their launches.
        struct Verdict { uint32_t ring_slot, it, n_shared; };
        std::vector<Verdict> verdicts;     // oldest first; at most RING - 2 outstanding
        uint32_t ring_next = 0;            // next h_ctl / chk_ev slot this lane uses (jobs of either kind take them in turn)
    };
    std::vector<WfLane> lanes;
    hipEvent_t fork_ev = nullptr;
    WaveTuning tune{};
    uint32_t trace_grid = 0;
    uint32_t pool_cap = 1u << 23;       // most slots a sub-pipeline's pool may have (MCPT_WF_POOL_LOG2); pools are allocated on first use, sized to the job
    uint32_t items_per_slot = 1;        // pool sizing knob: a job of n work items gets n / items_per_slot slots, at most pool_cap (MCPT_WF_ITEMS_PER_SLOT), see render_wavefront
    int n_cus = 0;
    uint32_t time_kernels = 0;          // MCPT_TIME_KERNELS=N: bracket the two kernels of every Nth iteration with HIP events (0 = off)
    double last_trace_ms = 0.0, total_trace_ms = 0.0, last_shade_ms = 0.0, total_shade_ms = 0.0;
    uint64_t total_iterations = 0;
    bool binary_ok = true;                // the binary cross-check tree fits its kernels' stack (false: a deep device-built tree)
    uint32_t wide_depth = 0;              // depth of the 8-wide tree the wavefront trace kernel walks
    std::vector<int32_t> h_tri_face;      // leaf order -> face index, fetched on first use by mcpt_probe_trace4
    // Scene::getPixelsColor every frame (main.cpp:26-33): the tonemapped film's device buffer and its pinned host image live as long as the
    // context (allocated by the first tonemap call) -- no hipMalloc / hipMemset / hipFree per frame
    DevBuf tone_dev; uint8_t* tone_host = nullptr;
};

namespace {

mcpt_status fail(mcpt_status s, const std::string& msg) { g_err = msg; return s; }
mcpt_status hip_fail(hipError_t e, const char* what) { g_err = std::string(what) + ": " + hipGetErrorString(e); return MCPT_ERR_HIP; }
#define HIP_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(e_, #call); } while (0)

template <class T>
hipError_t upload(DevBuf& b, const std::vector<T>& v) {
    hipError_t e = b.alloc(v.size() * sizeof(T));
    if (e != hipSuccess) return e;
    if (!v.empty()) e = hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

void destroy_ctx(mcpt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    c->nodes.free_(); c->nodes8.free_(); c->tri_isect.free_(); c->tri_shade.free_(); c->tri_pos64.free_(); c->tri_face.free_();
    c->mats.free_(); c->lights.free_(); c->light_pos64.free_(); c->texels.free_(); c->accum_own.free_(); c->counters.free_();
    for (auto& L : c->lanes) {
        for (auto& b : L.pool_bufs) b.free_();
        L.ctl_buf.free_(); L.ovf_buf.free_();
        if (L.h_ctl) (void)hipHostFree(L.h_ctl);
        for (auto e : L.chk_ev) (void)hipEventDestroy(e);
        for (auto e : L.k_ev) (void)hipEventDestroy(e);
        if (L.done_ev) (void)hipEventDestroy(L.done_ev);
        if (L.stream) (void)hipStreamDestroy(L.stream);
    }
    c->tone_dev.free_(); if (c->tone_host) (void)hipHostFree(c->tone_host);
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    for (uint32_t i = 0; i < mcpt_ctx::TIMED; i++) { if (c->ev0[i]) (void)hipEventDestroy(c->ev0[i]); if (c->ev1[i]) (void)hipEventDestroy(c->ev1[i]); }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

mcpt_status use(mcpt_ctx* c) {
    if (!c) return fail(MCPT_ERR_INVALID_ARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    return MCPT_OK;
}

uint32_t env_u32(const char* name, uint32_t dflt) {
    const char* v = std::getenv(name);
    return (v && *v) ? uint32_t(std::strtoul(v, nullptr, 10)) : dflt;
}

// Probe entry points take and return WORLD coordinates; the device works relative to DevScene::centre.
std::vector<double> to_local(const mcpt_ctx* c, const double* p, size_t n) {
    std::vector<double> v(3 * n);
    for (size_t i = 0; i < n; i++) for (int a = 0; a < 3; a++) v[3 * i + a] = p[3 * i + a] - c->dev.centre[a];
    return v;
}

// scratch device buffer for probes
// Device scratch of one probe / tonemap call.  The fills below go through the legacy default stream, the kernels that use the buffers
// run on the context's stream, which is NON-BLOCKING (no implicit ordering with the default stream): hipMemset on device memory
// returns before it has executed, so without the explicit wait a kernel could write its results first and have them zeroed afterwards
// (seen as black regions in `mcpt_cli --save-every` images: the tonemap kernel ran ahead of its buffer's memset).
struct Scratch {
    std::vector<void*> ptrs;
    ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T> hipError_t in(const T* host, size_t n, T** dev) {
        hipError_t e = hipMalloc((void**)dev, (n ? n : 1) * sizeof(T)); if (e != hipSuccess) return e;
        ptrs.push_back(*dev);
        if (!n) return hipSuccess;
        e = hipMemcpy(*dev, host, n * sizeof(T), hipMemcpyHostToDevice); if (e != hipSuccess) return e;
        return hipStreamSynchronize(nullptr);
    }
    template <class T> hipError_t out(size_t n, T** dev) {
        hipError_t e = hipMalloc((void**)dev, (n ? n : 1) * sizeof(T)); if (e != hipSuccess) return e;
        ptrs.push_back(*dev);
        e = hipMemset(*dev, 0, (n ? n : 1) * sizeof(T)); if (e != hipSuccess) return e;
        return hipStreamSynchronize(nullptr);
    }
};

}  // namespace

// Everything of a context that is not the scene: streams, events, film, counters, the wavefront sub-pipelines, and the device pointers of
// c->dev (the scene streams c->nodes ... c->texels are on the device already: uploaded by mcpt_create or copied by mcpt_clone_to_device).
static mcpt_status finish_ctx(mcpt_ctx* c) {
    hipError_t e = hipSuccess;
    auto bail = [&](hipError_t he, const char* what) { return hip_fail(he, what); };
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    c->stream = c->own_stream;
    for (uint32_t i = 0; i < mcpt_ctx::TIMED; i++) if ((e = hipEventCreate(&c->ev0[i])) != hipSuccess || (e = hipEventCreate(&c->ev1[i])) != hipSuccess) return bail(e, "hipEventCreate");
    const size_t accum_bytes = size_t(c->width) * c->height * sizeof(float4);
    if ((e = c->accum_own.alloc(accum_bytes)) != hipSuccess) return bail(e, "alloc accumulator");
    if ((e = hipMemset(c->accum_own.p, 0, accum_bytes)) != hipSuccess) return bail(e, "clear accumulator");
    if ((e = c->counters.alloc(sizeof(DevCounters) * WF_COUNTER_REPLICAS)) != hipSuccess) return bail(e, "alloc counters");
    if ((e = hipMemset(c->counters.p, 0, sizeof(DevCounters) * WF_COUNTER_REPLICAS)) != hipSuccess) return bail(e, "clear counters");
    {   // ---- wavefront pool.  Tunables are developer knobs (environment), not part of the ABI.
        hipDeviceProp_t prop;
        if ((e = hipGetDeviceProperties(&prop, c->device)) != hipSuccess) return bail(e, "hipGetDeviceProperties");
        c->n_cus = prop.multiProcessorCount;
        c->pool_cap = 1u << std::min(26u, env_u32("MCPT_WF_POOL_LOG2", 23));
        c->pool_cap = env_u32("MCPT_WF_POOL_SLOTS", c->pool_cap) & ~uint32_t(16 * WF_SHADE_BLOCK - 1);   // (developer knob: any multiple of 4096 slots)
        if (c->pool_cap < 4096) c->pool_cap = 4096;
        c->items_per_slot = std::max(1u, env_u32("MCPT_WF_ITEMS_PER_SLOT", 1));
        c->tune.refill_at = env_u32("MCPT_WF_REFILL", 28); c->tune.leaf_at = env_u32("MCPT_WF_LEAF", 16);
        c->tune.inner_keep = env_u32("MCPT_WF_INNER", 24); c->tune.policy = env_u32("MCPT_WF_POLICY", 0); c->tune.pend_cap = 48;      // speculative traversal: refined below once the scene's size is known
        c->time_kernels = env_u32("MCPT_TIME_KERNELS", 0);
        if (c->use_wavefront) {
            uint32_t n_lanes = env_u32("MCPT_WF_LANES", 2);
            if (n_lanes < 1) n_lanes = 1;
            if (c->opts.flags & MCPT_FLAG_DETERMINISTIC) n_lanes = 1;          // one owner per pixel, plain stores
            // Persistent trace grid: one 1024-thread block (16 waves, 78 VGPRs) per CU -- what the register file admits beside two shade
            // waves per SIMD (4 x 80 + 2 x 96 = 512).  Rounds 1-2 launched 3/4 and 7/8 of the CUs while the traversal data was
            // cache-resident (their 72-register kernel left room for a second block on some CUs); r03, 8-wide kernel, S-cornell 512 spp:
            // 214.6 / 217.5 / 211.6 / 213.4 / 208.5 ms at 192 / 208 / 224 / 240 / 256 blocks.  MCPT_WF_GRID overrides.
            const uint32_t per_cu = uint32_t(wf_trace_blocks_per_cu((c->opts.flags & MCPT_FLAG_COUNT_TRAVERSAL) != 0));
            c->trace_grid = uint32_t(c->n_cus) * per_cu;
            c->trace_grid = std::min(env_u32("MCPT_WF_GRID", c->trace_grid), uint32_t(c->n_cus) * per_cu);
            // speculative traversal: S-cornell 469 -> 450 ms; on the 4 M-triangle configuration, where the extra node visits are HBM
            // traffic, it is neutral within the noise (same box: 367 ms with, 371 ms without) -- on everywhere; MCPT_WF_PEND=0 turns it off
            c->tune.pend_cap = env_u32("MCPT_WF_PEND", 48u);
            if ((e = hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
            c->lanes.resize(n_lanes);
            for (auto& L : c->lanes) {
                L.pool.P = 0;                                             // allocated by ensure_pool() when the first job arrives
                if ((e = L.ctl_buf.alloc(sizeof(IterCtl))) != hipSuccess) return bail(e, "alloc IterCtl");
                if ((e = hipHostMalloc((void**)&L.h_ctl, 8 * sizeof(IterCtl), hipHostMallocDefault)) != hipSuccess) return bail(e, "hipHostMalloc");
                L.chk_ev.resize(8);
                for (auto& ev : L.chk_ev) if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
                if ((e = hipEventCreateWithFlags(&L.done_ev, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
                if ((e = hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
                // the global overflow area of the traversal stack is sized from the wide tree's depth (2 x depth + 3 entries of 8 B per trace lane): a
                // pathologically deep device-built tree (depth in the hundreds) would ask for a GB per sub-pipeline -- refuse instead of allocating it
                const size_t ovf_bytes = size_t(c->trace_grid) * wf_trace_block_threads() * wf_trace_overflow_bytes_per_lane(c->wide_depth);
                if (ovf_bytes > (size_t(512) << 20)) return fail(MCPT_ERR_BVH_DEPTH, "wide BVH of depth " + std::to_string(c->wide_depth) + " needs a traversal-stack overflow area of " + std::to_string(ovf_bytes >> 20) + " MB per sub-pipeline: build the tree with the host builder (no MCPT_FLAG_GPU_BVH_BUILD)");
                if ((e = L.ovf_buf.alloc(ovf_bytes)) != hipSuccess)
                    return bail(e, "alloc stack overflow area");
            }
        }
    }
    if ((e = hipDeviceSynchronize()) != hipSuccess) return bail(e, "sync after upload");
    c->accum = static_cast<float4*>(c->accum_own.p);
    DevScene& d = c->dev;
    d.nodes = static_cast<const float4*>(c->nodes.p); d.nodes8 = static_cast<const float4*>(c->nodes8.p);
    d.tri_isect = static_cast<const float4*>(c->tri_isect.p);
    d.tri_shade = static_cast<const float4*>(c->tri_shade.p); d.tri_pos64 = static_cast<const double*>(c->tri_pos64.p);
    d.tri_face = static_cast<const int32_t*>(c->tri_face.p); d.mats = static_cast<const DevMaterial*>(c->mats.p);
    d.lights = static_cast<const DevLight*>(c->lights.p); d.light_pos64 = static_cast<const double*>(c->light_pos64.p); d.texels = static_cast<const float4*>(c->texels.p);
    c->info.device_bytes = c->nodes.bytes + c->nodes8.bytes + c->tri_isect.bytes + c->tri_shade.bytes + c->tri_pos64.bytes + c->tri_face.bytes +
                           c->mats.bytes + c->lights.bytes + c->light_pos64.bytes + c->texels.bytes + accum_bytes;
    return MCPT_OK;
}

static void fill_wide_info(mcpt_scene_info& in, const HostScene& hs) {
    in.wide_width = 8; in.wide_nodes = uint32_t(hs.nodes8.size() / 5);
    in.wide_depth = hs.bvh8_depth;
    in.traversal_bytes = (hs.nodes8.size() + hs.tri_isect.size()) * sizeof(f4h);
    for (int a = 0; a < 3; a++) in.centre[a] = hs.centre[a];
    uint64_t h = 1469598103934665603ull;                                  // FNV-1a, 4 bytes at a time
    auto mix = [&](const void* p, size_t bytes) { const uint32_t* w = static_cast<const uint32_t*>(p); for (size_t i = 0; i < bytes / 4; i++) { h ^= w[i]; h *= 1099511628211ull; } };
    mix(hs.nodes8.data(), hs.nodes8.size() * sizeof(f4h)); mix(hs.tri_face.data(), hs.tri_face.size() * 4);
    in.wide_tree_hash = h;
}

extern "C" {

uint32_t mcpt_abi_version(void) { return MCPT_ABI_VERSION; }
const char* mcpt_last_error(void) { return g_err.c_str(); }

mcpt_status mcpt_check_scene(const mcpt_scene_desc* scene, mcpt_scene_info* out_info) {
    if (!scene) return fail(MCPT_ERR_INVALID_ARG, "mcpt_check_scene: null argument");
    HostScene hs; std::string err;
    mcpt_status st = build_host_scene(scene, hs, err);
    if (st != MCPT_OK) return fail(st, err);
    const std::string bad = validate_wide_bvh(hs);
    if (!bad.empty()) return fail(MCPT_ERR_UNSUPPORTED, "internal: wide BVH failed its self-check: " + bad);
    if (out_info) {
        std::memset(out_info, 0, sizeof *out_info);
        out_info->n_tris = uint32_t(hs.tri_face.size()); out_info->n_lights = uint32_t(hs.lights.size()); out_info->n_nodes = uint32_t(hs.nodes.size() / 4);
        out_info->bvh_depth = hs.bvh_depth; out_info->max_leaf = hs.max_leaf; out_info->width = uint32_t(scene->camera.width); out_info->height = uint32_t(scene->camera.height);
        out_info->bvh_build_ms = hs.bvh_build_ms;
        fill_wide_info(*out_info, hs);
    }
    return MCPT_OK;
}

mcpt_status mcpt_create(const mcpt_scene_desc* scene, const mcpt_opts* opts, mcpt_ctx** out_ctx) {
    if (!scene || !out_ctx) return fail(MCPT_ERR_INVALID_ARG, "mcpt_create: null argument");
    *out_ctx = nullptr;
    mcpt_opts o; std::memset(&o, 0, sizeof o);
    if (opts) std::memcpy(&o, opts, std::min<size_t>(sizeof o, opts->struct_size ? opts->struct_size : sizeof o));
    if (o.integrator > MCPT_INTEGRATOR_RECURSIVE_NEE) return fail(MCPT_ERR_INVALID_ARG, "unknown integrator");

    HostScene hs; std::string err;
    hs.reference_tie_order = (o.flags & MCPT_FLAG_REFERENCE_TIE_ORDER) != 0;
    // which pipeline this context runs is decided ONCE, here: it sets how deep a device-built binary tree may be (below) and which kernels
    // mcpt_render launches -- the two must agree, or a megakernel context could walk a tree deeper than its LDS stack
    const bool use_wavefront = [&]() { const char* pipe = std::getenv("MCPT_PIPELINE"); return !(pipe && std::string(pipe) == "mega") && o.integrator == MCPT_INTEGRATOR_MIS; }();
    if ((o.flags & MCPT_FLAG_REFERENCE_TIE_ORDER) && !use_wavefront)
        return fail(MCPT_ERR_UNSUPPORTED, "MCPT_FLAG_REFERENCE_TIE_ORDER needs the wavefront pipeline (MCPT_INTEGRATOR_MIS, no MCPT_PIPELINE=mega): the binary-tree kernels have no tie rule");
    int ndev = 0;
    hipError_t e = hipSuccess;
    auto check_device = [&]() -> mcpt_status {
        e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0) return fail(MCPT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
        if (o.device < 0 || o.device >= ndev) return fail(MCPT_ERR_NO_DEVICE, "device ordinal out of range");
        return MCPT_OK;
    };
    mcpt_status st;
    if (o.flags & MCPT_FLAG_GPU_BVH_BUILD) {                              // the tree is built on the device the context will render on
        if ((st = check_device()) != MCPT_OK) return st;
        if ((e = hipSetDevice(o.device)) != hipSuccess) return hip_fail(e, "hipSetDevice");
        // An agglomerative (PLOC) tree over millions of triangles can be deeper than the binary-tree kernels' 64-entry stack.  Only the
        // cross-check kernels (megakernel, recursive integrator, mcpt_probe_trace) walk the binary tree; the wavefront pipeline walks
        // the wide collapse of it, whose stack is sized from its own depth -- so a wavefront-only context keeps the deep tree.
        hs.allow_deep_binary = use_wavefront;
        st = build_host_scene(scene, hs, err, [&](const float* boxes, uint32_t n, std::vector<f4h>& nodes, std::vector<int>& order, uint32_t& depth,
                                                   uint32_t& max_leaf, std::string& berr) {
            GpuBvh g;
            const char* kind = std::getenv("MCPT_GPU_BVH");                        // developer knob: "lbvh" = the plain Karras tree
            const bool lbvh = kind && std::string(kind) == "lbvh";
            if (!(lbvh ? gpu_build_bvh2(boxes, n, g, berr) : gpu_build_ploc(boxes, n, g, berr))) return false;
            nodes.swap(g.nodes); order.assign(g.order.begin(), g.order.end()); depth = g.depth; max_leaf = g.max_leaf;
            return true;
        }, env_u32("MCPT_HOST_COLLAPSE", 0) ? Collapse8Fn(nullptr) : Collapse8Fn(gpu_collapse_bvh8));
        if (st != MCPT_OK) return fail(st, err);
        if (env_u32("MCPT_VALIDATE_BVH", 0)) { const std::string bad = validate_wide_bvh(hs); if (!bad.empty()) return fail(MCPT_ERR_HIP, "device-built BVH failed validation: " + bad); }
    } else {
        st = build_host_scene(scene, hs, err);
        if (st != MCPT_OK) return fail(st, err);
        if ((st = check_device()) != MCPT_OK) return st;
    }

    // the 8-wide trace kernel addresses node and triangle records with 32-bit byte offsets
    if ((hs.nodes8.size() * sizeof(f4h) >= (1ull << 32) || hs.tri_isect.size() * sizeof(f4h) >= (1ull << 32)))
        return fail(MCPT_ERR_UNSUPPORTED, "scene too large for the 8-wide traversal kernel (more than 89 M triangles)");
    mcpt_ctx* c = new mcpt_ctx();
    c->device = o.device; c->opts = o; c->width = scene->camera.width; c->height = scene->camera.height;
    c->wide_depth = hs.bvh8_depth; c->binary_ok = hs.binary_ok; c->use_wavefront = use_wavefront;
    auto bail = [&](hipError_t he, const char* what) { mcpt_status s = hip_fail(he, what); destroy_ctx(c); return s; };
    if ((e = hipSetDevice(c->device)) != hipSuccess) return bail(e, "hipSetDevice");
    auto t0 = std::chrono::steady_clock::now();
    if ((e = upload(c->nodes, hs.nodes)) != hipSuccess) return bail(e, "upload nodes");
    if ((e = upload(c->nodes8, hs.nodes8)) != hipSuccess) return bail(e, "upload nodes8");
    if ((e = upload(c->tri_isect, hs.tri_isect)) != hipSuccess) return bail(e, "upload tri_isect");
    if ((e = upload(c->tri_shade, hs.tri_shade)) != hipSuccess) return bail(e, "upload tri_shade");
    if ((e = upload(c->tri_pos64, hs.tri_pos64)) != hipSuccess) return bail(e, "upload tri_pos64");
    if ((e = upload(c->tri_face, hs.tri_face)) != hipSuccess) return bail(e, "upload tri_face");
    if ((e = upload(c->mats, hs.mats)) != hipSuccess) return bail(e, "upload materials");
    if ((e = upload(c->lights, hs.lights)) != hipSuccess) return bail(e, "upload lights");
    if ((e = upload(c->light_pos64, hs.light_pos64)) != hipSuccess) return bail(e, "upload light corners");
    if ((e = upload(c->texels, hs.texels)) != hipSuccess) return bail(e, "upload texels");
    DevScene& d = c->dev;
    d.n_nodes8 = int32_t(hs.nodes8.size() / 5);
    d.cam = hs.cam;
    for (int a = 0; a < 3; a++) d.centre[a] = hs.centre[a];
    d.n_tris = int32_t(hs.tri_face.size()); d.n_lights = int32_t(hs.lights.size()); d.n_nodes = int32_t(hs.nodes.size() / 4); d.n_mats = int32_t(hs.mats.size());
    mcpt_scene_info& in = c->info;
    in.n_tris = uint32_t(d.n_tris); in.n_lights = uint32_t(d.n_lights); in.n_nodes = uint32_t(d.n_nodes);
    in.bvh_depth = hs.bvh_depth; in.max_leaf = hs.max_leaf; in.width = uint32_t(c->width); in.height = uint32_t(c->height);
    in.bvh_build_ms = hs.bvh_build_ms;
    fill_wide_info(in, hs);
    const mcpt_status fs = finish_ctx(c);
    if (fs != MCPT_OK) { destroy_ctx(c); return fs; }
    in.upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *out_ctx = c;
    return MCPT_OK;
}

/* A second context for the SAME scene on another device (or the same one): every scene stream is copied device to device -- no flatten, no
 * BVH build, no host copy of the scene is kept around for it.  `mcpt_cli --gpus N` builds once and clones N - 1 times. */
mcpt_status mcpt_clone_to_device(mcpt_ctx* src, int32_t device, mcpt_ctx** out_ctx) {
    if (!src || !out_ctx) return fail(MCPT_ERR_INVALID_ARG, "mcpt_clone_to_device: null argument");
    *out_ctx = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(MCPT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(MCPT_ERR_NO_DEVICE, "device ordinal out of range");
    mcpt_status st = use(src); if (st != MCPT_OK) return st;
    HIP_TRY(hipStreamSynchronize(src->stream));
    mcpt_ctx* c = new mcpt_ctx();
    c->device = device; c->opts = src->opts; c->opts.device = device; c->width = src->width; c->height = src->height;
    c->wide_depth = src->wide_depth; c->binary_ok = src->binary_ok; c->use_wavefront = src->use_wavefront;
    c->dev = src->dev; c->info = src->info; c->info.bvh_build_ms = 0.0;
    auto bail = [&](hipError_t he, const char* what) { mcpt_status s = hip_fail(he, what); destroy_ctx(c); return s; };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
    auto t0 = std::chrono::steady_clock::now();
    DevBuf* from[10] = {&src->nodes, &src->nodes8, &src->tri_isect, &src->tri_shade, &src->tri_pos64, &src->tri_face, &src->mats, &src->lights, &src->light_pos64, &src->texels};
    DevBuf* to[10] = {&c->nodes, &c->nodes8, &c->tri_isect, &c->tri_shade, &c->tri_pos64, &c->tri_face, &c->mats, &c->lights, &c->light_pos64, &c->texels};
    for (int i = 0; i < 10; i++) {
        if ((e = to[i]->alloc(from[i]->bytes)) != hipSuccess) return bail(e, "alloc scene stream");
        if (from[i]->bytes && (e = hipMemcpyPeer(to[i]->p, device, from[i]->p, src->device, from[i]->bytes)) != hipSuccess) return bail(e, "hipMemcpyPeer");
    }
    const mcpt_status fs = finish_ctx(c);
    if (fs != MCPT_OK) { destroy_ctx(c); return fs; }
    c->info.upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *out_ctx = c;
    return MCPT_OK;
}

mcpt_status mcpt_destroy(mcpt_ctx* ctx) {
    if (!ctx) return MCPT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    destroy_ctx(ctx);
    return MCPT_OK;
}

mcpt_status mcpt_get_scene_info(const mcpt_ctx* ctx, mcpt_scene_info* out) {
    if (!ctx || !out) return fail(MCPT_ERR_INVALID_ARG, "null argument");
    *out = ctx->info;
    return MCPT_OK;
}

static mcpt_status check_pending_jobs(mcpt_ctx* ctx);
// Reads the durations of finished render calls (oldest first).  `block`: wait for all of them -- every entry point that drains the stream
// anyway, and a render call when per-kernel timing is on (its sampled kernel events are per call).  Otherwise only what has finished.
static mcpt_status resolve_timing(mcpt_ctx* c, bool block = true) {
    while (c->timed_tail != c->timed_head) {
        const uint32_t k = c->timed_tail % mcpt_ctx::TIMED;
        if (block) HIP_TRY(hipEventSynchronize(c->ev1[k]));
        else {
            const hipError_t q = hipEventQuery(c->ev1[k]);
            if (q == hipErrorNotReady) break;
            if (q != hipSuccess) return hip_fail(q, "hipEventQuery");
        }
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0[k], c->ev1[k]));
        c->last_kernel_ms = ms; c->total_kernel_ms += ms; c->timed_tail++;
        if (c->use_wavefront && c->time_kernels && c->timed_tail == c->timed_head) {   // (per-kernel timing: calls are resolved one at a time, see mcpt_render_tiles)
            double sh = 0.0, tr = 0.0;
            for (auto& L : c->lanes) {
                if (L.last_timed) {                                          // sampled iterations stand for all of them
                    double s_ = 0.0, t_ = 0.0;
                    for (uint64_t i = 0; i < L.last_timed; i++) {
                        float a_ = 0.f, b_ = 0.f;
                        HIP_TRY(hipEventElapsedTime(&a_, L.k_ev[3 * i], L.k_ev[3 * i + 1]));
                        HIP_TRY(hipEventElapsedTime(&b_, L.k_ev[3 * i + 1], L.k_ev[3 * i + 2]));
                        s_ += a_; t_ += b_;
                    }
                    const double scale = double(L.last_iterations) / double(L.last_timed);
                    sh += s_ * scale; tr += t_ * scale;
                }
                L.last_iterations = 0; L.last_timed = 0;
            }
            c->last_shade_ms = sh; c->last_trace_ms = tr; c->total_shade_ms += sh; c->total_trace_ms += tr;
        }
    }
    return block ? check_pending_jobs(c) : MCPT_OK;                        // (everything has finished: the known-length jobs' snapshots are in)
}

// A sub-pipeline's path pool, allocated on first use and grown (never shrunk) to the largest job seen: a 48 x 48 film gets a few hundred KB,
// the 1024-spp bench job its 2^23 slots (1.4 GB per sub-pipeline).  All slot state is dead between render calls, so growing loses nothing.
static mcpt_status ensure_pool(mcpt_ctx* ctx, mcpt_ctx::WfLane& L, uint32_t P) {
    if (L.pool.P >= P) return MCPT_OK;
    HIP_TRY(hipStreamSynchronize(L.stream)); HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (auto& b : L.pool_bufs) { ctx->info.device_bytes -= b.bytes; b.free_(); }
    // the drain compaction's scratch: half a pool of the seven records a live slot carries from one iteration to the next (+ 4096 slots of rounding)
    const bool want_compact = P >= 4 * WF_COMPACT_MIN_SLOTS && !(ctx->opts.flags & MCPT_FLAG_DETERMINISTIC) && env_u32("MCPT_WF_COMPACT", 1) != 0;
    const uint32_t eighths = std::max(1u, std::min(7u, env_u32("MCPT_WF_COMPACT_EIGHTHS", 4)));      // compact when at most this many eighths of the swept slots are alive
    const uint32_t ccap = want_compact ? uint32_t(uint64_t(P) * eighths / 8) + 4096 : 0;
    L.compact = CompactBufs{}; L.compact.capacity = ccap; L.compact.eighths = eighths;
    L.pool_bufs.clear(); L.pool_bufs.resize(22);
    void** dst[22] = {(void**)&L.pool.ray_o, (void**)&L.pool.ray_d, (void**)&L.pool.hit, (void**)&L.pool.sq_d, (void**)&L.pool.nee,
                      (void**)&L.pool.L, (void**)&L.pool.beta, (void**)&L.pool.sum, (void**)&L.pool.ids, (void**)&L.pool.shadow_queue,
                      (void**)&L.pool.shadow_count, (void**)&L.pool.sq_o, (void**)&L.pool.block_items, (void**)&L.pool.live_cnt,
                      (void**)&L.compact.beta, (void**)&L.compact.L, (void**)&L.compact.ray_d, (void**)&L.compact.ray_o, (void**)&L.compact.hit, (void**)&L.compact.nee,
                      (void**)&L.compact.ids, (void**)&L.compact.dst_off};
    for (int i = 0; i < 22; i++) {
        const size_t bytes = i == 9 ? size_t(P) * sizeof(uint32_t) : (i == 10 || i == 13 || i == 21) ? size_t(P / WF_SHADE_BLOCK) * sizeof(uint32_t) : i == 12 ? size_t(P / WF_SHADE_BLOCK) * sizeof(uint2)
                           : i == 20 ? size_t(ccap) * sizeof(uint2) : (i >= 14 && i <= 19) ? size_t(ccap) * 16 : size_t(P) * 16;
        hipError_t e = L.pool_bufs[i].alloc(bytes);
        if (e != hipSuccess) { L.pool.P = 0; return hip_fail(e, "alloc path pool"); }
        if ((e = hipMemset(L.pool_bufs[i].p, 0, bytes)) != hipSuccess) { L.pool.P = 0; return hip_fail(e, "clear path pool"); }
        *dst[i] = L.pool_bufs[i].p;
        ctx->info.device_bytes += bytes;
    }
    HIP_TRY(hipStreamSynchronize(nullptr));                              // (the fills ran on the default stream)
    L.pool.P = P;
    return MCPT_OK;
}

// Look at the snapshots known-length jobs left behind.  `block`: wait for every one of them (the caller has synchronised, or is about to
// synchronise, the stream anyway); otherwise only those that have arrived.  A job that did not finish inside its bound, or whose trace kernel
// raised the watchdog flag, is an internal error and is reported by whichever call gets here first.
static mcpt_status check_lane_verdicts(mcpt_ctx::WfLane& L, bool block) {
    while (!L.verdicts.empty()) {
        const mcpt_ctx::WfLane::Verdict v = L.verdicts.front();
        if (block) HIP_TRY(hipEventSynchronize(L.chk_ev[v.ring_slot]));
        else {
            const hipError_t q = hipEventQuery(L.chk_ev[v.ring_slot]);
            if (q == hipErrorNotReady) break;
            if (q != hipSuccess) return hip_fail(q, "hipEventQuery");
        }
        L.verdicts.erase(L.verdicts.begin());
        const IterCtl& s = L.h_ctl[v.ring_slot];
        if (s.pad[0]) return fail(MCPT_ERR_HIP, "trace kernel watchdog: a wave did not finish its ray list (internal error)");
        bool items_left = false;
        for (uint32_t q = 0; q < WF_ITEM_SHARDS; q++) items_left |= s.item_cursor[q].v < wf_shard_capacity(v.n_shared, q);
        if (s.any_active[v.it & 3] != 0 || items_left) return fail(MCPT_ERR_HIP, "a known-length job did not finish within its iteration bound (internal error)");
    }
    return MCPT_OK;
}
static mcpt_status check_pending_jobs(mcpt_ctx* ctx) {
    for (auto& L : ctx->lanes) { mcpt_status st = check_lane_verdicts(L, true); if (st != MCPT_OK) return st; }
    return MCPT_OK;
}

static mcpt_status render_wavefront(mcpt_ctx* ctx, RenderParams& p0, float4* accum) {
    // One mcpt_render call = per sub-pipeline a loop of [shade, trace] launches over its slot pool until its work items are done.
    // The sample range is split contiguously over the sub-pipelines; their streams fork from and join the context's stream.
    const uint64_t tiles = p0.n_owned;
    const uint32_t n_lanes = p0.probe_n ? 1u : uint32_t(ctx->lanes.size());    // a probe (mcpt_probe_paths) runs on one sub-pipeline
    const bool count = (p0.flags & MCPT_FLAG_COUNT_TRAVERSAL) != 0;
    constexpr uint32_t CHECK = 4, RING = 8;
    const bool debug = env_u32("MCPT_WF_DEBUG", 0) != 0;
    const uint32_t max_it = env_u32("MCPT_WF_MAXIT", 1u << 20);
    DevCounters* cnt = static_cast<DevCounters*>(ctx->counters.p);
    struct Run { RenderParams p; PathPool pool; uint32_t n_items = 0, n_shared = 0, it = 0, issued = 0, seen = 0, bound = 0, snap_it[RING] = {0}; size_t kev = 0; bool active = false, done = false;
                 uint32_t grid = 0;         // blocks of this sub-pipeline's trace launches (below: small jobs share the CUs instead of queueing for them)
                 bool drain = false; };     // drain: a snapshot showed the shared work-item cursors exhausted -> the compaction launches follow every trace launch from here on
    std::vector<Run> runs(n_lanes);
    uint32_t n_active = 0;
    // A call with fewer samples than sub-pipelines (the reference's one-sample-per-call loop, Render.cpp:56-69) splits its TILES over them
    // instead of its samples -- pipeline k takes every n_lanes-th tile of this call's share -- so that the shade of one still runs
    // beside the trace of the other.  Their pixel sets are disjoint.
    const bool split_tiles = !p0.probe_n && p0.spp < n_lanes && tiles >= n_lanes;
    for (uint32_t k = 0; k < n_lanes; k++) {
        Run& r = runs[k];
        r.p = p0;
        uint64_t my_tiles = tiles;
        if (split_tiles) {
            r.p.tile_mod = p0.tile_mod * n_lanes; r.p.tile_rem = p0.tile_rem + k * p0.tile_mod;
            my_tiles = (tiles - k + n_lanes - 1) / n_lanes; r.p.n_owned = uint32_t(my_tiles);
        } else {
            const uint32_t lo = uint32_t(uint64_t(p0.spp) * k / n_lanes), hi = uint32_t(uint64_t(p0.spp) * (k + 1) / n_lanes);   // (equal shares: 40 / 60 and 35 / 65 splits, so that the two pools do not drain together, were 6 - 10 % slower)
            if (hi == lo) { r.done = true; ctx->lanes[k].last_iterations = 0; ctx->lanes[k].last_timed = 0; continue; }
            r.p.spp = hi - lo; r.p.first_sample = p0.first_sample + lo;
        }
        if (r.p.samples_per_item > r.p.spp) r.p.samples_per_item = r.p.spp;
        r.p.chunks = (r.p.spp + r.p.samples_per_item - 1) / r.p.samples_per_item;
        r.n_items = p0.probe_n ? p0.probe_n : uint32_t(my_tiles * 64 * r.p.chunks);
        // Pool slots for this job: one per work item, at most pool_cap (2^23 by default).  (items_per_slot, a developer knob, default 1: a job of more
        // than 2^20 items gets items / items_per_slot slots, at least 2^20 -- what a slot costs is the end-of-job drain, the last ~8 iterations sweep a
        // pool that is emptying; measured in round 3, fewer slots than the job can fill lose more in short launches than they save in the drain.)
        {
            const uint64_t want64 = ((uint64_t(r.n_items) + WF_SHADE_BLOCK - 1) / WF_SHADE_BLOCK) * WF_SHADE_BLOCK;
            uint64_t P = std::min<uint64_t>(want64, ctx->pool_cap);
            if (want64 > (1ull << 20)) {
                const uint64_t by_items = ((uint64_t(r.n_items) / ctx->items_per_slot) + 16 * WF_SHADE_BLOCK - 1) & ~uint64_t(16 * WF_SHADE_BLOCK - 1);   // (rounded UP: a job just over 2^20 items keeps one slot per item and its known length)
                P = std::min<uint64_t>(P, std::max<uint64_t>(by_items, 1ull << 20));
            }
            mcpt_status ps = ensure_pool(ctx, ctx->lanes[k], uint32_t(P)); if (ps != MCPT_OK) return ps;
            r.pool = ctx->lanes[k].pool;
            r.pool.P = uint32_t(P);                                        // (a smaller job sweeps only the slots it needs)
        }
        // Every item has a slot of its own and one sample: all paths start in iteration 0, vertex b is shaded in iteration b + 1, the
        // depth limit ends the path by iteration max_depth + 1 and a parked NEE term (SLOT_DRAIN) costs one more.  The loop then runs
        // exactly that many iterations before it looks at the control block for the first time -- no launches past the end of the job.
        if (r.n_items <= r.pool.P && r.p.chunks == 1 && r.p.samples_per_item == 1 && p0.max_depth != 0 && !p0.probe_n) r.bound = p0.max_depth + 3;
        // Work items: 90 % are split evenly into one private range per shade block -- the block advances a cursor only it touches, so
        // the returning atomic that used to sit between two barriers of every block is gone from the steady state -- and the last
        // 10 % still come from the shared cursors, which is what balances the blocks at the end of the call.
        r.p.priv_items = 0; r.p.shared_base = 0; r.n_shared = r.n_items;
        const uint32_t n_blocks = r.pool.P / WF_SHADE_BLOCK;
        if (!p0.probe_n && uint64_t(r.n_items) >= 4ull * r.pool.P && env_u32("MCPT_WF_PRIVATE_ITEMS", 1)) {
            r.p.priv_items = uint32_t(0.9 * double(r.n_items) / double(n_blocks)) & ~63u;   // whole 64-item units (block b owns unit k * n_blocks + b)
            r.p.shared_base = n_blocks * r.p.priv_items;
            r.n_shared = r.n_items - r.p.shared_base;
        }
        {   // snapshots of earlier known-length jobs on this sub-pipeline: a polled job starts with none outstanding (it takes the ring's slots
            // in turn from 0), a known-length one needs a free slot for its own
            mcpt_ctx::WfLane& L = ctx->lanes[k];
            mcpt_status vs = check_lane_verdicts(L, r.bound == 0); if (vs != MCPT_OK) return vs;
            if (L.verdicts.size() > RING - 2) { vs = check_lane_verdicts(L, true); if (vs != MCPT_OK) return vs; }
        }
        r.active = true; n_active++;
    }
    for (Run& r : runs) if (r.active) r.p.atomic_accum = ((n_active > 1 && !split_tiles) || r.p.chunks > 1) ? 1u : 0u;
    // Trace grid.  A CU holds ONE trace block (registers), so the trace launches of two sub-pipelines queue for each other's CUs.  That is what the steady
    // state wants (the other pipeline's SHADE runs beside a trace launch); a job with about a ray per trace lane -- the one-sample frame of the reference's
    // display loop: 320 k paths per sub-pipeline, 262 k trace lanes -- has nothing to hide and is a chain of 2 x (depth + 3) dependent launches: there each
    // sub-pipeline's launch takes its share of the CUs and the chains run side by side.  S-cornell 800x800, render + tonemapped read per frame: 2.43 -> 2.06 ms;
    // two samples per call 2.83 -> 2.76, four 3.95 -> 4.73 (profiles/r04_frame_knobs.txt): the split applies up to 2.5 paths per trace lane.  The same holds
    // at the end of a long job once the drain compaction has shrunk the sweep that far (poll, below).
    const uint32_t small_job = env_u32("MCPT_WF_SMALL_JOB_SPLIT", 1) && n_active > 1 ? uint32_t(std::min<uint64_t>(0xffffffffull, uint64_t(ctx->trace_grid) * wf_trace_block_threads() * 5 / 2)) : 0u;
    const uint32_t shared_grid = std::max(1u, ctx->trace_grid / std::max(1u, n_active));
    for (Run& r : runs) if (r.active) r.grid = (small_job && !p0.probe_n && r.n_items <= small_job) ? shared_grid : ctx->trace_grid;
    HIP_TRY(hipEventRecord(ctx->fork_ev, ctx->stream));
    for (uint32_t k = 0; k < n_lanes; k++) {
        if (!runs[k].active) continue;
        mcpt_ctx::WfLane& L = ctx->lanes[k]; Run& r = runs[k];
        HIP_TRY(hipStreamWaitEvent(L.stream, ctx->fork_ev, 0));
        HIP_TRY(launch_wf_pool_reset(r.pool, static_cast<IterCtl*>(L.ctl_buf.p), L.stream));    // every slot DEAD, control block zeroed
    }
    auto k_event = [&](mcpt_ctx::WfLane& L, Run& r, bool timed) -> hipError_t {
        if (!timed) return hipSuccess;
        if (r.kev == L.k_ev.size()) { hipEvent_t ev; hipError_t e = hipEventCreate(&ev); if (e != hipSuccess) return e; L.k_ev.push_back(ev); }
        return hipEventRecord(L.k_ev[r.kev++], L.stream);
    };
    // consume finished control-block snapshots of one sub-pipeline; `block` waits for the oldest one
    auto poll = [&](mcpt_ctx::WfLane& L, Run& r, bool block) -> mcpt_status {
        while (r.seen < r.issued) {
            const uint32_t k = r.seen % RING;
            if (block) { HIP_TRY(hipEventSynchronize(L.chk_ev[k])); block = false; }
            else {
                hipError_t q = hipEventQuery(L.chk_ev[k]);
                if (q == hipErrorNotReady) break;
                if (q != hipSuccess) return hip_fail(q, "hipEventQuery");
            }
            const IterCtl& s = L.h_ctl[k];
            const uint32_t it_of = r.snap_it[k];                           // snapshot taken after iteration it_of
            if (debug && (r.seen < 40 || s.pad[WF_CTL_P_ACTIVE]))
                fprintf(stderr, "[wf] it=%u active=%u head=%u cursor0=%u/%u swept=%u live@compaction=%u compactions=%u\n", it_of, s.any_active[it_of & 3],
                        s.trace_head[it_of & 3], s.item_cursor[0].v, wf_shard_capacity(r.n_shared, 0), s.pad[WF_CTL_P_ACTIVE], s.pad[WF_CTL_LIVE], s.pad[WF_CTL_COMPACTIONS]);
            if (s.pad[0]) return fail(MCPT_ERR_HIP, "trace kernel watchdog: a wave did not finish its ray list (internal error)");
            bool items_left = false;
            for (uint32_t q = 0; q < WF_ITEM_SHARDS; q++) items_left |= s.item_cursor[q].v < wf_shard_capacity(r.n_shared, q);
            if (s.any_active[it_of & 3] == 0 && !items_left) r.done = true;
            // the compaction launches start as soon as the SHARED cursors move at all: a block turns to them when its private range (90 % of the items) is
            // used up, i.e. in the last tenth of the job -- the host reads snapshots 4 - 8 iterations late, and a drain lasts about ten; the plan kernel
            // itself waits until every item has been handed out
            if (small_job && s.pad[WF_CTL_P_ACTIVE] != 0u && s.pad[WF_CTL_P_ACTIVE] <= small_job) r.grid = shared_grid;   // (the compacted sweep of a draining job)
            if (!r.drain) { uint64_t moved = 0; for (uint32_t q = 0; q < WF_ITEM_SHARDS; q++) moved += s.item_cursor[q].v; if (moved != 0 || !items_left) r.drain = true; }
            r.seen++;
        }
        return MCPT_OK;
    };
    bool all_done = n_active == 0;
    while (!all_done) {
        all_done = true;
        for (uint32_t k = 0; k < n_lanes; k++) {                            // one iteration of every live sub-pipeline per round
            Run& r = runs[k];
            if (!r.active || r.done) continue;
            mcpt_ctx::WfLane& L = ctx->lanes[k];
            IterCtl* ctl = static_cast<IterCtl*>(L.ctl_buf.p);
            const bool timed = ctx->time_kernels && r.it % ctx->time_kernels == 0;
            HIP_TRY(k_event(L, r, timed));
            HIP_TRY(launch_wf_shade(ctx->dev, r.p, r.pool, ctl, r.it, r.n_shared, accum, cnt, L.stream));
            HIP_TRY(k_event(L, r, timed));
            HIP_TRY(launch_wf_trace(ctx->dev, r.pool, ctl, r.it, ctx->tune, count, cnt, r.grid, static_cast<int*>(L.ovf_buf.p), L.stream));
            HIP_TRY(k_event(L, r, timed));
            // end-of-job drain: move the live slots to the front of the pool once at most half of the swept ones are alive (decided on the device)
            if (r.drain && !r.bound && L.compact.capacity && r.p.samples_per_item == 1 && !p0.probe_n)
                HIP_TRY(launch_wf_compact(r.pool, L.compact, ctl, r.it, r.n_shared, r.p.priv_items, L.stream));
            r.it++;
            if (r.bound && r.it == r.bound) {                               // known-length job: all of it is enqueued; its verdict is read later
                const uint32_t q = L.ring_next++ % RING;
                HIP_TRY(hipMemcpyAsync(&L.h_ctl[q], ctl, sizeof(IterCtl), hipMemcpyDeviceToHost, L.stream));
                HIP_TRY(hipEventRecord(L.chk_ev[q], L.stream));
                L.verdicts.push_back({q, r.it - 1, r.n_shared});
                r.done = true;
                continue;
            }
            if (!r.bound && r.it % CHECK == 0) {
                mcpt_status ps = poll(L, r, r.issued - r.seen >= 2); if (ps != MCPT_OK) return ps;   // at most 2 checks (8 iterations) ahead
                if (!r.done) {
                    const uint32_t q = r.issued % RING;
                    HIP_TRY(hipMemcpyAsync(&L.h_ctl[q], ctl, sizeof(IterCtl), hipMemcpyDeviceToHost, L.stream));
                    HIP_TRY(hipEventRecord(L.chk_ev[q], L.stream));
                    r.snap_it[q] = r.it - 1;
                    r.issued++;
                }
            }
            if (!r.done) { mcpt_status ps = poll(L, r, false); if (ps != MCPT_OK) return ps; }
            if (r.it > max_it) return fail(MCPT_ERR_HIP, "wavefront loop did not terminate within MCPT_WF_MAXIT iterations");
            if (!r.done) all_done = false;
        }
    }
    for (uint32_t k = 0; k < n_lanes; k++) {
        if (!runs[k].active) continue;
        mcpt_ctx::WfLane& L = ctx->lanes[k];
        L.last_iterations = runs[k].it; L.last_timed = runs[k].kev / 3;
        ctx->total_iterations += runs[k].it;
        HIP_TRY(hipEventRecord(L.done_ev, L.stream));
        HIP_TRY(hipStreamWaitEvent(ctx->stream, L.done_ev, 0));            // join: the caller's stream continues after every sub-pipeline
    }
    return MCPT_OK;
}

mcpt_status mcpt_render(mcpt_ctx* ctx, uint32_t spp, uint64_t seed, uint32_t first_sample) { return mcpt_render_tiles(ctx, spp, seed, first_sample, 1u, 0u); }

mcpt_status mcpt_render_tiles(mcpt_ctx* ctx, uint32_t spp, uint64_t seed, uint32_t first_sample, uint32_t tile_mod, uint32_t tile_rem) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    if (tile_mod == 0 || tile_rem >= tile_mod) return fail(MCPT_ERR_INVALID_ARG, "mcpt_render_tiles: need tile_rem < tile_mod");
    if (spp == 0) return MCPT_OK;
    if (!ctx->use_wavefront && !ctx->binary_ok) return fail(MCPT_ERR_BVH_DEPTH, "the binary tree of this (device-built) scene is deeper than the megakernel's traversal stack");
    // durations of earlier calls: read what has finished; wait only when the ring of event pairs is full -- or when per-kernel timing is on,
    // whose sampled kernel events belong to one call at a time
    st = resolve_timing(ctx, ctx->time_kernels != 0 || ctx->timed_head - ctx->timed_tail >= mcpt_ctx::TIMED - 1); if (st != MCPT_OK) return st;
    RenderParams p; std::memset(&p, 0, sizeof p);
    p.spp = spp; p.first_sample = first_sample;
    p.tiles_x = uint32_t((ctx->width + 7) / 8); p.tiles_y = uint32_t((ctx->height + 7) / 8);
    p.tile_mod = tile_mod; p.tile_rem = tile_rem;
    {   const uint64_t all = uint64_t(p.tiles_x) * p.tiles_y;
        p.n_owned = all > tile_rem ? uint32_t((all - tile_rem + tile_mod - 1) / tile_mod) : 0u; }
    if (p.n_owned == 0) return MCPT_OK;                                   // more shards than tiles: nothing for this one
    const uint64_t tiles = p.n_owned;
    uint32_t spi = ctx->opts.samples_per_item;
    if (ctx->opts.flags & MCPT_FLAG_DETERMINISTIC) spi = spp;               // one lane owns a pixel for the whole call
    else if (spi == 0) {
        if (ctx->use_wavefront) {
            // auto: one sample per item, longer ones only to keep the item count in the cursors' range.  Short items keep the end-of-render
            // drain short (a slot works its item off sample after sample: 8-sample items cost 2.7 % at 1024 spp on the bench workload).
            // Rounds 1-2 also grew the items when many pool slots would share a film pixel (more than 32 per pixel), for fear of the film's
            // float atomics; measured in round 3 that rule was the problem, not the atomics: 64 x 64 x 4096 spp 88 -> 17 ms without it
            // (4 096 slots per pixel), 16 x 16 x 16 384 spp 49 -> 9 ms, 256 x 256 x 1024 spp 60 -> 51 ms, and an interleaved-tile share of
            // the bench job (1/8 of the pixels) 72 -> 58 ms -- the atomics execute at the memory side and 10^4 adders per address are fine.
            spi = 1;
            while (tiles * ((spp + spi - 1) / spi) > 0x3ffffffull && spi < spp) spi <<= 1;
        } else {
            // megakernel: long enough that per-item overheads vanish, short enough that the work balances across the chip
            spi = 64;
            const uint64_t want_items = 256ull * 16 * 16;
            while (spi > 8 && tiles * ((spp + spi - 1) / spi) < want_items) spi >>= 1;
        }
        if (spi > spp) spi = spp;
    }
    if (spi > spp) spi = spp;
    p.samples_per_item = spi; p.chunks = (spp + spi - 1) / spi;
    p.atomic_accum = p.chunks > 1 ? 1u : 0u;
    p.max_depth = ctx->opts.max_depth; p.flags = ctx->opts.flags; p.integrator = ctx->opts.integrator;
    p.seed_lo = uint32_t(seed); p.seed_hi = uint32_t(seed >> 32);
    if (tiles * p.chunks > 0x3ffffffull) return fail(MCPT_ERR_UNSUPPORTED, "launch too large: lower spp per call or raise samples_per_item");
    const uint32_t tk = ctx->timed_head % mcpt_ctx::TIMED;
    HIP_TRY(hipEventRecord(ctx->ev0[tk], ctx->stream));
    if (ctx->use_wavefront) {
        st = render_wavefront(ctx, p, ctx->accum); if (st != MCPT_OK) return st;
    } else {
        HIP_TRY(launch_render(ctx->dev, p, ctx->accum, static_cast<DevCounters*>(ctx->counters.p), ctx->stream));
    }
    HIP_TRY(hipEventRecord(ctx->ev1[tk], ctx->stream));
    ctx->timed_head++; ctx->launches++;
    return MCPT_OK;
}

mcpt_status mcpt_sync(mcpt_ctx* ctx) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return resolve_timing(ctx);
}

mcpt_status mcpt_read_accum(mcpt_ctx* ctx, float* rgba_host) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    if (!rgba_host) return fail(MCPT_ERR_INVALID_ARG, "null output");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(rgba_host, ctx->accum, size_t(ctx->width) * ctx->height * sizeof(float4), hipMemcpyDeviceToHost));
    return resolve_timing(ctx);
}
mcpt_status mcpt_write_accum(mcpt_ctx* ctx, const float* rgba_host) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    if (!rgba_host) return fail(MCPT_ERR_INVALID_ARG, "null input");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(ctx->accum, rgba_host, size_t(ctx->width) * ctx->height * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return MCPT_OK;
}
mcpt_status mcpt_clear_accum(mcpt_ctx* ctx) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    HIP_TRY(hipMemsetAsync(ctx->accum, 0, size_t(ctx->width) * ctx->height * sizeof(float4), ctx->stream));
    return MCPT_OK;
}

// tonemap_kernel over `film` into the context's persistent u8 buffer, copied to its pinned host image on the context's stream; returns after
// the stream has drained (the image is complete).  The kernel writes every byte: nothing to clear.
static mcpt_status tonemap_to_pinned(mcpt_ctx* ctx, const float4* film, int flip_y) {
    const size_t n = size_t(ctx->width) * ctx->height;
    if (!ctx->tone_dev.p) {
        HIP_TRY(ctx->tone_dev.alloc(3 * n));
        HIP_TRY(hipHostMalloc((void**)&ctx->tone_host, 3 * n ? 3 * n : 16, hipHostMallocDefault));
    }
    HIP_TRY(launch_tonemap(film, static_cast<uint8_t*>(ctx->tone_dev.p), ctx->width, ctx->height, flip_y, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->tone_host, ctx->tone_dev.p, 3 * n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return resolve_timing(ctx);
}

mcpt_status mcpt_tonemap(mcpt_ctx* ctx, uint8_t* rgb_host, int flip_y) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    if (!rgb_host) return fail(MCPT_ERR_INVALID_ARG, "null output");
    st = tonemap_to_pinned(ctx, ctx->accum, flip_y); if (st != MCPT_OK) return st;
    std::memcpy(rgb_host, ctx->tone_host, 3 * size_t(ctx->width) * ctx->height);
    return MCPT_OK;
}

/* The same without the last copy: *out_rgb points at the context's own pinned host image (width * height * 3 bytes), valid until the next
 * tonemap call on this context or its destruction -- what Scene::getPixelsColor hands out (Scene.cpp:23-33 returns a pointer into the
 * Scene's own vector, overwritten by the next call). */
mcpt_status mcpt_tonemap_map(mcpt_ctx* ctx, int flip_y, const uint8_t** out_rgb) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    if (!out_rgb) return fail(MCPT_ERR_INVALID_ARG, "null output");
    *out_rgb = nullptr;
    st = tonemap_to_pinned(ctx, ctx->accum, flip_y); if (st != MCPT_OK) return st;
    *out_rgb = ctx->tone_host;
    return MCPT_OK;
}

/* Scene::getPixelsColor of ANY film of this context's size that lives on its device -- e.g. the sum of several devices' films, reduced
 * into a scratch buffer for a progressive image of a multi-GPU render (mcpt_cli --gpus N --save-every K). */
mcpt_status mcpt_tonemap_buffer(mcpt_ctx* ctx, const void* device_rgba, uint8_t* rgb_host, int flip_y) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    if (!rgb_host || !device_rgba) return fail(MCPT_ERR_INVALID_ARG, "null argument");
    st = tonemap_to_pinned(ctx, static_cast<const float4*>(device_rgba), flip_y); if (st != MCPT_OK) return st;
    std::memcpy(rgb_host, ctx->tone_host, 3 * size_t(ctx->width) * ctx->height);
    return MCPT_OK;
}

mcpt_status mcpt_get_counters(mcpt_ctx* ctx, mcpt_counters* out) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    if (!out) return fail(MCPT_ERR_INVALID_ARG, "null output");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    st = resolve_timing(ctx); if (st != MCPT_OK) return st;
    std::vector<DevCounters> rep(WF_COUNTER_REPLICAS);               // kernels spread their atomics over replicas; sum them here
    HIP_TRY(hipMemcpy(rep.data(), ctx->counters.p, sizeof(DevCounters) * WF_COUNTER_REPLICAS, hipMemcpyDeviceToHost));
    DevCounters d; std::memset(&d, 0, sizeof d);
    for (const DevCounters& r : rep) {
        d.paths += r.paths; d.rays_primary += r.rays_primary; d.rays_continuation += r.rays_continuation; d.rays_shadow += r.rays_shadow;
        d.box_tests += r.box_tests; d.tri_tests += r.tri_tests; d.shaded_hits += r.shaded_hits; d.texel_fetches += r.texel_fetches;
        d.self_shadow_tests += r.self_shadow_tests; d.self_shadow_hits += r.self_shadow_hits; d.stack_spills += r.stack_spills; for (int q = 0; q < 4; q++) d.debug[q] += r.debug[q];
    }
    std::memset(out, 0, sizeof *out);
    out->paths = d.paths; out->rays_primary = d.rays_primary; out->rays_continuation = d.rays_continuation; out->rays_shadow = d.rays_shadow;
    out->box_tests = d.box_tests; out->tri_tests = d.tri_tests; out->shaded_hits = d.shaded_hits; out->texel_fetches = d.texel_fetches;
    out->self_shadow_tests = d.self_shadow_tests; out->self_shadow_hits = d.self_shadow_hits; out->stack_spills = d.stack_spills; for (int q = 0; q < 4; q++) out->debug[q] = d.debug[q];
    out->kernel_ms = ctx->last_kernel_ms; out->kernel_ms_total = ctx->total_kernel_ms; out->launches = ctx->launches;
    out->trace_ms_total = ctx->total_trace_ms; out->shade_ms_total = ctx->total_shade_ms; out->iterations = ctx->total_iterations;
    return MCPT_OK;
}
mcpt_status mcpt_reset_counters(mcpt_ctx* ctx) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemset(ctx->counters.p, 0, sizeof(DevCounters) * WF_COUNTER_REPLICAS));
    HIP_TRY(hipStreamSynchronize(nullptr));                          // (default-stream fill vs kernels on the non-blocking context stream: see Scratch)
    st = resolve_timing(ctx); if (st != MCPT_OK) return st;
    ctx->total_kernel_ms = 0.0; ctx->launches = 0; ctx->total_trace_ms = 0.0; ctx->total_shade_ms = 0.0; ctx->total_iterations = 0;
    return MCPT_OK;
}

mcpt_status mcpt_bind_accum(mcpt_ctx* ctx, void* device_rgba) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->accum = device_rgba ? static_cast<float4*>(device_rgba) : static_cast<float4*>(ctx->accum_own.p);
    return MCPT_OK;
}
mcpt_status mcpt_accum_device_ptr(mcpt_ctx* ctx, void** out_device_rgba) {
    if (!ctx || !out_device_rgba) return fail(MCPT_ERR_INVALID_ARG, "null argument");
    *out_device_rgba = ctx->accum;
    return MCPT_OK;
}
mcpt_status mcpt_set_stream(mcpt_ctx* ctx, void* hip_stream) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    st = resolve_timing(ctx); if (st != MCPT_OK) return st;
    ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return MCPT_OK;
}

mcpt_status mcpt_set_null_stream(mcpt_ctx* ctx) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    st = resolve_timing(ctx); if (st != MCPT_OK) return st;
    ctx->stream = nullptr;                                             // the device's legacy default stream
    return MCPT_OK;
}

// ------------------------------------------------------------------------------------------------ probes
mcpt_status mcpt_probe_trace(mcpt_ctx* ctx, uint32_t n, const double* origin, const double* dir, const double* t1, const double* t2, int any_hit,
                             float* out_t, int32_t* out_tri, float* out_u, float* out_v) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    if (!origin || !dir || !t1 || !t2 || !out_t || !out_tri || !out_u || !out_v) return fail(MCPT_ERR_INVALID_ARG, "null argument");
    if (!ctx->binary_ok) return fail(MCPT_ERR_BVH_DEPTH, "the binary cross-check tree of this (device-built) scene is deeper than its kernels' stack: use mcpt_probe_trace4");
    if (n == 0) return MCPT_OK;
    Scratch s; double *d_o, *d_d, *d_t1, *d_t2; float *d_t, *d_u, *d_v; int* d_tri;
    const std::vector<double> lo = to_local(ctx, origin, n);
    HIP_TRY(s.in(lo.data(), 3 * size_t(n), &d_o)); HIP_TRY(s.in(dir, 3 * size_t(n), &d_d)); HIP_TRY(s.in(t1, n, &d_t1)); HIP_TRY(s.in(t2, n, &d_t2));
    HIP_TRY(s.out(n, &d_t)); HIP_TRY(s.out(n, &d_tri)); HIP_TRY(s.out(n, &d_u)); HIP_TRY(s.out(n, &d_v));
    HIP_TRY(launch_probe_trace(ctx->dev, n, d_o, d_d, d_t1, d_t2, any_hit, d_t, d_tri, d_u, d_v, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(out_t, d_t, n * sizeof(float), hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(out_tri, d_tri, n * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_u, d_u, n * sizeof(float), hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(out_v, d_v, n * sizeof(float), hipMemcpyDeviceToHost));
    return MCPT_OK;
}

// The PRODUCTION traversal: the caller's rays are written into a path pool exactly as wf_shade_kernel would leave them (extend rays
// in ray_o / ray_d, shadow rays as sq_o / sq_d records of the per-block shadow queue), wf_trace8_kernel runs once over that pool, and the results
// are read back from where wf_shade_kernel would pick them up (pool.hit; for shadow rays the L += nee the unoccluded ones perform).
mcpt_status mcpt_probe_trace4(mcpt_ctx* ctx, uint32_t n, const double* origin, const double* dir, const double* t2, int any_hit,
                              float* out_t, int32_t* out_tri, float* out_u, float* out_v) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    if (!origin || !dir || !out_t || !out_tri || !out_u || !out_v || (any_hit && !t2)) return fail(MCPT_ERR_INVALID_ARG, "null argument");
    if (!ctx->use_wavefront || ctx->lanes.empty()) return fail(MCPT_ERR_UNSUPPORTED, "mcpt_probe_trace4 needs the wavefront pipeline (MIS integrator)");
    if (n == 0) return MCPT_OK;
    mcpt_ctx::WfLane& L = ctx->lanes[0];
    const uint32_t P = uint32_t(((uint64_t(n) + WF_SHADE_BLOCK - 1) / WF_SHADE_BLOCK) * WF_SHADE_BLOCK);
    if (P > ctx->pool_cap) return fail(MCPT_ERR_UNSUPPORTED, "mcpt_probe_trace4: more rays than pool slots");
    st = ensure_pool(ctx, L, P); if (st != MCPT_OK) return st;
    PathPool pool = L.pool;
    pool.P = P;
    std::vector<float> ro(4 * size_t(P), 0.f), rd(4 * size_t(P), 0.f), sd(4 * size_t(P), 0.f), hit(4 * size_t(P), 0.f);
    std::vector<uint32_t> queue(P, 0u), qcount(P / WF_SHADE_BLOCK, 0u);
    const int32_t no_skip = -1; float no_skip_f; std::memcpy(&no_skip_f, &no_skip, 4);
    for (uint32_t i = 0; i < P; i++) {
        float* o4 = &ro[4 * size_t(i)]; float* d4 = &rd[4 * size_t(i)]; float* s4 = &sd[4 * size_t(i)];
        o4[3] = no_skip_f; d4[2] = 1.f; s4[2] = 1.f;
        if (i >= n) continue;
        // a closest-hit ray with an all-zero direction stands for "this slot has no pending extend ray" (reported as a miss), like a dead slot of a
        // job that is running out: the trace kernel must look past it
        if (!any_hit && dir[3 * size_t(i)] == 0.0 && dir[3 * size_t(i) + 1] == 0.0 && dir[3 * size_t(i) + 2] == 0.0) continue;
        for (int k = 0; k < 3; k++) { o4[k] = float(origin[3 * size_t(i) + k] - ctx->dev.centre[k]); d4[k] = float(dir[3 * size_t(i) + k]); s4[k] = d4[k]; }
        if (any_hit) {
            s4[3] = t2[i] > 3.0e38 ? 3.0e38f : float(t2[i]);
            queue[i] = i;                                               // shade block b queues its own slots in order
            qcount[i / WF_SHADE_BLOCK]++;
        } else { const uint32_t one = 1u; std::memcpy(&d4[3], &one, 4); }   // bit 0 of ray_d.w: "an extend ray is pending"
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(pool.ray_o, ro.data(), ro.size() * 4, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(pool.ray_d, rd.data(), rd.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(pool.sq_d, sd.data(), sd.size() * 4, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(pool.sq_o, ro.data(), ro.size() * 4, hipMemcpyHostToDevice));   // queue entry i = slot i
    HIP_TRY(hipMemset(pool.hit, 0xff, size_t(P) * 16));
    HIP_TRY(hipMemset(pool.nee, 0, size_t(P) * 16));                 // nee.w == 0 afterwards <=> the trace kernel did not flag the ray as blocked
    HIP_TRY(hipMemcpy(pool.shadow_queue, queue.data(), queue.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(pool.shadow_count, qcount.data(), qcount.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(L.ctl_buf.p, 0, sizeof(IterCtl)));
    { const uint32_t one = 1u; HIP_TRY(hipMemcpy(&static_cast<IterCtl*>(L.ctl_buf.p)->any_active[0], &one, 4, hipMemcpyHostToDevice)); }   // "iteration 0 left live slots": the trace kernel returns at once otherwise
    HIP_TRY(hipStreamSynchronize(nullptr));                          // the fills above ran on the default stream; the kernel below does not wait for it by itself
    const bool count = (ctx->opts.flags & MCPT_FLAG_COUNT_TRAVERSAL) != 0;
    HIP_TRY(launch_wf_trace(ctx->dev, pool, static_cast<IterCtl*>(L.ctl_buf.p), 0u, ctx->tune, count, static_cast<DevCounters*>(ctx->counters.p), ctx->trace_grid,
                            static_cast<int*>(L.ovf_buf.p), ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    {   IterCtl snap; HIP_TRY(hipMemcpy(&snap, L.ctl_buf.p, sizeof snap, hipMemcpyDeviceToHost));
        if (snap.pad[0]) return fail(MCPT_ERR_HIP, "trace kernel watchdog: a wave did not finish its ray list (internal error)"); }
    if (any_hit) {
        HIP_TRY(hipMemcpy(hit.data(), pool.nee, hit.size() * 4, hipMemcpyDeviceToHost));     // blocked <=> the trace kernel set nee.w
        for (uint32_t i = 0; i < n; i++) {
            uint32_t flag; std::memcpy(&flag, &hit[4 * size_t(i) + 3], 4);
            out_tri[i] = flag ? 1 : 0; out_t[i] = 0.f; out_u[i] = 0.f; out_v[i] = 0.f;
        }
    } else {
        if (ctx->h_tri_face.empty()) {
            ctx->h_tri_face.resize(size_t(ctx->dev.n_tris));
            HIP_TRY(hipMemcpy(ctx->h_tri_face.data(), ctx->dev.tri_face, ctx->h_tri_face.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        }
        HIP_TRY(hipMemcpy(hit.data(), pool.hit, hit.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; i++) {
            int32_t tri; std::memcpy(&tri, &hit[4 * size_t(i)], 4);
            if (tri >= 0) tri &= HIT_TRI_MASK;                          // the upper bits carry the hit's lobe class for the shade kernel
            if (tri >= ctx->dev.n_tris) return fail(MCPT_ERR_HIP, "mcpt_probe_trace4: trace kernel returned an out-of-range triangle");
            out_tri[i] = tri < 0 ? -1 : ctx->h_tri_face[size_t(tri)];
            out_u[i] = tri < 0 ? 0.f : hit[4 * size_t(i) + 1]; out_v[i] = tri < 0 ? 0.f : hit[4 * size_t(i) + 2]; out_t[i] = tri < 0 ? 0.f : hit[4 * size_t(i) + 3];
        }
    }
    return MCPT_OK;
}

mcpt_status mcpt_probe_cast_ray(mcpt_ctx* ctx, uint32_t n, const int32_t* xy, const float* xi, float* out6) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    if (!xy || !xi || !out6) return fail(MCPT_ERR_INVALID_ARG, "null argument");
    if (n == 0) return MCPT_OK;
    Scratch s; int* d_xy; float *d_xi, *d_out;
    HIP_TRY(s.in(xy, 2 * size_t(n), &d_xy)); HIP_TRY(s.in(xi, 2 * size_t(n), &d_xi)); HIP_TRY(s.out(6 * size_t(n), &d_out));
    HIP_TRY(launch_probe_cast_ray(ctx->dev, n, d_xy, d_xi, d_out, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(out6, d_out, 6 * size_t(n) * sizeof(float), hipMemcpyDeviceToHost));
    return MCPT_OK;
}

mcpt_status mcpt_probe_hit_shade(mcpt_ctx* ctx, uint32_t n, const int32_t* face, const float* u, const float* v, const double* dir, float* out6) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    if (!face || !u || !v || !dir || !out6) return fail(MCPT_ERR_INVALID_ARG, "null argument");
    if (n == 0) return MCPT_OK;
    if (ctx->h_tri_face.empty()) {
        ctx->h_tri_face.resize(size_t(ctx->dev.n_tris));
        HIP_TRY(hipMemcpy(ctx->h_tri_face.data(), ctx->dev.tri_face, ctx->h_tri_face.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    std::vector<int32_t> leaf_of_face(ctx->h_tri_face.size(), -1), tri(n);
    for (size_t i = 0; i < ctx->h_tri_face.size(); i++) leaf_of_face[size_t(ctx->h_tri_face[i])] = int32_t(i);
    for (uint32_t i = 0; i < n; i++) {
        if (face[i] < 0 || size_t(face[i]) >= leaf_of_face.size()) return fail(MCPT_ERR_INVALID_ARG, "mcpt_probe_hit_shade: face index out of range");
        tri[i] = leaf_of_face[size_t(face[i])];
    }
    Scratch s; int* d_tri; float *d_u, *d_v, *d_out; double* d_d;
    HIP_TRY(s.in(tri.data(), n, &d_tri)); HIP_TRY(s.in(u, n, &d_u)); HIP_TRY(s.in(v, n, &d_v)); HIP_TRY(s.in(dir, 3 * size_t(n), &d_d)); HIP_TRY(s.out(6 * size_t(n), &d_out));
    HIP_TRY(launch_probe_hit_shade(ctx->dev, n, d_tri, d_u, d_v, d_d, d_out, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(out6, d_out, 6 * size_t(n) * sizeof(float), hipMemcpyDeviceToHost));
    return MCPT_OK;
}

mcpt_status mcpt_probe_bsdf(mcpt_ctx* ctx, uint32_t n, const float* normal, const float* wi, const float* kd, const float* ks, const float* ns,
                            const float* wo, const float* xi, float* out12) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    if (!normal || !wi || !kd || !ks || !ns || !wo || !xi || !out12) return fail(MCPT_ERR_INVALID_ARG, "null argument");
    if (n == 0) return MCPT_OK;
    Scratch s; float *d_n, *d_wi, *d_kd, *d_ks, *d_ns, *d_wo, *d_xi, *d_out;
    HIP_TRY(s.in(normal, 3 * size_t(n), &d_n)); HIP_TRY(s.in(wi, 3 * size_t(n), &d_wi)); HIP_TRY(s.in(kd, 3 * size_t(n), &d_kd));
    HIP_TRY(s.in(ks, 3 * size_t(n), &d_ks)); HIP_TRY(s.in(ns, n, &d_ns)); HIP_TRY(s.in(wo, 3 * size_t(n), &d_wo)); HIP_TRY(s.in(xi, 3 * size_t(n), &d_xi));
    HIP_TRY(s.out(12 * size_t(n), &d_out));
    HIP_TRY(launch_probe_bsdf(n, d_n, d_wi, d_kd, d_ks, d_ns, d_wo, d_xi, d_out, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(out12, d_out, 12 * size_t(n) * sizeof(float), hipMemcpyDeviceToHost));
    return MCPT_OK;
}

mcpt_status mcpt_probe_sample_light(mcpt_ctx* ctx, uint32_t n, const double* point, const float* xi, float* out10) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    if (!point || !xi || !out10) return fail(MCPT_ERR_INVALID_ARG, "null argument");
    if (n == 0) return MCPT_OK;
    Scratch s; double* d_p; float *d_xi, *d_out;
    const std::vector<double> lp = to_local(ctx, point, n);
    HIP_TRY(s.in(lp.data(), 3 * size_t(n), &d_p)); HIP_TRY(s.in(xi, 3 * size_t(n), &d_xi)); HIP_TRY(s.out(10 * size_t(n), &d_out));
    HIP_TRY(launch_probe_sample_light(ctx->dev, n, d_p, d_xi, d_out, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(out10, d_out, 10 * size_t(n) * sizeof(float), hipMemcpyDeviceToHost));
    return MCPT_OK;
}

mcpt_status mcpt_probe_paths(mcpt_ctx* ctx, uint32_t n, const double* origin, const double* dir, uint64_t seed, float* out_L3) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    if (!origin || !dir || !out_L3) return fail(MCPT_ERR_INVALID_ARG, "null argument");
    if (n == 0) return MCPT_OK;
    if (ctx->opts.integrator != MCPT_INTEGRATOR_MIS) return fail(MCPT_ERR_UNSUPPORTED, "mcpt_probe_paths drives the MIS integrator only");
    Scratch s; double *d_o, *d_d; float* d_out; DevCounters* d_cnt;
    const std::vector<double> lo = to_local(ctx, origin, n);
    HIP_TRY(s.in(lo.data(), 3 * size_t(n), &d_o)); HIP_TRY(s.in(dir, 3 * size_t(n), &d_d)); HIP_TRY(s.out(3 * size_t(n), &d_out)); HIP_TRY(s.out(1, &d_cnt));
    RenderParams p; std::memset(&p, 0, sizeof p);
    p.spp = 1; p.first_sample = 0; p.samples_per_item = 1; p.chunks = 1; p.tiles_x = 0x7fffffffu; p.tiles_y = 1; p.tile_mod = 1; p.tile_rem = 0; p.n_owned = 0x7fffffffu;
    p.max_depth = ctx->opts.max_depth; p.flags = ctx->opts.flags & ~MCPT_FLAG_COUNT_TRAVERSAL; p.integrator = MCPT_INTEGRATOR_MIS;
    p.seed_lo = uint32_t(seed); p.seed_hi = uint32_t(seed >> 32);
    if (ctx->use_wavefront) {
        // the production pipeline: [wf_shade, wf_trace] iterations over the path pool, item i = entry i of an n x 1 film
        if (n > 0x3ffffffu) return fail(MCPT_ERR_UNSUPPORTED, "mcpt_probe_paths: too many paths for one call");
        float4* d_film = nullptr;
        HIP_TRY(s.out(size_t(n), &d_film));
        p.probe_n = n; p.probe_o = d_o; p.probe_d = d_d;
        st = render_wavefront(ctx, p, d_film); if (st != MCPT_OK) return st;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for (auto& L : ctx->lanes) { L.last_iterations = 0; L.last_timed = 0; }
        std::vector<float> film(4 * size_t(n));
        HIP_TRY(hipMemcpy(film.data(), d_film, film.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; i++) {
            if (film[4 * size_t(i) + 3] != 1.f) return fail(MCPT_ERR_HIP, "mcpt_probe_paths: a probe path did not finish exactly once");
            for (int k = 0; k < 3; k++) out_L3[3 * size_t(i) + k] = film[4 * size_t(i) + k];
        }
        return MCPT_OK;
    }
    if (!ctx->binary_ok) return fail(MCPT_ERR_BVH_DEPTH, "the binary tree of this (device-built) scene is deeper than the megakernel's traversal stack");
    HIP_TRY(launch_probe_paths(ctx->dev, p, n, d_o, d_d, d_out, d_cnt, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(out_L3, d_out, 3 * size_t(n) * sizeof(float), hipMemcpyDeviceToHost));
    return MCPT_OK;
}

mcpt_status mcpt_probe_texture(mcpt_ctx* ctx, uint32_t material, uint32_t n, const float* uv2, float* out_rgb3) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    if (!uv2 || !out_rgb3) return fail(MCPT_ERR_INVALID_ARG, "null argument");
    if (material >= uint32_t(ctx->dev.n_mats)) return fail(MCPT_ERR_INVALID_ARG, "material index out of range");
    if (n == 0) return MCPT_OK;
    Scratch s; float *d_uv, *d_out;
    HIP_TRY(s.in(uv2, 2 * size_t(n), &d_uv)); HIP_TRY(s.out(3 * size_t(n), &d_out));
    HIP_TRY(launch_probe_texture(ctx->dev, int(material), n, d_uv, d_out, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(out_rgb3, d_out, 3 * size_t(n) * sizeof(float), hipMemcpyDeviceToHost));
    return MCPT_OK;
}

mcpt_status mcpt_probe_rng(mcpt_ctx* ctx, uint32_t n, const uint32_t* key3, uint64_t seed, float* out4) {
    mcpt_status st = use(ctx); if (st != MCPT_OK) return st;
    if (!key3 || !out4) return fail(MCPT_ERR_INVALID_ARG, "null argument");
    if (n == 0) return MCPT_OK;
    Scratch s; uint32_t* d_k; float* d_out;
    HIP_TRY(s.in(key3, 3 * size_t(n), &d_k)); HIP_TRY(s.out(4 * size_t(n), &d_out));
    HIP_TRY(launch_probe_rng(n, d_k, uint32_t(seed), uint32_t(seed >> 32), d_out, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(out4, d_out, 4 * size_t(n) * sizeof(float), hipMemcpyDeviceToHost));
    return MCPT_OK;
}

}  // extern "C"
