/* mcpt.h -- C ABI of the MI355X-native path-tracing hot path (libmcpt_hip.so).
 *
 * Drop-in boundary for laizesheng1/Monte-Carlo-Path-Tracer's `Render` class: everything a caller hands
 * over is the reference's own `Model` data (src/model.h:51-60) as plain pointers + counts, and what it
 * gets back is the reference's film accumulator `Pixels{vec3 color; float spp}` (src/Scene.h:7-12).
 * The reference has no FFI layer (it is one C++ executable); the entry points below are what a binding
 * for its Render::Render / Render::render pair would call -- see INTEGRATION.md for the ~40-line
 * `Render` replacement a maintainer would add.
 *
 * Plain C, no HIP / torch / STL types.  Every function returns an mcpt_status; nothing throws.
 * There is NO CPU fallback: without a usable HIP device mcpt_create fails with MCPT_ERR_NO_DEVICE.
 */
#ifndef MCPT_H
#define MCPT_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCPT_ABI_VERSION 4

typedef enum mcpt_status {
    MCPT_OK = 0,
    MCPT_ERR_INVALID_ARG = 1,   /* null pointer, out-of-range index in `face`, zero-sized image ... */
    MCPT_ERR_NO_DEVICE = 2,     /* no HIP device / device ordinal out of range */
    MCPT_ERR_HIP = 3,           /* a HIP runtime call failed; see mcpt_last_error() */
    MCPT_ERR_NO_LIGHTS = 4,     /* scene has no emissive triangle (reference: UB, Render.cpp:204-206) */
    MCPT_ERR_BVH_DEPTH = 5,     /* BVH deeper than the traversal stack the kernels were built for */
    MCPT_ERR_UNSUPPORTED = 6
} mcpt_status;

/* Texture (src/model.h:21-30): `image_color` as w*h RGB fp32 texels, row 0 first as stb_image returns
 * them (src/model.cpp:8-23).  A 1x1 texture is the constant Kd colour (model.cpp:25-28,32-35). */
typedef struct mcpt_texture {
    int32_t width, height;
    const float* rgb;
} mcpt_texture;

/* Material (src/model.h:32-40).  Tr and Ni are parsed by the reference but never read (SURVEY A-21). */
typedef struct mcpt_material {
    double ks[3];
    double ns;
    double radiance[3];   /* from the XML <light mtlname radiance> (model.cpp:181-182) */
    int32_t map_kd;       /* index into textures[] (Map_Kd) */
    int32_t reserved;
} mcpt_material;

/* CameraInfo (src/model.h:42-49) */
typedef struct mcpt_camera {
    double eye[3], lookat[3], up[3];
    double fovy;          /* vertical field of view in degrees (Render.cpp:73) */
    int32_t width, height;
} mcpt_camera;

/* Model (src/model.h:51-60) */
typedef struct mcpt_scene_desc {
    const double* vertex;   uint32_t n_vertex;    /* xyz per vertex   (Model::vertex)  */
    const double* normal;   uint32_t n_normal;    /* xyz per normal   (Model::normal)  */
    const double* texcoord; uint32_t n_texcoord;  /* uv per entry     (Model::texture) */
    const int32_t* face;    uint32_t n_face;      /* 12 ints per face = glm::imat3x4 (Model::face): for each of the
                                                     3 corners {vertex idx, normal idx, texcoord idx, material idx},
                                                     0-based; the material of a face is corner 0's (Render.cpp:33) */
    const mcpt_material* materials; uint32_t n_materials;
    const mcpt_texture* textures;   uint32_t n_textures;
    mcpt_camera camera;
} mcpt_scene_desc;

/* integrators */
#define MCPT_INTEGRATOR_MIS            0u  /* Render::ray_tracing(Ray&)      Render.cpp:111-175 -- the one that ships */
#define MCPT_INTEGRATOR_RECURSIVE_NEE  1u  /* Render::ray_tracing(Ray&,int)  Render.cpp:83-109 + sample_light :177-200 */

/* flags */
#define MCPT_FLAG_CORRECT_SHADOW_T2   0x1u  /* do NOT reproduce the reference's light self-occlusion (SURVEY A-9):
                                               the sampled light triangle is ignored by its own shadow ray */
#define MCPT_FLAG_DETERMINISTIC       0x2u  /* one thread owns a pixel for the whole call: no float atomics,
                                               bit-reproducible accumulator, worse tail balance */
#define MCPT_FLAG_COUNT_TRAVERSAL     0x4u  /* also count box tests / triangle tests / shaded hits (roofline input) */
#define MCPT_FLAG_GPU_BVH_BUILD       0x8u  /* build the BVH on the device -- SAH-costed agglomerative clustering over the Morton order (PLOC) --
                                               instead of the host's binned-SAH builder: ~2x faster construction, within ~5 % of the host
                                               tree's render speed; rendered results are the same (closest hit does not depend on the tree) */

#define MCPT_FLAG_REFERENCE_TIE_ORDER 0x10u /* among triangles hit at EXACTLY the same distance the one that comes first in the reference's own
                                               BVH::triangles order wins (BVH.cpp:15-54 + :95-113: left, right, own triangles; `t < t2` strict) --
                                               mcpt_create then replays the reference's midpoint partition to learn that order (O(n log n) on the
                                               host).  Default: the lowest index in this library's leaf order wins (any fixed rule gives the same
                                               image up to measure-zero ties; this flag is for tie-break-exact known-answer tests).
                                               The tie rule lives in the production pipeline only (wavefront trace kernel, MCPT_INTEGRATOR_MIS): mcpt_create
                                               refuses the flag with MCPT_ERR_UNSUPPORTED for MCPT_INTEGRATOR_RECURSIVE_NEE and for the cross-check megakernel
                                               (MCPT_PIPELINE=mega), whose binary-tree traversal lets the first triangle IT tests win -- as does mcpt_probe_trace */

typedef struct mcpt_opts {
    uint32_t struct_size;       /* = sizeof(mcpt_opts) */
    int32_t  device;            /* HIP device ordinal */
    uint32_t max_depth;         /* 0 = unbounded like the reference; N = stop before shading vertex N
                                   (`for (bounces = 0; bounces < N; ...)`, Render.cpp:116) */
    uint32_t integrator;        /* MCPT_INTEGRATOR_* */
    uint32_t flags;             /* MCPT_FLAG_* */
    uint32_t samples_per_item;  /* samples of one pixel traced back-to-back by one lane; 0 = auto */
    uint32_t reserved[4];
} mcpt_opts;

typedef struct mcpt_counters {
    uint64_t paths;             /* pixel-samples finished */
    uint64_t rays_primary;      /* camera rays traced (Render.cpp:64) */
    uint64_t rays_continuation; /* BSDF-sampled rays traced (Render.cpp:144); the reference's duplicate re-trace
                                   at Render.cpp:118 is never performed and never counted */
    uint64_t rays_shadow;       /* shadow rays traced, i.e. light samples with pdf != 0 (Render.cpp:125) */
    uint64_t box_tests;         /* AABB slab tests        (only with MCPT_FLAG_COUNT_TRAVERSAL) */
    uint64_t tri_tests;         /* triangle tests         (only with MCPT_FLAG_COUNT_TRAVERSAL) */
    uint64_t shaded_hits;       /* hits whose shading record was fetched (only with COUNT_TRAVERSAL) */
    uint64_t texel_fetches;     /* image-texture lookups  (only with COUNT_TRAVERSAL) */
    uint64_t self_shadow_tests; /* light samples that reached the fp64 self-hit predicate (A-9) */
    uint64_t self_shadow_hits;  /* ... and were rejected by it */
    double   kernel_ms;         /* HIP-event duration of all render kernels of the LAST mcpt_render call */
    double   kernel_ms_total;   /* sum of those durations over all mcpt_render calls since the last reset */
    uint64_t launches;          /* mcpt_render calls since the last reset */
    double   trace_ms_total;    /* wavefront pipeline, detailed timing on: summed duration of the traversal kernel ... */
    double   shade_ms_total;    /* ... and of the shade kernel since the last reset (0 when detailed timing is off) */
    uint64_t iterations;        /* [shade, trace] iterations since the last reset (each is one launch of either kernel) */
    uint64_t stack_spills;      /* traversal-stack entries that left LDS for the global overflow area (only with COUNT_TRAVERSAL) */
    uint64_t debug[4];          /* diagnostic library builds only (tools/sched_stats.py); 0 otherwise */
} mcpt_counters;

typedef struct mcpt_scene_info {
    uint32_t n_tris, n_lights, n_nodes, bvh_depth, max_leaf;
    uint32_t width, height;
    uint64_t device_bytes;      /* HBM held by the scene (nodes + triangle streams + textures + accumulator + path pools allocated so far) */
    double   bvh_build_ms, upload_ms;
    /* ABI 3: the wide tree the wavefront trace kernel walks (8 children per node; wide_width is always 8 since round 4, when the round-2 4-wide kernel was removed) */
    uint32_t wide_width, wide_nodes, wide_depth, reserved0;
    uint64_t traversal_bytes;   /* wide nodes + triangle intersection records: what a ray's traversal can touch */
    double   centre[3];         /* device coordinates are relative to this point (the fp64 centre of the scene's bounding box) */
    uint64_t wide_tree_hash;    /* FNV-1a over the wide tree's records and the leaf order: equal hashes = the same tree and triangle order
                                   (how the tests tell that the device collapse reproduces the host collapse bit for bit) */
} mcpt_scene_info;

typedef struct mcpt_ctx mcpt_ctx;

/* ---- lifecycle -------------------------------------------------------------------------------------- */
/* Replaces Render::Render(Model&) (Render.cpp:5-10): copies what it needs from `scene` (the caller may free it
 * afterwards, like the reference's by-value `model` member, Render.h:57), flattens faces into triangles and
 * collects emissive ones as lights (tranform_triangle, Render.cpp:12-44), builds the BVH (BVH.cpp:6-54) and
 * uploads everything to HBM.  Allocates a zeroed width*height accumulator. */
mcpt_status mcpt_create(const mcpt_scene_desc* scene, const mcpt_opts* opts, mcpt_ctx** out_ctx);
mcpt_status mcpt_destroy(mcpt_ctx* ctx);
/* A second context for the same scene on device `device` (may equal the source's): the scene streams are copied device to device, nothing is
 * flattened or built again.  The clone has its own film, counters, stream and options (those of `src`, device replaced). */
mcpt_status mcpt_clone_to_device(mcpt_ctx* src, int32_t device, mcpt_ctx** out_ctx);
/* Host-only half of mcpt_create: validates `scene` (same error codes) and runs the same flatten + BVH build, without
 * touching a device.  Fills n_tris / n_lights / n_nodes / bvh_depth / max_leaf / width / height / bvh_build_ms. */
mcpt_status mcpt_check_scene(const mcpt_scene_desc* scene, mcpt_scene_info* out_info);
mcpt_status mcpt_get_scene_info(const mcpt_ctx* ctx, mcpt_scene_info* out);
const char* mcpt_last_error(void);   /* thread-local, valid until the next failing call on this thread */
uint32_t    mcpt_abi_version(void);

/* ---- the hot path ----------------------------------------------------------------------------------- */
/* Replaces `spp` consecutive calls of Render::render(Scene&) (Render.cpp:56-69): adds `spp` samples to EVERY
 * pixel of the device accumulator (sum rgb + sample count, NaN components zeroed first like Scene::set_Pixel,
 * Scene.cpp:12-21).  Samples are numbered first_sample .. first_sample+spp-1; a sample's random numbers depend
 * only on (seed, pixel, sample index), so any split of a sample range over calls, GPUs or ranks yields the
 * same image up to fp32 summation order.  Asynchronous on the context's stream. */
mcpt_status mcpt_render(mcpt_ctx* ctx, uint32_t spp, uint64_t seed, uint32_t first_sample);
/* The same for the pixels of ONE interleaved share of the image only: 8x8-pixel tiles are numbered row-major and this call renders the
 * tiles t with t % tile_mod == tile_rem (every other pixel of the film is left untouched).  GPU g of G renders (G, g): the
 * "pixel-tile shard" of BASELINE.json's bathroom2 configuration -- the films of the G shares are disjoint and their sum (the same RCCL
 * all-reduce as for sample sharding) is the full image.  (1, 0) = mcpt_render. */
mcpt_status mcpt_render_tiles(mcpt_ctx* ctx, uint32_t spp, uint64_t seed, uint32_t first_sample, uint32_t tile_mod, uint32_t tile_rem);
mcpt_status mcpt_sync(mcpt_ctx* ctx);

/* Film = Scene::m_Pixels (Scene.h:7-12,25): width*height records {r_sum, g_sum, b_sum, spp}, index y*width+x,
 * y = 0 at the image bottom (Render.cpp:63, Scene.cpp:14). */
mcpt_status mcpt_read_accum(mcpt_ctx* ctx, float* rgba_host);     /* synchronises, then D2H */
mcpt_status mcpt_write_accum(mcpt_ctx* ctx, const float* rgba_host);   /* resume / merge */
mcpt_status mcpt_clear_accum(mcpt_ctx* ctx);
/* Scene::getPixelsColor (Scene.cpp:23-33) on the device: mean -> clamp[0,1] -> pow(.,0.5) -> *255.99 -> u8.
 * flip_y != 0 additionally applies Scene::save_image's vertical flip (Scene.cpp:40-46). */
mcpt_status mcpt_tonemap(mcpt_ctx* ctx, uint8_t* rgb_host, int flip_y);
/* ABI 4: the same without the last host copy -- *out_rgb points at the context's own pinned host image (width * height * 3 bytes), valid until
 * the next tonemap call on this context or mcpt_destroy.  This is Scene::getPixelsColor's own contract (Scene.cpp:23-33 returns a pointer
 * into a vector the next call overwrites), and what the reference's loop calls after EVERY render(scene) (main.cpp:26-33). */
mcpt_status mcpt_tonemap_map(mcpt_ctx* ctx, int flip_y, const uint8_t** out_rgb);

/* mcpt_tonemap of any film of this context's size resident on its device (e.g. several devices' films summed into a scratch buffer). */
mcpt_status mcpt_tonemap_buffer(mcpt_ctx* ctx, const void* device_rgba, uint8_t* rgb_host, int flip_y);
mcpt_status mcpt_get_counters(mcpt_ctx* ctx, mcpt_counters* out);  /* synchronises */
mcpt_status mcpt_reset_counters(mcpt_ctx* ctx);

/* ---- plumbing for multi-GPU hosts (one context per GPU / rank) ---------------------------------------- */
/* Use a caller-owned device buffer of width*height*4 floats as the accumulator (e.g. a torch tensor that
 * torch.distributed / RCCL all-reduces in place).  NULL re-binds the internal buffer. */
mcpt_status mcpt_bind_accum(mcpt_ctx* ctx, void* device_rgba);
mcpt_status mcpt_accum_device_ptr(mcpt_ctx* ctx, void** out_device_rgba);
/* Launch on a caller-owned hipStream_t.  NULL = back to the context's own (non-blocking) stream -- NOT the device's legacy
 * default stream, whose handle is also 0: a caller that wants its work ordered with the default stream (e.g. torch's
 * default stream, `cuda_stream == 0`) says so with mcpt_set_null_stream(). */
mcpt_status mcpt_set_stream(mcpt_ctx* ctx, void* hip_stream);
mcpt_status mcpt_set_null_stream(mcpt_ctx* ctx);

/* ---- function-level probes (what the parity tests call; each maps to one reference function) ---------- */
/* BVH::hit (BVH.cpp:90-113) / BVH::has_hit (BVH.cpp:115-136) for n host rays.  origin/dir: 3 doubles per ray.
 * t1,t2: per-ray interval.  Outputs (closest): t (fp32), triangle index in face order (-1 = miss), barycentric
 * u,v.  any_hit != 0: out_tri[i] = 1/0 only.  This is the binary-tree cross-check traversal: among triangles at EXACTLY the same distance
 * the first one in ITS traversal order wins (neither of the production kernel's tie rules: on coincident geometry it may name another face
 * than mcpt_probe_trace4 at the same t). */
mcpt_status mcpt_probe_trace(mcpt_ctx* ctx, uint32_t n, const double* origin, const double* dir,
                             const double* t1, const double* t2, int any_hit,
                             float* out_t, int32_t* out_tri, float* out_u, float* out_v);
/* The same two reference functions through the PRODUCTION traversal kernel (wf_trace8_kernel over the 8-wide compressed tree:
 * LDS top levels, LDS + global overflow group stack, chunked ray list): the rays are placed in a path pool the way the shade kernel
 * leaves them, the trace kernel runs once, results come back from the pool.  t1 is the kernel's fixed 1e-4 (Render.h:30);
 * closest-hit rays are unbounded (t2 = DBL_MAX like cast_Ray / BSDF rays), any-hit rays use t2[i] (Render.cpp:219-221).
 * Same outputs as mcpt_probe_trace. */
mcpt_status mcpt_probe_trace4(mcpt_ctx* ctx, uint32_t n, const double* origin, const double* dir, const double* t2, int any_hit,
                              float* out_t, int32_t* out_tri, float* out_u, float* out_v);
/* Render::cast_Ray (Render.cpp:71-80) for n (x,y) pixels with the xi the caller supplies (2 per ray). */
/* Triangle::hit's shading record (Triangle.cpp:68-76: interplote_Normal, normalize, interplote_uv, front = dot(n, d) < 0) for hits the
 * caller got from mcpt_probe_trace4: `face` = Model::face index, (u, v) = the hit's barycentrics, dir = the ray's direction.
 * out6 per hit = normal xyz | uv | front (1 / 0). */
mcpt_status mcpt_probe_hit_shade(mcpt_ctx* ctx, uint32_t n, const int32_t* face, const float* u, const float* v, const double* dir, float* out6);
mcpt_status mcpt_probe_cast_ray(mcpt_ctx* ctx, uint32_t n, const int32_t* xy, const float* xi, float* out_origin_dir6);
/* BSDF (BSDF.cpp:87-202) on synthetic hits: per item normal[3], wi[3], kd[3], ks[3], ns, wo[3] (world) and 3 xi
 * {lobe, xi1, xi2}.  out per item: Fx(wo)[3], Pdf(wo), sample.wo[3], sample.f[3], sample.pdf, isMirror = 12 floats */
mcpt_status mcpt_probe_bsdf(mcpt_ctx* ctx, uint32_t n, const float* normal, const float* wi, const float* kd,
                            const float* ks, const float* ns, const float* wo, const float* xi, float* out12);
/* Render::sample (Render.cpp:202-223) from n shading points (3 doubles each) with 3 xi each.
 * out per item: wo[3], radiance[3], pdf, t2, light triangle index (as float), self_hit (0/1; the fp64 predicate
 * of SURVEY A-9 evaluated on the sampled triangle only) = 10 floats */
mcpt_status mcpt_probe_sample_light(mcpt_ctx* ctx, uint32_t n, const double* point, const float* xi, float* out10);
/* One full path per item through the shipping integrator from a given ray, random numbers from the counter-based
 * generator keyed (seed, pixel = item, sample = 0).  out: L[3].  Runs the production wavefront pipeline (wf_shade_kernel +
 * wf_trace8_kernel over a path pool, item = entry of an n x 1 film); the cross-check megakernel only under MCPT_PIPELINE=mega. */
mcpt_status mcpt_probe_paths(mcpt_ctx* ctx, uint32_t n, const double* origin, const double* dir, uint64_t seed, float* out_L3);
/* Texture::get_color (model.cpp:30-41) of material `material`'s Map_Kd for n (u, v) pairs (fp32, as the device interpolates them):
 * nearest texel, fract + clamp01's 0.999 cap, no v flip; a 1x1 texture returns its constant colour. */
mcpt_status mcpt_probe_texture(mcpt_ctx* ctx, uint32_t material, uint32_t n, const float* uv2, float* out_rgb3);
/* The generator itself: n*4 uniforms for (pixel, sample, block) triples -- pins oracle and device to one stream. */
mcpt_status mcpt_probe_rng(mcpt_ctx* ctx, uint32_t n, const uint32_t* pixel_sample_block3, uint64_t seed, float* out4);

#ifdef __cplusplus
}
#endif
#endif /* MCPT_H */
