// HBM layout of a scene, shared by the host (upload) and the gfx950 kernels.  See DESIGN.md §Data layout.
//
// The reference keeps a pointer-linked BVH of heap nodes over shared_ptr<Triangle> (BVH.h:11-20, 88 B/node +
// 256 B AoS triangles, Triangle.h:11-14).  Here everything is a flat, index-linked array, split into a HOT
// stream touched by traversal and COLD streams touched once per shaded hit ("structure of streams"; a
// per-component x[] y[] z[] split would turn each gather into 3x more cache lines, so inside a stream the
// record of one node / triangle is contiguous and 16-B aligned for global_load_dwordx4):
//
//   nodes      64 B / inner node : the TWO CHILD boxes + two child links  (one fetch -> two slab tests)
//                f4[0] = c0.lo.x c0.hi.x c0.lo.y c0.hi.y
//                f4[1] = c1.lo.x c1.hi.x c1.lo.y c1.hi.y
//                f4[2] = c0.lo.z c0.hi.z c1.lo.z c1.hi.z
//                f4[3] = (int) child0, child1, 0, 0     child >= 0: inner node index; child < 0: leaf, ~child =
//                                                         first_triangle << 3 | count  (count <= 7, leaf order)
//   tri_isect  48 B / triangle   : v0.xyz,c | e1.xyz,_ | e2.xyz,_   (fp32; e = float(v_k - v_0) like Triangle.cpp:25-26)
//                                  c = uint bits HIT_CLASS_* << 28: the lobe set BSDF::BSDF will build for this triangle's
//                                  material (BSDF.cpp:95-107); the trace kernel ORs it into the triangle index of a closest hit
//   tri_shade 128 B / triangle   : ONE 128-B-aligned record = one cache line per shaded hit (round 4; r03: a 64-B shading record and a 32-B fp64 plane in two
//                                  streams = two lines -- the L2 fetches whole 128-B lines for a gather, profiles/r04_fetch_calibration.txt):
//                                  f4[0..3] n0.xyz uv0.x | n1.xyz uv0.y | n2.xyz uv1.x | uv1.y uv2.x uv2.y material
//                                  f4[4..5] the triangle's fp64 plane (n.xyz, n.v0): the fp64 hit point of a shaded hit   f4[6..7] spare
//   tri_pos64  72 B / triangle   : v0 v1 v2 in fp64 -- read once per shaded hit to form the fp64 hit point the
//                                  reference's shadow-ray self-occlusion depends on (SURVEY A-9), and per light sample
//   tri_face    4 B / triangle   : leaf order -> face index of the input (Model::face order)
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

// 8-wide tree: every quantised child plane lies at least this many quantisation steps outside the child's box (build_bvh8,
// gpu_collapse_bvh8) -- the margin the trace kernel's fp16-mix plane arithmetic needs (wavefront.hip, WF8_CHILD)
#ifndef MCPT_Q_MARGIN
#define MCPT_Q_MARGIN (1.0 / 1024.0)
#endif
#ifndef MCPT_LEAF_MAX
#define MCPT_LEAF_MAX 2
#endif
#ifndef MCPT_STACK_DEPTH
#define MCPT_STACK_DEPTH 64        // LDS traversal stack entries per lane of the binary-tree kernels (host SAH trees: depth <= 30; device LBVH trees: <= 63, checked)
#endif
#ifndef MCPT_BLOCK
#define MCPT_BLOCK 256             // threads per workgroup = 4 waves of 64
#endif
#ifndef MCPT_TOP_NODES
#define MCPT_TOP_NODES 256         // nodes numbered breadth-first by the builder; the trace kernel serves them from LDS
#endif
#define MCPT_NODE_SENTINEL ((int)0x80000000)
#define HIT_CLASS_SHIFT 28         // pool.hit.x = triangle index (< 2^28, checked at scene build) | HIT_CLASS_* << 28; -1 = miss
#define HIT_TRI_MASK 0x0fffffff
#define HIT_CLASS_DIFFUSE 0u       // Diffuse lobe only                         (BSDF.cpp:105)
#define HIT_CLASS_PHONG 1u         // Blinn-Phong + Diffuse                     (BSDF.cpp:99-105)
#define HIT_CLASS_MIRROR 2u        // specular_reflection + Diffuse (Ns >= 10000, BSDF.cpp:97-98)

struct DevMaterial {               // Material (model.h:32-40) + its Texture header (model.h:21-30)
    float ks[3]; float ns;
    float radiance[3]; uint32_t flags;       // MAT_*
    int32_t tex_off, tex_w, tex_h, pad;      // texel offset into DevScene::texels (float4 per texel)
    float kd[3]; float pad2;                 // the colour of a 1x1 (constant) texture, so shading needs no texel fetch
};
#define MAT_HAS_SPEC   1u   // glm::length(Ks) != 0                (BSDF.cpp:96)
#define MAT_MIRROR     2u   // ... and Ns >= 10000                 (BSDF.cpp:98)
#define MAT_EMISSIVE   4u   // glm::length(radiance) != 0          (Triangle.cpp:75, Render.cpp:146)
#define MAT_EMIT_0     8u   // glm::length(radiance) > 0.0001      (Render.cpp:121)
#define MAT_CONST_KD  32u   // Map_Kd is a constant colour
#define MAT_EMIT_REC  16u   // glm::length(radiance) > 0.01        (Render.cpp:94, light list :41)

struct DevLight {                  // one entry of Render::lights (Render.cpp:41-42)
    int32_t tri;                   // leaf-order triangle index
    float area;                    // Triangle::area() (Triangle.cpp:24-28), fp32
    float radiance[3];
    float n0[3], n1[3], n2[3];     // vertex normals (fp32) for interplote_Normal
    int32_t pad[2];                // 64 B: read as four 16-B records
};
static_assert(sizeof(DevLight) == 64 && sizeof(DevMaterial) == 64, "records are fetched as 4 x float4");

struct DevCamera {                 // Render::cast_Ray's per-frame constants hoisted (Render.cpp:73-75), fp64
    double eye[3], front[3], right[3], up[3];
    double h;                      // 2*tan(fovy/2)
    int32_t width, height;
};

// nodes8: 80 B / node of the 8-wide compressed tree (breadth-first numbering; Ylitie, Karras & Laine 2017 re-laid for gfx950), five 16-B records:
//   r[0] = origin x y z (fp32: the frame's origin LESS 1024 steps -- plane q lies at origin + (1024 + q) scale, see WF8_CHILD in wavefront.hip)
//          | u32: scale_x << 16 | scale_y   (each scale = 2^e as a bfloat16, i.e. the top half of the fp32)
//   r[1] = u32 child_base | u32 tri_base | u32 scale_z << 16 | u32 imask | p0 << 8 | p1 << 16 | (p0 | p1) << 24
//          slot s holds an inner child iff imask bit s: its record = child_base + popcount(imask & below(s));
//          a leaf child of p0[s] + 2 p1[s] triangles otherwise: its first triangle = tri_base + popcount(p0 & below(s)) + 2 popcount(p1 & below(s))
//   r[2] = x planes, four u32, word j = the planes of slots 2j and 2j + 1: bytes { lo[2j], lo[2j+1], hi[2j], hi[2j+1] }  (MCPT_N8_* below; box = origin +
//          (1024 + q) * scale, >= MCPT_Q_MARGIN steps outside the child's box; an empty slot keeps lo = 255, hi = 0).  Low and high planes of a slot pair
//   r[3] = y planes, r[4] = z planes     share a word so that ONE v_perm_b32 with a per-ray selector picks the pair's ENTRY planes (low planes if the ray
//          travels along +axis, high planes otherwise) and one more its EXIT planes -- no selects by direction sign (round 4; r03: 12 v_cndmask per node)
//   Slots are OCTANT slots: bit a of s set = the child lies towards +a of the node's centre, so a ray meets the children roughly front to
//   back in the order of s ^ (its direction octant).
#define MCPT_N8_WORD(slot) ((slot) >> 1)                 // which of an axis record's four words holds the slot's two planes
#define MCPT_N8_LO_SHIFT(slot) (8 * ((slot) & 1))        // bit position of its low-plane byte in that word
#define MCPT_N8_HI_SHIFT(slot) (16 + 8 * ((slot) & 1))   // ... and of its high-plane byte
#define MCPT_N8_EMPTY_WORD 0x0000ffffu                   // two empty slots: lo = 255, hi = 0 (an inverted box)
#define MCPT_TRI_SHADE_F4 8                               // float4s per tri_shade record (128 B); the fp64 plane starts at float4 4
struct DevScene {
    const float4* nodes;
    const float4* nodes8;
    const float4* tri_isect;
    const float4* tri_shade;
    const double* tri_pos64;
    const int32_t* tri_face;
    const DevMaterial* mats;
    const DevLight* lights;
    const double* light_pos64;     // 9 doubles / light: the fp64 corners of lights[i].tri again, contiguous (no lights[i].tri -> tri_pos64 chain)
    const float4* texels;
    DevCamera cam;
    double centre[3];              // every coordinate on the device (vertices, boxes, planes, camera eye) is RELATIVE to this point -- the fp64 centre
                                   // of the scene's bounding box, subtracted on the host in fp64 before anything is rounded to fp32: the reference
                                   // works in fp64 world coordinates, where its absolute ray epsilon t1 = 1e-4 (Render.h:30) is translation-invariant;
                                   // fp32 keeps that property only while |coordinate| * 2^-24 << 1e-4.  World-space values the reference itself
                                   // rounds to fp32 (the light point and hit point of Render::sample) get the centre added back first (sample_light).
    int32_t n_tris, n_lights, n_nodes, n_mats, n_nodes8;
};

struct RenderParams {
    uint32_t spp;                  // samples per pixel in this launch
    uint32_t first_sample;
    uint32_t samples_per_item;     // samples one lane traces back-to-back
    uint32_t chunks;               // ceil(spp / samples_per_item)
    uint32_t tiles_x, tiles_y;     // 8x8 pixel tiles
    uint32_t tile_mod, tile_rem;   // this call owns the tiles t with t % tile_mod == tile_rem (1, 0 = all: the default; interleaved tile
    uint32_t n_owned;              //   sharding across GPUs uses world, rank) -- n_owned of them: tile k of the call = tile_rem + k * tile_mod
    uint32_t max_depth;            // 0 = unbounded
    uint32_t flags;                // MCPT_FLAG_*
    uint32_t integrator;
    uint32_t seed_lo, seed_hi;
    uint32_t atomic_accum;         // 1: several items per pixel -> float atomics; 0: plain read-modify-write
    uint32_t priv_items;           // wavefront: shade block b alone owns priv_items work items (64-item units k * n_blocks + b: no atomics);
    uint32_t shared_base;          //   items from shared_base on are handed out through the shared cursors (load balance at the end of a call)
    uint32_t probe_n;              // mcpt_probe_paths through the wavefront pipeline: item i (< probe_n) = "pixel" i of an n x 1 film,
    const double* probe_o;         //   whose one sample starts from the caller's ray (probe_o/probe_d: 3 doubles each) instead of
    const double* probe_d;         //   cast_Ray.  All three are 0 in a render.
    // x / d for the three run-time divisors of the shade kernel's work-item decode as multiply + shift (filled by launch_wf_shade):
    // {m, s} with x / d == (uint64(x) * m) >> s for every x < 2^30 (Granlund & Montgomery 1994: m = floor(2^(30 + L) / d) + 1, s = 30 + L,
    // L = ceil(log2 d); m < 2^32).  The compiler's general 32-bit division is ~20 VALU instructions, and every shade wave runs all three.
    uint32_t div_owned_m, div_owned_s, div_tiles_x_m, div_tiles_x_s, div_width_m, div_width_s;
};
#define MCPT_FASTDIV_MAX (1u << 30)   // exclusive bound on the dividends (work-item units, tiles, pixels: mcpt_create refuses larger films)

struct DevCounters {               // mirrors the integer part of mcpt_counters
    unsigned long long paths, rays_primary, rays_continuation, rays_shadow, box_tests, tri_tests, shaded_hits,
        texel_fetches, self_shadow_tests, self_shadow_hits, stack_spills;
    unsigned long long debug[4];   // diagnostic builds (-DWF_SCHED_STATS) only
};
