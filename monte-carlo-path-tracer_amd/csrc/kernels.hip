// gfx950 kernels: the render megakernel (Render::render -> cast_Ray -> ray_tracing, Render.cpp:56-175), the dead
// recursive integrator as an iterative kernel (Render.cpp:83-109,177-200), the film tonemap (Scene.cpp:23-33) and
// the function-level probe kernels the parity tests drive through the C ABI.
#include "pt_device.h"
#include "kernels.h"

#ifndef MCPT_MIN_WAVES
#define MCPT_MIN_WAVES 2      // waves per SIMD the register allocator must leave room for (launch_bounds 2nd arg)
#endif

// ======================================================================================================
// Work decomposition.  One work item = (pixel, chunk of `samples_per_item` consecutive samples).  One LANE owns one
// item and traces its samples back-to-back inside ONE flattened loop: the moment a lane's path ends it starts its
// next sample in the same loop iteration ("in-lane path regeneration"), so a 64-wide wave stays full until lanes
// run out of samples, instead of draining to ~3 % occupancy after five bounces as a one-path-per-lane mapping
// would (SURVEY §3.2 path-length histogram).  A wave = one 8x8 pixel tile (coherent primary rays); the
// blockIdx -> tile map keeps each XCD (blockIdx % 8) on a contiguous band of the image so its 4 MiB L2 holds
// the BVH subtrees that band's primary/shadow rays share.
// ======================================================================================================
__device__ __forceinline__ uint32_t xcd_band_block(uint32_t b, uint32_t nblocks) {
    // blocks are dealt round-robin over the 8 XCDs; give XCD x the contiguous logical range [x*per, (x+1)*per)
    const uint32_t per = nblocks >> 3;
    if (per == 0 || b >= (per << 3)) return b;           // tail blocks keep their index
    return (b & 7u) * per + (b >> 3);
}

struct LaneCounters { uint32_t paths, prim, cont, shadow, shaded, texel, self_t, self_h; TravCount tc; };

__device__ __forceinline__ unsigned long long wave_sum(uint32_t v) {
    unsigned long long s = v;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}
__device__ __forceinline__ void flush_counters(const LaneCounters& c, DevCounters* g, bool detail) {
    const unsigned long long paths = wave_sum(c.paths), prim = wave_sum(c.prim), cont = wave_sum(c.cont), sh = wave_sum(c.shadow);
    const unsigned long long st = wave_sum(c.self_t), shh = wave_sum(c.self_h);
    unsigned long long box = 0, tri = 0, shaded = 0, texel = 0;
    if (detail) { box = wave_sum(c.tc.box); tri = wave_sum(c.tc.tri); shaded = wave_sum(c.shaded); texel = wave_sum(c.texel); }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g->paths, paths); atomicAdd(&g->rays_primary, prim); atomicAdd(&g->rays_continuation, cont);
        atomicAdd(&g->rays_shadow, sh); atomicAdd(&g->self_shadow_tests, st); atomicAdd(&g->self_shadow_hits, shh);
        if (detail) { atomicAdd(&g->box_tests, box); atomicAdd(&g->tri_tests, tri); atomicAdd(&g->shaded_hits, shaded); atomicAdd(&g->texel_fetches, texel); }
    }
}

__device__ __forceinline__ f3 scrub_nan(f3 c) {          // Scene::set_Pixel (Scene.cpp:16-18)
    if (c.x != c.x) c.x = 0.f;
    if (c.y != c.y) c.y = 0.f;
    if (c.z != c.z) c.z = 0.f;
    return c;
}

// ---------------------------------------------------------------------------------------------- MIS integrator
// PROBE = true turns the same loop into mcpt_probe_paths: item i traces ONE path from the caller's ray (probe_o/probe_d),
// random numbers keyed (pixel = i, sample = 0), radiance written to probe_out instead of the film.
template <bool COUNT, bool PROBE>
__global__ void __launch_bounds__(MCPT_BLOCK, MCPT_MIN_WAVES) render_mis_kernel(DevScene sc, RenderParams p, float4* __restrict__ accum, DevCounters* gcnt,
                                                                const double* probe_o, const double* probe_d, float* probe_out, uint32_t probe_n) {
    __shared__ int s_stack[MCPT_STACK_DEPTH * MCPT_BLOCK];
    int* stk = s_stack + threadIdx.x;

    const uint32_t lb = PROBE ? blockIdx.x : xcd_band_block(blockIdx.x, gridDim.x);
    const uint32_t wave = lb * (MCPT_BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n_tiles = p.n_owned;
    const uint32_t chunk = PROBE ? 0u : wave / n_tiles, tile = p.tile_rem + (wave - chunk * n_tiles) * p.tile_mod;
    const int px = (int)((tile % p.tiles_x) * 8 + (lane & 7)), py = (int)((tile / p.tiles_x) * 8 + (lane >> 3));
    const bool valid = PROBE ? (wave * 64 + lane < probe_n) : (chunk < p.chunks && px < sc.cam.width && py < sc.cam.height);
    const uint32_t pixel = PROBE ? (wave * 64 + lane) : (uint32_t)(py * sc.cam.width + px);
    uint32_t s_next = p.first_sample + chunk * p.samples_per_item;
    uint32_t s_end = p.first_sample + min(p.spp, (chunk + 1) * p.samples_per_item);
    if (!valid) s_end = s_next;
    const uint32_t n_samples = s_end - s_next;
    const float nl = (float)sc.n_lights;
    const bool correct_t2 = (p.flags & MCPT_FLAG_CORRECT_SHADOW_T2) != 0;

    LaneCounters lc = {};
    f3 sum = mk3(0.f, 0.f, 0.f);

    // per-path state
    bool alive = false;
    uint32_t sample = 0; int bounce = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), beta = mk3(1, 1, 1), L = mk3(0, 0, 0), prev_p = mk3(0, 0, 0);
    d3 o64 = mkd(0, 0, 0);                     // fp64 origin of the ray in flight (reference: Ray::start is a dvec3)
    float prev_pdf = 0.f; bool prev_mirror = false;

    for (;;) {
        if (!alive) {
            if (s_next == s_end) break;
            sample = s_next++;
            if (PROBE) {
                o64 = mkd(probe_o[3 * pixel], probe_o[3 * pixel + 1], probe_o[3 * pixel + 2]); o = to_f3(o64);
                d = mk3((float)probe_d[3 * pixel], (float)probe_d[3 * pixel + 1], (float)probe_d[3 * pixel + 2]);
            } else {
                const Rng4 r = rng_block(pixel, sample, 0u, p.seed_lo, p.seed_hi);
                cast_ray(sc.cam, px, py, r.v[0], r.v[1], o64, o, d);                          // Render.cpp:64
            }
            beta = mk3(1.f, 1.f, 1.f); L = mk3(0.f, 0.f, 0.f); bounce = 0; alive = true;
            lc.paths++; lc.prim++;
        } else {
            lc.cont++;
        }
        // ---- closest hit: bvh->hit (Render.cpp:118 for the camera ray, :144 for BSDF-sampled rays)
        int tri = -1; float ht = 0.f, hu = 0.f, hv = 0.f;
        const bool hit = bvh_traverse<false, COUNT>(sc, o, d, 1e-4f, 3.0e38f, -1, stk, tri, ht, hu, hv, lc.tc);
        if (!hit) { sum = sum + scrub_nan(L); alive = false; continue; }                      // Render.cpp:118-119,144-145

        const d3 p64 = hit_point64(sc, tri, o64, d, hu, hv);
        const f3 p32 = to_f3(p64);
        const HitShade hs = load_hit_shade(sc, tri, hu, hv, d);
        const DevMaterial& mat = sc.mats[hs.mat];
        if (COUNT) lc.shaded++;

        if (bounce > 0) {
            // ---- BSDF-sampled ray reached an emitter: MIS against light sampling (Render.cpp:146-162)
            if ((mat.flags & MAT_EMISSIVE) && hs.front) {
                const f3 rad = mk3(mat.radiance[0], mat.radiance[1], mat.radiance[2]);
                if (prev_mirror) L = L + beta * rad;
                else {
                    const f3 dd = prev_p - p32;
                    const float len = length(dd);
                    const float dist2 = len * len;
                    const float cosine = dot(normalize(dd), hs.n);
                    float light_pdf = 0.f;
                    if (cosine != 0.f) light_pdf = dist2 / cosine / nl / tri_area(sc, tri);
                    L = L + beta * rad * power_heuristic(prev_pdf, light_pdf);
                }
            }
            // ---- Russian roulette of the PREVIOUS vertex (Render.cpp:164-170: `bounces > 3`, q = min(max(beta), .95))
            if (bounce - 1 > 3) {
                const float q = fminf(max3(beta), 0.95f);
                const Rng4 r = rng_block(pixel, sample, 2u + 2u * (uint32_t)(bounce - 1), p.seed_lo, p.seed_hi);
                if (r.v[2] > q) { sum = sum + scrub_nan(L); alive = false; continue; }
                beta = beta / q;
            }
        }
        // ---- loop condition of the depth-bounded variant: `bounces < max_depth` (mcpt.h)
        if (p.max_depth != 0 && (uint32_t)bounce >= p.max_depth) { sum = sum + scrub_nan(L); alive = false; continue; }

        if (bounce == 0 && (mat.flags & MAT_EMIT_0)) L = L + mk3(mat.radiance[0], mat.radiance[1], mat.radiance[2]);   // :121-122

        // ---- shade: BSDF, one light sample, one BSDF sample -- everything except the shadow ray itself
        const f3 kd = tex_color(sc, mat, hs.tu, hs.tv, lc.texel);
        const Bsdf bsdf = make_bsdf(mat, kd, hs.n, -d);
        const Rng4 ra = rng_block(pixel, sample, 1u + 2u * (uint32_t)bounce, p.seed_lo, p.seed_hi);
        const Rng4 rb = rng_block(pixel, sample, 2u + 2u * (uint32_t)bounce, p.seed_lo, p.seed_hi);
        const LightSample ls = sample_light(sc, p64, ra.v[0], ra.v[1], ra.v[2], true);        // Render.cpp:124
        f3 nee = mk3(0.f, 0.f, 0.f);
        bool need_shadow = ls.pdf != 0.f;
        if (need_shadow) {
            lc.self_t++;
            if (ls.self_hit) lc.self_h++;
            if (!correct_t2 && ls.self_hit) need_shadow = false;        // the sampled light triangle blocks its own ray (A-9)
            else {
                f3 fx; float bpdf;
                bsdf_eval(bsdf, ls.wo, fx, bpdf);
                const float cos_theta = fabsf(dot(hs.n, ls.wo));
                const float weight = power_heuristic(ls.pdf / nl, bpdf);
                nee = weight * beta * ls.rad * fx * cos_theta / ls.pdf * nl;                  // Render.cpp:127-129
            }
        }
        const Scatter sc_ = bsdf_sample(bsdf, ra.v[3], rb.v[0], rb.v[1]);                     // Render.cpp:133-134

        // ---- shadow ray: bvh->has_hit (Render.cpp:125).  The sampled triangle is excluded from the fp32 traversal: its
        // verdict was taken in fp64 above (reference-faithful mode) or is "never blocks" (MCPT_FLAG_CORRECT_SHADOW_T2).
        if (need_shadow) {
            int st = -1; float t_, u_, v_;
            lc.shadow++;                                                 // counted only when actually traversed
            const bool blocked = bvh_traverse<true, COUNT>(sc, p32, ls.wo, 1e-4f, ls.t2, ls.tri, stk, st, t_, u_, v_, lc.tc);
            if (!blocked) L = L + nee;
        }
        if (sc_.pdf == 0.f) { sum = sum + scrub_nan(L); alive = false; continue; }            // Render.cpp:135-136
        const float cos_theta = fabsf(dot(hs.n, sc_.wo));
        beta = beta * (sc_.f * cos_theta / sc_.pdf);                                          // Render.cpp:140
        prev_p = p32; prev_pdf = sc_.pdf; prev_mirror = sc_.mirror;
        o = p32; o64 = p64; d = sc_.wo;
        bounce++;
    }

    // ---- film: Scene::set_Pixel's sum + count (Scene.cpp:19-20), once per item instead of once per sample
    if (PROBE) {
        if (valid) { probe_out[3 * pixel] = sum.x; probe_out[3 * pixel + 1] = sum.y; probe_out[3 * pixel + 2] = sum.z; }
    } else if (valid && n_samples) {
        float* a = reinterpret_cast<float*>(accum + pixel);
        if (p.atomic_accum) {
            atomicAdd(a + 0, sum.x); atomicAdd(a + 1, sum.y); atomicAdd(a + 2, sum.z); atomicAdd(a + 3, (float)n_samples);
        } else {
            float4 cur = accum[pixel];
            cur.x += sum.x; cur.y += sum.y; cur.z += sum.z; cur.w += (float)n_samples;
            accum[pixel] = cur;
        }
    }
    flush_counters(lc, gcnt, COUNT);
}

// ---------------------------------------------------------------------------------------------- recursive NEE integrator
// Render::ray_tracing(Ray&,int) (Render.cpp:83-109) unrolled into a loop: the recursion's return value
//   L_k + f_k |n.wo| / pdf_k * (next level)          becomes   total += T_k * L_k,  T_{k+1} = T_k * f_k |n.wo| / pdf_k.
// A level whose BSDF sample fails returns 0 INCLUDING its own L_k (:105-106), so L_k is added only after the sample.
template <bool COUNT>
__global__ void __launch_bounds__(MCPT_BLOCK, MCPT_MIN_WAVES) render_recursive_kernel(DevScene sc, RenderParams p, float4* __restrict__ accum, DevCounters* gcnt) {
    __shared__ int s_stack[MCPT_STACK_DEPTH * MCPT_BLOCK];
    int* stk = s_stack + threadIdx.x;
    const uint32_t lb = xcd_band_block(blockIdx.x, gridDim.x);
    const uint32_t wave = lb * (MCPT_BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n_tiles = p.n_owned;
    const uint32_t chunk = wave / n_tiles, tile = p.tile_rem + (wave - chunk * n_tiles) * p.tile_mod;
    const int px = (int)((tile % p.tiles_x) * 8 + (lane & 7)), py = (int)((tile / p.tiles_x) * 8 + (lane >> 3));
    const bool valid = chunk < p.chunks && px < sc.cam.width && py < sc.cam.height;
    const uint32_t pixel = (uint32_t)(py * sc.cam.width + px);
    uint32_t s_next = p.first_sample + chunk * p.samples_per_item;
    uint32_t s_end = p.first_sample + min(p.spp, (chunk + 1) * p.samples_per_item);
    if (!valid) s_end = s_next;
    const uint32_t n_samples = s_end - s_next;
    const int maxd = p.max_depth ? (int)p.max_depth : 10;                                     // MAX_DEPTH, Render.h:11
    const bool correct_t2 = (p.flags & MCPT_FLAG_CORRECT_SHADOW_T2) != 0;

    LaneCounters lc = {};
    f3 sum = mk3(0.f, 0.f, 0.f);
    bool alive = false; uint32_t sample = 0; int depth = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), T = mk3(1, 1, 1), L = mk3(0, 0, 0);
    d3 o64 = mkd(0, 0, 0);

    for (;;) {
        if (!alive) {
            if (s_next == s_end) break;
            sample = s_next++;
            const Rng4 r = rng_block(pixel, sample, 0u, p.seed_lo, p.seed_hi);
            cast_ray(sc.cam, px, py, r.v[0], r.v[1], o64, o, d);
            T = mk3(1.f, 1.f, 1.f); L = mk3(0.f, 0.f, 0.f); depth = 0; alive = true;
            lc.paths++; lc.prim++;
        } else lc.cont++;
        int tri = -1; float ht = 0.f, hu = 0.f, hv = 0.f;
        const bool hit = bvh_traverse<false, COUNT>(sc, o, d, 1e-4f, 3.0e38f, -1, stk, tri, ht, hu, hv, lc.tc);
        if (!hit) { sum = sum + scrub_nan(L); alive = false; continue; }                      // :90-93
        const d3 p64 = hit_point64(sc, tri, o64, d, hu, hv);
        const f3 p32 = to_f3(p64);
        const HitShade hs = load_hit_shade(sc, tri, hu, hv, d);
        const DevMaterial& mat = sc.mats[hs.mat];
        if (COUNT) lc.shaded++;
        if (mat.flags & MAT_EMIT_REC) {                                                        // :94-97
            L = L + T * mk3(mat.radiance[0], mat.radiance[1], mat.radiance[2]);
            sum = sum + scrub_nan(L); alive = false; continue;
        }
        const f3 kd = tex_color(sc, mat, hs.tu, hs.tv, lc.texel);
        const Rng4 ra = rng_block(pixel, sample, 1u + 2u * (uint32_t)depth, p.seed_lo, p.seed_hi);
        const Rng4 rb = rng_block(pixel, sample, 2u + 2u * (uint32_t)depth, p.seed_lo, p.seed_hi);
        // sample_light (Render.cpp:177-200): no cos guard, radiance * Kd * |n.l| / pdf / 2
        const LightSample ls = sample_light(sc, p64, ra.v[0], ra.v[1], ra.v[2], false);
        f3 Lk = mk3(0.f, 0.f, 0.f);
        lc.self_t++;
        if (ls.self_hit) lc.self_h++;
        if (correct_t2 || !ls.self_hit) {
            int st = -1; float t_, u_, v_;
            lc.shadow++;
            const bool blocked = bvh_traverse<true, COUNT>(sc, p32, ls.wo, 1e-4f, ls.t2, ls.tri, stk, st, t_, u_, v_, lc.tc);
            if (!blocked) Lk = ls.rad * kd * fabsf(dot(hs.n, ls.wo)) / ls.pdf / 2.0f;
        }
        const Bsdf bsdf = make_bsdf(mat, kd, hs.n, -d);
        const Scatter s = bsdf_sample(bsdf, ra.v[3], rb.v[0], rb.v[1]);
        if (length(s.wo) < 0.00001f) { sum = sum + scrub_nan(L); alive = false; continue; }    // :105-106 (drops L_k too)
        L = L + T * Lk;
        T = T * (s.f * fabsf(dot(hs.n, s.wo))) / s.pdf;                                       // :108
        o = p32; o64 = p64; d = s.wo; depth++;
        if (depth > maxd) { sum = sum + scrub_nan(L); alive = false; continue; }              // :85-87
    }
    if (valid && n_samples) {
        float* a = reinterpret_cast<float*>(accum + pixel);
        if (p.atomic_accum) { atomicAdd(a + 0, sum.x); atomicAdd(a + 1, sum.y); atomicAdd(a + 2, sum.z); atomicAdd(a + 3, (float)n_samples); }
        else { float4 cur = accum[pixel]; cur.x += sum.x; cur.y += sum.y; cur.z += sum.z; cur.w += (float)n_samples; accum[pixel] = cur; }
    }
    flush_counters(lc, gcnt, COUNT);
}

// ---------------------------------------------------------------------------------------------- film
// Scene::getPixelsColor (Scene.cpp:23-33) (+ the vertical flip of Scene::save_image, Scene.cpp:40-46)
__global__ void tonemap_kernel(const float4* __restrict__ accum, uint8_t* __restrict__ rgb, int w, int h, int flip) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    const float4 a = accum[i];
    const int y = i / w, x = i - y * w;
    const int o = flip ? ((h - 1 - y) * w + x) : i;
    const float c[3] = {a.x / a.w, a.y / a.w, a.z / a.w};
    for (int k = 0; k < 3; k++) {
        float m = fminf(fmaxf(c[k], 0.f), 1.f);      // glm::clamp = min(max(x, lo), hi); NaN -> 0 through fmaxf
        m = powf(m, 0.5f);
        rgb[3 * o + k] = (uint8_t)(m * 255.99f);
    }
}

// ---------------------------------------------------------------------------------------------- probes
__global__ void __launch_bounds__(MCPT_BLOCK) probe_trace_kernel(DevScene sc, uint32_t n, const double* origin, const double* dir, const double* t1,
                                                                 const double* t2, int any_hit, float* out_t, int* out_tri, float* out_u, float* out_v) {
    __shared__ int s_stack[MCPT_STACK_DEPTH * MCPT_BLOCK];
    int* stk = s_stack + threadIdx.x;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 o = mk3((float)origin[3 * i], (float)origin[3 * i + 1], (float)origin[3 * i + 2]);
    const f3 d = mk3((float)dir[3 * i], (float)dir[3 * i + 1], (float)dir[3 * i + 2]);
    const float tmin = (float)t1[i];
    const float tmax = t2[i] > 3.0e38 ? 3.0e38f : (float)t2[i];
    int tri = -1; float t = 0.f, u = 0.f, v = 0.f; TravCount tc = {0, 0};
    bool hit;
    if (any_hit) hit = bvh_traverse<true, false>(sc, o, d, tmin, tmax, -1, stk, tri, t, u, v, tc);
    else hit = bvh_traverse<false, false>(sc, o, d, tmin, tmax, -1, stk, tri, t, u, v, tc);
    out_t[i] = hit ? t : 0.f;
    out_tri[i] = any_hit ? (hit ? 1 : 0) : (hit ? sc.tri_face[tri] : -1);
    out_u[i] = hit ? u : 0.f; out_v[i] = hit ? v : 0.f;
}

__global__ void probe_cast_ray_kernel(DevScene sc, uint32_t n, const int* xy, const float* xi, float* out6) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 o, d; d3 o64;
    cast_ray(sc.cam, xy[2 * i], xy[2 * i + 1], xi[2 * i], xi[2 * i + 1], o64, o, d);
    o = to_f3(o64 + mkd(sc.centre[0], sc.centre[1], sc.centre[2]));      // the caller's world coordinates (the device works relative to DevScene::centre)
    out6[6 * i + 0] = o.x; out6[6 * i + 1] = o.y; out6[6 * i + 2] = o.z; out6[6 * i + 3] = d.x; out6[6 * i + 4] = d.y; out6[6 * i + 5] = d.z;
}

// Triangle::hit's shading record (Triangle.cpp:68-76) for a given hit: interpolated + normalised vertex normal, uv, front flag
__global__ void probe_hit_shade_kernel(DevScene sc, uint32_t n, const int* tri, const float* u, const float* v, const double* dir, float* out6) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const HitShade h = load_hit_shade(sc, tri[i], u[i], v[i], mk3((float)dir[3 * i], (float)dir[3 * i + 1], (float)dir[3 * i + 2]));
    out6[6 * i + 0] = h.n.x; out6[6 * i + 1] = h.n.y; out6[6 * i + 2] = h.n.z; out6[6 * i + 3] = h.tu; out6[6 * i + 4] = h.tv; out6[6 * i + 5] = h.front ? 1.f : 0.f;
}

__global__ void probe_bsdf_kernel(uint32_t n, const float* normal, const float* wi, const float* kd, const float* ks, const float* ns,
                                  const float* wo, const float* xi, float* out12) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    DevMaterial m = {};
    m.ks[0] = ks[3 * i]; m.ks[1] = ks[3 * i + 1]; m.ks[2] = ks[3 * i + 2]; m.ns = ns[i];
    // flags as the host derives them (scene_prep.cpp): length(Ks) != 0, Ns >= 10000
    if (m.ks[0] != 0.f || m.ks[1] != 0.f || m.ks[2] != 0.f) m.flags |= MAT_HAS_SPEC | (m.ns >= 10000.f ? MAT_MIRROR : 0u);
    const f3 nrm = mk3(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
    const Bsdf b = make_bsdf(m, mk3(kd[3 * i], kd[3 * i + 1], kd[3 * i + 2]), nrm, mk3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]));
    f3 fx; float pdf;
    bsdf_eval(b, mk3(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]), fx, pdf);
    const Scatter s = bsdf_sample(b, xi[3 * i], xi[3 * i + 1], xi[3 * i + 2]);
    float* o = out12 + 12 * i;
    o[0] = fx.x; o[1] = fx.y; o[2] = fx.z; o[3] = pdf; o[4] = s.wo.x; o[5] = s.wo.y; o[6] = s.wo.z;
    o[7] = s.f.x; o[8] = s.f.y; o[9] = s.f.z; o[10] = s.pdf; o[11] = s.mirror ? 1.f : 0.f;
}

__global__ void probe_sample_light_kernel(DevScene sc, uint32_t n, const double* point, const float* xi, float* out10) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const LightSample ls = sample_light(sc, mkd(point[3 * i], point[3 * i + 1], point[3 * i + 2]), xi[3 * i], xi[3 * i + 1], xi[3 * i + 2], true);
    float* o = out10 + 10 * i;
    o[0] = ls.wo.x; o[1] = ls.wo.y; o[2] = ls.wo.z; o[3] = ls.rad.x; o[4] = ls.rad.y; o[5] = ls.rad.z; o[6] = ls.pdf; o[7] = ls.t2;
    o[8] = (float)sc.tri_face[ls.tri]; o[9] = ls.self_hit ? 1.f : 0.f;
}

__global__ void probe_texture_kernel(DevScene sc, int material, uint32_t n, const float* uv, float* out3) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t fetches = 0;
    const f3 c = tex_color(sc, sc.mats[material], uv[2 * i], uv[2 * i + 1], fetches);
    out3[3 * i] = c.x; out3[3 * i + 1] = c.y; out3[3 * i + 2] = c.z;
}

__global__ void probe_rng_kernel(uint32_t n, const uint32_t* key3, uint32_t seed_lo, uint32_t seed_hi, float* out4) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Rng4 r = rng_block(key3[3 * i], key3[3 * i + 1], key3[3 * i + 2], seed_lo, seed_hi);
    out4[4 * i] = r.v[0]; out4[4 * i + 1] = r.v[1]; out4[4 * i + 2] = r.v[2]; out4[4 * i + 3] = r.v[3];
}

// ---------------------------------------------------------------------------------------------- launchers
hipError_t launch_render(const DevScene& sc, const RenderParams& p, float4* accum, DevCounters* cnt, hipStream_t stream) {
    const uint64_t waves = (uint64_t)p.n_owned * p.chunks;
    const uint32_t blocks = (uint32_t)((waves + (MCPT_BLOCK / 64) - 1) / (MCPT_BLOCK / 64));
    if (blocks == 0) return hipSuccess;
    const bool count = (p.flags & MCPT_FLAG_COUNT_TRAVERSAL) != 0;
    if (p.integrator == MCPT_INTEGRATOR_RECURSIVE_NEE) {
        if (count) hipLaunchKernelGGL(render_recursive_kernel<true>, dim3(blocks), dim3(MCPT_BLOCK), 0, stream, sc, p, accum, cnt);
        else hipLaunchKernelGGL(render_recursive_kernel<false>, dim3(blocks), dim3(MCPT_BLOCK), 0, stream, sc, p, accum, cnt);
    } else {
        if (count) hipLaunchKernelGGL((render_mis_kernel<true, false>), dim3(blocks), dim3(MCPT_BLOCK), 0, stream, sc, p, accum, cnt, nullptr, nullptr, nullptr, 0u);
        else hipLaunchKernelGGL((render_mis_kernel<false, false>), dim3(blocks), dim3(MCPT_BLOCK), 0, stream, sc, p, accum, cnt, nullptr, nullptr, nullptr, 0u);
    }
    return hipGetLastError();
}
hipError_t launch_probe_paths(const DevScene& sc, const RenderParams& p, uint32_t n, const double* o, const double* d, float* out3, DevCounters* cnt, hipStream_t stream) {
    hipLaunchKernelGGL((render_mis_kernel<false, true>), dim3((n + MCPT_BLOCK - 1) / MCPT_BLOCK), dim3(MCPT_BLOCK), 0, stream, sc, p, nullptr, cnt, o, d, out3, n);
    return hipGetLastError();
}
hipError_t launch_tonemap(const float4* accum, uint8_t* rgb, int w, int h, int flip, hipStream_t stream) {
    const int n = w * h;
    hipLaunchKernelGGL(tonemap_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, accum, rgb, w, h, flip);
    return hipGetLastError();
}
hipError_t launch_probe_trace(const DevScene& sc, uint32_t n, const double* o, const double* d, const double* t1, const double* t2, int any_hit,
                              float* out_t, int* out_tri, float* out_u, float* out_v, hipStream_t stream) {
    hipLaunchKernelGGL(probe_trace_kernel, dim3((n + MCPT_BLOCK - 1) / MCPT_BLOCK), dim3(MCPT_BLOCK), 0, stream, sc, n, o, d, t1, t2, any_hit, out_t, out_tri, out_u, out_v);
    return hipGetLastError();
}
hipError_t launch_probe_cast_ray(const DevScene& sc, uint32_t n, const int* xy, const float* xi, float* out6, hipStream_t stream) {
    hipLaunchKernelGGL(probe_cast_ray_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, sc, n, xy, xi, out6);
    return hipGetLastError();
}
hipError_t launch_probe_hit_shade(const DevScene& sc, uint32_t n, const int* tri, const float* u, const float* v, const double* dir, float* out6, hipStream_t stream) {
    hipLaunchKernelGGL(probe_hit_shade_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, sc, n, tri, u, v, dir, out6);
    return hipGetLastError();
}
hipError_t launch_probe_bsdf(uint32_t n, const float* normal, const float* wi, const float* kd, const float* ks, const float* ns, const float* wo,
                             const float* xi, float* out12, hipStream_t stream) {
    hipLaunchKernelGGL(probe_bsdf_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, normal, wi, kd, ks, ns, wo, xi, out12);
    return hipGetLastError();
}
hipError_t launch_probe_sample_light(const DevScene& sc, uint32_t n, const double* point, const float* xi, float* out10, hipStream_t stream) {
    hipLaunchKernelGGL(probe_sample_light_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, sc, n, point, xi, out10);
    return hipGetLastError();
}
hipError_t launch_probe_texture(const DevScene& sc, int material, uint32_t n, const float* uv, float* out3, hipStream_t stream) {
    hipLaunchKernelGGL(probe_texture_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, sc, material, n, uv, out3);
    return hipGetLastError();
}
hipError_t launch_probe_rng(uint32_t n, const uint32_t* key3, uint32_t seed_lo, uint32_t seed_hi, float* out4, hipStream_t stream) {
    hipLaunchKernelGGL(probe_rng_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, key3, seed_lo, seed_hi, out4);
    return hipGetLastError();
}
