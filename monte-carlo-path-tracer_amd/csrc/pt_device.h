// gfx950 device functions of the path-tracing hot path.  Each block cites the reference function whose
// behaviour it reproduces (paths relative to /root/reference/src).  Arithmetic: fp32 everywhere the reference
// uses vec3 (shading) AND for ray/box/triangle tests (the reference uses dvec3 there); fp64 only for the three
// quantities whose low-order bits decide a branch in the reference -- the camera ray (Render.cpp:71-80), the hit
// point and light point (Triangle.cpp:35-38) and the sampled light's own shadow test (SURVEY A-9).
#pragma once
#include <hip/hip_runtime.h>
#include "device_scene.h"
#include "../../include/mcpt.h"

#define DEV __device__ __forceinline__

struct f3 { float x, y, z; };
struct d3 { double x, y, z; };

DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
DEV f3 operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
// Division / normalisation of CONTINUOUS shading quantities: v_rcp_f32 / v_rsq_f32 (1 ulp) instead of the ~10-instruction IEEE
// sequence hipcc emits for `/` -- the reference divides, but a 1-ulp difference here moves a pixel by ~1e-7 relative.  The few
// operations whose last bit decides a branch of the reference (sample_light) use IEEE ops explicitly, see there.
DEV float rcp(float s) { return __builtin_amdgcn_rcpf(s); }
DEV f3 operator/(f3 a, float s) { const float r = rcp(s); return mk3(a.x * r, a.y * r, a.z * r); }
DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
DEV float length(f3 a) { return __builtin_amdgcn_sqrtf(dot(a, a)); }
DEV f3 normalize(f3 a) { return a * __builtin_amdgcn_rsqf(dot(a, a)); }   // glm: v * inversesqrt(dot(v,v))
DEV float max3(f3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }

DEV d3 mkd(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV d3 operator+(d3 a, d3 b) { return mkd(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV d3 operator-(d3 a, d3 b) { return mkd(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV d3 operator*(double s, d3 a) { return mkd(a.x * s, a.y * s, a.z * s); }
DEV double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV d3 cross(d3 a, d3 b) { return mkd(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
DEV f3 to_f3(d3 a) { return mk3((float)a.x, (float)a.y, (float)a.z); }
DEV d3 to_d3(f3 a) { return mkd((double)a.x, (double)a.y, (double)a.z); }
DEV d3 ld_d3(const double* p) { return mkd(p[0], p[1], p[2]); }
// fp64 reciprocal / reciprocal square root: hardware seed (v_rcp_f64 / v_rsq_f64) + two Newton steps = full double accuracy to
// ~1 ulp in ~8 instructions instead of the ~30-instruction IEEE divide / sqrt sequences.  Every consumer compares against
// quantities that differ at the 1e-8 level (SURVEY A-9) or rounds to fp32, so the last fp64 ulp is irrelevant.
DEV double rcp64(double x) { double r = __builtin_amdgcn_rcp(x); r = fma(fma(-x, r, 1.0), r, r); r = fma(fma(-x, r, 1.0), r, r); return r; }
DEV double rsq64(double x) { double r = __builtin_amdgcn_rsq(x); r = fma(fma(-0.5 * x * r, r, 0.5), r, r); r = fma(fma(-0.5 * x * r, r, 0.5), r, r); return r; }

#define PT_PI 3.1415926f                       // utils.h:20 -- the reference's truncated pi, used in all shading
#define PT_INV_PI (1.0f / 3.1415926f)
#define PT_INV_2PI (1.0f / (2.0f * 3.1415926f))

// ---------------------------------------------------------------------------------------------- RNG
// Counter-based replacement for the reference's global mt19937 (utils.h:23-28): pcg4d (Jarzynski & Olano 2020)
// of (pixel, sample, block, seed).  Block layout per path: DESIGN.md §RNG (0 = camera; 1+2b / 2+2b = bounce b).
struct Rng4 { float v[4]; };
DEV Rng4 rng_block(uint32_t pixel, uint32_t sample, uint32_t block, uint32_t seed_lo, uint32_t seed_hi) {
    uint32_t x = pixel, y = sample, z = block ^ (seed_hi * 0x9E3779B9u), w = seed_lo;
    x = x * 1664525u + 1013904223u; y = y * 1664525u + 1013904223u; z = z * 1664525u + 1013904223u; w = w * 1664525u + 1013904223u;
    x += y * w; y += z * x; z += x * y; w += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
    x += y * w; y += z * x; z += x * y; w += y * z;
    Rng4 r;
    const float s = 1.0f / 16777216.0f;
    r.v[0] = (float)(x >> 8) * s; r.v[1] = (float)(y >> 8) * s; r.v[2] = (float)(z >> 8) * s; r.v[3] = (float)(w >> 8) * s;
    return r;
}

// ---------------------------------------------------------------------------------------------- camera
// Render::cast_Ray (Render.cpp:71-80): (x + xi)/w in float, the rest in double; tan/normalize/cross hoisted to
// DevCamera by the host.  `up` is used raw (SURVEY A-17).
DEV void cast_ray(const DevCamera& c, int x, int y, float xi_x, float xi_y, d3& o64, f3& o, f3& d) {
    double u = ((double)(((float)x + xi_x) / (float)c.width) - 0.5) * c.h * (double)c.width / (double)c.height;
    double v = ((double)(((float)y + xi_y) / (float)c.height) - 0.5) * c.h;
    double dx = c.front[0] + u * c.right[0] + v * c.up[0];
    double dy = c.front[1] + u * c.right[1] + v * c.up[1];
    double dz = c.front[2] + u * c.right[2] + v * c.up[2];
    double inv = rsq64(dx * dx + dy * dy + dz * dz);
    d = mk3((float)(dx * inv), (float)(dy * inv), (float)(dz * inv));
    o64 = mkd(c.eye[0], c.eye[1], c.eye[2]);
    o = to_f3(o64);
}

// ---------------------------------------------------------------------------------------------- traversal
// Replaces BVH_node::hit / has_hit (BVH.cpp:95-136: unordered recursion over heap nodes), AABB::Intersection
// (AABB.cpp:25-36: six fp64 divides per node) and Triangle::hit / isIntersect (Triangle.cpp:48-106) by an
// iterative, near-child-first loop over the flat arrays of device_scene.h, with the per-lane stack in LDS
// (stk[level * MCPT_BLOCK]: lane l always hits bank l % 32 -> conflict-free at any per-lane depth).
// Acceptance rules are the reference's: closest  t1 <= t < t2, |a| >= 1e-5, u,v,1-u-v >= 0;
//                                       any-hit  t1 <= t <= t2, |det| >= 1e-6, 0<=u<=1, v>=0, u+v<=1.
struct TravCount { uint32_t box, tri; };

// Moller-Trumbore against one 48-B {v0, e1, e2} record -- the ONE triangle test of this library, shared by the binary-tree
// traversal below and the wavefront trace kernel (wf_trace8_kernel, wavefront.hip).  Triangle::hit / isIntersect compute the same
// quantities in fp64 (Triangle.cpp:48-66, :83-104); the acceptance rules are the reference's, see tri_accept_*.
struct TriTest { float a, t, u, v; };
DEV TriTest tri_test(const float4 v0, const float4 e1, const float4 e2, const f3 o, const f3 d) {
    // Every product-sum is spelled as an explicit fma chain (nothing is left for -ffp-contract to decide): each inlined copy (the two
    // triangles of a leaf pair, the binary-tree kernels) then rounds identically, so a triangle's (t, u, v) do not depend on which tree
    // or which slot of a pair it was reached through -- deterministic renders agree bit for bit across differently built trees.
    const float hx = fmaf(d.y, e2.z, -(e2.y * d.z)), hy = fmaf(d.z, e2.x, -(e2.z * d.x)), hz = fmaf(d.x, e2.y, -(e2.x * d.y));   // h = d x e2
    TriTest r;
    r.a = fmaf(e1.z, hz, fmaf(e1.y, hy, e1.x * hx));
    const float sx = o.x - v0.x, sy = o.y - v0.y, sz = o.z - v0.z;
    const float qx = fmaf(sy, e1.z, -(e1.y * sz)), qy = fmaf(sz, e1.x, -(e1.z * sx)), qz = fmaf(sx, e1.y, -(e1.x * sy));         // q = s x e1
    const float inv_a = __builtin_amdgcn_rcpf(r.a);
    r.u = fmaf(sz, hz, fmaf(sy, hy, sx * hx)) * inv_a;
    r.v = fmaf(d.z, qz, fmaf(d.y, qy, d.x * qx)) * inv_a;
    r.t = fmaf(e2.z, qz, fmaf(e2.y, qy, e2.x * qx)) * inv_a;
    return r;
}
// Triangle::isIntersect (Triangle.cpp:85-104): |det| >= 1e-6, 0 <= u <= 1, v >= 0, u + v <= 1, t1 <= t <= t2 (inclusive)
DEV bool tri_accept_any(const TriTest& r, float tmin, float tmax) {
    return fabsf(r.a) >= 1e-6f && r.u >= 0.0f && r.u <= 1.0f && r.v >= 0.0f && r.u + r.v <= 1.0f && r.t >= tmin && r.t <= tmax;
}
// Triangle::hit (Triangle.cpp:54,66): |a| >= 1e-5, t1 <= t < t2, u, v, 1-u-v >= 0
DEV bool tri_accept_closest(const TriTest& r, float tmin, float tmax) {
    return fabsf(r.a) >= 0.00001f && r.t >= tmin && r.t < tmax && r.u >= 0.0f && r.v >= 0.0f && (1.0f - r.u - r.v) >= 0.0f;
}
// (Exact ties: this traversal lets the first triangle it tests win.  The production kernel has a defined winner -- lowest tie rank, see the leaf
// block of wf_trace8_kernel -- and MCPT_FLAG_REFERENCE_TIE_ORDER is refused for the kernels that traverse with this function.)
template <bool ANY, bool COUNT>
DEV bool bvh_traverse(const DevScene& sc, f3 o, f3 d, float tmin, float tmax, int skip_tri, int* stk,
                      int& hit_tri, float& hit_t, float& hit_u, float& hit_v, TravCount& tc) {
    const float tiny = 1e-30f;
    const float idx = 1.0f / (fabsf(d.x) > tiny ? d.x : copysignf(tiny, d.x));
    const float idy = 1.0f / (fabsf(d.y) > tiny ? d.y : copysignf(tiny, d.y));
    const float idz = 1.0f / (fabsf(d.z) > tiny ? d.z : copysignf(tiny, d.z));
    const float oodx = o.x * idx, oody = o.y * idy, oodz = o.z * idz;
    bool found = false;
    int sp = 1;
    stk[0] = MCPT_NODE_SENTINEL;
    int node = 0;
    while (node != MCPT_NODE_SENTINEL) {
        while (node >= 0) {                                   // inner nodes: one 64-B record = two child boxes
            const float4* n = sc.nodes + 4 * (size_t)node;
            const float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
            const float c0x0 = fmaf(n0.x, idx, -oodx), c0x1 = fmaf(n0.y, idx, -oodx);
            const float c0y0 = fmaf(n0.z, idy, -oody), c0y1 = fmaf(n0.w, idy, -oody);
            const float c0z0 = fmaf(n2.x, idz, -oodz), c0z1 = fmaf(n2.y, idz, -oodz);
            const float c1x0 = fmaf(n1.x, idx, -oodx), c1x1 = fmaf(n1.y, idx, -oodx);
            const float c1y0 = fmaf(n1.z, idy, -oody), c1y1 = fmaf(n1.w, idy, -oody);
            const float c1z0 = fmaf(n2.z, idz, -oodz), c1z1 = fmaf(n2.w, idz, -oodz);
            const float c0n = fmaxf(fmaxf(fminf(c0x0, c0x1), fminf(c0y0, c0y1)), fmaxf(fminf(c0z0, c0z1), tmin));
            const float c0f = fminf(fminf(fmaxf(c0x0, c0x1), fmaxf(c0y0, c0y1)), fminf(fmaxf(c0z0, c0z1), tmax));
            const float c1n = fmaxf(fmaxf(fminf(c1x0, c1x1), fminf(c1y0, c1y1)), fmaxf(fminf(c1z0, c1z1), tmin));
            const float c1f = fminf(fminf(fmaxf(c1x0, c1x1), fmaxf(c1y0, c1y1)), fminf(fmaxf(c1z0, c1z1), tmax));
            const bool h0 = c0n <= c0f, h1 = c1n <= c1f;
            if (COUNT) tc.box += 2;
            const int ch0 = __float_as_int(n3.x), ch1 = __float_as_int(n3.y);
            if (h0 && h1) {
                const bool swp = c1n < c0n;
                node = swp ? ch1 : ch0;
                stk[sp * MCPT_BLOCK] = swp ? ch0 : ch1;
                sp++;
            } else if (h0) node = ch0;
            else if (h1) node = ch1;
            else { sp--; node = stk[sp * MCPT_BLOCK]; }
        }
        while (node < 0 && node != MCPT_NODE_SENTINEL) {     // leaves
            const uint32_t leaf = (uint32_t)~node;
            const uint32_t first = leaf >> 3, cnt = leaf & 7u;
            for (uint32_t i = 0; i < cnt; i++) {
                const int ti = (int)(first + i);
                if (ti == skip_tri) continue;
                const float4* T = sc.tri_isect + 3 * (size_t)ti;
                const float4 v0 = T[0], e1 = T[1], e2 = T[2];
                if (COUNT) tc.tri++;
                const TriTest r = tri_test(v0, e1, e2, o, d);
                if (ANY) {
                    if (tri_accept_any(r, tmin, tmax)) { hit_tri = ti; hit_t = r.t; hit_u = r.u; hit_v = r.v; return true; }
                } else {
                    if (tri_accept_closest(r, tmin, tmax)) { tmax = r.t; hit_tri = ti; hit_t = r.t; hit_u = r.u; hit_v = r.v; found = true; }
                }
            }
            sp--; node = stk[sp * MCPT_BLOCK];
        }
    }
    return found;
}

// ---------------------------------------------------------------------------------------------- shading data
struct HitShade { f3 n; float tu, tv; int mat; bool front; };

// Triangle::hit's record (Triangle.cpp:68-76): interpolated+normalised vertex normal (the shading normal, A-2),
// uv, material.  One 64-B fetch (the first half of the triangle's 128-B shading record; its fp64 plane is the second half).
DEV HitShade load_hit_shade(const DevScene& sc, int tri, float u, float v, f3 d) {
    const float4* S = sc.tri_shade + MCPT_TRI_SHADE_F4 * (size_t)tri;
    const float4 a = S[0], b = S[1], c = S[2], e = S[3];
    const float w = 1.0f - u - v;
    HitShade h;
    h.n = normalize(mk3(w * a.x + u * b.x + v * c.x, w * a.y + u * b.y + v * c.y, w * a.z + u * b.z + v * c.z));
    h.tu = w * a.w + u * c.w + v * e.y;      // uv0.x, uv1.x, uv2.x
    h.tv = w * b.w + u * e.x + v * e.z;      // uv0.y, uv1.y, uv2.y
    h.mat = __float_as_int(e.w);
    h.front = dot(h.n, d) < 0.0f;
    return h;
}
// The hit record's fp64 part, exactly as Triangle::hit computes it (Triangle.cpp:48-68): the fp32 traversal only SELECTS the
// triangle; u, v and the hit point are then recomputed by one fp64 Moller-Trumbore from the fp64 ray origin (the previous
// hit point, or the eye) and the fp32-valued direction.  This keeps the low-order bits of the hit point statistically
// identical to the reference's -- they decide the light self-occlusion of SURVEY A-9 (with fp32 barycentrics the point is
// exactly fp32-representable on axis-aligned walls and the image comes out 1.2 % darker than the reference).
DEV d3 hit_point64(const DevScene& sc, int tri, d3 o64, f3 dir, float& u_out, float& v_out) {
    const double* P = sc.tri_pos64 + 9 * (size_t)tri;
    const d3 v0 = ld_d3(P), v1 = ld_d3(P + 3), v2 = ld_d3(P + 6);
    const d3 e1 = v1 - v0, e2 = v2 - v0, dd = to_d3(dir);
    const d3 h = cross(dd, e2);
    const double inv_a = rcp64(dot(e1, h));
    const d3 s = o64 - v0;
    const double u = dot(s, h) * inv_a;
    const d3 q = cross(s, e1);
    const double v = dot(dd, q) * inv_a;
    u_out = (float)u; v_out = (float)v;
    return (1.0 - u - v) * v0 + u * v1 + v * v2;
}
// Triangle::area (Triangle.cpp:24-28) from the fp32 edges already stored for intersection
DEV float tri_area(const DevScene& sc, int tri) {
    const float4* T = sc.tri_isect + 3 * (size_t)tri;
    const float4 e1 = T[1], e2 = T[2];
    return 0.5f * length(cross(mk3(e1.x, e1.y, e1.z), mk3(e2.x, e2.y, e2.z)));
}
// Texture::get_color (model.cpp:30-41) + clamp01 (utils.h:30-34): nearest texel, fract + 0.999 cap, no v flip
DEV f3 tex_color(const DevScene& sc, const DevMaterial& m, float tu, float tv, uint32_t& texel_fetches) {
    if (m.flags & MAT_CONST_KD) return mk3(m.kd[0], m.kd[1], m.kd[2]);      // Texture(Color3f): image_color.size() == 1 (model.cpp:32-35)
    int idx = m.tex_off;
    if (m.tex_w * m.tex_h != 1) {
        float fu = tu - floorf(tu), fv = tv - floorf(tv);
        double cu = fu > 0.999f ? 0.999 : (fu < 0.0f ? 0.0 : (double)fu);
        double cv = fv > 0.999f ? 0.999 : (fv < 0.0f ? 0.0 : (double)fv);
        int x = (int)(cu * m.tex_w), y = (int)(cv * m.tex_h);
        idx += y * m.tex_w + x;
        texel_fetches++;
    }
    const float4 t = sc.texels[idx];
    return mk3(t.x, t.y, t.z);
}

// ---------------------------------------------------------------------------------------------- BSDF
// BSDF::BSDF (BSDF.cpp:87-110): ONB (BSDF.h:14-18), lobes [Phong | mirror] + Diffuse, luminance sampling weights
// taken BEFORE the energy rescale (:108-109,165-202).  No heap, no virtual calls: `kind` selects the lobe pair.
#define BSDF_DIFFUSE 0
#define BSDF_PHONG 1
#define BSDF_MIRROR 2
struct Bsdf {
    f3 u, v, w;        // local frame, w = shading normal
    f3 m_wo;           // local direction towards the previous vertex (the lobes' m_wo)
    f3 kd, ks;         // lobe `reflect` after energy_conservation
    float ns, w_spec, w_diff;
    int kind;
};
DEV f3 to_local(const Bsdf& b, f3 t) { return mk3(dot(t, b.u), dot(t, b.v), dot(t, b.w)); }
DEV f3 to_world(const Bsdf& b, f3 a) { return a.x * b.u + a.y * b.v + a.z * b.w; }

DEV Bsdf make_bsdf(const DevMaterial& m, f3 kd_tex, f3 n, f3 wi_world) {
    Bsdf b;
    b.w = n;
    const f3 a = (fabsf(n.x) > 0.9f) ? mk3(0.f, 1.f, 0.f) : mk3(1.f, 0.f, 0.f);
    b.v = normalize(cross(b.w, a));
    b.u = cross(b.w, b.v);
    b.m_wo = to_local(b, wi_world);
    b.kd = kd_tex; b.ks = mk3(0.f, 0.f, 0.f); b.ns = m.ns;
    b.kind = BSDF_DIFFUSE;
    if (m.flags & MAT_HAS_SPEC) {
        b.kind = (m.flags & MAT_MIRROR) ? BSDF_MIRROR : BSDF_PHONG;
        b.ks = (m.flags & MAT_MIRROR) ? mk3(1.f, 1.f, 1.f) : mk3(m.ks[0], m.ks[1], m.ks[2]);
    }
    const float lum_d = b.kd.x * 0.212671f + b.kd.y * 0.715160f + b.kd.z * 0.072169f;
    const float lum_s = b.ks.x * 0.212671f + b.ks.y * 0.715160f + b.ks.z * 0.072169f;
    const float sum = (b.kind == BSDF_DIFFUSE) ? lum_d : (lum_s + lum_d);
    b.w_spec = 0.f; b.w_diff = 0.f;                    // sum == 0: reference leaves weights uninitialised (A-12) -> path ends
    if (sum != 0.f) { const float inv = rcp(sum); b.w_spec = lum_s * inv; b.w_diff = lum_d * inv; }
    const f3 tot = b.kd + b.ks;                         // energy_conservation (BSDF.cpp:188-202)
    const float maxc = max3(tot);
    if (!(maxc < 1.0f)) { b.kd = b.kd / maxc; b.ks = b.ks / maxc; }
    return b;
}
// pow for x >= 0 as exp2(y * log2 x) on the transcendental unit (v_log_f32 + v_exp_f32, ~1 ulp each): the OCML powf behind both
// `powf` and `__powf` is a ~180-instruction extended-precision expansion, and shade inlines seven of them that run for the few
// lanes of a wave sitting on a glossy surface -- they were ~55 % of the kernel's VALU instructions at 28 % lane utilisation.
// Relative error ~ |y log2 x| * 2^-22: 1e-5 where the lobe is not negligible, even at Ns = 10^4.
DEV float pow_pos(float x, float y) { return y == 0.f ? 1.f : __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
// Specular::Fx (BSDF.cpp:33-40) / Specular::Pdf (:67-76): normalised Blinn-Phong on the half vector
DEV f3 phong_fx(const Bsdf& b, f3 wi) {
    if (wi.z < 0.f || b.m_wo.z < 0.f) return mk3(0.f, 0.f, 0.f);
    const f3 H = normalize(wi + b.m_wo);
    const float factor = (b.ns + 2.0f) * PT_INV_2PI;
    return b.ks * factor * pow_pos(H.z, b.ns);
}
DEV float phong_pdf(const Bsdf& b, f3 wi) {
    if (b.m_wo.z < 0.f || wi.z < 0.f) return 0.f;
    const f3 H = normalize(wi + b.m_wo);
    return (b.ns + 1.0f) * PT_INV_2PI * pow_pos(H.z, b.ns);
}
DEV float diffuse_pdf(const Bsdf& b, f3 wi) { return (wi.z < 0.f || b.m_wo.z < 0.f) ? 0.f : (wi.z * PT_INV_PI); }   // BSDF.cpp:28-31
// BSDF::Fx (BSDF.cpp:112-121) and BSDF::Pdf (:153-163) for a world direction; Diffuse::Fx has no hemisphere test (A-22)
DEV void bsdf_eval(const Bsdf& b, f3 dir_world, f3& fx, float& pdf) {
    const f3 wo = to_local(b, dir_world);
    fx = b.kd * PT_INV_PI;
    pdf = diffuse_pdf(b, wo) * b.w_diff;
    if (b.kind == BSDF_PHONG) { fx = phong_fx(b, wo) + fx; pdf = phong_pdf(b, wo) * b.w_spec + pdf; }
}
struct Scatter { f3 wo, f; float pdf; bool mirror; };
// BSDF::Sample (BSDF.cpp:123-151) with Diffuse::Sample (:11-26), Specular::Sample (:42-65),
// specular_reflection::Sample (:78-85).  xi_lobe picks the lobe through the weight prefix sums (lower_bound).
DEV Scatter bsdf_sample(const Bsdf& b, float xi_lobe, float xi1, float xi2) {
    Scatter s; s.wo = mk3(0.f, 0.f, 0.f); s.f = mk3(0.f, 0.f, 0.f); s.pdf = 0.f; s.mirror = false;
    const bool two = b.kind != BSDF_DIFFUSE;
    const float total = two ? (b.w_spec + b.w_diff) : b.w_diff;
    const bool pick_spec = two && (b.w_spec >= xi_lobe * total);
    if (!pick_spec) {                                                    // Diffuse lobe sampled
        if (!(b.m_wo.z < 0.f)) {
            const float phi = xi1 * 2.f * PT_PI;
            // theta = acos(1 - 2 xi2) / 2 (BSDF.cpp:15-16)  =>  cos theta = sqrt(1 - xi2), sin theta = sqrt(xi2): two v_sqrt_f32 instead of
            // an acosf expansion and two more sin/cos
            const float st = __builtin_amdgcn_sqrtf(xi2), ct = __builtin_amdgcn_sqrtf(1.f - xi2), sp = __sinf(phi), cp = __cosf(phi);
            s.wo = mk3(st * cp, st * sp, ct);
            s.f = b.kd * PT_INV_PI;
            s.pdf = fabsf(ct) * PT_INV_PI;
        }
        s.pdf *= b.w_diff;
        if (b.kind == BSDF_PHONG) { s.f = s.f + phong_fx(b, s.wo); s.pdf += phong_pdf(b, s.wo) * b.w_spec; }
        // mirror companion: Fx = 0, Pdf = 0 (BSDF.h:80-82)
    } else if (b.kind == BSDF_PHONG) {
        if (!(b.m_wo.z < 0.f)) {
            const float phi = 2.f * PT_PI * xi1;
            const float cosT = pow_pos(xi2, rcp(b.ns + 1.f));
            const float sinT = __builtin_amdgcn_sqrtf(fmaxf(1.f - cosT * cosT, 0.f));
            const float sp = __sinf(phi), cp = __cosf(phi);
            const f3 H = mk3(sinT * cp, sinT * sp, cosT);
            const f3 wi = -b.m_wo + H * 2.f * dot(H, b.m_wo);
            if (!(wi.z < 0.f)) {
                s.wo = wi; s.f = phong_fx(b, wi);
                s.pdf = (b.ns + 1.f) * PT_INV_2PI * pow_pos(cosT, b.ns);
            }
        }
        s.pdf *= b.w_spec;
        s.f = s.f + b.kd * PT_INV_PI;                                    // other lobe: Diffuse::Fx(wo), no test
        s.pdf += diffuse_pdf(b, s.wo) * b.w_diff;
    } else {                                                             // mirror
        if (!(b.m_wo.z < 0.f)) {
            s.wo = mk3(-b.m_wo.x, -b.m_wo.y, b.m_wo.z);
            s.f = mk3(1.f, 1.f, 1.f) / b.m_wo.z;
            s.pdf = 1.f; s.mirror = true;
        }
        s.pdf *= b.w_spec;
        s.f = s.f + b.kd * PT_INV_PI;
        s.pdf += diffuse_pdf(b, s.wo) * b.w_diff;
    }
    s.wo = to_world(b, s.wo);
    return s;
}

DEV float power_heuristic(float p1, float p2) { const float s = p1 * p1 + p2 * p2; return s == 0.f ? 0.f : p1 * p1 * rcp(s); }   // utils.h:56-60

// ---------------------------------------------------------------------------------------------- light sampling
struct LightSample { f3 wo, rad; float pdf, t2; int tri; bool self_hit; };
// The light pick depends only on a random number, so its record and fp64 corners can be requested as soon as the path's
// RNG key is known -- light_fetch() is issued together with the hit's shading loads (one memory round trip, not two).
struct LightData { float4 a, b, c, e; d3 v0, v1, v2; };   // a..e = the 64-B DevLight record
DEV LightData light_fetch(const DevScene& sc, float xi_l) {
    const int cnt = sc.n_lights;
    int idx = (int)(xi_l * (float)cnt); idx = idx < cnt - 1 ? idx : cnt - 1;          // Render.cpp:204-205
    const float4* L = reinterpret_cast<const float4*>(sc.lights + idx);
    LightData d;
    d.a = L[0]; d.b = L[1]; d.c = L[2]; d.e = L[3];
    const double* P = sc.light_pos64 + 9 * (size_t)idx;            // (not tri_pos64[lights[idx].tri]: that would wait for the record)
    d.v0 = ld_d3(P); d.v1 = ld_d3(P + 3); d.v2 = ld_d3(P + 6);
    return d;
}
// Render::sample (Render.cpp:202-223) [guard = true] / the sampling half of sample_light (:177-200) [guard = false].
// The light point is interpolated in fp64 and rounded to fp32 exactly like `vec3 point = light->interplote_Vertex(..)`.
// self_hit = Triangle::isIntersect (Triangle.cpp:83-106) of the SAMPLED triangle against the shadow ray in fp64 with the
// reference's inclusive t <= t2 = float(|d|): the rounding-level self-occlusion of SURVEY A-9.
DEV LightSample sample_light(const LightData& ld, d3 p64, float xi_u, float xi_v, bool guard, d3 centre) {
    // DevLight layout: tri, area, radiance[3], n0[3], n1[3], n2[3], pad  ->  a = {tri, area, r, g}  b = {b, n0x, n0y, n0z}  c = {n1x, n1y, n1z, n2x}  e = {n2y, n2z, pad, -}
    const int ltri = __float_as_int(ld.a.x); const float area = ld.a.y;
    const f3 rad = mk3(ld.a.z, ld.a.w, ld.b.x);
    const f3 n0 = mk3(ld.b.y, ld.b.z, ld.b.w), n1 = mk3(ld.c.x, ld.c.y, ld.c.z), n2 = mk3(ld.c.w, ld.e.x, ld.e.y);
    float u = xi_u, v = xi_v;
    if (u + v > 1.f) { u = 1.f - u; v = 1.f - v; }                      // Triangle.cpp:15-22
    const d3 v0 = ld.v0, v1 = ld.v1, v2 = ld.v2;
    const double b1 = (double)u, b2 = (double)v;
    // (`centre`: device coordinates are relative to DevScene::centre; the two points the reference rounds to fp32 are WORLD points, and the
    // self-occlusion coin below hangs on the last bit of exactly those roundings -- so the centre is added back, in fp64, before them)
    const f3 point = to_f3(((1.0 - b1 - b2) * v0 + b1 * v1 + b2 * v2) + centre);
    const float w = 1.f - u - v;
    const f3 normal = normalize(mk3(w * n0.x + u * n1.x + v * n2.x, w * n0.y + u * n1.y + v * n2.y, w * n0.z + u * n1.z + v * n2.z));
    // The next five values feed the fp64 self-hit predicate below, whose verdict hangs on their LAST BIT (SURVEY A-9): they are
    // computed with the reference's exact rounding sequence (glm: products and sums rounded one by one, (x*x + y*y) + z*z,
    // v * (1 / sqrt(dot)); Render.cpp:208-213,217): contraction is switched off for this block and sqrt / divide are IEEE.
    const f3 po = to_f3(p64 + centre);
    f3 d, dir; float d2, t2;
    {
#pragma clang fp contract(off)
        d = mk3(point.x - po.x, point.y - po.y, point.z - po.z);
        d2 = (d.x * d.x + d.y * d.y) + d.z * d.z;
        const float len = __builtin_sqrtf(d2);             // IEEE (correctly rounded) sqrt and divide: hipcc default for sqrtf and `/`
        const float inv_len = 1.0f / len;
        dir = mk3(d.x * inv_len, d.y * inv_len, d.z * inv_len);
        t2 = len;
    }
    const float cs = dot(-dir, normal);
    LightSample ls;
    ls.pdf = 0.f;
    if (!guard || cs != 0.f) ls.pdf = d2 * rcp(cs) * rcp(area);
    ls.wo = dir; ls.rad = rad; ls.t2 = t2; ls.tri = ltri;
    // fp64 Moller-Trumbore any-hit on the sampled triangle only
    const d3 e1 = v1 - v0, e2 = v2 - v0, dd = to_d3(dir);
    const d3 h = cross(dd, e2);
    const double det = dot(e1, h);
    bool self = false;
    if (!(fabs(det) < 1e-6)) {
        const double inv = rcp64(det);
        const d3 s = p64 - v0;
        const double uu = inv * dot(s, h);
        if (!(uu < 0.0 || uu > 1.0)) {
            const d3 q = cross(s, e1);
            const double vv = inv * dot(dd, q);
            if (!(vv < 0.0 || uu + vv > 1.0)) {
                const double t = inv * dot(e2, q);
                self = !(t < 0.0001 || t > (double)ls.t2);
            }
        }
    }
    ls.self_hit = self;
    return ls;
}
// The fp64 hit point from the triangle's fp64 PLANE: p = o + t d with t = (n.v0 - n.o) / (n.d), all in double.  Like the reference's
// interpolated point (Triangle.cpp:35-38 with fp64 barycentrics) it lies on the triangle's plane to ~1e-16 and carries full fp64
// rounding noise in its low bits -- the two properties the self-occlusion statistics of SURVEY A-9 depend on (checked against the
// real reference's images and self-occlusion rate by the tests) -- but costs two 16-B loads instead of five and a third of the flops.
DEV d3 hit_point64_plane(const DevScene& sc, int tri, d3 o64, f3 dir) {
    const double* P = reinterpret_cast<const double*>(sc.tri_shade + MCPT_TRI_SHADE_F4 * (size_t)tri + 4);
    const d3 n = ld_d3(P); const double nd0 = P[3];
    const d3 dd = to_d3(dir);
    const double t = (nd0 - dot(n, o64)) * rcp64(dot(n, dd));
    return mkd(fma(t, dd.x, o64.x), fma(t, dd.y, o64.y), fma(t, dd.z, o64.z));
}
// the same with the plane record fetched by the caller (the shade kernel requests it a phase early, beside the hit's shading record)
DEV d3 hit_point64_plane(const double4 pl, d3 o64, f3 dir) {
    const d3 n = mkd(pl.x, pl.y, pl.z);
    const d3 dd = to_d3(dir);
    const double t = (pl.w - dot(n, o64)) * rcp64(dot(n, dd));
    return mkd(fma(t, dd.x, o64.x), fma(t, dd.y, o64.y), fma(t, dd.z, o64.z));
}
DEV LightSample sample_light(const DevScene& sc, d3 p64, float xi_l, float xi_u, float xi_v, bool guard) {
    return sample_light(light_fetch(sc, xi_l), p64, xi_u, xi_v, guard, mkd(sc.centre[0], sc.centre[1], sc.centre[2]));
}
