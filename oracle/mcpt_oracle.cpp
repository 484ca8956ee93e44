// ======================================================================================================
// TEST INFRASTRUCTURE ONLY -- CPU oracle for the MI355X path tracer.  NOT part of the product: nothing under
// monte-carlo-path-tracer_amd/ includes, links or calls this file.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load liboracle.so.
//
// What it is: a from-scratch restatement, in plain C++ (no glm, no shared_ptr, no virtual lobes), of the
// hot path of laizesheng1/Monte-Carlo-Path-Tracer: Render::render -> cast_Ray -> ray_tracing (+ sample /
// sample_light), BVH build / hit / has_hit, AABB::Intersection, Triangle::hit / isIntersect, the BSDF lobes,
// Texture::get_color, Scene::set_Pixel / getPixelsColor.  Every function cites the reference lines it follows.
// Precision follows the reference exactly: fp64 geometry (dvec3), fp32 shading (vec3), same operation order
// as the glm expressions it replaces (dot = (x+y)+z, normalize = v * (1/sqrt(dot)), func_geometric.inl:48-90).
//
// Parity pin: tests/test_oracle_vs_reference.py drives this file and the REAL reference (oracle/_ref, built by
// oracle/build_ref.sh) with the same injected random numbers (RNG mode SEQ below, reference side:
// oracle/ref/ref_shim.h) and requires equal results function by function and path by path; the outputs of
// the real reference are committed as tests/golden/*.npz so the pin also holds where /root/reference is absent.
//
// Random numbers.  The reference draws from one global mt19937 (utils.h:23-28), so a pixel's numbers depend
// on every pixel before it; no seed exists.  This oracle therefore has two modes:
//   SEQ     : numbers are popped from a caller-supplied array in the reference's own draw order (SURVEY A-20)
//             -- used only for the pin above.
//   COUNTER : xi = pcg4d(pixel, sample, block, seed) -- the SPEC shared with the HIP kernels:
//             block 0 = {camera xi_x, xi_y}; bounce b: block 1+2b = {light pick, light u, light v, lobe pick},
//             block 2+2b = {lobe xi_1, lobe xi_2, russian roulette, unused}.  A skipped draw shifts nothing.
// ======================================================================================================
#include "../include/mcpt.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>
#include <chrono>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ------------------------------------------------------------------------------------------- tiny vectors
struct D3 { double x, y, z; double& operator[](int i) { return (&x)[i]; } double operator[](int i) const { return (&x)[i]; } };
struct F3 { float x, y, z; };
struct D2 { double x, y; };

inline D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline D3 operator-(D3 a) { return {-a.x, -a.y, -a.z}; }
inline D3 operator*(double s, D3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline D3 operator*(D3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline D3 operator/(D3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline D3 operator/(D3 a, D3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline double dot(D3 a, D3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }           // func_geometric.inl:48-55
inline D3 cross(D3 a, D3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }   // :68-79
inline double length(D3 a) { return std::sqrt(dot(a, a)); }
inline D3 normalize(D3 a) { return a * (1.0 / std::sqrt(dot(a, a))); }                  // :82-90
inline D3 vmin(D3 a, D3 b) { return {b.x < a.x ? b.x : a.x, b.y < a.y ? b.y : a.y, b.z < a.z ? b.z : a.z}; }
inline D3 vmax(D3 a, D3 b) { return {a.x < b.x ? b.x : a.x, a.y < b.y ? b.y : a.y, a.z < b.z ? b.z : a.z}; }

inline F3 operator+(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline F3 operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline F3 operator-(F3 a) { return {-a.x, -a.y, -a.z}; }
inline F3 operator*(F3 a, F3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline F3 operator*(float s, F3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline F3 operator*(F3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline F3 operator/(F3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline F3& operator+=(F3& a, F3 b) { a = a + b; return a; }
inline F3& operator*=(F3& a, F3 b) { a = a * b; return a; }
inline float dot(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline F3 cross(F3 a, F3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline float length(F3 a) { return std::sqrt(dot(a, a)); }
inline F3 normalize(F3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline F3 toF(D3 a) { return {float(a.x), float(a.y), float(a.z)}; }
inline D3 toD(F3 a) { return {double(a.x), double(a.y), double(a.z)}; }

const float PI_F = 3.1415926f;   // utils.h:20  (shading); cast_Ray alone uses full double pi (Render.cpp:73)
const double PI_D = 3.14159265358979323846;

// ------------------------------------------------------------------------------------------- RNG
inline void pcg4d(uint32_t v[4]) {   // Jarzynski & Olano 2020, "Hash Functions for GPU Rendering", pcg4d
    for (int i = 0; i < 4; i++) v[i] = v[i] * 1664525u + 1013904223u;
    v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
    for (int i = 0; i < 4; i++) v[i] ^= v[i] >> 16;
    v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
}
inline void counter_block(uint32_t pixel, uint32_t sample, uint32_t block, uint64_t seed, float out[4]) {
    uint32_t v[4] = {pixel, sample, block ^ (uint32_t(seed >> 32) * 0x9E3779B9u), uint32_t(seed)};
    pcg4d(v);
    for (int i = 0; i < 4; i++) out[i] = float(v[i] >> 8) * (1.0f / 16777216.0f);   // [0,1), 24 bits, never 1.0
}

struct Rng {
    int mode = 0;                 // 0 = COUNTER, 1 = SEQ
    uint32_t pixel = 0, sample = 0; uint64_t seed = 0;
    uint32_t cached = 0xffffffffu; float cache[4];
    const float* q = nullptr; int n = 0, pos = 0, underflow = 0;
    float get(uint32_t block, int lane) {
        if (mode == 1) { if (pos < n) return q[pos++]; underflow++; return 0.f; }
        if (block != cached) { counter_block(pixel, sample, block, seed, cache); cached = block; }
        return cache[lane];
    }
};
enum { L_LIGHT = 0, L_U = 1, L_V = 2, L_LOBE = 3, L_XI1 = 0, L_XI2 = 1, L_RR = 2 };
inline uint32_t blockA(int bounce) { return 1u + 2u * uint32_t(bounce); }
inline uint32_t blockB(int bounce) { return 2u + 2u * uint32_t(bounce); }

// ------------------------------------------------------------------------------------------- scene
struct Tex { int w, h; std::vector<F3> texel; };
struct Mat { D3 ks; double ns; D3 radiance; int tex; };
struct Tri {                       // Triangle.h:11-14
    D3 v[3], vn[3]; D2 uv[3]; D3 A, B; int mat;
};
struct Ray { D3 start, direction; double t1 = 0.0001, t2 = std::numeric_limits<double>::max(); };   // Render.h:26-32
struct Hit {                       // hitInfo, Render.h:14-24
    double t = 0; D3 wi{0, 0, 0}, point{0, 0, 0}, normal{0, 0, 0}; D2 uv{0, 0}; bool front = false; int mat = -1; float lightarea = 0.f;
    int tri = -1;
};
struct Node { int left = -1, right = -1, first = 0, count = 0; D3 A, B; };   // BVH.h:11-20
struct Counters { uint64_t paths = 0, rays_primary = 0, rays_cont = 0, rays_shadow = 0, box = 0, tri = 0, self_tests = 0, self_hits = 0; };

struct Scene {
    std::vector<Tri> tris; std::vector<Mat> mats; std::vector<Tex> texs; std::vector<int> lights;
    std::vector<Node> nodes; std::vector<int> order;   // order = BVH::triangles after all partitions
    mcpt_camera cam;
    uint32_t max_depth = 0, flags = 0, integrator = 0;
};

// Triangle.cpp:24-28
inline float tri_area(const Tri& t) { F3 a = toF(t.v[1] - t.v[0]), b = toF(t.v[2] - t.v[0]); return 0.5f * length(cross(a, b)); }
inline D3 tri_center(const Tri& t) { return (t.v[0] + t.v[1] + t.v[2]) / 3.0; }                       // Triangle.cpp:30-33
inline D3 interp_vertex(const Tri& t, double b1, double b2) { return (1 - b1 - b2) * t.v[0] + b1 * t.v[1] + b2 * t.v[2]; }   // :35-38
inline D3 interp_normal(const Tri& t, double b1, double b2) { return normalize((1 - b1 - b2) * t.vn[0] + b1 * t.vn[1] + b2 * t.vn[2]); }   // :39-42
inline D2 interp_uv(const Tri& t, double b1, double b2) {                                              // :43-46
    double w = 1 - b1 - b2;
    return {w * t.uv[0].x + b1 * t.uv[1].x + b2 * t.uv[2].x, w * t.uv[0].y + b1 * t.uv[1].y + b2 * t.uv[2].y};
}

// AABB.cpp:25-36 -- divides per node, far plane *1.001 after the swap, strict tmin < tmax
inline bool aabb_intersect(const D3& A, const D3& B, const Ray& ray) {
    D3 v0 = (A - ray.start) / ray.direction;
    D3 v1 = (B - ray.start) / ray.direction;
    double tmin = ray.t1, tmax = ray.t2;
    for (int i = 0; i < 3; i++) {
        if (v0[i] > v1[i]) std::swap(v0[i], v1[i]);
        v1[i] *= 1.001;
        tmin = tmin < v0[i] ? v0[i] : tmin;
        tmax = tmax > v1[i] ? v1[i] : tmax;
    }
    return tmin < tmax;
}

// Triangle.cpp:48-80 -- Moller-Trumbore, no culling, |a| < 1e-5f rejected, t in [t1,t2), u,v,1-u-v >= 0
inline bool tri_hit(const Scene& s, int ti, const Ray& ray, Hit& info) {
    const Tri& T = s.tris[ti];
    D3 edge1 = T.v[1] - T.v[0], edge2 = T.v[2] - T.v[0];
    D3 h = cross(ray.direction, edge2);
    double a = dot(edge1, h);
    if (std::abs(a) < 0.00001f) return false;
    D3 sv = ray.start - T.v[0];
    double u = dot(sv, h);
    D3 q = cross(sv, edge1);
    double v = dot(ray.direction, q);
    double t = dot(edge2, q);
    double inv_a = 1.0 / a;
    u *= inv_a; v *= inv_a; t *= inv_a;
    if (t >= ray.t1 && t < ray.t2 && u >= 0 && v >= 0 && (1 - u - v) >= 0) {
        info.t = t;
        info.point = interp_vertex(T, u, v);
        info.normal = interp_normal(T, u, v);
        info.front = dot(info.normal, ray.direction) < 0.0;
        info.wi = -ray.direction;
        info.uv = interp_uv(T, u, v);
        info.mat = T.mat; info.tri = ti;
        if (length(s.mats[T.mat].radiance)) info.lightarea = tri_area(T);   // Triangle.cpp:75-76
        return true;
    }
    return false;
}

// Triangle.cpp:83-106 -- any-hit, |det| < 1e-6, t in [t1,t2] INCLUSIVE (the root of SURVEY A-9)
inline bool tri_any(const Tri& T, const Ray& ray) {
    const double EPSILON = 1e-6;
    D3 edge1 = T.v[1] - T.v[0], edge2 = T.v[2] - T.v[0];
    D3 h = cross(ray.direction, edge2);
    double det = dot(edge1, h);
    if (std::fabs(det) < EPSILON) return false;
    double invDet = 1.0 / det;
    D3 sv = ray.start - T.v[0];
    double u = invDet * dot(sv, h);
    if (u < 0.0 || u > 1.0) return false;
    D3 q = cross(sv, edge1);
    double v = invDet * dot(ray.direction, q);
    if (v < 0.0 || u + v > 1.0) return false;
    double t = invDet * dot(edge2, q);
    if (t < ray.t1 || t > ray.t2) return false;
    return true;
}

// BVH.cpp:15-54 -- midpoint split on the longest axis of the CENTROID box, float mid value, leaf <= 5,
// std::partition (same libstdc++ algorithm as the reference build => same leaf order and tie-breaking).
int bvh_build(Scene& s, int l, int r) {
    int id = int(s.nodes.size());
    s.nodes.emplace_back();
    const double mx = std::numeric_limits<double>::max(), lo = std::numeric_limits<double>::lowest();
    D3 A{mx, mx, mx}, B{lo, lo, lo};                                        // AABB.h:14
    for (int i = l; i < r; i++) { A = vmin(A, s.tris[s.order[i]].A); B = vmax(B, s.tris[s.order[i]].B); }
    s.nodes[id].A = A; s.nodes[id].B = B;
    if (r - l <= 5) { s.nodes[id].first = l; s.nodes[id].count = r - l; return id; }
    D3 cA{mx, mx, mx}, cB{lo, lo, lo};
    for (int i = l; i < r; i++) { D3 c = tri_center(s.tris[s.order[i]]); cA = vmin(cA, c); cB = vmax(cB, c); }
    int axis = 0; double len = cB.x - cA.x;                                 // AABB.cpp:12-23
    for (int i = 0; i < 3; i++) { double t = cB[i] - cA[i]; if (len < t) { len = t; axis = i; } }
    float mid_val = (cA[axis] + cB[axis]) / 2.0f;                           // BVH.cpp:39 (double expr narrowed to float)
    auto mid = std::partition(s.order.begin() + l, s.order.begin() + r,
                              [&](int ti) { return tri_center(s.tris[ti])[axis] < mid_val; });
    int mid_idx = int(mid - s.order.begin());
    if (mid_idx == l || mid_idx == r) mid_idx = (l + r) / 2;
    int L = bvh_build(s, l, mid_idx);
    int R = bvh_build(s, mid_idx, r);
    s.nodes[id].left = L; s.nodes[id].right = R;
    return id;
}

// BVH.cpp:95-113 -- left, then right, then own triangles; accepted hit shrinks ray.t2
bool bvh_hit(const Scene& s, int ni, Ray& ray, Hit& info, Counters& c) {
    const Node& n = s.nodes[ni];
    bool is_hit = false;
    c.box++;
    if (!aabb_intersect(n.A, n.B, ray)) return false;
    if (n.left >= 0) is_hit |= bvh_hit(s, n.left, ray, info, c);
    if (n.right >= 0) is_hit |= bvh_hit(s, n.right, ray, info, c);
    for (int i = 0; i < n.count; i++) {
        c.tri++;
        if (tri_hit(s, s.order[n.first + i], ray, info)) { ray.t2 = info.t; is_hit = true; }
    }
    return is_hit;
}
// BVH.cpp:120-136.  `skip` (>= 0) removes one triangle from the test: only used with MCPT_FLAG_CORRECT_SHADOW_T2.
bool bvh_any(const Scene& s, int ni, const Ray& ray, Counters& c, int skip) {
    const Node& n = s.nodes[ni];
    c.box++;
    if (!aabb_intersect(n.A, n.B, ray)) return false;
    if (n.left >= 0 && bvh_any(s, n.left, ray, c, skip)) return true;
    if (n.right >= 0 && bvh_any(s, n.right, ray, c, skip)) return true;
    for (int i = 0; i < n.count; i++) {
        int ti = s.order[n.first + i];
        if (ti == skip) continue;
        c.tri++;
        if (tri_any(s.tris[ti], ray)) return true;
    }
    return false;
}

// model.cpp:30-41 + utils.h:30-34 (clamp01 takes a FLOAT and returns double; 0.999 cap; nearest texel, no v flip)
inline double clamp01(float d) { if (d > 0.999f) return 0.999; if (d < 0.0f) return 0.0; return d; }
inline F3 tex_color(const Tex& t, D2 uv) {
    if (t.texel.size() == 1) return t.texel[0];
    double u = clamp01(float(uv.x - std::floor(uv.x)));
    double v = clamp01(float(uv.y - std::floor(uv.y)));
    int x = int(u * t.w), y = int(v * t.h);
    return t.texel[size_t(y) * t.w + x];
}

// ------------------------------------------------------------------------------------------- BSDF
struct Onb { F3 u, v, w; };                                  // BSDF.h:9-27
inline Onb make_onb(F3 n) {
    Onb o; o.w = n;
    F3 a = (std::fabs(n.x) > 0.9f) ? F3{0, 1, 0} : F3{1, 0, 0};
    o.v = normalize(cross(o.w, a));
    o.u = cross(o.w, o.v);
    return o;
}
inline F3 to_world(const Onb& o, F3 a) { return a.x * o.u + a.y * o.v + a.z * o.w; }
inline F3 to_local(const Onb& o, F3 t) { return {dot(t, o.u), dot(t, o.v), dot(t, o.w)}; }

enum { LOBE_DIFFUSE = 0, LOBE_PHONG = 1, LOBE_MIRROR = 2 };
struct Lobe { int kind; F3 reflect; float coeff; float weight; };
struct Scatter { F3 wo{0, 0, 0}, f{0, 0, 0}; float pdf = 0.f; bool mirror = false; };   // BSDF.h:29-38
struct Bsdf { Onb onb; F3 m_wo; Lobe lobe[2]; int n = 0; };

inline F3 lobe_fx(const Lobe& l, F3 m_wo, F3 wi) {
    if (l.kind == LOBE_DIFFUSE) return l.reflect / PI_F;                                 // BSDF.cpp:4-9 (no hemisphere test)
    if (l.kind == LOBE_PHONG) {                                                          // BSDF.cpp:33-40
        if (wi.z < 0.f || m_wo.z < 0.f) return {0, 0, 0};
        F3 H = normalize(wi + m_wo);
        float factor = (l.coeff + 2) / (2.f * PI_F);
        return l.reflect * factor * std::pow(H.z, l.coeff);
    }
    return {0, 0, 0};                                                                    // BSDF.h:80
}
inline float lobe_pdf(const Lobe& l, F3 m_wo, F3 wi) {
    if (l.kind == LOBE_DIFFUSE) return (wi.z < 0.f || m_wo.z < 0.f) ? 0.f : (wi.z / PI_F);   // BSDF.cpp:28-31
    if (l.kind == LOBE_PHONG) {                                                          // BSDF.cpp:67-76
        if (m_wo.z < 0.f || wi.z < 0.f) return 0.f;
        F3 H = normalize(wi + m_wo);
        return (l.coeff + 1) / (2.f * PI_F) * std::pow(H.z, l.coeff);
    }
    return 0.f;                                                                          // BSDF.h:82
}
inline Scatter lobe_sample(const Lobe& l, F3 m_wo, Rng& rng, int bounce) {
    Scatter s;
    if (l.kind == LOBE_DIFFUSE) {                                                        // BSDF.cpp:11-26
        if (m_wo.z < 0) return s;
        F3 f = l.reflect / PI_F;
        float phi = rng.get(blockB(bounce), L_XI1) * 2 * PI_F;
        float theta = 0.5f * std::acos(1 - 2 * rng.get(blockB(bounce), L_XI2));
        F3 dir{std::sin(theta) * std::cos(phi), std::sin(theta) * std::sin(phi), std::cos(theta)};
        s.wo = dir; s.f = f; s.pdf = std::abs(dir.z) / PI_F;
        return s;
    }
    if (l.kind == LOBE_PHONG) {                                                          // BSDF.cpp:42-65
        if (m_wo.z < 0.f) return s;
        float u = rng.get(blockB(bounce), L_XI1), v = rng.get(blockB(bounce), L_XI2);
        float phi = 2 * PI_F * u;
        float cosTheta = std::pow(v, 1.f / (l.coeff + 1));
        float sinTheta = std::sqrt(1.f - cosTheta * cosTheta);
        F3 H{sinTheta * std::cos(phi), sinTheta * std::sin(phi), cosTheta};
        F3 wi = -m_wo + H * 2.f * dot(H, m_wo);
        if (wi.z < 0.f) return s;
        float pdf = (l.coeff + 1) / (2.f * PI_F) * std::pow(cosTheta, l.coeff);
        s.wo = wi; s.f = lobe_fx(l, m_wo, wi); s.pdf = pdf;
        return s;
    }
    if (m_wo.z < 0.f) return s;                                                          // BSDF.cpp:78-85
    s.wo = {-m_wo.x, -m_wo.y, m_wo.z};
    s.f = F3{1.f, 1.f, 1.f} / m_wo.z;
    s.pdf = 1.f; s.mirror = true;
    return s;
}

// BSDF::BSDF (BSDF.cpp:87-110) + get_sample_weight (:165-186) + energy_conservation (:188-202)
inline Bsdf make_bsdf(const Scene& s, const Hit& info) {
    Bsdf b;
    b.onb = make_onb(toF(info.normal));
    b.m_wo = to_local(b.onb, toF(info.wi));
    const Mat& m = s.mats[info.mat];
    F3 kd = tex_color(s.texs[m.tex], info.uv);
    if (length(m.ks)) {
        if (m.ns >= 10000) b.lobe[b.n++] = {LOBE_MIRROR, F3{1, 1, 1}, 0.f, 0.f};
        else b.lobe[b.n++] = {LOBE_PHONG, toF(m.ks), float(m.ns), 0.f};
    }
    b.lobe[b.n++] = {LOBE_DIFFUSE, kd, 0.f, 0.f};
    float lum[2], sum = 0.f;
    for (int i = 0; i < b.n; i++) { F3 r = b.lobe[i].reflect; lum[i] = r.x * 0.212671f + r.y * 0.715160f + r.z * 0.072169f; sum += lum[i]; }
    if (sum != 0) { float inv = 1.f / sum; for (int i = 0; i < b.n; i++) b.lobe[i].weight = lum[i] * inv; }
    // sum == 0: the reference leaves `weight` uninitialised (SURVEY A-12); defined here as 0 => the path ends.
    F3 tot{0, 0, 0};
    for (int i = 0; i < b.n; i++) tot += b.lobe[i].reflect;
    float maxc = std::max(tot.x, std::max(tot.y, tot.z));
    if (!(maxc < 1.f)) for (int i = 0; i < b.n; i++) b.lobe[i].reflect = b.lobe[i].reflect / maxc;
    return b;
}
inline F3 bsdf_fx(const Bsdf& b, F3 wi_world) {                                          // BSDF.cpp:112-121
    F3 wo = to_local(b.onb, wi_world), ret{0, 0, 0};
    for (int i = 0; i < b.n; i++) ret += lobe_fx(b.lobe[i], b.m_wo, wo);
    return ret;
}
inline float bsdf_pdf(const Bsdf& b, F3 wi_world) {                                      // BSDF.cpp:153-163
    F3 wo = to_local(b.onb, wi_world); float ret = 0.f;
    for (int i = 0; i < b.n; i++) ret += lobe_pdf(b.lobe[i], b.m_wo, wo) * b.lobe[i].weight;
    return ret;
}
inline Scatter bsdf_sample(const Bsdf& b, Rng& rng, int bounce) {                        // BSDF.cpp:123-151
    float prefix[2];
    for (int i = 0; i < b.n; i++) { prefix[i] = b.lobe[i].weight; if (i) prefix[i] += prefix[i - 1]; }
    float r = rng.get(blockA(bounce), L_LOBE) * prefix[b.n - 1];
    int index = int(std::lower_bound(prefix, prefix + b.n, r) - prefix);
    index = std::min(index, b.n - 1);
    Scatter s = lobe_sample(b.lobe[index], b.m_wo, rng, bounce);
    s.pdf *= b.lobe[index].weight;
    for (int i = 0; i < b.n; i++) {
        if (i == index) continue;
        s.f += lobe_fx(b.lobe[i], b.m_wo, s.wo);
        s.pdf += lobe_pdf(b.lobe[i], b.m_wo, s.wo) * b.lobe[i].weight;
    }
    s.wo = to_world(b.onb, s.wo);
    return s;
}

inline float power_heuristic(float p1, float p2) { float sum = p1 * p1 + p2 * p2; return sum == 0.f ? 0.f : p1 * p1 / sum; }   // utils.h:56-60

// ------------------------------------------------------------------------------------------- light sampling
struct LightSample { F3 wo, f; float pdf; Ray ray; int tri; };
// Render::sample (Render.cpp:202-223); guard = the `if (cos != 0)` that sample_light (:177-200) lacks
inline LightSample sample_light_point(const Scene& s, const Hit& info, Rng& rng, int bounce, bool guard) {
    int cnt = int(s.lights.size());
    int idx = std::min(int(rng.get(blockA(bounce), L_LIGHT) * cnt), cnt - 1);
    int ti = s.lights[idx];
    const Tri& T = s.tris[ti];
    float u = rng.get(blockA(bounce), L_U), v = rng.get(blockA(bounce), L_V);           // Triangle.cpp:15-22
    if (u + v > 1) { u = 1 - u; v = 1 - v; }
    F3 point = toF(interp_vertex(T, u, v));
    F3 normal = toF(interp_normal(T, u, v));
    F3 d = point - toF(info.point);
    F3 dir = normalize(d);
    float d2 = dot(d, d);
    float cs = dot(-dir, normal);
    float pdf = 0.f;
    if (!guard || cs != 0) pdf = d2 / cs / tri_area(T);
    LightSample ls;
    ls.wo = dir; ls.f = toF(s.mats[T.mat].radiance); ls.pdf = pdf; ls.tri = ti;
    ls.ray.start = info.point; ls.ray.direction = toD(dir); ls.ray.t2 = length(d);
    return ls;
}
// shadow test: BVH::has_hit as the reference does it, or -- with MCPT_FLAG_CORRECT_SHADOW_T2 -- ignoring the
// sampled light triangle itself.  Also counts how often the sampled triangle alone rejects the ray (A-9).
inline bool shadow_blocked(const Scene& s, const LightSample& ls, Counters& c) {
    c.rays_shadow++;
    c.self_tests++;
    bool self = tri_any(s.tris[ls.tri], ls.ray);
    if (self) c.self_hits++;
    if (s.flags & MCPT_FLAG_CORRECT_SHADOW_T2) return bvh_any(s, 0, ls.ray, c, ls.tri);
    return bvh_any(s, 0, ls.ray, c, -1);
}

// ------------------------------------------------------------------------------------------- integrators
// Render::cast_Ray (Render.cpp:71-80)
inline Ray cast_ray(const Scene& s, int x, int y, Rng& rng) {
    const mcpt_camera& c = s.cam;
    D3 eye{c.eye[0], c.eye[1], c.eye[2]}, lookat{c.lookat[0], c.lookat[1], c.lookat[2]}, up{c.up[0], c.up[1], c.up[2]};
    double h = std::tan(c.fovy * PI_D / 180.0 * 0.5) * 2.0;
    D3 front = normalize(lookat - eye);
    D3 right = normalize(cross(front, up));
    double u = ((x + rng.get(0, 0)) / c.width - 0.5) * h * c.width / c.height;
    double v = ((y + rng.get(0, 1)) / c.height - 0.5) * h;
    D3 dir = normalize(front + u * right + v * up);
    Ray r; r.start = eye; r.direction = dir;
    return r;
}

// Render::ray_tracing(Ray&) (Render.cpp:111-175).  The reference re-traces `ray` at the top of every iteration
// (:118); for bounces >= 1 that repeats the trace that produced nextInfo (:144) with an identical result, so it
// is skipped here (and not counted as a ray).
F3 trace_mis(const Scene& s, Ray ray, Rng& rng, Counters& c) {
    F3 L{0, 0, 0}, beta{1, 1, 1};
    Hit info;
    const float nl = float(s.lights.size());
    for (int bounces = 0; s.max_depth == 0 || bounces < int(s.max_depth); bounces++) {
        if (bounces == 0) { c.rays_primary++; if (!bvh_hit(s, 0, ray, info, c)) break; }
        const Mat& mat = s.mats[info.mat];
        if (bounces == 0 && length(mat.radiance) > 0.0001) L += toF(mat.radiance);          // :121-122
        Bsdf bsdf = make_bsdf(s, info);
        LightSample ls = sample_light_point(s, info, rng, bounces, true);
        if (ls.pdf != 0 && !shadow_blocked(s, ls, c)) {                                     // :125-130
            float cos_theta = std::fabs(dot(toF(info.normal), ls.wo));
            float weight = power_heuristic(ls.pdf / nl, bsdf_pdf(bsdf, ls.wo));
            L += weight * beta * ls.f * bsdf_fx(bsdf, ls.wo) * cos_theta / ls.pdf * nl;
        }
        Scatter sc = bsdf_sample(bsdf, rng, bounces);                                       // :133-136
        if (sc.pdf == 0.f) break;
        Ray new_ray; new_ray.start = info.point; new_ray.direction = toD(sc.wo);
        float cos_theta = std::fabs(dot(toF(info.normal), sc.wo));
        beta *= sc.f * cos_theta / sc.pdf;                                                   // :140
        Hit next; Ray tmp = new_ray;
        c.rays_cont++;
        if (!bvh_hit(s, 0, tmp, next, c)) break;                                            // :144
        if (length(s.mats[next.mat].radiance) && next.front) {                               // :146-162
            if (sc.mirror) L += beta * toF(s.mats[next.mat].radiance);
            else {
                D3 d = info.point - next.point;
                double dist2 = length(d) * length(d);
                double cosine = dot(normalize(d), next.normal);
                float lightPdf = 0.f;
                if (cosine != 0) lightPdf = float(dist2 / cosine / nl / next.lightarea);
                float weight = power_heuristic(sc.pdf, lightPdf);
                L += beta * toF(s.mats[next.mat].radiance) * weight;
            }
        }
        if (bounces > 3) {                                                                   // :164-170
            float q = std::min(std::max(std::max(beta.x, beta.y), beta.z), 0.95f);
            if (rng.get(blockB(bounces), L_RR) > q) break;
            beta = beta / q;
        }
        ray = new_ray; info = next;
    }
    return L;
}

// Render::sample_light (Render.cpp:177-200)
F3 sample_light_recursive(const Scene& s, const Hit& info, Rng& rng, int depth, Counters& c) {
    LightSample ls = sample_light_point(s, info, rng, depth, false);
    if (!shadow_blocked(s, ls, c)) {
        F3 kd = tex_color(s.texs[s.mats[info.mat].tex], info.uv);
        return ls.f * kd * std::fabs(dot(toF(info.normal), ls.wo)) / ls.pdf / 2.0f;
    }
    return {0, 0, 0};
}
// Render::ray_tracing(Ray&,int) (Render.cpp:83-109), MAX_DEPTH 10 (Render.h:11) unless max_depth overrides it
F3 trace_recursive(const Scene& s, Ray& ray, int depth, Rng& rng, Counters& c) {
    int maxd = s.max_depth ? int(s.max_depth) : 10;
    if (depth > maxd) return {0, 0, 0};
    Hit info;
    if (depth == 0) c.rays_primary++; else c.rays_cont++;
    if (!bvh_hit(s, 0, ray, info, c)) return {0, 0, 0};
    const Mat& mat = s.mats[info.mat];
    if (length(mat.radiance) > 0.01) return toF(mat.radiance);
    F3 L = sample_light_recursive(s, info, rng, depth, c);
    Bsdf bsdf = make_bsdf(s, info);
    Scatter sc = bsdf_sample(bsdf, rng, depth);
    if (length(sc.wo) < 0.00001f) return {0, 0, 0};
    Ray nr; nr.start = info.point; nr.direction = toD(sc.wo);
    return L + sc.f * std::fabs(dot(toF(info.normal), sc.wo)) * trace_recursive(s, nr, depth + 1, rng, c) / sc.pdf;
}

inline F3 trace(const Scene& s, Ray ray, Rng& rng, Counters& c) {
    c.paths++;
    if (s.integrator == MCPT_INTEGRATOR_RECURSIVE_NEE) return trace_recursive(s, ray, 0, rng, c);
    return trace_mis(s, ray, rng, c);
}

Scene* build_scene(const mcpt_scene_desc* d, uint32_t max_depth, uint32_t integrator, uint32_t flags) {
    Scene* s = new Scene();
    s->cam = d->camera; s->max_depth = max_depth; s->integrator = integrator; s->flags = flags;
    for (uint32_t i = 0; i < d->n_textures; i++) {
        Tex t; t.w = d->textures[i].width; t.h = d->textures[i].height;
        for (int k = 0; k < t.w * t.h; k++) t.texel.push_back({d->textures[i].rgb[3 * k], d->textures[i].rgb[3 * k + 1], d->textures[i].rgb[3 * k + 2]});
        s->texs.push_back(t);
    }
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const mcpt_material& m = d->materials[i];
        s->mats.push_back({{m.ks[0], m.ks[1], m.ks[2]}, m.ns, {m.radiance[0], m.radiance[1], m.radiance[2]}, m.map_kd});
    }
    for (uint32_t f = 0; f < d->n_face; f++) {                                               // Render.cpp:12-44
        Tri t;
        for (int k = 0; k < 3; k++) {
            const int32_t* c = d->face + 12 * f + 4 * k;
            t.v[k] = {d->vertex[3 * c[0]], d->vertex[3 * c[0] + 1], d->vertex[3 * c[0] + 2]};
            t.vn[k] = {d->normal[3 * c[1]], d->normal[3 * c[1] + 1], d->normal[3 * c[1] + 2]};
            t.uv[k] = {d->texcoord[2 * c[2]], d->texcoord[2 * c[2] + 1]};
        }
        t.mat = d->face[12 * f + 3];
        t.A = vmin(t.v[0], vmin(t.v[1], t.v[2])); t.B = vmax(t.v[0], vmax(t.v[1], t.v[2]));
        s->tris.push_back(t);
        if (length(s->mats[t.mat].radiance) > 0.01) s->lights.push_back(int(f));
    }
    s->order.resize(s->tris.size());
    for (size_t i = 0; i < s->order.size(); i++) s->order[i] = int(i);
    if (!s->tris.empty()) bvh_build(*s, 0, int(s->tris.size()));
    return s;
}

inline D3 P3(const double* p) { return {p[0], p[1], p[2]}; }
inline void put3(double* o, D3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
inline void put3f(float* o, F3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
inline Rng seq_rng(const float* xi, int n) { Rng r; r.mode = 1; r.q = xi; r.n = n; return r; }

Scene* probe_scene(const float* kd, const double* ks, double ns) {   // one material, one constant texture
    Scene* s = new Scene();
    Tex t; t.w = t.h = 1; t.texel.push_back({kd[0], kd[1], kd[2]}); s->texs.push_back(t);
    s->mats.push_back({{ks[0], ks[1], ks[2]}, ns, {0, 0, 0}, 0});
    return s;
}
Hit probe_hit(const double* n3, const double* wi3) { Hit h; h.normal = P3(n3); h.wi = P3(wi3); h.uv = {0.25, 0.75}; h.mat = 0; return h; }

}  // namespace

// =========================================================================================== C entry points
extern "C" {

void* oracle_create(const mcpt_scene_desc* d, uint32_t max_depth, uint32_t integrator, uint32_t flags) {
    return build_scene(d, max_depth, integrator, flags);
}
void oracle_destroy(void* h) { delete static_cast<Scene*>(h); }
void oracle_set_opts(void* h, uint32_t max_depth, uint32_t integrator, uint32_t flags) {
    Scene* s = static_cast<Scene*>(h); s->max_depth = max_depth; s->integrator = integrator; s->flags = flags;
}
void oracle_info(void* h, long* out5) {   // tris, lights, nodes, leaves, max depth
    Scene* s = static_cast<Scene*>(h);
    out5[0] = long(s->tris.size()); out5[1] = long(s->lights.size()); out5[2] = long(s->nodes.size());
    long leaves = 0; for (auto& n : s->nodes) if (n.left < 0 && n.right < 0) leaves++;
    out5[3] = leaves;
    // depth by iterative walk
    long maxd = 0; std::vector<std::pair<int, int>> st; if (!s->nodes.empty()) st.push_back({0, 0});
    while (!st.empty()) { auto [ni, dd] = st.back(); st.pop_back(); maxd = std::max<long>(maxd, dd); if (s->nodes[ni].left >= 0) st.push_back({s->nodes[ni].left, dd + 1}); if (s->nodes[ni].right >= 0) st.push_back({s->nodes[ni].right, dd + 1}); }
    out5[4] = maxd;
}
int oracle_light_tri(void* h, int i) { return static_cast<Scene*>(h)->lights[i]; }

// Render::render x spp (Render.cpp:56-69) into a Scene::m_Pixels-shaped accumulator (Scene.cpp:12-21),
// COUNTER rng.  counters_out: 8 x uint64 {paths, primary, continuation, shadow, box, tri, self_tests, self_hits}.
// Returns wall seconds.  threads <= 0: all cores.
double oracle_render(void* h, uint32_t spp, uint64_t seed, uint32_t first_sample, float* rgba, uint64_t* counters_out, int threads) {
    Scene* s = static_cast<Scene*>(h);
    const int w = s->cam.width, ht = s->cam.height, cnt = w * ht;
    auto t0 = std::chrono::steady_clock::now();
    Counters total;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        Counters c;
#pragma omp for schedule(dynamic, 64)
        for (int i = 0; i < cnt; i++) {
            int x = i % w, y = i / w;
            for (uint32_t k = 0; k < spp; k++) {
                Rng rng; rng.pixel = uint32_t(i); rng.sample = first_sample + k; rng.seed = seed;
                Ray ray = cast_ray(*s, x, y, rng);
                F3 col = trace(*s, ray, rng, c);
                if (col.x != col.x) col.x = 0.f;                                             // Scene.cpp:16-18
                if (col.y != col.y) col.y = 0.f;
                if (col.z != col.z) col.z = 0.f;
                rgba[4 * i] += col.x; rgba[4 * i + 1] += col.y; rgba[4 * i + 2] += col.z; rgba[4 * i + 3] += 1.0f;
            }
        }
#pragma omp critical
        {
            total.paths += c.paths; total.rays_primary += c.rays_primary; total.rays_cont += c.rays_cont; total.rays_shadow += c.rays_shadow;
            total.box += c.box; total.tri += c.tri; total.self_tests += c.self_tests; total.self_hits += c.self_hits;
        }
    }
    if (counters_out) {
        counters_out[0] = total.paths; counters_out[1] = total.rays_primary; counters_out[2] = total.rays_cont; counters_out[3] = total.rays_shadow;
        counters_out[4] = total.box; counters_out[5] = total.tri; counters_out[6] = total.self_tests; counters_out[7] = total.self_hits;
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// Scene::getPixelsColor (Scene.cpp:23-33): mean -> clamp -> pow(0.5) -> *255.99 -> u8
void oracle_tonemap(const float* rgba, int n, unsigned char* rgb) {
    for (int i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) {
            float m = rgba[4 * i + c] / rgba[4 * i + 3];
            m = std::min(std::max(m, 0.f), 1.f);
            m = std::pow(m, 0.5f);
            rgb[3 * i + c] = (unsigned char)(m * 255.99f);
        }
}

void oracle_rng_block(uint32_t pixel, uint32_t sample, uint32_t block, uint64_t seed, float* out4) { counter_block(pixel, sample, block, seed, out4); }

// ---- SEQ-mode probes: same argument lists as oracle/ref/ref_driver.cpp -------------------------------
int oracle_cast_ray(void* h, int x, int y, const float* xi, int n, double* od6) {
    Rng r = seq_rng(xi, n); Ray ray = cast_ray(*static_cast<Scene*>(h), x, y, r);
    put3(od6, ray.start); put3(od6 + 3, ray.direction); return r.pos;
}
int oracle_aabb_intersect(const double* A, const double* B, const double* o, const double* d, double t1, double t2) {
    Ray r; r.start = P3(o); r.direction = P3(d); r.t1 = t1; r.t2 = t2; return aabb_intersect(P3(A), P3(B), r) ? 1 : 0;
}
static Scene* one_tri_scene(const double* v9, const double* vn9, const double* uv6, int emissive) {
    Scene* s = new Scene();
    Tex t; t.w = t.h = 1; t.texel.push_back({0.5f, 0.5f, 0.5f}); s->texs.push_back(t);
    s->mats.push_back({{0, 0, 0}, 1.0, {emissive ? 1.0 : 0.0, 0, 0}, 0});
    Tri T; for (int k = 0; k < 3; k++) { T.v[k] = P3(v9 + 3 * k); T.vn[k] = P3(vn9 + 3 * k); T.uv[k] = {uv6[2 * k], uv6[2 * k + 1]}; }
    T.mat = 0; T.A = vmin(T.v[0], vmin(T.v[1], T.v[2])); T.B = vmax(T.v[0], vmax(T.v[1], T.v[2]));
    s->tris.push_back(T); return s;
}
int oracle_tri_hit(const double* v9, const double* vn9, const double* uv6, int emissive, const double* o, const double* d, double t1, double t2, double* out13) {
    Scene* s = one_tri_scene(v9, vn9, uv6, emissive);
    Ray r; r.start = P3(o); r.direction = P3(d); r.t1 = t1; r.t2 = t2;
    Hit info; bool hit = tri_hit(*s, 0, r, info);
    out13[0] = info.t; put3(out13 + 1, info.point); put3(out13 + 4, info.normal); out13[7] = info.uv.x; out13[8] = info.uv.y;
    out13[9] = info.front ? 1.0 : 0.0; out13[10] = info.lightarea;
    delete s; return hit ? 1 : 0;
}
int oracle_tri_any(const double* v9, const double* o, const double* d, double t1, double t2) {
    double z9[9] = {0, 0, 1, 0, 0, 1, 0, 0, 1}, z6[6] = {0, 0, 0, 0, 0, 0};
    Scene* s = one_tri_scene(v9, z9, z6, 0);
    Ray r; r.start = P3(o); r.direction = P3(d); r.t1 = t1; r.t2 = t2;
    bool hit = tri_any(s->tris[0], r); delete s; return hit ? 1 : 0;
}
float oracle_tri_area(const double* v9) {
    double z9[9] = {0, 0, 1, 0, 0, 1, 0, 0, 1}, z6[6] = {0, 0, 0, 0, 0, 0};
    Scene* s = one_tri_scene(v9, z9, z6, 0); float a = tri_area(s->tris[0]); delete s; return a;
}
int oracle_bvh_hit(void* h, const double* o, const double* d, double t1, double t2, double* out12) {
    Scene* s = static_cast<Scene*>(h); Counters c;
    Ray r; r.start = P3(o); r.direction = P3(d); r.t1 = t1; r.t2 = t2;
    Hit info; bool hit = bvh_hit(*s, 0, r, info, c);
    out12[0] = info.t; put3(out12 + 1, info.point); put3(out12 + 4, info.normal); out12[7] = info.uv.x; out12[8] = info.uv.y;
    out12[9] = info.front ? 1.0 : 0.0; out12[10] = info.lightarea; out12[11] = hit ? double(info.tri) : -1.0;
    return hit ? 1 : 0;
}
int oracle_bvh_has_hit(void* h, const double* o, const double* d, double t1, double t2) {
    Scene* s = static_cast<Scene*>(h); Counters c;
    Ray r; r.start = P3(o); r.direction = P3(d); r.t1 = t1; r.t2 = t2;
    return bvh_any(*s, 0, r, c, -1) ? 1 : 0;
}
void oracle_bsdf_setup(const double* n3, const double* wi3, const double* kd, const double* ks, double ns, float* out18) {
    float kdf[3] = {float(kd[0]), float(kd[1]), float(kd[2])};
    Scene* s = probe_scene(kdf, ks, ns); Hit hit = probe_hit(n3, wi3);
    Bsdf b = make_bsdf(*s, hit);
    for (int i = 0; i < 18; i++) out18[i] = 0.f;
    out18[0] = float(b.n);
    for (int i = 0; i < b.n; i++) { out18[1 + i] = b.lobe[i].weight; put3f(out18 + 3 + 3 * i, b.lobe[i].reflect); }
    put3f(out18 + 9, b.onb.u); put3f(out18 + 12, b.onb.v); put3f(out18 + 15, b.onb.w);
    delete s;
}
void oracle_bsdf_eval(const double* n3, const double* wi3, const double* kd, const double* ks, double ns, const float* wo3, float* out4) {
    float kdf[3] = {float(kd[0]), float(kd[1]), float(kd[2])};
    Scene* s = probe_scene(kdf, ks, ns); Hit hit = probe_hit(n3, wi3);
    Bsdf b = make_bsdf(*s, hit); F3 wo{wo3[0], wo3[1], wo3[2]};
    put3f(out4, bsdf_fx(b, wo)); out4[3] = bsdf_pdf(b, wo);
    delete s;
}
int oracle_bsdf_sample(const double* n3, const double* wi3, const double* kd, const double* ks, double ns, const float* xi, int n, float* out8) {
    float kdf[3] = {float(kd[0]), float(kd[1]), float(kd[2])};
    Scene* s = probe_scene(kdf, ks, ns); Hit hit = probe_hit(n3, wi3);
    Bsdf b = make_bsdf(*s, hit); Rng r = seq_rng(xi, n);
    Scatter sc = bsdf_sample(b, r, 0);
    put3f(out8, sc.wo); put3f(out8 + 3, sc.f); out8[6] = sc.pdf; out8[7] = sc.mirror ? 1.f : 0.f;
    delete s; return r.pos;
}
float oracle_power_heuristic(float a, float b) { return power_heuristic(a, b); }
double oracle_clamp01(float d) { return clamp01(d); }
void oracle_texture_get_color(int w, int h, const float* rgb, double u, double v, float* out3) {
    Tex t; t.w = w; t.h = h; for (int i = 0; i < w * h; i++) t.texel.push_back({rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]});
    put3f(out3, tex_color(t, {u, v}));
}
int oracle_sample_light(void* h, const double* p3, const float* xi, int n, double* out14) {
    Scene* s = static_cast<Scene*>(h); Hit info; info.point = P3(p3); Rng r = seq_rng(xi, n);
    LightSample ls = sample_light_point(*s, info, r, 0, true);
    out14[0] = ls.wo.x; out14[1] = ls.wo.y; out14[2] = ls.wo.z; out14[3] = ls.f.x; out14[4] = ls.f.y; out14[5] = ls.f.z;
    out14[6] = ls.pdf; out14[7] = ls.ray.t2; put3(out14 + 8, ls.ray.start); put3(out14 + 11, ls.ray.direction);
    return r.pos;
}
int oracle_trace_path(void* h, const double* o, const double* d, const float* xi, int n, float* L3) {
    Scene* s = static_cast<Scene*>(h); Rng r = seq_rng(xi, n); Counters c;
    Ray ray; ray.start = P3(o); ray.direction = P3(d);
    uint32_t keep = s->integrator; s->integrator = MCPT_INTEGRATOR_MIS;
    put3f(L3, trace(*s, ray, r, c)); s->integrator = keep;
    return r.pos;
}
int oracle_trace_path_recursive(void* h, const double* o, const double* d, const float* xi, int n, float* L3) {
    Scene* s = static_cast<Scene*>(h); Rng r = seq_rng(xi, n); Counters c;
    Ray ray; ray.start = P3(o); ray.direction = P3(d);
    uint32_t keep = s->integrator; s->integrator = MCPT_INTEGRATOR_RECURSIVE_NEE;
    put3f(L3, trace(*s, ray, r, c)); s->integrator = keep;
    return r.pos;
}
int oracle_trace_pixel(void* h, int x, int y, const float* xi, int n, float* L3) {
    Scene* s = static_cast<Scene*>(h); Rng r = seq_rng(xi, n); Counters c;
    Ray ray = cast_ray(*s, x, y, r);
    put3f(L3, trace(*s, ray, r, c));
    return r.pos;
}
// COUNTER-mode single path (what mcpt_probe_paths computes on the device): key (seed, pixel=item, sample=0)
void oracle_trace_path_counter(void* h, const double* o, const double* d, uint32_t item, uint64_t seed, float* L3) {
    Scene* s = static_cast<Scene*>(h); Rng r; r.pixel = item; r.sample = 0; r.seed = seed; Counters c;
    Ray ray; ray.start = P3(o); ray.direction = P3(d);
    put3f(L3, trace(*s, ray, r, c));
}
int oracle_shadow_probe(void* h, const double* o, const double* d, const float* xi, int n, double shrink, int* out2) {
    Scene* s = static_cast<Scene*>(h); Counters c; Rng r = seq_rng(xi, n);
    Ray ray; ray.start = P3(o); ray.direction = P3(d);
    Hit info; if (!bvh_hit(*s, 0, ray, info, c)) return 0;
    LightSample ls = sample_light_point(*s, info, r, 0, true);
    if (ls.pdf == 0) return 0;
    out2[0] = bvh_any(*s, 0, ls.ray, c, -1) ? 1 : 0;
    Ray b = ls.ray; b.t2 = ls.ray.t2 * (1.0 - shrink); out2[1] = bvh_any(*s, 0, b, c, -1) ? 1 : 0;
    return 1;
}

}  // extern "C"
