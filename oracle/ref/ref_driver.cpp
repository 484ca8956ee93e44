// TEST INFRASTRUCTURE ONLY.  A C-ABI driver linked against the reference's OWN object files
// (compiled by oracle/build_ref.sh from /root/reference/src, which is never copied into this
// repo).  It only *calls* reference functions so that tests can (a) generate golden vectors,
// (b) pin oracle/mcpt_oracle.cpp to the real thing, and (c) time the reference CPU path as
// bench.py's cpu_baseline (kind "reference").  Compiled with -fno-access-control because the
// reference keeps cast_Ray / ray_tracing / sample / m_Pixels private (Render.h:56-68, Scene.h:23-27).
#include "model.h"
#include "Render.h"
#include "BVH.h"
#include "BSDF.h"
#include "Scene.h"
#include "Triangle.h"
#include <unordered_map>
#include <cstring>
#include <climits>
#include <chrono>

namespace mcpt_refshim {
rng_control g_rng;
int g_max_bounces = INT_MAX;
}

namespace {
std::unique_ptr<Model> g_model;
std::unique_ptr<Render> g_render;
std::unique_ptr<Scene> g_scene;
std::unordered_map<const Material*, int> g_tri_of_mtl;   // each Triangle owns its own Material copy (Render.cpp:34)
std::vector<uint32_t> g_queue;

dvec3 D3(const double* p) { return dvec3(p[0], p[1], p[2]); }
void put3(double* o, const dvec3& v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
void put3f(float* o, const vec3& v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

std::shared_ptr<Material> make_material(const double* kd, const double* ks, double ns, const double* radiance) {
    auto m = std::make_shared<Material>();
    m->Ks = D3(ks);
    m->Ns = ns;
    m->radiance = D3(radiance);
    m->Map_Kd = std::make_shared<Texture>(Color3f(float(kd[0]), float(kd[1]), float(kd[2])));
    return m;
}
}  // namespace

extern "C" {

// ---------------------------------------------------------------- RNG control (ref_shim.h)
void ref_rng_mode(int mode) { mcpt_refshim::g_rng.mode = mode; }
// xi[i] in [0,1) as float; each becomes the pair (lo=0, hi=xi*2^32) that libstdc++'s
// generate_canonical<double,53> folds back to exactly xi (utils.h:23-28 then casts to float).
void ref_rng_inject(const float* xi, int n) {
    g_queue.resize(size_t(n) * 2);
    for (int i = 0; i < n; i++) {
        g_queue[2 * i] = 0u;
        g_queue[2 * i + 1] = uint32_t(double(xi[i]) * 4294967296.0);
    }
    auto& c = mcpt_refshim::g_rng;
    c.queue = g_queue.data(); c.n = long(g_queue.size()); c.pos = 0; c.underflow = 0;
}
int ref_rng_consumed() { return int(mcpt_refshim::g_rng.pos / 2); }
int ref_rng_underflow() { return int(mcpt_refshim::g_rng.underflow); }
void ref_set_max_bounces(int n) { mcpt_refshim::g_max_bounces = n <= 0 ? INT_MAX : n; }
float ref_rand1f() { return rand1f(); }

// ---------------------------------------------------------------- scene (main.cpp:13-17)
// ---------------------------------------------------------------- the loader alone (model.cpp:44-281): what Model(filename) parsed
// A Model that is never handed to Render, so files that exercise the parser's quirks (a material without Kd leaves Map_Kd null,
// model.h:38) can be inspected without running into the renderer's own undefined behaviour.
std::unique_ptr<Model> g_parsed;
int ref_model_load(const char* obj_path) { g_parsed.reset(new Model(obj_path)); return 0; }
void ref_model_counts(int* out7) {
    out7[0] = int(g_parsed->vertex.size()); out7[1] = int(g_parsed->normal.size()); out7[2] = int(g_parsed->texture.size());
    out7[3] = int(g_parsed->face.size()); out7[4] = int(g_parsed->materials.size());
    out7[5] = g_parsed->camerainfo.width; out7[6] = g_parsed->camerainfo.height;
}
void ref_model_arrays(double* v3, double* vn3, double* vt2, int* f12) {
    for (size_t i = 0; i < g_parsed->vertex.size(); i++) put3(v3 + 3 * i, g_parsed->vertex[i]);
    for (size_t i = 0; i < g_parsed->normal.size(); i++) put3(vn3 + 3 * i, g_parsed->normal[i]);
    for (size_t i = 0; i < g_parsed->texture.size(); i++) { vt2[2 * i] = g_parsed->texture[i].x; vt2[2 * i + 1] = g_parsed->texture[i].y; }
    for (size_t i = 0; i < g_parsed->face.size(); i++)
        for (int c = 0; c < 3; c++) for (int k = 0; k < 4; k++) f12[12 * i + 4 * c + k] = g_parsed->face[i][c][k];
}
// per material: Ks[3], Tr[3], Ns, Ni, radiance[3], number of texels of Map_Kd (0 = Map_Kd is null), image_w, image_h
void ref_model_material(int i, double* out14) {
    const Material& m = g_parsed->materials[size_t(i)];
    put3(out14, m.Ks); put3(out14 + 3, m.Tr); out14[6] = m.Ns; out14[7] = m.Ni; put3(out14 + 8, m.radiance);
    out14[11] = m.Map_Kd ? double(m.Map_Kd->image_color.size()) : 0.0;
    out14[12] = (m.Map_Kd && m.Map_Kd->image_color.size() > 1) ? double(m.Map_Kd->image_w) : 1.0;
    out14[13] = (m.Map_Kd && m.Map_Kd->image_color.size() > 1) ? double(m.Map_Kd->image_h) : 1.0;
}
void ref_model_texels(int i, float* rgb) {
    const Material& m = g_parsed->materials[size_t(i)];
    if (!m.Map_Kd) return;
    for (size_t k = 0; k < m.Map_Kd->image_color.size(); k++) put3f(rgb + 3 * k, m.Map_Kd->image_color[k]);
}
void ref_model_camera(double* out10) {
    const CameraInfo& c = g_parsed->camerainfo;
    put3(out10, c.eye); put3(out10 + 3, c.lookat); put3(out10 + 6, c.up); out10[9] = c.fovy;
}

int ref_load(const char* obj_path) {
    g_model.reset(new Model(obj_path));
    if (g_model->face.empty()) return -1;
    g_scene.reset(new Scene(g_model->camerainfo.width, g_model->camerainfo.height));
    g_render.reset(new Render(*g_model));
    g_tri_of_mtl.clear();
    for (size_t i = 0; i < g_render->triangles.size(); i++)
        g_tri_of_mtl[g_render->triangles[i]->mtl.get()] = int(i);
    return 0;
}
int ref_width() { return g_render->camera.w; }
int ref_height() { return g_render->camera.h; }
int ref_num_tris() { return int(g_render->triangles.size()); }
int ref_num_lights() { return int(g_render->lights.size()); }
void ref_set_resolution(int w, int h) {   // camera XML override; Scene re-created to match
    g_render->camera.w = w; g_render->camera.h = h;
    g_scene.reset(new Scene(w, h));
}
int ref_light_tri(int i) { return g_tri_of_mtl[g_render->lights[i]->mtl.get()]; }

// flattened triangle i as Render::tranform_triangle built it (Render.cpp:12-44)
void ref_get_triangle(int i, double* v9, double* vn9, double* uv6, double* radiance3, double* ks3, double* ns, float* kd3) {
    auto& t = *g_render->triangles[i];
    for (int k = 0; k < 3; k++) { put3(v9 + 3 * k, t.v[k]); put3(vn9 + 3 * k, t.vn[k]); uv6[2 * k] = t.uv[k].x; uv6[2 * k + 1] = t.uv[k].y; }
    put3(radiance3, t.mtl->radiance); put3(ks3, t.mtl->Ks); *ns = t.mtl->Ns;
    put3f(kd3, t.mtl->Map_Kd->image_color[0]);
}

// ---------------------------------------------------------------- film (Scene.cpp)
void ref_clear() { g_scene.reset(new Scene(g_render->camera.w, g_render->camera.h)); }
double ref_render(int frames) {   // Render::render x frames (Render.cpp:56-69); returns wall seconds
    auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; f++) g_render->render(*g_scene);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
void ref_get_accum(float* rgba) {   // Pixels{vec3 color; float spp} (Scene.h:7-12)
    std::memcpy(rgba, g_scene->m_Pixels.get(), sizeof(float) * 4 * size_t(g_scene->w) * g_scene->h);
}
void ref_set_pixel(int x, int y, float r, float g, float b) { Color3f c(r, g, b); g_scene->set_Pixel({x, y}, c); }
void ref_get_pixels_u8(unsigned char* rgb) {   // Scene::getPixelsColor (Scene.cpp:23-33)
    const Color3b* p = g_scene->getPixelsColor();
    std::memcpy(rgb, p, 3 * size_t(g_scene->w) * g_scene->h);
}

// ---------------------------------------------------------------- function-level KATs
void ref_cast_ray(int x, int y, double* od6) {   // Render::cast_Ray (Render.cpp:71-80); consumes 2 xi
    Ray r = g_render->cast_Ray(x, y);
    put3(od6, r.start); put3(od6 + 3, r.direction);
}
int ref_aabb_intersect(const double* A, const double* B, const double* o, const double* d, double t1, double t2) {
    AABB box(D3(A), D3(B)); Ray r(D3(o), D3(d)); r.t1 = t1; r.t2 = t2;
    return box.Intersection(r) ? 1 : 0;   // AABB.cpp:25-36
}
static Triangle make_tri(const double* v9, const double* vn9, const double* uv6, std::shared_ptr<Material> m) {
    Triangle t;
    for (int k = 0; k < 3; k++) { t.v[k] = D3(v9 + 3 * k); t.vn[k] = D3(vn9 + 3 * k); t.uv[k] = dvec2(uv6[2 * k], uv6[2 * k + 1]); }
    t.A = glm::min(t.v[0], glm::min(t.v[1], t.v[2])); t.B = glm::max(t.v[0], glm::max(t.v[1], t.v[2]));
    t.mtl = m;
    return t;
}
// Triangle::hit (Triangle.cpp:48-80).  out13 = t, point3, normal3, uv2, front, lightarea, (2 spare)
int ref_tri_hit(const double* v9, const double* vn9, const double* uv6, int emissive,
                const double* o, const double* d, double t1, double t2, double* out13) {
    double kd[3] = {0.5, 0.5, 0.5}, ks[3] = {0, 0, 0}, rad[3] = {emissive ? 1.0 : 0.0, 0, 0};
    Triangle t = make_tri(v9, vn9, uv6, make_material(kd, ks, 1.0, rad));
    Ray r(D3(o), D3(d)); r.t1 = t1; r.t2 = t2;
    hitInfo info;
    bool h = t.hit(r, info);
    out13[0] = info.t; put3(out13 + 1, info.point); put3(out13 + 4, info.normal);
    out13[7] = info.uv.x; out13[8] = info.uv.y; out13[9] = info.front ? 1.0 : 0.0; out13[10] = info.lightarea;
    return h ? 1 : 0;
}
int ref_tri_any(const double* v9, const double* o, const double* d, double t1, double t2) {   // Triangle.cpp:83-106
    double z9[9] = {0, 0, 1, 0, 0, 1, 0, 0, 1}, z6[6] = {0, 0, 0, 0, 0, 0}, kd[3] = {0.5, 0.5, 0.5}, ks[3] = {0, 0, 0}, rad[3] = {0, 0, 0};
    Triangle t = make_tri(v9, z9, z6, make_material(kd, ks, 1.0, rad));
    Ray r(D3(o), D3(d)); r.t1 = t1; r.t2 = t2;
    return t.isIntersect(r) ? 1 : 0;
}
float ref_tri_area(const double* v9) {   // Triangle.cpp:24-28
    double z9[9] = {0, 0, 1, 0, 0, 1, 0, 0, 1}, z6[6] = {0, 0, 0, 0, 0, 0};
    Triangle t = make_tri(v9, z9, z6, nullptr);
    return t.area();
}
// BVH::hit (BVH.cpp:90-113) on the loaded scene.  out12 = t, point3, normal3, uv2, front, lightarea, tri index
int ref_bvh_hit(const double* o, const double* d, double t1, double t2, double* out12) {
    Ray r(D3(o), D3(d)); r.t1 = t1; r.t2 = t2;
    hitInfo info;
    bool h = g_render->bvh->hit(r, info);
    out12[0] = info.t; put3(out12 + 1, info.point); put3(out12 + 4, info.normal);
    out12[7] = info.uv.x; out12[8] = info.uv.y; out12[9] = info.front ? 1.0 : 0.0; out12[10] = info.lightarea;
    out12[11] = h ? double(g_tri_of_mtl[info.mtl.get()]) : -1.0;
    return h ? 1 : 0;
}
int ref_bvh_has_hit(const double* o, const double* d, double t1, double t2) {   // BVH.cpp:115-136
    Ray r(D3(o), D3(d)); r.t1 = t1; r.t2 = t2;
    return g_render->bvh->has_hit(r) ? 1 : 0;
}
static void bvh_walk(BVH_node* n, int depth, long* nodes, long* leaves, long* maxdepth, long* maxleaf) {
    if (!n) return;
    (*nodes)++;
    if (depth > *maxdepth) *maxdepth = depth;
    if (!n->left && !n->right) { (*leaves)++; if (long(n->contain_tri.size()) > *maxleaf) *maxleaf = long(n->contain_tri.size()); }
    bvh_walk(n->left, depth + 1, nodes, leaves, maxdepth, maxleaf);
    bvh_walk(n->right, depth + 1, nodes, leaves, maxdepth, maxleaf);
}
void ref_bvh_stats(long* out4) { out4[0] = out4[1] = out4[2] = out4[3] = 0; bvh_walk(g_render->bvh->root, 0, out4, out4 + 1, out4 + 2, out4 + 3); }

// BSDF on a synthetic hit (BSDF.cpp:87-202).  Constant-colour Kd texture.
static hitInfo make_hit(const double* n3, const double* wi3, const double* kd, const double* ks, double ns) {
    double rad[3] = {0, 0, 0};
    hitInfo info; info.normal = D3(n3); info.wi = D3(wi3); info.uv = dvec2(0.25, 0.75); info.mtl = make_material(kd, ks, ns, rad);
    return info;
}
// out = nlobes, weight[0..1], reflect[0..1] rgb (after energy rescale), onb u v w
void ref_bsdf_setup(const double* n3, const double* wi3, const double* kd, const double* ks, double ns, float* out18) {
    hitInfo info = make_hit(n3, wi3, kd, ks, ns);
    BSDF b(info);
    for (int i = 0; i < 18; i++) out18[i] = 0.f;
    out18[0] = float(b.bxdfs.size());
    for (size_t i = 0; i < b.bxdfs.size() && i < 2; i++) { out18[1 + i] = b.bxdfs[i]->weight; put3f(out18 + 3 + 3 * i, b.bxdfs[i]->reflect); }
    put3f(out18 + 9, b.onb.u); put3f(out18 + 12, b.onb.v); put3f(out18 + 15, b.onb.w);
}
void ref_bsdf_eval(const double* n3, const double* wi3, const double* kd, const double* ks, double ns, const float* wo3, float* out4) {
    hitInfo info = make_hit(n3, wi3, kd, ks, ns);
    BSDF b(info);
    vec3 wo(wo3[0], wo3[1], wo3[2]);
    put3f(out4, b.Fx(wo)); out4[3] = b.Pdf(wo);   // BSDF.cpp:112-121, 153-163
}
// BSDF::Sample (BSDF.cpp:123-151); consumes 1 + {0,2} xi.  out8 = wo3, f3, pdf, isMirror
void ref_bsdf_sample(const double* n3, const double* wi3, const double* kd, const double* ks, double ns, float* out8) {
    hitInfo info = make_hit(n3, wi3, kd, ks, ns);
    BSDF b(info);
    Scatterinfo s = b.Sample();
    put3f(out8, s.wo); put3f(out8 + 3, s.f); out8[6] = s.pdf; out8[7] = s.isMirrorReflect ? 1.f : 0.f;
}
float ref_power_heuristic(float a, float b) { return power_heuristic(a, b); }   // utils.h:56-60
double ref_clamp01(float d) { return clamp01(d); }                               // utils.h:30-34
// Texture::get_color (model.cpp:30-41) on a w*h rgb float image
void ref_texture_get_color(int w, int h, const float* rgb, double u, double v, float* out3) {
    Texture t(Color3f(0.f));
    t.image_color.clear();
    for (int i = 0; i < w * h; i++) t.image_color.push_back(Color3f(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]));
    t.image_w = w; t.image_h = h;
    put3f(out3, t.get_color(dvec2(u, v)));
}
// Render::sample on the loaded scene (Render.cpp:202-223); consumes 3 xi.  out = wo3, f3, pdf, t2, start3, dir3
void ref_sample_light(const double* p3, double* out14) {
    hitInfo info; info.point = D3(p3);
    lightinfo li = g_render->sample(info);
    out14[0] = li.wo.x; out14[1] = li.wo.y; out14[2] = li.wo.z; out14[3] = li.f.x; out14[4] = li.f.y; out14[5] = li.f.z;
    out14[6] = li.pdf; out14[7] = li.ray.t2; put3(out14 + 8, li.ray.start); put3(out14 + 11, li.ray.direction);
}

// ---------------------------------------------------------------- path-level KATs
// Render::ray_tracing(Ray&) (Render.cpp:111-175), the shipping iterative MIS integrator.
void ref_trace_path(const double* o, const double* d, float* L3) {
    Ray r(D3(o), D3(d));
    put3f(L3, g_render->ray_tracing(r));
}
// Render::ray_tracing(Ray&,int) (Render.cpp:83-109), the dead recursive NEE integrator.
void ref_trace_path_recursive(const double* o, const double* d, float* L3) {
    Ray r(D3(o), D3(d));
    put3f(L3, g_render->ray_tracing(r, 0));
}
// cast_Ray + ray_tracing for one pixel = one iteration of Render::render's loop body (Render.cpp:62-66)
void ref_trace_pixel(int x, int y, float* L3) {
    Ray r = g_render->cast_Ray(x, y);
    put3f(L3, g_render->ray_tracing(r));
}
// Render::sample_light (Render.cpp:177-200) for a hit found by tracing (o,d); consumes 3 xi. returns 0 on miss.
int ref_sample_light_recursive(const double* o, const double* d, float* L3) {
    Ray r(D3(o), D3(d));
    hitInfo info;
    if (!g_render->bvh->hit(r, info)) return 0;
    put3f(L3, g_render->sample_light(info));
    return 1;
}
// Light self-occlusion probe (SURVEY A-9): from hit point of (o,d), draw a light sample (3 xi) and report
// has_hit of the shadow ray with the reference's inclusive t2 and with t2 shortened by `shrink`.
int ref_shadow_probe(const double* o, const double* d, double shrink, int* out2) {
    Ray r(D3(o), D3(d));
    hitInfo info;
    if (!g_render->bvh->hit(r, info)) return 0;
    lightinfo li = g_render->sample(info);
    if (li.pdf == 0) return 0;
    Ray a = li.ray; out2[0] = g_render->bvh->has_hit(a) ? 1 : 0;
    Ray b = li.ray; b.t2 = li.ray.t2 * (1.0 - shrink); out2[1] = g_render->bvh->has_hit(b) ? 1 : 0;
    return 1;
}

}  // extern "C"
