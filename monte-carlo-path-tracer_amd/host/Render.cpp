#include "Render.h"

#include <cstring>
#include <iostream>

void model_to_desc(Model& m, std::vector<mcpt_material>& mats, std::vector<mcpt_texture>& texs, mcpt_scene_desc& d) {
    std::memset(&d, 0, sizeof d);
    mats.resize(m.materials.size()); texs.resize(m.materials.size());
    for (size_t i = 0; i < m.materials.size(); i++) {
        const Material& s = m.materials[i];
        mcpt_material& o = mats[i]; std::memset(&o, 0, sizeof o);
        o.ks[0] = s.Ks.x; o.ks[1] = s.Ks.y; o.ks[2] = s.Ks.z; o.ns = s.Ns;
        o.radiance[0] = s.radiance.x; o.radiance[1] = s.radiance.y; o.radiance[2] = s.radiance.z;
        o.map_kd = int32_t(i);
        texs[i].width = s.Map_Kd->image_w; texs[i].height = s.Map_Kd->image_h;
        texs[i].rgb = reinterpret_cast<const float*>(s.Map_Kd->image_color.data());
    }
    static_assert(sizeof(dvec3) == 24 && sizeof(dvec2) == 16 && sizeof(imat3x4) == 48 && sizeof(Color3f) == 12, "Model arrays are passed through as-is");
    d.vertex = reinterpret_cast<const double*>(m.vertex.data()); d.n_vertex = uint32_t(m.vertex.size());
    d.normal = reinterpret_cast<const double*>(m.normal.data()); d.n_normal = uint32_t(m.normal.size());
    d.texcoord = reinterpret_cast<const double*>(m.texture.data()); d.n_texcoord = uint32_t(m.texture.size());
    d.face = reinterpret_cast<const int32_t*>(m.face.data()); d.n_face = uint32_t(m.face.size());
    d.materials = mats.data(); d.n_materials = uint32_t(mats.size());
    d.textures = texs.data(); d.n_textures = uint32_t(texs.size());
    const CameraInfo& c = m.camerainfo;
    d.camera.eye[0] = c.eye.x; d.camera.eye[1] = c.eye.y; d.camera.eye[2] = c.eye.z;
    d.camera.lookat[0] = c.lookat.x; d.camera.lookat[1] = c.lookat.y; d.camera.lookat[2] = c.lookat.z;
    d.camera.up[0] = c.up.x; d.camera.up[1] = c.up.y; d.camera.up[2] = c.up.z;
    d.camera.fovy = c.fovy; d.camera.width = c.width; d.camera.height = c.height;
}

void Render::create(Model& m, const mcpt_opts& opts) {
    std::vector<mcpt_material> mats; std::vector<mcpt_texture> texs; mcpt_scene_desc d;
    model_to_desc(m, mats, texs, d);
    if (mcpt_create(&d, &opts, &ctx) != MCPT_OK) { std::cerr << "Error: mcpt_create: " << mcpt_last_error() << std::endl; ctx = nullptr; return; }
    film.resize(size_t(d.camera.width) * d.camera.height * 4);
}
Render::Render(Model& m) { mcpt_opts o; std::memset(&o, 0, sizeof o); o.struct_size = sizeof o; create(m, o); }
Render::Render(Model& m, const mcpt_opts& opts) { create(m, opts); }
Render::Render(Render& other, int device) {
    seed = other.seed;
    if (!other.ctx || mcpt_clone_to_device(other.ctx, device, &ctx) != MCPT_OK) { std::cerr << "Error: mcpt_clone_to_device: " << mcpt_last_error() << std::endl; ctx = nullptr; return; }
    film.resize(other.film.size());
}
Render::~Render() {
    if (target) { flush_into(*target); target->detach(this); }
    if (ctx) mcpt_destroy(ctx);
}

void Render::render(Scene& scene) { render(scene, 1); }
void Render::render(Scene& scene, uint32_t spp) {
    if (!ctx || spp == 0) return;
    if (scene.width() * scene.height() * 4 != int(film.size())) { std::cerr << "Error: Render::render: the Scene's size differs from the camera's" << std::endl; return; }
    // the film lives in `scene` (several Renders may share one Scene); this Render's share of it stays on the device until read
    if (target != &scene) {
        if (target) { flush_into(*target); target->detach(this); }
        target = &scene;
    }
    scene.attach(this);                               // (flushes whichever other Render held samples for `scene`)
    if (mcpt_render(ctx, spp, seed, next_sample) != MCPT_OK) { std::cerr << "Error: mcpt_render: " << mcpt_last_error() << std::endl; return; }
    next_sample += spp; dirty = true;
}
void Render::flush_into(Scene& scene) {
    if (!ctx || !dirty || &scene != target) return;
    dirty = false;
    if (mcpt_read_accum(ctx, film.data()) != MCPT_OK || mcpt_clear_accum(ctx) != MCPT_OK) { std::cerr << "Error: film read-back: " << mcpt_last_error() << std::endl; return; }
    scene.add_film(film.data());
}
const Color3b* Render::tonemapped(Scene& scene) {
    static_assert(sizeof(Color3b) == 3, "mcpt_tonemap_map's image is read as Color3b[]");
    if (!ctx || !dirty || &scene != target) return nullptr;          // (nothing held: the host path shows what an empty film shows)
    const uint8_t* px = nullptr;
    if (mcpt_tonemap_map(ctx, 0, &px) != MCPT_OK) { std::cerr << "Error: mcpt_tonemap_map: " << mcpt_last_error() << std::endl; return nullptr; }
    return reinterpret_cast<const Color3b*>(px);
}
void Render::displaced(Scene& scene) { if (&scene == target) target = nullptr; }   // (already flushed by Scene::attach)
void Render::scene_gone(Scene& scene) {
    if (&scene != target) return;
    target = nullptr;
    if (ctx && dirty && mcpt_clear_accum(ctx) != MCPT_OK) std::cerr << "Error: mcpt_clear_accum: " << mcpt_last_error() << std::endl;
    dirty = false;
}
