// `Render` with the reference's surface (src/Render.h:51-69) on top of the C ABI (include/mcpt.h).  No HIP here.
#pragma once
#include <cstdint>
#include <vector>
#include "../../include/mcpt.h"
#include "Model.h"
#include "Scene.h"

class Render : public FilmSource {
public:
    explicit Render(Model& m_model);                  // Render.cpp:5-10: flatten + BVH + upload (inside mcpt_create)
    Render(Model& m_model, const mcpt_opts& opts);
    Render(Render& same_scene, int device);           // the same scene on another GPU: copied device to device, nothing is built again
    ~Render();
    void render(Scene& scene);                        // Render.cpp:56-69: adds exactly ONE sample to every pixel of `scene`
    void render(Scene& scene, uint32_t spp);          // the same `spp` times in one call
    // The samples stay in HBM (the Scene is told: Scene::attach) until the Scene is read -- getPixelsColor, save_image, pixels(),
    // Scene::sync -- or rendered into by another Render, or either object goes away.
    void flush_into(Scene& scene) override;
    void scene_gone(Scene& scene) override;
    void displaced(Scene& scene) override;
    const Color3b* tonemapped(Scene& scene) override;
    Render(const Render&) = delete;
    Render& operator=(const Render&) = delete;
    bool ok() const { return ctx != nullptr; }
    mcpt_ctx* handle() { return ctx; }
    uint64_t seed = 20251004;                         // the reference seeds from random_device; here reproducible by default
private:
    mcpt_ctx* ctx = nullptr;
    uint32_t next_sample = 0;
    std::vector<float> film;
    Scene* target = nullptr;                          // the Scene the device film belongs to
    bool dirty = false;                               // the device film holds samples `target` has not seen
    void create(Model& m, const mcpt_opts& opts);
};
// Fills an mcpt_scene_desc that points INTO `m` (and into the two scratch vectors); valid while all three live.
void model_to_desc(Model& m, std::vector<mcpt_material>& mats, std::vector<mcpt_texture>& texs, mcpt_scene_desc& d);
