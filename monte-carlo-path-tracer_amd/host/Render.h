// `Render` with the reference's surface (src/Render.h:51-69) on top of the C ABI (include/mcpt.h).  No HIP here.
#pragma once
#include <cstdint>
#include <vector>
#include "../../include/mcpt.h"
#include "Model.h"
#include "Scene.h"

class Render {
public:
    explicit Render(Model& m_model);                  // Render.cpp:5-10: flatten + BVH + upload (inside mcpt_create)
    Render(Model& m_model, const mcpt_opts& opts);
    ~Render();
    void render(Scene& scene);                        // Render.cpp:56-69: adds exactly ONE sample to every pixel of `scene`
    void render(Scene& scene, uint32_t spp);          // the same `spp` times, with one film read-back instead of `spp`
    bool ok() const { return ctx != nullptr; }
    mcpt_ctx* handle() { return ctx; }
    uint64_t seed = 20251004;                         // the reference seeds from random_device; here reproducible by default
private:
    mcpt_ctx* ctx = nullptr;
    uint32_t next_sample = 0;
    std::vector<float> film;
    void create(Model& m, const mcpt_opts& opts);
};
// Fills an mcpt_scene_desc that points INTO `m` (and into the two scratch vectors); valid while all three live.
void model_to_desc(Model& m, std::vector<mcpt_material>& mats, std::vector<mcpt_texture>& texs, mcpt_scene_desc& d);
