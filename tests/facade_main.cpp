// The reference's own usage pattern (main.cpp:13-17,28-30,37) against the facade classes of host/: Model -> Scene -> Render,
// render(scene) once per sample, then read the film.  Built and run by tests/test_gpu_parity.py::test_facade_classes_*.
//   facade_main scene.obj frames depth out.bin
// Writes w*h {sum r, g, b, count} floats.  Sample bookkeeping the test reproduces through the C ABI:
//   render A (seed 11): samples 0 .. frames-1 into `scene`, sample `frames` into a short-lived Scene, sample frames+1 into `scene`
//   render B (seed 12): samples 0, 1 into `scene` in one call (two Renders share one Scene), destroyed before the film is read
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include "Model.h"
#include "Render.h"
#include "Scene.h"

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    Model model(argv[1], true);
    if (!model.ok) return 3;
    const int frames = std::atoi(argv[2]);
    mcpt_opts o; std::memset(&o, 0, sizeof o); o.struct_size = sizeof o; o.max_depth = uint32_t(std::atoi(argv[3])); o.flags = MCPT_FLAG_CORRECT_SHADOW_T2;
    const int w = model.camerainfo.width, h = model.camerainfo.height;
    Scene scene(w, h);
    Render a(model, o); a.seed = 11;
    if (!a.ok()) return 4;
    for (int f = 0; f < frames; f++) a.render(scene);                    // nothing is read back here
    { Scene brief(w, h); a.render(brief); if (brief.pixels()[0].spp != 1.f) return 5; a.render(brief); }   // dies holding an unread sample of `a`
    {
        auto b = std::make_unique<Render>(model, o); b->seed = 12;
        if (!b->ok()) return 4;
        b->render(scene, 2);                                             // takes the Scene over: a's samples are folded in first
    }                                                                    // b dies: its two samples are folded in
    a.render(scene);                                                     // sample index frames + 2 (brief consumed two)
    Color3f c{0.25f, 0.5f, 0.75f};
    scene.set_Pixel(Point2i{0, 0}, c);                                   // host-side write while samples are pending on the device
    const Color3b* px = scene.getPixelsColor();                          // reader: folds the device film in
    if (!px) return 6;
    FILE* f = std::fopen(argv[4], "wb");
    if (!f) return 7;
    std::fwrite(scene.pixels(), sizeof(Pixels), size_t(w) * h, f);
    std::fclose(f);
    std::printf("%d %d %u %u %u\n", w, h, unsigned(px[0].x), unsigned(px[0].y), unsigned(px[0].z));
    return 0;
}
