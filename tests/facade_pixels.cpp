// The reference's display loop (main.cpp:26-33) against the facade classes: render(scene); getPixelsColor(); every frame.  Checks, for
// tests/test_gpu_parity.py::test_facade_getPixelsColor_runs_on_the_device, that the tonemapped image handed out while the whole film is on
// the device (i) is there after every frame, (ii) equals the host path's image of the same film, and (iii) leaves the samples where they are.
//   facade_pixels scene.obj frames depth out_device.rgb out_host.rgb out_film.bin
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "Model.h"
#include "Render.h"
#include "Scene.h"

int main(int argc, char** argv) {
    if (argc < 7) return 2;
    Model model(argv[1], true);
    if (!model.ok) return 3;
    const int frames = std::atoi(argv[2]);
    mcpt_opts o; std::memset(&o, 0, sizeof o); o.struct_size = sizeof o; o.max_depth = uint32_t(std::atoi(argv[3])); o.flags = MCPT_FLAG_CORRECT_SHADOW_T2;
    const int w = model.camerainfo.width, h = model.camerainfo.height;
    const size_t n = size_t(w) * h;
    Scene scene(w, h);
    Render a(model, o); a.seed = 21;
    if (!a.ok()) return 4;
    std::vector<Color3b> dev(n);
    for (int f = 0; f < frames; f++) {
        a.render(scene);
        const Color3b* px = scene.getPixelsColor();                      // device path: nothing has been written into the Scene's host part
        if (!px) return 5;
        std::memcpy(dev.data(), px, n * 3);
    }
    { FILE* f = std::fopen(argv[4], "wb"); if (!f) return 7; std::fwrite(dev.data(), 3, n, f); std::fclose(f); }
    { FILE* f = std::fopen(argv[6], "wb"); if (!f) return 7; std::fwrite(scene.pixels(), sizeof(Pixels), n, f); std::fclose(f); }   // folds the device film in: all samples must be there
    const Color3b* px = scene.getPixelsColor();                          // host path now (m_Pixels is no longer empty)
    { FILE* f = std::fopen(argv[5], "wb"); if (!f) return 7; std::fwrite(px, 3, n, f); std::fclose(f); }
    a.render(scene);                                                     // one more sample on top of a host part: the reader must see the sum
    px = scene.getPixelsColor();
    std::printf("%d %d %.0f\n", w, h, scene.pixels()[0].spp);
    return 0;
}
