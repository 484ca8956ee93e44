// Developer tool / measurement (GPU box): the reference's own display loop (main.cpp:26-33) through the facade classes of host/ --
//     while (...) { render.render(scene); px = scene.getPixelsColor(); ... }
// one sample per pixel per call, the tonemapped film read after EVERY call -- timed per frame.
//   frame_loop scene.obj frames depth [warmup]
// Prints one JSON line: mean / median / p95 ms per frame of the pair, of render() alone and of getPixelsColor() alone, and a checksum of the
// last image (so that the work cannot be skipped).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "Model.h"
#include "Render.h"
#include "Scene.h"

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: frame_loop scene.obj frames depth [warmup]\n"); return 2; }
    Model model(argv[1], true);
    if (!model.ok) return 3;
    const int frames = std::atoi(argv[2]), warm = argc > 4 ? std::atoi(argv[4]) : 20;
    mcpt_opts o; std::memset(&o, 0, sizeof o); o.struct_size = sizeof o; o.max_depth = uint32_t(std::atoi(argv[3]));
    const int w = model.camerainfo.width, h = model.camerainfo.height;
    Scene scene(w, h);
    Render render(model, o);
    if (!render.ok()) return 4;
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::vector<double> pair, t_render, t_pixels;
    const Color3b* px = nullptr;
    for (int f = 0; f < warm + frames; f++) {
        const auto t0 = clk::now();
        render.render(scene);                                            // main.cpp:29
        const auto t1 = clk::now();
        px = scene.getPixelsColor();                                     // main.cpp:31
        const auto t2 = clk::now();
        if (!px) return 5;
        if (f >= warm) { pair.push_back(ms(t0, t2)); t_render.push_back(ms(t0, t1)); t_pixels.push_back(ms(t1, t2)); }
    }
    unsigned long long sum = 0;
    for (size_t i = 0; i < size_t(w) * h; i++) sum += px[i].x + 3u * px[i].y + 7u * px[i].z;
    auto stat = [](std::vector<double> v, double& mean, double& med, double& p95) {
        std::sort(v.begin(), v.end()); mean = 0; for (double x : v) mean += x; mean /= double(v.size()); med = v[v.size() / 2]; p95 = v[size_t(double(v.size()) * 0.95)];
    };
    double m0, d0, p0, m1, d1, p1, m2, d2, p2;
    stat(pair, m0, d0, p0); stat(t_render, m1, d1, p1); stat(t_pixels, m2, d2, p2);
    const float spp = scene.pixels()[size_t(w) * h / 2].spp;             // (folds the device film in: every frame's sample must be there)
    std::printf("{\"loop\": \"render(scene); getPixelsColor();\", \"width\": %d, \"height\": %d, \"depth\": %u, \"frames\": %d, \"ms_per_frame_mean\": %.4f, \"ms_per_frame_median\": %.4f, "
                "\"ms_per_frame_p95\": %.4f, \"render_ms_mean\": %.4f, \"getPixelsColor_ms_mean\": %.4f, \"getPixelsColor_ms_median\": %.4f, \"samples_in_film\": %.0f, \"image_checksum\": %llu}\n",
                w, h, o.max_depth, frames, m0, d0, p0, m1, m2, d2, spp, sum);
    return spp == float(warm + frames) ? 0 : 6;
}
