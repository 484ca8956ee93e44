// mcpt_cli -- headless replacement for the reference's GLFW shell (src/main.cpp:4-39): load a scene, render N frames
// (= spp), print the reference's per-frame line, save <name><frames>.png.  `--gpus N` shards the sample range over N
// devices of this node from ONE process and sums the films with RCCL (ncclAllReduce over xGMI).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "Render.h"

static void usage() {
    std::cout << "usage: mcpt_cli scene.obj [--spp N] [--batch B] [--depth D] [--gpus G] [--shard samples|tiles] [--out prefix] [--seed S] [--recursive] [--corrected]\n"
                 "                          [--deterministic] [--ref-index-order] [--ref-tie-order] [--gpu-bvh] [--check] [--dump-model file] [--save-every K]\n"
                 "       mcpt_cli --decode-image texture.(png|jpg|ppm|bmp|tga|hdr) out.(ppm|pfm)\n";
}

int main(int argc, char** argv) {
    if (argc < 2) { usage(); return 2; }
    if (std::string(argv[1]) == "--decode-image") {              // host-only helper: texture file -> binary PPM (what map_Kd textures decode to)
        if (argc != 4) { usage(); return 2; }
        int w = 0, h = 0; std::vector<unsigned char> rgb;
        {   // a Radiance .hdr decodes to linear floats: written as a binary PFM-like dump ("PF\nw h\n-1.0\n" + w*h*3 little-endian floats, top row first)
            std::vector<float> lin;
            if (load_image_hdr(argv[2], w, h, lin)) {
                FILE* f = std::fopen(argv[3], "wb");
                if (!f) return 1;
                std::fprintf(f, "PF\n%d %d\n-1.0\n", w, h); std::fwrite(lin.data(), 4, lin.size(), f); std::fclose(f);
                return 0;
            }
        }
        if (!load_image_rgb8(argv[2], w, h, rgb)) { std::cerr << "Error: cannot decode " << argv[2] << std::endl; return 1; }
        FILE* f = std::fopen(argv[3], "wb");
        if (!f) return 1;
        std::fprintf(f, "P6\n%d %d\n255\n", w, h); std::fwrite(rgb.data(), 1, rgb.size(), f); std::fclose(f);
        return 0;
    }
    std::string filename = argv[1], out, dump_model;
    uint32_t spp = 64, batch = 0, depth = 0, gpus = 1, save_every = 0; uint64_t seed = 20251004; uint32_t flags = 0, integrator = 0; bool ref_order = false, check_only = false, shard_tiles = false;
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i]; auto next = [&]() { return i + 1 < argc ? argv[++i] : (char*)"0"; };
        if (a == "--spp") spp = uint32_t(std::atoi(next())); else if (a == "--batch") batch = uint32_t(std::atoi(next()));
        else if (a == "--depth") depth = uint32_t(std::atoi(next())); else if (a == "--gpus") gpus = uint32_t(std::atoi(next()));
        else if (a == "--out") out = next(); else if (a == "--seed") seed = std::strtoull(next(), nullptr, 10);
        else if (a == "--recursive") integrator = MCPT_INTEGRATOR_RECURSIVE_NEE; else if (a == "--corrected") flags |= MCPT_FLAG_CORRECT_SHADOW_T2;
        else if (a == "--deterministic") flags |= MCPT_FLAG_DETERMINISTIC; else if (a == "--ref-index-order") ref_order = true;
        else if (a == "--gpu-bvh") flags |= MCPT_FLAG_GPU_BVH_BUILD;
        else if (a == "--ref-tie-order") flags |= MCPT_FLAG_REFERENCE_TIE_ORDER;
        else if (a == "--check") check_only = true;
        else if (a == "--dump-model") dump_model = next();
        else if (a == "--save-every") save_every = uint32_t(std::atoi(next()));
        else if (a == "--shard") shard_tiles = std::string(next()) == "tiles";
        else { usage(); return 2; }
    }
    Model model(filename, ref_order);
    if (!model.ok) { std::cerr << "Error: scene did not load" << std::endl; return 1; }
    std::cout << model.face.size() << " " << model.normal.size() << " " << model.vertex.size() << std::endl;   // main.cpp:14
    if (!dump_model.empty()) {   // host-only: everything Model(filename) parsed, as text (tests compare it with the reference's own parse)
        FILE* f = std::fopen(dump_model.c_str(), "w");
        if (!f) { std::cerr << "Error: cannot write " << dump_model << std::endl; return 1; }
        std::fprintf(f, "counts %zu %zu %zu %zu %zu %d %d\n", model.vertex.size(), model.normal.size(), model.texture.size(), model.face.size(), model.materials.size(),
                     model.camerainfo.width, model.camerainfo.height);
        for (auto& v : model.vertex) std::fprintf(f, "v %.17g %.17g %.17g\n", v.x, v.y, v.z);
        for (auto& v : model.normal) std::fprintf(f, "vn %.17g %.17g %.17g\n", v.x, v.y, v.z);
        for (auto& v : model.texture) std::fprintf(f, "vt %.17g %.17g\n", v.x, v.y);
        for (auto& fc : model.face) { std::fprintf(f, "f"); for (int c = 0; c < 3; c++) for (int k = 0; k < 4; k++) std::fprintf(f, " %d", fc[c][k]); std::fprintf(f, "\n"); }
        for (auto& m : model.materials) {
            std::fprintf(f, "m %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %zu %d %d\n", m.Ks.x, m.Ks.y, m.Ks.z, m.Tr.x, m.Tr.y, m.Tr.z, m.Ns, m.Ni,
                         m.radiance.x, m.radiance.y, m.radiance.z, m.Map_Kd->image_color.size(), m.Map_Kd->image_w, m.Map_Kd->image_h);
            std::fprintf(f, "t"); for (auto& c : m.Map_Kd->image_color) std::fprintf(f, " %.9g %.9g %.9g", c.x, c.y, c.z); std::fprintf(f, "\n");
        }
        const CameraInfo& c = model.camerainfo;
        std::fprintf(f, "c %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", c.eye.x, c.eye.y, c.eye.z, c.lookat.x, c.lookat.y, c.lookat.z, c.up.x, c.up.y, c.up.z, c.fovy);
        std::fclose(f);
        if (!check_only) return 0;
    }
    if (check_only) {   // host-only: what the loader produced + what the library's flatten / BVH build makes of it (no GPU needed)
        std::vector<mcpt_material> mats; std::vector<mcpt_texture> texs; mcpt_scene_desc d; mcpt_scene_info info;
        model_to_desc(model, mats, texs, d);
        const mcpt_status st = mcpt_check_scene(&d, &info);
        double sv = 0, sn = 0, st_ = 0; long long sf = 0; double stex = 0;
        for (auto& v : model.vertex) sv += v.x + 2 * v.y + 3 * v.z;
        for (auto& v : model.normal) sn += v.x + 2 * v.y + 3 * v.z;
        for (auto& v : model.texture) st_ += v.x + 2 * v.y;
        for (auto& f : model.face) for (int i = 0; i < 3; i++) sf += f[i][0] + 3LL * f[i][1] + 5LL * f[i][2] + 7LL * f[i][3];
        for (auto& m : model.materials) for (auto& c : m.Map_Kd->image_color) stex += c.x + c.y + c.z;
        std::printf("{\"status\": %d, \"faces\": %zu, \"materials\": %zu, \"width\": %d, \"height\": %d, \"fovy\": %.17g, \"sum_v\": %.17g, \"sum_vn\": %.17g, "
                    "\"sum_vt\": %.17g, \"sum_f\": %lld, \"sum_tex\": %.9g, \"n_tris\": %u, \"n_lights\": %u, \"n_nodes\": %u, \"bvh_depth\": %u}\n",
                    int(st), model.face.size(), model.materials.size(), model.camerainfo.width, model.camerainfo.height, model.camerainfo.fovy, sv, sn, st_, sf, stex,
                    info.n_tris, info.n_lights, info.n_nodes, info.bvh_depth);
        return st == MCPT_OK ? 0 : 1;
    }
    const int w = model.camerainfo.width, h = model.camerainfo.height;
    Scene scene(w, h);
    const size_t slash = filename.rfind('/');
    const std::string file_name = filename.substr(slash == std::string::npos ? 0 : slash + 1);
    if (out.empty()) out = file_name;
    if (batch == 0) batch = spp;
    if (gpus < 1) gpus = 1;

    // The scene is flattened and its BVH built ONCE (Render::Render(Model&), Render.cpp:5-10); the other devices get device-to-device
    // copies of the finished streams (mcpt_clone_to_device): 8 GPUs cost one build + 8 uploads, not 8 builds.
    std::vector<Render*> renders(gpus, nullptr);
    for (uint32_t g = 0; g < gpus; g++) {
        if (g == 0) {
            mcpt_opts o; std::memset(&o, 0, sizeof o); o.struct_size = sizeof o; o.device = 0; o.max_depth = depth; o.flags = flags; o.integrator = integrator;
            renders[0] = new Render(model, o); renders[0]->seed = seed;
        } else renders[g] = new Render(*renders[0], int(g));
        if (!renders[g]->ok()) return 1;
    }
    std::vector<ncclComm_t> comms(gpus);
    if (gpus > 1) {
        std::vector<int> devs(gpus); for (uint32_t g = 0; g < gpus; g++) devs[g] = int(g);
        if (ncclCommInitAll(comms.data(), int(gpus), devs.data()) != ncclSuccess) { std::cerr << "Error: ncclCommInitAll" << std::endl; return 1; }
    }
    uint32_t frame = 0, batches_done = 0; uint64_t rays = 0; double total_s = 0;
    void* progress_film = nullptr;                    // device 0: the sum of all devices' films, for --save-every with several devices
    std::vector<float> film(size_t(w) * h * 4);
    std::atomic<int> failed{0};                       // any device error or failed collective: no image, non-zero exit
    auto fail_with = [&](const std::string& what) { std::cerr << "Error: " << what << std::endl; failed.store(1); };
    // The films stay on the devices from batch to batch (the reference's loop reads its film every frame only to display it): per batch
    // one mcpt_render per device, at the end the path's one exchange step and one read-back.
    while (frame < spp && !failed.load()) {
        const uint32_t n = std::min(batch, spp - frame);
        auto t0 = std::chrono::steady_clock::now();
        // sample range [frame, frame+n) split contiguously over the devices; one host thread per device (mcpt_render blocks
        // until its device's work is enqueued and nearly finished)
        auto work = [&](uint32_t g) {
            const uint32_t lo = frame + uint32_t(uint64_t(n) * g / gpus), hi = frame + uint32_t(uint64_t(n) * (g + 1) / gpus);
            mcpt_ctx* c = renders[g]->handle();
            // --shard tiles: device g renders ALL n samples of its interleaved share of the 8x8 tiles (BASELINE.json's "pixel-tile shard");
            // default: its contiguous share of the sample range for every pixel.  Either way the films add up to the frame.
            const mcpt_status rs = shard_tiles ? mcpt_render_tiles(c, n, seed, frame, gpus, g) : (hi > lo ? mcpt_render(c, hi - lo, seed, lo) : MCPT_OK);
            if (rs != MCPT_OK) return fail_with(std::string("mcpt_render: ") + mcpt_last_error());
            if (mcpt_sync(c) != MCPT_OK) return fail_with(std::string("mcpt_sync: ") + mcpt_last_error());
        };
        if (gpus == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (uint32_t g = 0; g < gpus; g++) th.emplace_back(work, g);
            for (auto& t : th) t.join();
        }
        if (failed.load()) break;
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        total_s += s; frame += n;
        std::cout << "frame: " << frame << "    frame cost: " << s << "s\n";                       // main.cpp:31
        // --save-every K: a progressive image every K batches, like the reference's window shows every frame (main.cpp:33-36) -- tonemapped
        // on the device (Scene::getPixelsColor as a kernel: mean, clamp, sqrt, x255.99), 3 bytes per pixel read back
        batches_done++;
        if (save_every && batches_done % save_every == 0 && frame < spp) {
            std::vector<uint8_t> rgb(size_t(w) * h * 3);
            bool ok_img;
            if (gpus == 1) ok_img = mcpt_tonemap(renders[0]->handle(), rgb.data(), 1) == MCPT_OK;
            else {
                // the whole film so far, like the reference's window (main.cpp:26-36): the devices' films are summed into a scratch film on
                // device 0 (ncclReduce; the films themselves keep accumulating untouched) and that one is tonemapped
                bool ok = hipSetDevice(0) == hipSuccess && (progress_film || hipMalloc(&progress_film, size_t(w) * h * 16) == hipSuccess);
                ok = ok && ncclGroupStart() == ncclSuccess;
                for (uint32_t g = 0; g < gpus && ok; g++) {
                    void* p = nullptr;
                    ok = mcpt_accum_device_ptr(renders[g]->handle(), &p) == MCPT_OK && hipSetDevice(int(g)) == hipSuccess &&
                         ncclReduce(p, progress_film, size_t(w) * h * 4, ncclFloat, ncclSum, 0, comms[g], nullptr) == ncclSuccess;
                }
                ok = (ncclGroupEnd() == ncclSuccess) && ok;
                for (uint32_t g = 0; g < gpus && ok; g++) ok = hipSetDevice(int(g)) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
                ok_img = ok && mcpt_tonemap_buffer(renders[0]->handle(), progress_film, rgb.data(), 1) == MCPT_OK;
            }
            if (!ok_img) { fail_with(std::string("progressive image: ") + mcpt_last_error()); break; }
            const std::string file = out + std::to_string(frame) + ".png";
            if (write_png_rgb8(file, w, h, rgb.data())) std::cout << "Image saved successfully: " << file << std::endl;
            else std::cerr << "Failed to save image: " << file << std::endl;
        }
    }
    if (!failed.load()) {
        auto t0 = std::chrono::steady_clock::now();
        if (gpus > 1) {                               // sum of the per-device films over xGMI
            bool ok = ncclGroupStart() == ncclSuccess;
            for (uint32_t g = 0; g < gpus && ok; g++) {
                void* p = nullptr;
                ok = mcpt_accum_device_ptr(renders[g]->handle(), &p) == MCPT_OK && hipSetDevice(int(g)) == hipSuccess &&
                     ncclAllReduce(p, p, size_t(w) * h * 4, ncclFloat, ncclSum, comms[g], nullptr) == ncclSuccess;
            }
            ok = (ncclGroupEnd() == ncclSuccess) && ok;
            for (uint32_t g = 0; g < gpus && ok; g++) ok = hipSetDevice(int(g)) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
            if (!ok) fail_with("RCCL all-reduce of the films failed");
        }
        if (!failed.load()) {
            if (mcpt_read_accum(renders[0]->handle(), film.data()) != MCPT_OK) fail_with(std::string("mcpt_read_accum: ") + mcpt_last_error());
            else scene.add_film(film.data());
        }
        total_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    if (progress_film) { (void)hipSetDevice(0); (void)hipFree(progress_film); }
    if (gpus > 1) for (auto& c : comms) (void)ncclCommDestroy(c);
    if (failed.load()) { for (auto r : renders) delete r; return 1; }
    for (uint32_t g = 0; g < gpus; g++) {
        mcpt_counters c;
        if (mcpt_get_counters(renders[g]->handle(), &c) != MCPT_OK) { std::cerr << "Error: " << mcpt_last_error() << std::endl; return 1; }
        rays += c.rays_primary + c.rays_continuation + c.rays_shadow;
    }
    std::printf("%u spp, %dx%d, %u GPU(s): %.3f s, %.1f Mray/s\n", spp, w, h, gpus, total_s, rays / total_s / 1e6);
    scene.save_image(int(frame), out);                                                              // main.cpp:37
    for (auto r : renders) delete r;
    return 0;
}
