// Sanitizer harness for the host half of mcpt_scene_create: build_host_scene (validation, flattening, light
// collection, binned-SAH builder, 8-wide collapse + quantisation) and validate_wide_bvh, fed with hostile scene
// descriptions -- indices out of range, NaN / inf / huge coordinates, degenerate and duplicated triangles, zero-sized
// films and textures, no lights.  A description may be rejected (status != MCPT_OK); an accepted one must yield a
// tree validate_wide_bvh calls sound.  Built by tools/fuzz_scene_build.sh with -fsanitize=address,undefined (CPU only).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <string>
#include <vector>
#include "../monte-carlo-path-tracer_amd/csrc/scene_build.h"

int main(int argc, char** argv) {
    const int cases = argc > 1 ? std::atoi(argv[1]) : 400;
    const unsigned seed = argc > 2 ? unsigned(std::atoi(argv[2])) : 1u;
    std::mt19937 rng(seed);
    auto uni = [&](double a, double b) { return std::uniform_real_distribution<double>(a, b)(rng); };
    auto pick = [&](int n) { return int(rng() % unsigned(n)); };
    int accepted = 0, rejected = 0, unsound = 0;
    for (int c = 0; c < cases; c++) {
        const int kind = c % 8;
        if (std::getenv("FUZZ_VERBOSE")) { std::printf("case %d kind %d\n", c, kind); std::fflush(stdout); }
        const uint32_t nv = 3 + pick(kind == 7 ? 30000 : 400), nn = 1 + pick(50), nt = 1 + pick(50), nf = 1 + pick(kind == 7 ? 60000 : 900);
        const uint32_t nm = 1 + pick(6), ntex = 1 + pick(4);
        std::vector<double> vertex(3 * nv), normal(3 * nn), tc(2 * nt);
        const double span = kind == 3 ? 1e17 : kind == 4 ? 1e-30 : 10.0;
        for (auto& x : vertex) x = uni(-span, span);
        if (kind == 5) for (uint32_t i = 0; i < nv; i++) vertex[3 * i + 1] = 0.0;                 // everything coplanar
        if (kind == 6) for (uint32_t i = 3; i < 3 * nv; i++) vertex[i] = vertex[i % 3];            // every vertex identical
        for (auto& x : normal) x = uni(-1, 1);
        for (auto& x : tc) x = uni(-3, 3);
        if (kind == 1) {                                                                           // non-finite coordinates
            const double bad[3] = {std::numeric_limits<double>::quiet_NaN(), std::numeric_limits<double>::infinity(), -std::numeric_limits<double>::infinity()};
            for (int k = 0; k < 5; k++) vertex[pick(int(vertex.size()))] = bad[pick(3)];
        }
        std::vector<int32_t> face(12 * nf);
        for (uint32_t f = 0; f < nf; f++)
            for (int k = 0; k < 3; k++) {
                face[12 * f + 4 * k + 0] = pick(int(nv)); face[12 * f + 4 * k + 1] = pick(int(nn));
                face[12 * f + 4 * k + 2] = pick(int(nt)); face[12 * f + 4 * k + 3] = pick(int(nm));
            }
        if (kind == 2) {                                                                           // indices out of range / negative
            const int32_t bad[4] = {-1, int32_t(nv), std::numeric_limits<int32_t>::max(), std::numeric_limits<int32_t>::min()};
            for (int k = 0; k < 3; k++) face[pick(int(face.size()))] = bad[pick(4)];
        }
        std::vector<std::vector<float>> texel(ntex);
        std::vector<mcpt_texture> tex(ntex);
        for (uint32_t t = 0; t < ntex; t++) {
            int w = 1 + pick(9), h = 1 + pick(9);
            texel[t].assign(size_t(3) * w * h, 0.5f);
            tex[t] = {w, h, texel[t].data()};
            if (kind == 2 && pick(6) == 0) tex[t].width = pick(2) ? 0 : -4;
        }
        std::vector<mcpt_material> mat(nm);
        for (uint32_t m = 0; m < nm; m++) {
            mat[m] = {};
            mat[m].ks[0] = mat[m].ks[1] = mat[m].ks[2] = pick(2) ? 0.0 : uni(0, 1);
            mat[m].ns = pick(3) ? uni(0, 2000) : 1.0;
            if (m == 0 && c % 11 != 10) mat[m].radiance[0] = mat[m].radiance[1] = mat[m].radiance[2] = 10.0;   // usually one emitter
            mat[m].map_kd = pick(int(ntex));
            if (kind == 2 && pick(8) == 0) mat[m].map_kd = pick(2) ? -1 : int32_t(ntex);
        }
        mcpt_scene_desc d{};
        d.vertex = vertex.data(); d.n_vertex = nv; d.normal = normal.data(); d.n_normal = nn; d.texcoord = tc.data(); d.n_texcoord = nt;
        d.face = face.data(); d.n_face = nf; d.materials = mat.data(); d.n_materials = nm; d.textures = tex.data(); d.n_textures = ntex;
        d.camera = {{0, 0, 5}, {0, 0, 0}, {0, 1, 0}, 40.0, 64, 48};
        if (kind == 2 && pick(5) == 0) d.camera.width = pick(2) ? 0 : -1;
        HostScene hs; std::string err;
        hs.reference_tie_order = (c % 3) == 0;                 // MCPT_FLAG_REFERENCE_TIE_ORDER: the reference's std::partition order replayed on hostile input too
        const mcpt_status st = build_host_scene(&d, hs, err);
        if (st != MCPT_OK) { rejected++; continue; }
        accepted++;
        std::string why = validate_wide_bvh(hs);
        if (why.empty()) {                                     // the tie ranks (low 28 bits of every intersection record's .w) are a permutation of 0 .. n-1
            const size_t nt = hs.tri_face.size();
            std::vector<uint8_t> seen(nt, 0);
            for (size_t i = 0; i < nt && why.empty(); i++) {
                uint32_t w; std::memcpy(&w, &hs.tri_isect[3 * i].w, 4); w &= 0x0fffffffu;
                if (w >= nt || seen[w]++) why = "tie ranks are not a permutation";
                else if (!hs.reference_tie_order && w != i) why = "default tie rank is not the leaf-order index";
            }
        }
        if (!why.empty()) { unsound++; std::printf("case %d kind %d: accepted but tree unsound: %s\n", c, kind, why.c_str()); }
    }
    std::printf("%d cases: %d accepted, %d rejected, %d unsound\n", cases, accepted, rejected, unsound);
    return unsound ? 1 : 0;
}
