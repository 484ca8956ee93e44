// Host-only what-if: how many traversal steps per ray would a k-wide tree need?  Builds the production binary SAH tree (scene_build.cpp)
// for an OBJ scene, collapses it to k = 2, 4, 6, 8 children per node with the production rule (adopt the children of the largest inner
// child until the node is full) and walks secondary-ray-like rays through each the way the trace kernel does (all child boxes tested
// against [1e-4, tmax], hits ordered by entry distance, nearest first, no culling at the pop), counting node visits ("steps"), child-box
// tests, leaf visits and triangle tests.  Exact fp32 boxes (the quantised frames of the device tree are a few % looser).
//   usage: wide_bvh_probe scene.obj [rays=200000]            build: tools/wide_bvh_probe.sh
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../monte-carlo-path-tracer_amd/csrc/scene_build.h"
#include "../monte-carlo-path-tracer_amd/host/Model.h"
#include "../monte-carlo-path-tracer_amd/host/Render.h"

struct B3 { float lo[3], hi[3]; };
static int child2(const std::vector<f4h>& n2, int n, int k) { int c; std::memcpy(&c, k == 0 ? &n2[4 * size_t(n) + 3].x : &n2[4 * size_t(n) + 3].y, 4); return c; }
static B3 box2(const std::vector<f4h>& n2, int n, int k) {
    const f4h a = n2[4 * size_t(n) + k], z = n2[4 * size_t(n) + 2]; B3 b;
    b.lo[0] = a.x; b.hi[0] = a.y; b.lo[1] = a.z; b.hi[1] = a.w; b.lo[2] = k == 0 ? z.x : z.z; b.hi[2] = k == 0 ? z.y : z.w; return b;
}
static double area(const B3& b) { const double x = double(b.hi[0]) - b.lo[0], y = double(b.hi[1]) - b.lo[1], z = double(b.hi[2]) - b.lo[2]; return 2 * (x * y + y * z + z * x); }

struct Wide { std::vector<B3> box; std::vector<int> code; std::vector<int> first, count; };   // node i: children [first[i], first[i]+count[i])
static Wide collapse(const std::vector<f4h>& n2, int K) {
    Wide w; struct Work { int node2, slot; };
    std::vector<Work> q{{0, 0}}; w.first.push_back(0); w.count.push_back(0);
    for (size_t h = 0; h < q.size(); h++) {
        std::vector<std::pair<int, B3>> kids;
        for (int k = 0; k < 2; k++) kids.push_back({child2(n2, q[h].node2, k), box2(n2, q[h].node2, k)});
        while (int(kids.size()) < K) {
            int best = -1; double ba = -1;
            for (size_t i = 0; i < kids.size(); i++) if (kids[i].first >= 0 && area(kids[i].second) > ba) { ba = area(kids[i].second); best = int(i); }
            if (best < 0) break;
            const int n = kids[best].first;
            kids[best] = {child2(n2, n, 0), box2(n2, n, 0)}; kids.push_back({child2(n2, n, 1), box2(n2, n, 1)});
        }
        w.first[q[h].slot] = int(w.box.size()); w.count[q[h].slot] = int(kids.size());
        for (auto& kd : kids) {
            int code = kd.first;
            if (code >= 0) { const int slot = int(w.first.size()); w.first.push_back(0); w.count.push_back(0); q.push_back({code, slot}); code = slot; }
            w.box.push_back(kd.second); w.code.push_back(code);
        }
    }
    return w;
}
// SAH-optimal collapse (dynamic programme over the binary tree, Wald et al. 2008 / Ylitie et al. 2017, leaves kept as they are):
// cost[n][i] = least expected number of node visits for the subtree of binary node n when it may occupy i child slots of its k-wide
// parent (i = 1: n becomes a k-wide node itself; i > 1: n is dissolved and its two children share the i slots).
static Wide collapse_dp(const std::vector<f4h>& n2, int K) {
    const int N = int(n2.size() / 4);
    std::vector<double> A(N);
    std::vector<int> order; order.reserve(N);                           // parents before children
    { std::vector<int> st{0}; while (!st.empty()) { int n = st.back(); st.pop_back(); order.push_back(n); for (int k = 0; k < 2; k++) { int c = child2(n2, n, k); if (c >= 0) st.push_back(c); } } }
    auto uni = [&](int n) { B3 a = box2(n2, n, 0), b = box2(n2, n, 1); for (int x = 0; x < 3; x++) { a.lo[x] = std::min(a.lo[x], b.lo[x]); a.hi[x] = std::max(a.hi[x], b.hi[x]); } return a; };
    const double root_area = area(uni(0));
    for (int n = 0; n < N; n++) A[n] = area(uni(n)) / root_area;
    std::vector<double> cost(size_t(N) * (K + 1), 0.0);                 // cost[n * (K + 1) + i], i = 1 .. K
    std::vector<unsigned char> split(size_t(N) * (K + 1), 0);           // slots given to the left child when n is dissolved over i slots; 0 = "use i - 1"
    auto C = [&](int code, int i) -> double { return code < 0 ? 0.0 : cost[size_t(code) * (K + 1) + i]; };   // a leaf costs no node visit in any number of slots
    for (int idx = N - 1; idx >= 0; idx--) {
        const int n = order[idx], l = child2(n2, n, 0), r = child2(n2, n, 1);
        auto distribute = [&](int j, int& best_a) { double best = 1e300; for (int a = 1; a < j; a++) { const double c = C(l, a) + C(r, j - a); if (c < best) { best = c; best_a = a; } } return best; };
        int a = 1;
        cost[size_t(n) * (K + 1) + 1] = A[n] + distribute(K, a); split[size_t(n) * (K + 1) + 1] = (unsigned char)a;
        for (int i = 2; i <= K; i++) {
            const double d = distribute(i, a), keep = cost[size_t(n) * (K + 1) + i - 1];
            if (d < keep) { cost[size_t(n) * (K + 1) + i] = d; split[size_t(n) * (K + 1) + i] = (unsigned char)a; } else { cost[size_t(n) * (K + 1) + i] = keep; split[size_t(n) * (K + 1) + i] = 0; }
        }
    }
    Wide w; struct Work { int node2, slot; };
    std::vector<Work> q{{0, 0}}; w.first.push_back(0); w.count.push_back(0);
    for (size_t h = 0; h < q.size(); h++) {
        std::vector<std::pair<int, B3>> kids;
        // gather the roots of the forest below binary node `n` that may use `i` slots
        struct Item { int parent, k, slots; };
        std::vector<Item> todo;
        { const int n = q[h].node2; const int a = split[size_t(n) * (K + 1) + 1]; todo.push_back({n, 0, a}); todo.push_back({n, 1, K - a}); }
        while (!todo.empty()) {
            const Item it = todo.back(); todo.pop_back();
            const int c = child2(n2, it.parent, it.k);
            int i = it.slots;
            if (c < 0) { kids.push_back({c, box2(n2, it.parent, it.k)}); continue; }
            while (i > 1 && split[size_t(c) * (K + 1) + i] == 0) i--;     // "use i - 1"
            if (i == 1) { kids.push_back({c, box2(n2, it.parent, it.k)}); continue; }
            const int a = split[size_t(c) * (K + 1) + i];
            todo.push_back({c, 0, a}); todo.push_back({c, 1, i - a});
        }
        w.first[q[h].slot] = int(w.box.size()); w.count[q[h].slot] = int(kids.size());
        for (auto& kd : kids) {
            int code = kd.first;
            if (code >= 0) { const int slot = int(w.first.size()); w.first.push_back(0); w.count.push_back(0); q.push_back({code, slot}); code = slot; }
            w.box.push_back(kd.second); w.code.push_back(code);
        }
    }
    return w;
}
struct Stats { double steps = 0, boxes = 0, leaves = 0, tris = 0, hits = 0; };
static void trace(const Wide& w, const HostScene& hs, const float o[3], const float d[3], bool any, float tmax, Stats& st) {
    float id[3]; for (int a = 0; a < 3; a++) id[a] = 1.0f / (std::fabs(d[a]) > 1e-30f ? d[a] : std::copysign(1e-30f, d[a]));
    std::vector<int> stack{0};
    while (!stack.empty()) {
        const int code = stack.back(); stack.pop_back();
        if (code < 0) {
            const uint32_t leaf = uint32_t(~code), first = leaf >> 3, cnt = leaf & 7u; st.leaves++;
            for (uint32_t i = 0; i < cnt; i++) {
                st.tris++;
                const f4h v0 = hs.tri_isect[3 * size_t(first + i)], e1 = hs.tri_isect[3 * size_t(first + i) + 1], e2 = hs.tri_isect[3 * size_t(first + i) + 2];
                const float px = d[1] * e2.z - d[2] * e2.y, py = d[2] * e2.x - d[0] * e2.z, pz = d[0] * e2.y - d[1] * e2.x;
                const float det = e1.x * px + e1.y * py + e1.z * pz;
                if (std::fabs(det) < 1e-5f) continue;
                const float inv = 1.0f / det, tx = o[0] - v0.x, ty = o[1] - v0.y, tz = o[2] - v0.z;
                const float u = (tx * px + ty * py + tz * pz) * inv; if (u < 0 || u > 1) continue;
                const float qx = ty * e1.z - tz * e1.y, qy = tz * e1.x - tx * e1.z, qz = tx * e1.y - ty * e1.x;
                const float v = (d[0] * qx + d[1] * qy + d[2] * qz) * inv; if (v < 0 || u + v > 1) continue;
                const float t = (e2.x * qx + e2.y * qy + e2.z * qz) * inv;
                if (t >= 1e-4f && t < tmax) { tmax = t; st.hits++; if (any) return; }
            }
            continue;
        }
        st.steps++;
        std::pair<float, int> hit[8]; int nh = 0;
        for (int k = 0; k < w.count[code]; k++) {
            const B3& b = w.box[w.first[code] + k]; st.boxes++;
            float tn = 1e-4f, tf = tmax;
            for (int a = 0; a < 3; a++) { float t0 = (b.lo[a] - o[a]) * id[a], t1 = (b.hi[a] - o[a]) * id[a]; if (t0 > t1) std::swap(t0, t1); tn = std::max(tn, t0); tf = std::min(tf, t1); }
            if (tn <= tf) hit[nh++] = {tn, w.code[w.first[code] + k]};
        }
        if (!any) std::sort(hit, hit + nh, [](auto& a, auto& b) { return a.first > b.first; });   // far ... near: the nearest is popped first
        for (int k = 0; k < nh; k++) stack.push_back(hit[k].second);
    }
}

// ---- what the device's 8-wide kernel will actually do (round 3): children sit in OCTANT SLOTS (slot bit a set = the child lies towards +a
// of the node's centre; greedy assignment), a ray visits the hit inner children in the order of slot ^ octant (no distance sort), and the
// hit LEAF children of a node are tested before any of its inner children (Ylitie, Karras & Laine 2017).  Counts the same things as trace().
struct Slots { std::vector<int> slot; };                               // per child entry of a Wide: its slot 0..7
static Slots assign_slots(const Wide& w) {
    Slots s; s.slot.assign(w.box.size(), 0);
    for (size_t n = 0; n < w.first.size(); n++) {
        const int f = w.first[n], c = w.count[n];
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
        for (int k = 0; k < c; k++) for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], w.box[f + k].lo[a]); hi[a] = std::max(hi[a], w.box[f + k].hi[a]); }
        double cost[8][8];
        for (int k = 0; k < c; k++) for (int sl = 0; sl < 8; sl++) {
            double v = 0; for (int a = 0; a < 3; a++) { const double cc = 0.5 * (double(w.box[f + k].lo[a]) + w.box[f + k].hi[a]) - 0.5 * (double(lo[a]) + hi[a]); v += ((sl >> a) & 1) ? cc : -cc; }
            cost[k][sl] = v;
        }
        bool ck[8] = {0}, cs[8] = {0};
        for (int r = 0; r < c; r++) { int bk = -1, bs = -1; double bv = -1e300; for (int k = 0; k < c; k++) if (!ck[k]) for (int sl = 0; sl < 8; sl++) if (!cs[sl] && cost[k][sl] > bv) { bv = cost[k][sl]; bk = k; bs = sl; }
            ck[bk] = cs[bs] = true; s.slot[f + bk] = bs; }
    }
    return s;
}
static bool tri_hit(const HostScene& hs, uint32_t ti, const float o[3], const float d[3], float& tmax) {
    const f4h v0 = hs.tri_isect[3 * size_t(ti)], e1 = hs.tri_isect[3 * size_t(ti) + 1], e2 = hs.tri_isect[3 * size_t(ti) + 2];
    const float px = d[1] * e2.z - d[2] * e2.y, py = d[2] * e2.x - d[0] * e2.z, pz = d[0] * e2.y - d[1] * e2.x;
    const float det = e1.x * px + e1.y * py + e1.z * pz;
    if (std::fabs(det) < 1e-5f) return false;
    const float inv = 1.0f / det, tx = o[0] - v0.x, ty = o[1] - v0.y, tz = o[2] - v0.z;
    const float u = (tx * px + ty * py + tz * pz) * inv; if (u < 0 || u > 1) return false;
    const float qx = ty * e1.z - tz * e1.y, qy = tz * e1.x - tx * e1.z, qz = tx * e1.y - ty * e1.x;
    const float v = (d[0] * qx + d[1] * qy + d[2] * qz) * inv; if (v < 0 || u + v > 1) return false;
    const float t = (e2.x * qx + e2.y * qy + e2.z * qz) * inv;
    if (t >= 1e-4f && t < tmax) { tmax = t; return true; }
    return false;
}
// order: 0 = octant order, leaves of a node first; 1 = exact distance order for the inner children (leaves first); 2 = nearest inner child exactly, rest in octant order
static void trace_octant(const Wide& w, const Slots& sl, const HostScene& hs, const float o[3], const float d[3], bool any, float tmax, int order, Stats& st) {
    float id[3]; for (int a = 0; a < 3; a++) id[a] = 1.0f / (std::fabs(d[a]) > 1e-30f ? d[a] : std::copysign(1e-30f, d[a]));
    const int oct = (d[0] < 0 ? 1 : 0) | (d[1] < 0 ? 2 : 0) | (d[2] < 0 ? 4 : 0);
    std::vector<int> stack{0};
    while (!stack.empty()) {
        const int node = stack.back(); stack.pop_back();
        st.steps++;
        std::pair<float, int> inner[8]; int ni = 0; int leaves[8]; int nl = 0;
        for (int k = 0; k < w.count[node]; k++) {
            const B3& b = w.box[w.first[node] + k]; st.boxes++;
            float tn = 1e-4f, tf = tmax;
            for (int a = 0; a < 3; a++) { float t0 = (b.lo[a] - o[a]) * id[a], t1 = (b.hi[a] - o[a]) * id[a]; if (t0 > t1) std::swap(t0, t1); tn = std::max(tn, t0); tf = std::min(tf, t1); }
            if (!(tn <= tf)) continue;
            const int code = w.code[w.first[node] + k];
            if (code < 0) leaves[nl++] = code;
            else { const int prio = sl.slot[w.first[node] + k] ^ oct; inner[ni++] = {order == 1 ? tn : float(prio), code}; if (order == 2) inner[ni - 1].first = float(prio) + 1000.f * 0; }
        }
        for (int k = 0; k < nl; k++) {
            const uint32_t leaf = uint32_t(~leaves[k]), first = leaf >> 3, cnt = leaf & 7u; st.leaves++;
            for (uint32_t i = 0; i < cnt; i++) { st.tris++; if (tri_hit(hs, first + i, o, d, tmax)) { st.hits++; if (any) return; } }
        }
        std::sort(inner, inner + ni, [](auto& a, auto& b) { return a.first > b.first; });   // smallest key popped first
        for (int k = 0; k < ni; k++) stack.push_back(inner[k].second);
    }
}
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const int n_rays = argc > 2 ? std::atoi(argv[2]) : 200000;
    Model model(argv[1], true); if (!model.ok) return 3;
    std::vector<mcpt_material> mats; std::vector<mcpt_texture> texs; mcpt_scene_desc d; model_to_desc(model, mats, texs, d);
    HostScene hs; std::string err;
    if (build_host_scene(&d, hs, err) != MCPT_OK) { std::fprintf(stderr, "%s\n", err.c_str()); return 4; }
    const size_t nt = hs.tri_face.size();
    std::mt19937 rng(7); std::uniform_real_distribution<float> U(0.f, 1.f);
    // secondary-ray-like rays: origin = a random point of an area-weighted random triangle, direction uniform on the sphere
    std::vector<float> O(3 * size_t(n_rays)), D(3 * size_t(n_rays));
    std::vector<double> cdf(nt);                                        // origins area-weighted: path vertices land on surfaces in proportion to their area
    { double acc = 0; for (size_t t = 0; t < nt; t++) { const f4h e1 = hs.tri_isect[3 * t + 1], e2 = hs.tri_isect[3 * t + 2];
        const double cx = double(e1.y) * e2.z - double(e1.z) * e2.y, cy = double(e1.z) * e2.x - double(e1.x) * e2.z, cz = double(e1.x) * e2.y - double(e1.y) * e2.x;
        acc += 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz); cdf[t] = acc; } }
    for (int i = 0; i < n_rays; i++) {
        const size_t t = std::min(nt - 1, size_t(std::lower_bound(cdf.begin(), cdf.end(), double(U(rng)) * cdf.back()) - cdf.begin())); float u = U(rng), v = U(rng); if (u + v > 1) { u = 1 - u; v = 1 - v; }
        const f4h v0 = hs.tri_isect[3 * t], e1 = hs.tri_isect[3 * t + 1], e2 = hs.tri_isect[3 * t + 2];
        O[3 * i] = v0.x + u * e1.x + v * e2.x; O[3 * i + 1] = v0.y + u * e1.y + v * e2.y; O[3 * i + 2] = v0.z + u * e1.z + v * e2.z;
        const float z = 2 * U(rng) - 1, ph = 6.2831853f * U(rng), r = std::sqrt(std::max(0.f, 1 - z * z));
        D[3 * i] = r * std::cos(ph); D[3 * i + 1] = r * std::sin(ph); D[3 * i + 2] = z;
    }
    std::printf("%s: %zu triangles, %zu binary nodes, %d rays (closest-hit | any-hit)\n", argv[1], nt, hs.nodes.size() / 4, n_rays);
    for (int mode = 0; mode < 2; mode++)
    for (int K : {2, 4, 6, 8}) {
        if (mode == 1 && K == 2) continue;
        const Wide w = mode ? collapse_dp(hs.nodes, K) : collapse(hs.nodes, K);
        if (K == 4) std::printf("  -- %s collapse\n", mode ? "SAH-optimal (dynamic programme)" : "production (largest inner child first)");
        double fill = double(w.box.size()) / double(w.first.size());
        for (int any = 0; any < 2; any++) {
            Stats st;
            for (int i = 0; i < n_rays; i++) trace(w, hs, &O[3 * i], &D[3 * i], any != 0, 3.0e38f, st);
            std::printf("  k=%d %-8s nodes %8zu (%.2f children each)  steps/ray %6.2f  box tests/ray %7.2f  leaf visits/ray %5.2f  tri tests/ray %5.2f\n", K, any ? "any-hit" : "closest", w.first.size(), fill,
                        st.steps / n_rays, st.boxes / n_rays, st.leaves / n_rays, st.tris / n_rays);
        }
    }
    for (int K : {4, 8}) {
        const Wide w = collapse_dp(hs.nodes, K); const Slots sl = assign_slots(w);
        for (int order = 0; order < 2; order++) for (int any = 0; any < 2; any++) {
            Stats st;
            for (int i = 0; i < n_rays; i++) trace_octant(w, sl, hs, &O[3 * i], &D[3 * i], any != 0, 3.0e38f, order, st);
            std::printf("  k=%d SAH-optimal, leaves first, inner children in %s order, %-8s steps/ray %6.2f  box tests/ray %7.2f  leaf visits/ray %5.2f  tri tests/ray %5.2f\n", K,
                        order ? "distance" : "octant  ", any ? "any-hit" : "closest", st.steps / n_rays, st.boxes / n_rays, st.leaves / n_rays, st.tris / n_rays);
        }
    }
    return 0;
}
