// Host-side scene preparation (see scene_build.h).
//
//  * flatten():  Render::tranform_triangle (Render.cpp:12-44) -- faces -> triangles by index, material per face
//                from corner 0, lights = triangles whose |radiance| > 0.01 in face order.
//  * Builder:    replaces BVH::build (BVH.cpp:15-54: spatial-midpoint split of the centroid box, leaf <= 5,
//                pointer tree).  The traversal result (closest hit / any hit) does not depend on the tree, only its
//                cost does, so the tree here is built for the GPU: binned-SAH splits (16 bins x 3 axes), leaves
//                <= MCPT_LEAF_MAX triangles, child boxes stored in the parent (64-B nodes), depth capped so the LDS
//                stack of MCPT_STACK_DEPTH entries can never overflow (median splits take over when the remaining
//                depth budget gets tight).  Boxes are the fp64 triangle bounds rounded OUTWARD to fp32 and padded by
//                ~16 ulp so the fp32 slab test never rejects a box whose fp32 triangle test would accept.
#include "scene_build.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <future>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <system_error>
#include <thread>

static constexpr double kMaxCoord = 1e18;

namespace {

struct BTri { double lo[3], hi[3], c[3]; };
struct Box {
    double lo[3], hi[3];
    Box() { for (int a = 0; a < 3; a++) { lo[a] = std::numeric_limits<double>::max(); hi[a] = std::numeric_limits<double>::lowest(); } }
    void grow(const double* l, const double* h) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]); } }
    void grow(const Box& b) { grow(b.lo, b.hi); }
    void grow_pt(const double* p) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    double area() const {
        double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0;
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    }
};

inline int leaf_code(uint32_t first, uint32_t count) { return ~int((first << 3) | count); }
inline float as_float(int v) { float f; std::memcpy(&f, &v, 4); return f; }

inline float round_down(double v, float pad) { float f = float(v); if (double(f) > v) f = std::nextafterf(f, -INFINITY); return f - pad; }
inline float round_up(double v, float pad) { float f = float(v); if (double(f) < v) f = std::nextafterf(f, INFINITY); return f + pad; }

class Builder {
public:
    Builder(const std::vector<BTri>& t, std::vector<f4h>& nodes, int leaf_max = MCPT_LEAF_MAX) : leaf_max_(std::min(std::max(leaf_max, 1), MCPT_LEAF_MAX)), t_(t), nodes_(nodes), order_(t.size()) {
        for (size_t i = 0; i < order_.size(); i++) order_[i] = int(i);
    }
    void run() {
        const int n = int(t_.size());
        Box rb;
        if (n <= MCPT_LEAF_MAX) {            // keep the invariant "node 0 is an inner node"
            nodes_.resize(4);
            Box b; for (int i = 0; i < n; i++) b.grow(t_[i].lo, t_[i].hi);
            Box empty; for (int a = 0; a < 3; a++) { empty.lo[a] = 0; empty.hi[a] = 0; }
            write_node(0, b, leaf_code(0, uint32_t(n)), empty, leaf_code(0, 0));
            max_leaf = uint32_t(n); depth = 1;
            return;
        }
        // Subtrees are independent (disjoint ranges of order_, node records claimed from an atomic counter), so the top levels of
        // a large scene fork one task per child.  The topology does not depend on the schedule; node NUMBERS do, and the caller
        // renumbers breadth-first anyway.
        nodes_.resize(4 * size_t(n));                                  // upper bound: n - 1 inner nodes
        build(0, n, 0, rb);
        nodes_.resize(4 * size_t(next_node_.load()));
        depth = depth_.load(); max_leaf = max_leaf_.load();
    }
    const std::vector<int>& order() const { return order_; }
    uint32_t depth = 0, max_leaf = 0;

private:
    static constexpr int kBins = 16;
    static constexpr int kMaxDepth = 30;   // inner-node levels (the traversal stacks hold MCPT_STACK_DEPTH = 64: sentinel + one entry per level)

    int levels_needed(int n) const { int l = 0; while ((leaf_max_ << l) < n) l++; return l; }

    void write_node(int idx, const Box& b0, int c0, const Box& b1, int c1) {
        auto pad_of = [](const Box& b) {
            double m = 0; for (int a = 0; a < 3; a++) m = std::max(m, std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
            return float(m * 1e-6 + 1e-30);
        };
        const float p0 = pad_of(b0), p1 = pad_of(b1);
        f4h* n = &nodes_[4 * size_t(idx)];
        n[0] = {round_down(b0.lo[0], p0), round_up(b0.hi[0], p0), round_down(b0.lo[1], p0), round_up(b0.hi[1], p0)};
        n[1] = {round_down(b1.lo[0], p1), round_up(b1.hi[0], p1), round_down(b1.lo[1], p1), round_up(b1.hi[1], p1)};
        n[2] = {round_down(b0.lo[2], p0), round_up(b0.hi[2], p0), round_down(b1.lo[2], p1), round_up(b1.hi[2], p1)};
        n[3] = {as_float(c0), as_float(c1), 0.f, 0.f};
    }

    // returns child code (inner index >= 0 or leaf code < 0); `box` = bounds of [l,r)
    int build(int l, int r, int d, Box& box) {
        const int n = r - l;
        for (int i = l; i < r; i++) box.grow(t_[order_[i]].lo, t_[order_[i]].hi);
        if (n <= leaf_max_) {
            atomic_max(max_leaf_, uint32_t(n));
            return leaf_code(uint32_t(l), uint32_t(n));
        }
        atomic_max(depth_, uint32_t(d + 1));
        Box cb;
        for (int i = l; i < r; i++) cb.grow_pt(t_[order_[i]].c);
        int mid = -1;
        const int budget = kMaxDepth - (d + 1);               // levels left for each child
        // ---- binned SAH over the three axes
        double best_cost = std::numeric_limits<double>::max(); int best_axis = -1, best_bin = -1;
        for (int a = 0; a < 3; a++) {
            const double ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0)) continue;
            Box bb[kBins]; int cnt[kBins] = {0};
            const double scale = kBins / ext;
            for (int i = l; i < r; i++) {
                const BTri& T = t_[order_[i]];
                int b = int((T.c[a] - cb.lo[a]) * scale); b = std::min(std::max(b, 0), kBins - 1);
                bb[b].grow(T.lo, T.hi); cnt[b]++;
            }
            double la[kBins], ra[kBins]; int lc[kBins], rc[kBins];
            Box acc; int c = 0;
            for (int b = 0; b < kBins; b++) { acc.grow(bb[b]); c += cnt[b]; la[b] = acc.area(); lc[b] = c; }
            Box acc2; c = 0;
            for (int b = kBins - 1; b >= 0; b--) { acc2.grow(bb[b]); c += cnt[b]; ra[b] = acc2.area(); rc[b] = c; }
            for (int b = 0; b < kBins - 1; b++) {
                if (lc[b] == 0 || rc[b + 1] == 0) continue;
                if (levels_needed(lc[b]) > budget || levels_needed(rc[b + 1]) > budget) continue;
                const double cost = la[b] * lc[b] + ra[b + 1] * rc[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
            }
        }
        if (best_axis >= 0) {
            const int a = best_axis;
            const double scale = kBins / (cb.hi[a] - cb.lo[a]);
            auto it = std::partition(order_.begin() + l, order_.begin() + r, [&](int ti) {
                int b = int((t_[ti].c[a] - cb.lo[a]) * scale); b = std::min(std::max(b, 0), kBins - 1);
                return b <= best_bin;
            });
            mid = int(it - order_.begin());
        }
        if (mid <= l || mid >= r) {                            // degenerate or out of depth budget: object median
            int a = 0; double ext = cb.hi[0] - cb.lo[0];
            for (int k = 1; k < 3; k++) if (cb.hi[k] - cb.lo[k] > ext) { ext = cb.hi[k] - cb.lo[k]; a = k; }
            mid = l + n / 2;
            std::nth_element(order_.begin() + l, order_.begin() + mid, order_.begin() + r,
                             [&](int x, int y) { return t_[x].c[a] < t_[y].c[a]; });
        }
        const int idx = next_node_.fetch_add(1);
        Box b0, b1;
        int c0, c1;
        std::future<int> left;
        if (n >= kForkMin && d < kForkDepth) {
            try { left = std::async(std::launch::async, [&]() { return build(l, mid, d + 1, b0); }); }
            catch (const std::system_error&) {}                        // no thread to be had: build both children here
        }
        if (left.valid()) {
            c1 = build(mid, r, d + 1, b1);
            c0 = left.get();
        } else {
            c0 = build(l, mid, d + 1, b0);
            c1 = build(mid, r, d + 1, b1);
        }
        write_node(idx, b0, c0, b1, c1);
        return idx;
    }
    static void atomic_max(std::atomic<uint32_t>& a, uint32_t v) { uint32_t cur = a.load(); while (cur < v && !a.compare_exchange_weak(cur, v)) {} }
    static constexpr int kForkMin = 1 << 15, kForkDepth = 5;          // fork while a range has >= 32 k triangles, at most 32 tasks
    std::atomic<int> next_node_{0};
    std::atomic<uint32_t> depth_{0}, max_leaf_{0};

    const int leaf_max_;                                               // triangles per binary leaf (the wide collapse may merge small subtrees again)
    const std::vector<BTri>& t_;
    std::vector<f4h>& nodes_;
    std::vector<int> order_;
};

// static chunking over hardware threads for the embarrassingly parallel per-triangle loops of large scenes
template <class F> void parallel_for(uint32_t n, F&& body) {
    const uint32_t hw = std::max(1u, std::thread::hardware_concurrency()), nt = n < (1u << 16) ? 1u : std::min(hw, 16u);
    if (nt == 1) { body(0u, n); return; }
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < nt; t++) {
        const uint32_t b = uint32_t(uint64_t(n) * t / nt), e = uint32_t(uint64_t(n) * (t + 1) / nt);
        try { th.emplace_back([&body, b, e]() { body(b, e); }); }
        catch (const std::system_error&) { body(b, e); }              // no thread to be had: do this chunk here
    }
    for (auto& x : th) x.join();
}

inline double len3(const double* v) { return std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }

// ---- helpers of the wide collapse: the binary tree's records --------------------------------------------------------------------
struct Box3f { float lo[3], hi[3]; };
inline int child2(const std::vector<f4h>& n2, int n, int k) { int c; std::memcpy(&c, k == 0 ? &n2[4 * size_t(n) + 3].x : &n2[4 * size_t(n) + 3].y, 4); return c; }
inline Box3f box2(const std::vector<f4h>& n2, int n, int k) {
    const f4h a = n2[4 * size_t(n) + k], z = n2[4 * size_t(n) + 2];
    Box3f b;
    b.lo[0] = a.x; b.hi[0] = a.y; b.lo[1] = a.z; b.hi[1] = a.w;
    b.lo[2] = k == 0 ? z.x : z.z; b.hi[2] = k == 0 ? z.y : z.w;
    return b;
}
inline double area3(const Box3f& b) { const double x = double(b.hi[0]) - b.lo[0], y = double(b.hi[1]) - b.lo[1], z = double(b.hi[2]) - b.lo[2]; return 2.0 * (x * y + y * z + z * x); }
inline uint32_t as_u32(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float from_u32(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// Which binary nodes become K-wide nodes: SAH-optimal collapse (dynamic programme over the slot budget, Wald et al. 2008 / Ylitie et al.
// 2017; leaves stay as the builder made them).  cost[n][i] = least expected number of node visits (sum of relative surface areas of the
// K-wide nodes created) for the subtree of binary node n when it may occupy i of its K-wide parent's slots: i = 1 makes n a node of its
// own, i > 1 dissolves it and lets its two children share the slots.  split[n][i] = slots given to the left child (0 = "as for i - 1").
struct CollapsePlan {
    int K = 4;
    std::vector<unsigned char> split;                                   // [n * (K + 1) + i]
    std::vector<int> merged_leaf;                                       // leaf formation (LeafCosts): binary node -> the leaf code its whole subtree becomes, 0 = stays a node
    unsigned char at(int n, int i) const { return split[size_t(n) * (K + 1) + i]; }
    int leaf_of(int n) const { return merged_leaf.empty() ? 0 : merged_leaf[size_t(n)]; }
};
// Leaf formation inside the dynamic programme (Ylitie et al. 2017, sec. 3.1): with `on`, a leaf child costs (visit + tri * triangles) node
// visits per unit of relative area, and a binary subtree of <= MCPT_LEAF_MAX triangles that are consecutive in `order` may become ONE leaf
// child where that is cheaper than keeping it a node.  Off, leaves cost nothing and stay as the binary builder made them.
struct LeafCosts { bool on = false; double visit = 0.0, tri = 0.0; };
CollapsePlan plan_collapse(const std::vector<f4h>& n2, int K, const std::vector<int>& subtree_begin = std::vector<int>(), const LeafCosts lc = LeafCosts()) {
    const int N = int(n2.size() / 4);
    CollapsePlan p; p.K = K; p.split.assign(size_t(N) * (K + 1), 0);
    if (lc.on) p.merged_leaf.assign(size_t(N), 0);
    auto own_area = [&](int n) { Box3f a = box2(n2, n, 0); const Box3f b = box2(n2, n, 1); for (int x = 0; x < 3; x++) { a.lo[x] = std::min(a.lo[x], b.lo[x]); a.hi[x] = std::max(a.hi[x], b.hi[x]); } return area3(a); };
    const double root_area = std::max(own_area(0), 1e-300);
    std::vector<double> cost(size_t(N) * (K + 1), 0.0);
    std::vector<uint32_t> span(lc.on ? size_t(N) : 0, 0);             // (first << 3 | count) of a subtree that could be one leaf, 0 = cannot
    auto leaf_cost = [&](double area, uint32_t cnt) { return area / root_area * (lc.visit + lc.tri * double(cnt)); };
    // cost of child k of node n in i slots: a node's from the table; a leaf is no node visit, in any number of slots (with LeafCosts: its own price)
    auto C = [&](int n, int k, int i) {
        const int code = child2(n2, n, k);
        if (code >= 0) return cost[size_t(code) * (K + 1) + i];
        return lc.on ? leaf_cost(area3(box2(n2, n, k)), uint32_t(~code) & 7u) : 0.0;
    };
    auto span_of = [&](int code) { return code < 0 ? uint32_t(~code) : span[size_t(code)]; };
    // The builder's renumbering puts every parent before its children, so descending index order is children first; below the top levels
    // every subtree is one contiguous index range (subtree_begin), and disjoint subtrees do not read each other's costs.
    auto range = [&](int lo, int hi) {
        for (int n = hi - 1; n >= lo; n--) {
            auto distribute = [&](int j, int& best_a) { double best = 1e300; for (int a = 1; a < j; a++) { const double c = C(n, 0, a) + C(n, 1, j - a); if (c < best) { best = c; best_a = a; } } return best; };
            int a = 1;
            cost[size_t(n) * (K + 1) + 1] = own_area(n) / root_area + distribute(K, a); p.split[size_t(n) * (K + 1) + 1] = (unsigned char)a;
            if (lc.on) {
                const uint32_t sl = span_of(child2(n2, n, 0)), sr = span_of(child2(n2, n, 1));
                const uint32_t cl = sl & 7u, cr = sr & 7u;
                if (n != 0 && cl && cr && cl + cr <= uint32_t(MCPT_LEAF_MAX) && (sr >> 3) == (sl >> 3) + cl) {
                    span[size_t(n)] = ((sl >> 3) << 3) | (cl + cr);
                    const double as_leaf = leaf_cost(own_area(n), cl + cr);
                    if (as_leaf < cost[size_t(n) * (K + 1) + 1]) { cost[size_t(n) * (K + 1) + 1] = as_leaf; p.merged_leaf[size_t(n)] = ~int(span[size_t(n)]); }
                }
            }
            for (int i = 2; i <= K; i++) {
                const double d = distribute(i, a), keep = cost[size_t(n) * (K + 1) + i - 1];
                if (d < keep) { cost[size_t(n) * (K + 1) + i] = d; p.split[size_t(n) * (K + 1) + i] = (unsigned char)a; }
                else { cost[size_t(n) * (K + 1) + i] = keep; p.split[size_t(n) * (K + 1) + i] = 0; }
            }
        }
    };
    int top = N;
    if (subtree_begin.size() > 1 && N >= (1 << 16)) {
        top = subtree_begin[0];
        const uint32_t ns = uint32_t(subtree_begin.size());
        std::atomic<uint32_t> next{0};
        auto worker = [&]() { for (uint32_t k = next.fetch_add(1); k < ns; k = next.fetch_add(1)) range(subtree_begin[k], k + 1 < ns ? subtree_begin[k + 1] : N); };
        const uint32_t nt = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        std::vector<std::thread> th;
        for (uint32_t t = 1; t < nt; t++) { try { th.emplace_back(worker); } catch (const std::system_error&) { break; } }
        worker();
        for (auto& t : th) t.join();
    }
    range(0, top);
    return p;
}
// The children a K-wide node adopts under the plan: the roots of the forest below binary node `node2` that fills its K slots.
struct Kid { int code; Box3f box; };
void planned_kids(const std::vector<f4h>& n2, const CollapsePlan& plan, int node2, std::vector<Kid>& kids) {
    struct Item { int parent, k, slots; };
    std::vector<Item> todo;
    { const int a = plan.at(node2, 1); todo.push_back({node2, 1, plan.K - a}); todo.push_back({node2, 0, a}); }
    while (!todo.empty()) {
        const Item it = todo.back(); todo.pop_back();
        const int c = child2(n2, it.parent, it.k);
        int i = it.slots;
        if (c >= 0) while (i > 1 && plan.at(c, i) == 0) i--;
        if (c < 0 || i == 1) { kids.push_back({c >= 0 && plan.leaf_of(c) ? plan.leaf_of(c) : c, box2(n2, it.parent, it.k)}); continue; }
        const int a = plan.at(c, i);
        todo.push_back({c, 1, i - a}); todo.push_back({c, 0, a});
    }
}


// ---- 8-wide compressed BVH for the wavefront trace kernel (device_scene.h: nodes8) --------------------------------------------
// Ylitie, Karras & Laine 2017 ("Efficient incoherent ray traversal on GPUs through compressed wide BVHs"), laid out for gfx950.
// The binary SAH tree is collapsed SAH-optimally to 8 children per node (plan_collapse; binary leaves stay as the builder made them).
// A node's children sit in OCTANT SLOTS: slot bit a set = the child lies towards +a of the node's centre (greedy assignment), so a ray
// with direction octant `oct` meets the children roughly front to back in the order of slot ^ oct -- the kernel needs no distance sort and
// a whole node's pending children are ONE stack entry (child base + hit mask).  Inner children are numbered consecutively in slot order
// (child = child_base + popcount(imask & below(slot))), and the triangles of a node's leaf children are consecutive in slot order
// (triangle = tri_base + sum of the counts below the slot): the leaf order of every triangle stream is therefore defined HERE, and
// `order` plus the leaf codes of the binary tree are rewritten to it.  One node = 80 B = five 16-B records, see device_scene.h.
inline uint32_t bf16_pow2(int e) { return uint32_t(e + 127) << 7; }     // 2^e as a bfloat16 (the top half of the fp32)
void build_bvh8(HostScene& out, std::vector<int>& order) {
    static_assert(MCPT_LEAF_MAX <= 3, "a leaf child's triangle count is two bits in the 8-wide node");
    std::vector<f4h>& n2 = out.nodes;
    std::vector<f4h>& n8 = out.nodes8;
    const auto tb0 = std::chrono::steady_clock::now();
    LeafCosts lc;                                                       // developer knobs: leaf formation in the collapse (see LeafCosts)
    if (const char* e = std::getenv("MCPT_DP_LEAF_VISIT")) { lc.on = true; lc.visit = std::atof(e); }
    if (const char* e = std::getenv("MCPT_DP_LEAF_TRI")) { lc.on = true; lc.tri = std::atof(e); }
    const CollapsePlan plan = plan_collapse(n2, 8, out.subtree_begin, lc);
    if (std::getenv("MCPT_BUILD_DEBUG")) fprintf(stderr, "[build]   collapse plan (dynamic programme) %.0f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count());
    // Breadth-first, one level at a time: the nodes of a level are independent (children gathered from the plan, octant slots, quantised
    // planes: all threads), then one serial sweep over the level hands out the children's record numbers and the leaf triangles' positions
    // in node order -- so the numbering is the serial one, whatever the thread count.
    struct Work { int node2, rec; };
    struct Emit { int inner2[8]; uint32_t leaf_first[8]; unsigned char leaf_cnt[8]; unsigned char n_inner, n_leaf; };
    std::vector<Work> level{{0, 0}}, next_level;
    std::vector<Emit> emit;
    n8.assign(5, f4h{0.f, 0.f, 0.f, 0.f});
    out.bvh8_depth = 0;
    std::vector<int> new_order; new_order.reserve(order.size());
    std::vector<int> new_first(order.size() + 1, -1);                   // old first position of a leaf -> its new one
    while (!level.empty()) {
        out.bvh8_depth++;
        emit.assign(level.size(), Emit());
        parallel_for(uint32_t(level.size()), [&](uint32_t i_begin, uint32_t i_end) {
        std::vector<Kid> kids;
        for (uint32_t wi = i_begin; wi < i_end; wi++) {
        const Work w = level[wi];
        kids.clear();
        planned_kids(n2, plan, w.node2, kids);
        for (size_t i = 0; i < kids.size();) { if (kids[i].code < 0 && ((uint32_t(~kids[i].code)) & 7u) == 0) kids.erase(kids.begin() + i); else i++; }   // (the empty second child of a one-leaf scene)
        const int nk = int(kids.size());
        float lo[3], hi[3];
        for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; }
        for (const Kid& k : kids) for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], k.box.lo[a]); hi[a] = std::max(hi[a], k.box.hi[a]); }
        if (kids.empty()) { for (int a = 0; a < 3; a++) lo[a] = hi[a] = 0.f; }
        // octant slots: greedily give the (child, slot) pair with the largest projection of the child's offset from the node centre on the slot's diagonal
        int kid_in[8]; for (int sl = 0; sl < 8; sl++) kid_in[sl] = -1;
        {
            double cost[8][8]; bool ck[8] = {false}, cs[8] = {false};
            for (int k = 0; k < nk; k++) {
                double cc[3];
                for (int a = 0; a < 3; a++) cc[a] = 0.5 * (double(kids[k].box.lo[a]) + kids[k].box.hi[a]) - 0.5 * (double(lo[a]) + hi[a]);
                for (int sl = 0; sl < 8; sl++) { double v = 0; for (int a = 0; a < 3; a++) v += ((sl >> a) & 1) ? cc[a] : -cc[a]; cost[k][sl] = v; }
            }
            for (int r = 0; r < nk; r++) {
                int bk = -1, bs = -1; double bv = -INFINITY;
                for (int k = 0; k < nk; k++) if (!ck[k]) for (int sl = 0; sl < 8; sl++) if (!cs[sl] && (bk < 0 || cost[k][sl] > bv)) { bv = cost[k][sl]; bk = k; bs = sl; }
                ck[bk] = cs[bs] = true; kid_in[bs] = bk;
            }
        }
        // The frame: a power-of-two step and a STORED origin 1024 + 2 margins steps below the lowest child bound.  The trace kernel forms a
        // plane's distance as (1024 + q) a + c with q riding in the low mantissa byte of the fp16 value 1024 + q (wavefront.hip, WF8_CHILD), so
        // plane q lies at origin + (1024 + q) step; c carries a rounding of up to 6e-5 steps, and every quantised plane -- also those of children
        // that touch the node's faces -- is kept at least MCPT_Q_MARGIN steps outside its box.  The origin is rounded to fp32 FIRST and the
        // planes are quantised against the rounded value: its rounding costs nothing.
        int ebits[3]; double scale[3];
        for (int a = 0; a < 3; a++) {
            const double ext = double(hi[a]) - double(lo[a]);
            int e = ext > 0 ? int(std::ceil(std::log2(ext / 255.0))) : -100;
            e = std::max(-126, std::min(127, e));
            float org = lo[a];
            for (;; e++) {
                const double sc = std::ldexp(1.0, e), of = double(lo[a]) - (1024.0 + 2.0 * MCPT_Q_MARGIN) * sc;
                org = float(of); if (double(org) > of) org = std::nextafterf(org, -INFINITY);
                if (!(ext > 0) || e >= 127 || (double(hi[a]) - double(org)) / sc - 1024.0 + MCPT_Q_MARGIN <= 255.0) break;
            }
            lo[a] = org; ebits[a] = e; scale[a] = std::ldexp(1.0, e);
        }
        // empty slots keep an inverted box (lo 255, hi 0): no ray interval survives it
        uint32_t q[3][4];                                                            // [axis][slot pair]: bytes lo, lo, hi, hi (device_scene.h: MCPT_N8_*)
        for (int a = 0; a < 3; a++) for (int j = 0; j < 4; j++) q[a][j] = MCPT_N8_EMPTY_WORD;
        uint32_t imask = 0, p0 = 0, p1 = 0;
        Emit& em = emit[wi];
        for (int sl = 0; sl < 8; sl++) {
            const int k = kid_in[sl]; if (k < 0) continue;
            for (int a = 0; a < 3; a++) {
                double ql = std::floor((double(kids[k].box.lo[a]) - double(lo[a])) / scale[a] - 1024.0 - MCPT_Q_MARGIN);      // (lo[] holds the stored origin now)
                double qh = std::ceil((double(kids[k].box.hi[a]) - double(lo[a])) / scale[a] - 1024.0 + MCPT_Q_MARGIN);
                ql = std::min(255.0, std::max(0.0, ql)); qh = std::min(255.0, std::max(0.0, qh));
                uint32_t& word = q[a][MCPT_N8_WORD(sl)];
                word = (word & ~(0xffu << MCPT_N8_LO_SHIFT(sl)) & ~(0xffu << MCPT_N8_HI_SHIFT(sl))) | (uint32_t(ql) << MCPT_N8_LO_SHIFT(sl)) | (uint32_t(qh) << MCPT_N8_HI_SHIFT(sl));
            }
            if (kids[k].code >= 0) { imask |= 1u << sl; em.inner2[em.n_inner++] = kids[k].code; }        // inner children: records in slot order
            else {                                                                   // leaf children: triangles in slot order
                const uint32_t leaf = uint32_t(~kids[k].code), cnt = leaf & 7u;
                p0 |= (cnt & 1u) << sl; p1 |= ((cnt >> 1) & 1u) << sl;
                em.leaf_first[em.n_leaf] = leaf >> 3; em.leaf_cnt[em.n_leaf++] = (unsigned char)cnt;
            }
        }
        f4h* r = &n8[5 * size_t(w.rec)];
        r[0] = {lo[0], lo[1], lo[2], from_u32((bf16_pow2(ebits[0]) << 16) | bf16_pow2(ebits[1]))};
        r[1] = {0.f, 0.f, from_u32(bf16_pow2(ebits[2]) << 16), from_u32(imask | (p0 << 8) | (p1 << 16) | ((p0 | p1) << 24))};   // child_base / tri_base: the sweep below
        r[2] = {from_u32(q[0][0]), from_u32(q[0][1]), from_u32(q[0][2]), from_u32(q[0][3])};   // x: slots 0-1, 2-3, 4-5, 6-7
        r[3] = {from_u32(q[1][0]), from_u32(q[1][1]), from_u32(q[1][2]), from_u32(q[1][3])};   // y
        r[4] = {from_u32(q[2][0]), from_u32(q[2][1]), from_u32(q[2][2]), from_u32(q[2][3])};   // z
        }
        });
        // the serial sweep: record numbers of the next level, triangle positions
        next_level.clear();
        size_t n_rec = n8.size() / 5;
        for (size_t wi = 0; wi < level.size(); wi++) {
            const Emit& em = emit[wi];
            f4h* r = &n8[5 * size_t(level[wi].rec)];          // (n8 grows only after this loop)
            r[1].x = from_u32(uint32_t(n_rec)); r[1].y = from_u32(uint32_t(new_order.size()));
            for (int i = 0; i < em.n_inner; i++) next_level.push_back({em.inner2[i], int(n_rec++)});
            for (int i = 0; i < em.n_leaf; i++) {
                for (uint32_t t = 0; t < em.leaf_cnt[i]; t++) { new_first[em.leaf_first[i] + t] = int(new_order.size()); new_order.push_back(order[em.leaf_first[i] + t]); }   // (every position: a merged leaf holds several binary leaves)
            }
        }
        n8.resize(5 * n_rec, f4h{0.f, 0.f, 0.f, 0.f});
        if (std::getenv("MCPT_BUILD_DEBUG")) { size_t ni = 0, nl = 0, nt = 0; for (const Emit& em : emit) { ni += em.n_inner; nl += em.n_leaf; for (int i = 0; i < em.n_leaf; i++) nt += em.leaf_cnt[i]; } fprintf(stderr, "[build]   level %u: %zu nodes, %zu inner + %zu leaf children (%.2f of 8 slots), %zu triangles\n", out.bvh8_depth, emit.size(), ni, nl, double(ni + nl) / double(std::max<size_t>(1, emit.size())), nt); }
        level.swap(next_level);
    }
    // the new leaf order: every triangle stream and the binary tree's leaf codes follow it
    const size_t nb = n2.size() / 4;
    for (size_t n = 0; n < nb; n++) for (int k = 0; k < 2; k++) {
        const int c = child2(n2, int(n), k);
        if (c >= 0) continue;
        const uint32_t leaf = uint32_t(~c), first = leaf >> 3, cnt = leaf & 7u;
        if (cnt == 0) continue;
        const int code = leaf_code(uint32_t(new_first[first]), cnt);
        (k == 0 ? n2[4 * n + 3].x : n2[4 * n + 3].y) = as_float(code);
    }
    order.swap(new_order);
}

}  // namespace

// Self-check used by mcpt_check_scene -- the soundness walk of the 8-wide tree (every leaf triangle's fp32 test data inside every box on its
// root path, every triangle referenced exactly once, child links in range; empty string = sound): boxes dequantised the way wf_trace8_kernel does, children and triangles located the way it
// locates them (child_base + rank among the inner slots; tri_base + the counts of the lower leaf slots).
std::string validate_bvh8(const HostScene& hs) {
    const size_t n8 = hs.nodes8.size() / 5, nt = hs.tri_face.size();
    if (n8 == 0) return "empty nodes8";
    std::vector<uint8_t> seen(nt, 0);
    struct Item { uint32_t node; float lo[3], hi[3]; };
    std::vector<Item> stack;
    Item root; root.node = 0; for (int a = 0; a < 3; a++) { root.lo[a] = -INFINITY; root.hi[a] = INFINITY; }
    stack.push_back(root);
    size_t visited = 0;
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        if (size_t(it.node) >= n8) return "child link out of range";
        if (++visited > n8) return "cycle in nodes8";
        const f4h* r = &hs.nodes8[5 * size_t(it.node)];
        const uint32_t sxy = as_u32(r[0].w), szw = as_u32(r[1].z), masks = as_u32(r[1].w), child_base = as_u32(r[1].x), tri_base = as_u32(r[1].y);
        const float sc[3] = {from_u32(sxy & 0xffff0000u), from_u32(sxy << 16), from_u32(szw & 0xffff0000u)};
        const float org[3] = {r[0].x, r[0].y, r[0].z};
        const uint32_t imask = masks & 0xffu, p0 = (masks >> 8) & 0xffu, p1 = (masks >> 16) & 0xffu;
        if (imask & (p0 | p1)) return "a slot is both inner and leaf";
        const f4h* Q = r + 2;
        for (int sl = 0; sl < 8; sl++) {
            const bool inner = (imask >> sl) & 1u; const uint32_t cnt = ((p0 >> sl) & 1u) + 2u * ((p1 >> sl) & 1u);
            const uint32_t below = (1u << sl) - 1u;
            Item ch;
            float own_lo[3], own_hi[3];
            bool inverted = false;
            for (int a = 0; a < 3; a++) {
                const float* words = &Q[a].x;
                const uint32_t word = as_u32(words[MCPT_N8_WORD(sl)]);
                const float qlo = float((word >> MCPT_N8_LO_SHIFT(sl)) & 0xffu), qhi = float((word >> MCPT_N8_HI_SHIFT(sl)) & 0xffu);
                if (qlo > qhi) inverted = true;
                const float lo = org[a] + (1024.0f + qlo) * sc[a], hi = org[a] + (1024.0f + qhi) * sc[a];      // plane q = stored origin + (1024 + q) steps
                ch.lo[a] = std::max(it.lo[a], lo); ch.hi[a] = std::min(it.hi[a], hi);
                own_lo[a] = lo; own_hi[a] = hi;
            }
            if (!inner && cnt == 0) { if (!inverted) return "an empty slot has a box a ray could enter"; continue; }
            if (inverted) return "an occupied slot has an inverted box";
            if (inner) { ch.node = child_base + uint32_t(__builtin_popcount(imask & below)); stack.push_back(ch); continue; }
            const uint32_t first = tri_base + uint32_t(__builtin_popcount(p0 & below)) + 2u * uint32_t(__builtin_popcount(p1 & below));
            if (size_t(first) + cnt > nt) return "leaf range out of bounds";
            for (uint32_t t = first; t < first + cnt; t++) {
                if (seen[t]++) return "triangle referenced twice";
                const f4h v0 = hs.tri_isect[3 * size_t(t)], e1 = hs.tri_isect[3 * size_t(t) + 1], e2 = hs.tri_isect[3 * size_t(t) + 2];
                const float P[3][3] = {{v0.x, v0.y, v0.z}, {v0.x + e1.x, v0.y + e1.y, v0.z + e1.z}, {v0.x + e2.x, v0.y + e2.y, v0.z + e2.z}};
                for (int c = 0; c < 3; c++) for (int a = 0; a < 3; a++) {
                    if (!(P[c][a] >= ch.lo[a] && P[c][a] <= ch.hi[a])) return "triangle " + std::to_string(t) + " sticks out of a quantised box on its path";
                    // the leaf's own planes keep MCPT_Q_MARGIN steps of distance (half of it asked for here, less the fp32 rounding of this very reconstruction)
                    const float slack = float(0.5 * MCPT_Q_MARGIN) * sc[a] - 4.0f * std::max(std::fabs(org[a]), std::fabs(P[c][a])) * 1.2e-7f;
                    if (!(P[c][a] - own_lo[a] >= slack && own_hi[a] - P[c][a] >= slack)) return "triangle " + std::to_string(t) + " closer than the quantisation margin to a plane of its leaf box";
                }
            }
        }
    }
    for (size_t t = 0; t < nt; t++) if (!seen[t]) return "triangle " + std::to_string(t) + " not reachable";
    return "";
}

// MCPT_FLAG_REFERENCE_TIE_ORDER: where every face ends up in the reference's BVH::triangles.  BVH::build (BVH.cpp:15-54) partitions the
// range [l, r) about the midpoint (narrowed to float, :39) of the CENTROID box's longest axis (AABB::max_axis, AABB.cpp:12-23) with
// std::partition -- the same libstdc++ algorithm `oracle/_ref` is built with, so the order is the same element for element -- halves the
// range by count where that leaves a side empty (:46-48), and stops at ranges of <= 5 (BVH.h:32).  BVH_node::hit (BVH.cpp:95-113) visits
// left, right, then the node's own triangles with a strict `t < t2`: among exact ties the triangle that comes first in this order wins.
// Ranges are disjoint, so an explicit stack replaces the recursion (a midpoint split can be as lopsided as 1 : n - 1).
static void reference_triangle_order(const mcpt_scene_desc* d, std::vector<uint32_t>& rank) {
    const uint32_t nf = d->n_face;
    std::vector<double> cen(3 * size_t(nf));
    for (uint32_t f = 0; f < nf; f++) {
        const int32_t* c = d->face + 12 * size_t(f);
        const double* v0 = d->vertex + 3 * size_t(c[0]); const double* v1 = d->vertex + 3 * size_t(c[4]); const double* v2 = d->vertex + 3 * size_t(c[8]);
        for (int a = 0; a < 3; a++) cen[3 * size_t(f) + a] = (v0[a] + v1[a] + v2[a]) / 3.0;                 // Triangle::center (Triangle.cpp:30-33)
    }
    std::vector<int> ord(nf);
    for (uint32_t f = 0; f < nf; f++) ord[f] = int(f);
    std::vector<std::pair<int, int>> todo; todo.push_back({0, int(nf)});
    while (!todo.empty()) {
        const int l = todo.back().first, r = todo.back().second; todo.pop_back();
        if (r - l <= 5) continue;
        double A[3], B[3];
        for (int a = 0; a < 3; a++) { A[a] = std::numeric_limits<double>::max(); B[a] = std::numeric_limits<double>::lowest(); }          // AABB.h:14
        for (int i = l; i < r; i++) for (int a = 0; a < 3; a++) { const double c = cen[3 * size_t(ord[i]) + a]; A[a] = std::min(A[a], c); B[a] = std::max(B[a], c); }
        int axis = 0; double len = B[0] - A[0];
        for (int a = 0; a < 3; a++) { const double t = B[a] - A[a]; if (len < t) { len = t; axis = a; } }
        const float mid_val = float((A[axis] + B[axis]) / 2.0);
        auto mid = std::partition(ord.begin() + l, ord.begin() + r, [&](int f) { return cen[3 * size_t(f) + axis] < mid_val; });
        int m = int(mid - ord.begin());
        if (m == l || m == r) m = (l + r) / 2;
        todo.push_back({l, m}); todo.push_back({m, r});
    }
    rank.resize(nf);
    for (uint32_t i = 0; i < nf; i++) rank[size_t(ord[i])] = i;
}

mcpt_status build_host_scene(const mcpt_scene_desc* d, HostScene& out, std::string& err, const BvhBuildFn& custom_bvh, const Collapse8Fn& custom_collapse8) {
    if (!d || !d->vertex || !d->normal || !d->texcoord || !d->face || !d->materials || !d->textures) { err = "null pointer in mcpt_scene_desc"; return MCPT_ERR_INVALID_ARG; }
    if (d->n_face == 0 || d->n_materials == 0 || d->n_textures == 0) { err = "empty scene"; return MCPT_ERR_INVALID_ARG; }
    if (d->camera.width <= 0 || d->camera.height <= 0) { err = "camera width/height must be positive"; return MCPT_ERR_INVALID_ARG; }
    if (uint64_t(d->camera.width) * uint64_t(d->camera.height) >= uint64_t(MCPT_FASTDIV_MAX)) { err = "film too large (limit 2^30 - 1 pixels)"; return MCPT_ERR_UNSUPPORTED; }
    if (d->n_face >= (1u << 28)) { err = "too many faces (limit 2^28-1)"; return MCPT_ERR_UNSUPPORTED; }
    const uint32_t nf = d->n_face;

    // ---- materials + textures
    out.texels.clear(); out.mats.clear();
    std::vector<int32_t> tex_off(d->n_textures);
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const mcpt_texture& t = d->textures[i];
        if (t.width <= 0 || t.height <= 0 || !t.rgb) { err = "bad texture " + std::to_string(i); return MCPT_ERR_INVALID_ARG; }
        tex_off[i] = int32_t(out.texels.size());
        const size_t n = size_t(t.width) * t.height;
        for (size_t k = 0; k < n; k++) out.texels.push_back({t.rgb[3 * k], t.rgb[3 * k + 1], t.rgb[3 * k + 2], 0.f});
    }
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const mcpt_material& m = d->materials[i];
        if (m.map_kd < 0 || uint32_t(m.map_kd) >= d->n_textures) { err = "material " + std::to_string(i) + ": map_kd out of range (the reference dereferences a null Map_Kd here)"; return MCPT_ERR_INVALID_ARG; }
        DevMaterial dm; std::memset(&dm, 0, sizeof dm);
        for (int k = 0; k < 3; k++) { dm.ks[k] = float(m.ks[k]); dm.radiance[k] = float(m.radiance[k]); }
        dm.ns = float(m.ns);
        if (len3(m.ks) != 0.0) { dm.flags |= MAT_HAS_SPEC; if (m.ns >= 10000) dm.flags |= MAT_MIRROR; }   // BSDF.cpp:96-98
        const double rl = len3(m.radiance);
        if (rl != 0.0) dm.flags |= MAT_EMISSIVE;
        if (rl > 0.0001) dm.flags |= MAT_EMIT_0;
        if (rl > 0.01) dm.flags |= MAT_EMIT_REC;
        dm.tex_off = tex_off[m.map_kd]; dm.tex_w = d->textures[m.map_kd].width; dm.tex_h = d->textures[m.map_kd].height;
        if (dm.tex_w * dm.tex_h == 1) { dm.flags |= MAT_CONST_KD; for (int k = 0; k < 3; k++) dm.kd[k] = d->textures[m.map_kd].rgb[k]; }
        out.mats.push_back(dm);
    }

    // ---- flatten faces (Render.cpp:12-44).  Coordinates are taken relative to the fp64 centre of the scene's bounding box (DevScene::centre):
    // the first pass validates and finds the box, the second builds the triangle bounds in centred coordinates.
    {
        double lo[3] = {std::numeric_limits<double>::max(), std::numeric_limits<double>::max(), std::numeric_limits<double>::max()}, hi[3] = {-lo[0], -lo[1], -lo[2]};
        for (uint32_t f = 0; f < nf; f++) {
            const int32_t* c = d->face + 12 * size_t(f);
            for (int k = 0; k < 3; k++) {
                if (c[4 * k] < 0 || uint32_t(c[4 * k]) >= d->n_vertex) { err = "face " + std::to_string(f) + ": index out of range"; return MCPT_ERR_INVALID_ARG; }
                for (int a = 0; a < 3; a++) { const double x = d->vertex[3 * size_t(c[4 * k]) + a]; if (x < lo[a]) lo[a] = x; if (x > hi[a]) hi[a] = x; }   // (NaN compares false: caught below)
            }
        }
        const char* keep = std::getenv("MCPT_NO_RECENTRE");                // developer knob: world coordinates on the device, as in rounds 1-2
        for (int a = 0; a < 3; a++) {
            const double c = 0.5 * lo[a] + 0.5 * hi[a];
            out.centre[a] = (keep && *keep == '1') || !(std::fabs(c) <= kMaxCoord) ? 0.0 : c;
        }
    }
    const double* const ctr = out.centre;
    std::vector<BTri> bt(nf);
    for (uint32_t f = 0; f < nf; f++) {
        const int32_t* c = d->face + 12 * size_t(f);
        for (int k = 0; k < 3; k++) {
            if (c[4 * k] < 0 || uint32_t(c[4 * k]) >= d->n_vertex || c[4 * k + 1] < 0 || uint32_t(c[4 * k + 1]) >= d->n_normal ||
                c[4 * k + 2] < 0 || uint32_t(c[4 * k + 2]) >= d->n_texcoord) { err = "face " + std::to_string(f) + ": index out of range"; return MCPT_ERR_INVALID_ARG; }
        }
        if (c[3] < 0 || uint32_t(c[3]) >= d->n_materials) { err = "face " + std::to_string(f) + ": material out of range"; return MCPT_ERR_INVALID_ARG; }
        BTri& T = bt[f];
        for (int a = 0; a < 3; a++) {
            const double w0 = d->vertex[3 * size_t(c[0]) + a], w1 = d->vertex[3 * size_t(c[4]) + a], w2 = d->vertex[3 * size_t(c[8]) + a];
            const double x0 = w0 - ctr[a], x1 = w1 - ctr[a], x2 = w2 - ctr[a];
            // NaN / inf coordinates have no order (the builders' partitions need one) and anything past 1e18 overflows the fp32 boxes' areas
            if (!(std::fabs(w0) <= kMaxCoord && std::fabs(w1) <= kMaxCoord && std::fabs(w2) <= kMaxCoord)) {
                err = "face " + std::to_string(f) + ": vertex coordinate is not finite or exceeds 1e18"; return MCPT_ERR_INVALID_ARG;
            }
            T.lo[a] = std::min(x0, std::min(x1, x2)); T.hi[a] = std::max(x0, std::max(x1, x2));
            T.c[a] = (x0 + x1 + x2) / 3.0;
        }
    }

    // ---- BVH
    auto t0 = std::chrono::steady_clock::now();
    out.nodes.clear();
    std::vector<int> order;
    bool built = false;
    if (custom_bvh && nf > uint32_t(MCPT_LEAF_MAX)) {
        std::vector<float> boxes(6 * size_t(nf));
        for (uint32_t f = 0; f < nf; f++)
            for (int a = 0; a < 3; a++) { boxes[6 * size_t(f) + a] = round_down(bt[f].lo[a], 0.f); boxes[6 * size_t(f) + 3 + a] = round_up(bt[f].hi[a], 0.f); }
        if (!custom_bvh(boxes.data(), nf, out.nodes, order, out.bvh_depth, out.max_leaf, err)) return MCPT_ERR_HIP;
        if (order.size() != nf || out.nodes.empty() || out.nodes.size() % 4 != 0) { err = "custom BVH builder returned inconsistent arrays"; return MCPT_ERR_HIP; }
        // A Morton-code tree over many coincident centroids can come out deeper than the binary-tree kernels' stack: such a scene
        // is rebuilt by the depth-capped host builder instead of being refused.
        built = out.bvh_depth <= uint32_t(MCPT_STACK_DEPTH - 1) || (out.allow_deep_binary && out.bvh_depth <= 255u);
        if (!built) { out.nodes.clear(); order.clear(); }
    }
    if (!built) {
        Builder b(bt, out.nodes, std::getenv("MCPT_BIN_LEAF") ? std::atoi(std::getenv("MCPT_BIN_LEAF")) : MCPT_LEAF_MAX);
        b.run();
        if (std::getenv("MCPT_BUILD_DEBUG")) fprintf(stderr, "[build] SAH %.0f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        out.bvh_depth = b.depth; out.max_leaf = b.max_leaf;
        order = b.order();
    }
    {   // Renumber: the first MCPT_TOP_NODES nodes in breadth-first order (= the top ~10 levels, which the trace kernel keeps in
        // LDS), every subtree below them in depth-first order (children next to parents -> cache-line locality in L1/L2).
        const int n_nodes = int(out.nodes.size() / 4);
        std::vector<int> new_id(n_nodes, -1), order_new; order_new.reserve(n_nodes);
        auto child = [&](int n, int k) { int c; std::memcpy(&c, k == 0 ? &out.nodes[4 * size_t(n) + 3].x : &out.nodes[4 * size_t(n) + 3].y, 4); return c; };
        std::vector<int> queue{0}; size_t qh = 0;
        while (qh < queue.size() && int(order_new.size()) + int(queue.size() - qh) <= MCPT_TOP_NODES) {
            const int n = queue[qh++]; new_id[n] = int(order_new.size()); order_new.push_back(n);
            for (int k = 0; k < 2; k++) { const int c = child(n, k); if (c >= 0) queue.push_back(c); }
        }
        std::vector<int> stack;
        for (size_t i = queue.size(); i-- > qh;) stack.push_back(queue[i]);           // remaining frontier, in BFS order
        const size_t frontier_left = stack.size();
        out.subtree_begin.clear();
        while (!stack.empty()) {
            const int n = stack.back(); stack.pop_back();
            if (stack.size() < frontier_left - out.subtree_begin.size()) out.subtree_begin.push_back(int(order_new.size()));   // a frontier node: its whole subtree follows, contiguously
            new_id[n] = int(order_new.size()); order_new.push_back(n);
            const int c0 = child(n, 0), c1 = child(n, 1);
            if (c1 >= 0) stack.push_back(c1);
            if (c0 >= 0) stack.push_back(c0);
        }
        std::vector<f4h> renum(out.nodes.size());
        for (int i = 0; i < n_nodes; i++) {
            const int o = order_new[i];
            for (int q = 0; q < 4; q++) renum[4 * size_t(i) + q] = out.nodes[4 * size_t(o) + q];
            const int c0 = child(o, 0), c1 = child(o, 1);
            renum[4 * size_t(i) + 3].x = as_float(c0 >= 0 ? new_id[c0] : c0);
            renum[4 * size_t(i) + 3].y = as_float(c1 >= 0 ? new_id[c1] : c1);
        }
        out.nodes.swap(renum);
    }
    out.bvh_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    out.binary_ok = out.bvh_depth <= uint32_t(MCPT_STACK_DEPTH - 1);
    if (!out.binary_ok && !out.allow_deep_binary) { err = "BVH depth exceeds traversal stack"; return MCPT_ERR_BVH_DEPTH; }
    out.nodes8.clear(); out.bvh8_depth = 0;
    if (custom_collapse8) { if (!custom_collapse8(out.nodes, order, out.nodes8, out.bvh8_depth, err)) return MCPT_ERR_HIP; }
    else build_bvh8(out, order);                                          // (defines the leaf order: `order` and the binary leaf codes are rewritten)
    if (std::getenv("MCPT_BUILD_DEBUG")) fprintf(stderr, "[build] 8-wide collapse done at %.0f ms (%zu nodes, depth %u)\n",
                                                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), out.nodes8.size() / 5, out.bvh8_depth);
    std::vector<int> pos_of_face(nf);
    for (uint32_t i = 0; i < nf; i++) pos_of_face[order[i]] = int(i);

    // ---- streams in leaf order
    std::vector<uint32_t> ref_rank;
    if (out.reference_tie_order) reference_triangle_order(d, ref_rank);
    out.tri_isect.assign(3 * size_t(nf) + 3, f4h{0.f, 0.f, 0.f, 0.f});    /* + one spare record: the trace kernel fetches triangles in pairs */ out.tri_shade.assign(size_t(MCPT_TRI_SHADE_F4) * nf, f4h{0.f, 0.f, 0.f, 0.f}); out.tri_pos64.resize(9 * size_t(nf)); out.tri_face.resize(nf);
    parallel_for(nf, [&](uint32_t i_begin, uint32_t i_end) {
    for (uint32_t i = i_begin; i < i_end; i++) {
        const int f = order[i];
        const int32_t* c = d->face + 12 * size_t(f);
        const double* w0 = d->vertex + 3 * size_t(c[0]); const double* w1 = d->vertex + 3 * size_t(c[4]); const double* w2 = d->vertex + 3 * size_t(c[8]);
        const double v0[3] = {w0[0] - ctr[0], w0[1] - ctr[1], w0[2] - ctr[2]}, v1[3] = {w1[0] - ctr[0], w1[1] - ctr[1], w1[2] - ctr[2]}, v2[3] = {w2[0] - ctr[0], w2[1] - ctr[1], w2[2] - ctr[2]};
        const double* n0 = d->normal + 3 * size_t(c[1]); const double* n1 = d->normal + 3 * size_t(c[5]); const double* n2 = d->normal + 3 * size_t(c[9]);
        const double* t0_ = d->texcoord + 2 * size_t(c[2]); const double* t1_ = d->texcoord + 2 * size_t(c[6]); const double* t2_ = d->texcoord + 2 * size_t(c[10]);
        const uint32_t mflags = out.mats[size_t(c[3])].flags;
        const uint32_t lobe_class = !(mflags & MAT_HAS_SPEC) ? HIT_CLASS_DIFFUSE : (mflags & MAT_MIRROR) ? HIT_CLASS_MIRROR : HIT_CLASS_PHONG;
        // .w: lobe class | the triangle's TIE RANK -- among hits at exactly the same distance the lowest rank wins (tri_accept in the trace kernels)
        out.tri_isect[3 * size_t(i) + 0] = {float(v0[0]), float(v0[1]), float(v0[2]), as_float(int((lobe_class << HIT_CLASS_SHIFT) | (out.reference_tie_order ? ref_rank[size_t(f)] : i)))};
        out.tri_isect[3 * size_t(i) + 1] = {float(w1[0] - w0[0]), float(w1[1] - w0[1]), float(w1[2] - w0[2]), 0.f};      // (edges from the world coordinates: Triangle.cpp:25-26's values)
        out.tri_isect[3 * size_t(i) + 2] = {float(w2[0] - w0[0]), float(w2[1] - w0[1]), float(w2[2] - w0[2]), 0.f};
        f4h* S = &out.tri_shade[size_t(MCPT_TRI_SHADE_F4) * i];
        S[0] = {float(n0[0]), float(n0[1]), float(n0[2]), float(t0_[0])};
        S[1] = {float(n1[0]), float(n1[1]), float(n1[2]), float(t0_[1])};
        S[2] = {float(n2[0]), float(n2[1]), float(n2[2]), float(t1_[0])};
        S[3] = {float(t1_[1]), float(t2_[0]), float(t2_[1]), as_float(c[3])};
        for (int a = 0; a < 3; a++) { out.tri_pos64[9 * size_t(i) + a] = v0[a]; out.tri_pos64[9 * size_t(i) + 3 + a] = v1[a]; out.tri_pos64[9 * size_t(i) + 6 + a] = v2[a]; }
        {
            const double ax = v1[0] - v0[0], ay = v1[1] - v0[1], az = v1[2] - v0[2], bx = v2[0] - v0[0], by = v2[1] - v0[1], bz = v2[2] - v0[2];
            const double nx = ay * bz - by * az, ny = az * bx - bz * ax, nz = ax * by - bx * ay;
            const double pl[4] = {nx, ny, nz, nx * v0[0] + ny * v0[1] + nz * v0[2]};
            std::memcpy(&S[4], pl, sizeof pl);                              // the fp64 plane: second half of the record
        }
        out.tri_face[i] = f;
    }
    });

    // ---- lights in face order (Render.cpp:41-42)
    out.lights.clear(); out.light_pos64.clear();
    for (uint32_t f = 0; f < nf; f++) {
        const int32_t* c = d->face + 12 * size_t(f);
        const mcpt_material& m = d->materials[c[3]];
        if (!(len3(m.radiance) > 0.01)) continue;
        DevLight L; std::memset(&L, 0, sizeof L);
        L.tri = pos_of_face[f];
        const f4h e1 = out.tri_isect[3 * size_t(L.tri) + 1], e2 = out.tri_isect[3 * size_t(L.tri) + 2];
        const float cx = e1.y * e2.z - e2.y * e1.z, cy = e1.z * e2.x - e2.z * e1.x, cz = e1.x * e2.y - e2.x * e1.y;
        L.area = 0.5f * std::sqrt((cx * cx + cy * cy) + cz * cz);                       // Triangle.cpp:24-28
        for (int a = 0; a < 3; a++) {
            L.radiance[a] = float(m.radiance[a]);
            L.n0[a] = float(d->normal[3 * size_t(c[1]) + a]); L.n1[a] = float(d->normal[3 * size_t(c[5]) + a]); L.n2[a] = float(d->normal[3 * size_t(c[9]) + a]);
        }
        out.lights.push_back(L);
        out.light_pos64.insert(out.light_pos64.end(), out.tri_pos64.begin() + 9 * size_t(L.tri), out.tri_pos64.begin() + 9 * size_t(L.tri) + 9);
    }
    if (out.lights.empty()) { err = "scene has no emissive triangle (|radiance| > 0.01); the reference indexes lights[-1] here"; return MCPT_ERR_NO_LIGHTS; }

    // ---- camera constants (Render.cpp:73-75), fp64, glm operation order
    const mcpt_camera& cm = d->camera;
    DevCamera& cam = out.cam;
    const double PI_D = 3.14159265358979323846;
    cam.h = std::tan(cm.fovy * PI_D / 180.0 * 0.5) * 2.0;
    double fr[3] = {cm.lookat[0] - cm.eye[0], cm.lookat[1] - cm.eye[1], cm.lookat[2] - cm.eye[2]};
    double inv = 1.0 / len3(fr);
    for (int a = 0; a < 3; a++) { cam.front[a] = fr[a] * inv; cam.eye[a] = cm.eye[a] - ctr[a]; cam.up[a] = cm.up[a]; }   // (front from the world coordinates, like Render.cpp:74)
    double rt[3] = {cam.front[1] * cm.up[2] - cm.up[1] * cam.front[2], cam.front[2] * cm.up[0] - cm.up[2] * cam.front[0],
                    cam.front[0] * cm.up[1] - cm.up[0] * cam.front[1]};
    inv = 1.0 / len3(rt);
    for (int a = 0; a < 3; a++) cam.right[a] = rt[a] * inv;
    cam.width = cm.width; cam.height = cm.height;
    return MCPT_OK;
}
