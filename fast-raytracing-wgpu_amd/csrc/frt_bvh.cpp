// frt_bvh.cpp — host SAH-BVH over the flattened triangle list. Replaces the driver BLAS/TLAS build hidden behind
// wgpu's EXPERIMENTAL_RAY_QUERY (src/geometry.rs:35-44, src/scene/builder.rs:143-179, :454-468).
//
// Output 1: canonical BVH2 (frt_bvh2_node, 32 B) — binned SAH (16 bins, 3 axes), leaves <= 2 triangles, deterministic
//           (stable partitions, ties broken by triangle id); depth capped at kMaxBvhDepth by switching to median splits.
// Output 2: GPU layout — 64-byte pair nodes (both child boxes in the parent) + 48-byte triangle slots in leaf order.
// Boxes are padded so that box tests are strictly more permissive than the ray/triangle test (hit semantics do not
// depend on the tree: DESIGN.md §3).
#include "frt_scene.hpp"
#include "frt_shade.hpp"
#include "frt_bvh_opt.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <cstdlib>
#include <cstdio>
#include <memory>

namespace frt {

namespace {
struct Box {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; ++a) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -lo[a]; } }
    void grow(const float p[3]) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    void grow(const Box& b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};
struct Prim { Box box; float centroid[3]; uint32_t id; };

struct Builder {
    std::vector<Prim> prims;
    std::vector<frt_bvh2_node>& nodes;
    uint32_t max_depth = 0, leaves = 0, max_leaf = 0;
    float pad;
    explicit Builder(std::vector<frt_bvh2_node>& n) : nodes(n) {}

    void set_node_box(uint32_t ni, const Box& b) {
        for (int a = 0; a < 3; ++a) { nodes[ni].bmin[a] = b.lo[a] - pad; nodes[ni].bmax[a] = b.hi[a] + pad; }
    }
    static int ceil_log2(uint32_t n) { int l = 0; while ((1u << l) < n) ++l; return l; }

    void build(uint32_t ni, uint32_t first, uint32_t count, uint32_t depth) {
        max_depth = std::max(max_depth, depth);
        Box box, cbox;
        box.reset(); cbox.reset();
        for (uint32_t i = first; i < first + count; ++i) { box.grow(prims[i].box); cbox.grow(prims[i].centroid); }
        set_node_box(ni, box);
        // leaves of <= 2 triangles measured best on MI355X (4: -11 %, 1: -5 % Mrays/s on the Cornell Box); FRT_BVH_LEAF overrides for experiments
#if defined(FRT_EXPERIMENTS) && FRT_EXPERIMENTS
        const uint32_t kLeaf = getenv("FRT_BVH_LEAF") ? (uint32_t)std::max(1, std::min(4, atoi(getenv("FRT_BVH_LEAF")))) : 2u;      // (measured: 1 -> +3 %, 3 -> +4 %, 4 -> +15 % frame time)
#else
        const uint32_t kLeaf = 2u;
#endif
        if (count <= kLeaf) {
            nodes[ni].left_first = first; nodes[ni].count = count;
            ++leaves; max_leaf = std::max(max_leaf, count);
            return;
        }
        uint32_t mid = 0;
        // depth budget: a balanced finish needs ceil(log2(count / 1)) more levels at most
        bool force_median = (int)depth + ceil_log2(count) + 1 >= kMaxBvhDepth;
        int best_axis = -1;
        if (!force_median) {
            const int kBins = 16;
            float best_cost = std::numeric_limits<float>::infinity();
            int best_split = -1;
            for (int axis = 0; axis < 3; ++axis) {
                float lo = cbox.lo[axis], ext = cbox.hi[axis] - lo;
                if (!(ext > 0.0f)) continue;
                Box bb[kBins]; uint32_t bn[kBins];
                for (int b = 0; b < kBins; ++b) { bb[b].reset(); bn[b] = 0; }
                float scale = (float)kBins / ext;
                for (uint32_t i = first; i < first + count; ++i) {
                    int b = std::min(kBins - 1, (int)((prims[i].centroid[axis] - lo) * scale));
                    bb[b].grow(prims[i].box); ++bn[b];
                }
                float right_area[kBins]; uint32_t right_n[kBins];
                Box acc; acc.reset(); uint32_t n = 0;
                for (int b = kBins - 1; b > 0; --b) { acc.grow(bb[b]); n += bn[b]; right_area[b] = acc.half_area(); right_n[b] = n; }
                acc.reset(); n = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    acc.grow(bb[b]); n += bn[b];
                    if (n == 0 || right_n[b + 1] == 0) continue;
                    float cost = acc.half_area() * (float)n + right_area[b + 1] * (float)right_n[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = b; }
                }
            }
            if (best_axis >= 0) {
                float lo = cbox.lo[best_axis], scale = 16.0f / (cbox.hi[best_axis] - lo);
                auto it = std::stable_partition(prims.begin() + first, prims.begin() + first + count, [&](const Prim& p) {
                    int b = std::min(15, (int)((p.centroid[best_axis] - lo) * scale));
                    return b <= best_split;
                });
                mid = (uint32_t)(it - prims.begin());
            }
        }
        if (best_axis < 0 || mid == first || mid == first + count) {
            // median split on the widest centroid axis (ties / degenerate input / depth budget), order by (centroid, id)
            int axis = 0;
            float e0 = cbox.hi[0] - cbox.lo[0], e1 = cbox.hi[1] - cbox.lo[1], e2 = cbox.hi[2] - cbox.lo[2];
            if (e1 > e0 && e1 >= e2) axis = 1; else if (e2 > e0 && e2 > e1) axis = 2;
            mid = first + count / 2;
            std::sort(prims.begin() + first, prims.begin() + first + count, [&](const Prim& a, const Prim& b) {
                if (a.centroid[axis] != b.centroid[axis]) return a.centroid[axis] < b.centroid[axis];
                return a.id < b.id;
            });
        }
        uint32_t left = (uint32_t)nodes.size();
        nodes.push_back(frt_bvh2_node{}); nodes.push_back(frt_bvh2_node{});
        nodes[ni].left_first = left; nodes[ni].count = 0;
        build(left, first, mid - first, depth + 1);
        build(left + 1, mid, first + count - mid, depth + 1);
    }
};
} // namespace

#ifndef FRT_BVH_OPT_MIN
#define FRT_BVH_OPT_MIN 8192      // (A/B builds: 0 = every scene goes through the insertion pass)
#endif
static const size_t kBvhOptMinTris = FRT_BVH_OPT_MIN;

void SceneBuilder::build_bvh2() {
    bvh2.clear(); bvh2_tri_index.clear();
    bvh_depth = bvh_leaves = bvh_max_leaf = 0;
    if (tris.empty()) { error = "scene has no triangles"; return; }
    if (tris.size() >= (1u << 24)) { error = "more than 2^24 triangles"; return; }
    Builder b(bvh2);
    b.prims.resize(tris.size());
    Box scene_box; scene_box.reset();
    for (size_t i = 0; i < tris.size(); ++i) {
        const TriRec& t = tris[i];
        Prim& p = b.prims[i];
        // bounds of the triangle the intersector actually sees: v0, v0 + e1, v0 + e2
        float v1[3], v2[3];
        for (int a = 0; a < 3; ++a) { v1[a] = t.v0[a] + t.e1[a]; v2[a] = t.v0[a] + t.e2[a]; }
        p.box.reset(); p.box.grow(t.v0); p.box.grow(v1); p.box.grow(v2);
        for (int a = 0; a < 3; ++a) p.centroid[a] = 0.5f * (p.box.lo[a] + p.box.hi[a]);
        p.id = (uint32_t)i;
        scene_box.grow(p.box);
    }
    float ext = 0.0f;
    for (int a = 0; a < 3; ++a) ext = std::max(ext, std::max(fabsf(scene_box.lo[a]), fabsf(scene_box.hi[a])));
    b.pad = 1e-4f * std::max(ext, 1.0f);
    bvh2.reserve(tris.size() * 2);
    bvh2.push_back(frt_bvh2_node{});
    b.build(0, 0, (uint32_t)tris.size(), 1);
    bvh2_tri_index.resize(tris.size());
    for (size_t i = 0; i < tris.size(); ++i) bvh2_tri_index[i] = b.prims[i].id;
    bvh_depth = b.max_depth; bvh_leaves = b.leaves; bvh_max_leaf = b.max_leaf;
    // Larger scenes: one pass of insertion-based optimisation over the tree just built (frt_bvh_opt.hpp: what it buys per scene size, and why small
    // scenes are left alone). Same leaves, same triangles: pixels cannot change. Build time of the 246k-triangle scene 0.33 -> 1.1 s.
    if (tris.size() >= kBvhOptMinTris) bvh_depth = optimize_bvh2(bvh2, bvh2_tri_index, 1, (uint32_t)kMaxBvhDepth, bvh_depth);
    if ((int)bvh_depth > kMaxBvhDepth) error = "BVH depth exceeds the traversal stack";
}

// Quantized form of the pair nodes (frt_trace.hpp: QBvh): child boxes on a 16-bit grid over the box of all (padded) node boxes, rounded
// outward by one extra quantum, which covers the rounding of the grid-space slab test on the device (its error is a few ulp of a
// coordinate, a quantum is 2^-16 of the scene extent). Absent children get an inverted box.
static void quantize_pair_nodes(SceneBuilder& b) {
    const size_t n = b.pair_nodes.size();
    b.qnode_a.assign(n * 4, 0u); b.qnode_b.assign(n * 4, 0u);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (const PairNode& p : b.pair_nodes)
        for (int c = 0; c < 2; ++c) {
            uint32_t ref; memcpy(&ref, &p.q[12 + c], 4);
            if (ref == kNoChild) continue;
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], (double)p.q[4 * a + c]); hi[a] = std::max(hi[a], (double)p.q[4 * a + 2 + c]); }
        }
    for (int a = 0; a < 3; ++a) {
        if (!(hi[a] > lo[a])) { hi[a] = lo[a] + 1e-3; }
        const double ext = hi[a] - lo[a];
        b.qmin[a] = (float)(lo[a] - 1e-6 * ext);
        b.qstep[a] = (float)((ext * (1.0 + 4e-6)) / 65531.0);      // leaves room for the outward quanta at both ends
    }
    auto q_lo = [&](double v, int a) { double g = std::floor((v - (double)b.qmin[a]) / (double)b.qstep[a]) - 1.0; return (uint32_t)std::min(65535.0, std::max(0.0, g)); };
    auto q_hi = [&](double v, int a) { double g = std::ceil((v - (double)b.qmin[a]) / (double)b.qstep[a]) + 1.0; return (uint32_t)std::min(65535.0, std::max(0.0, g)); };
    for (size_t i = 0; i < n; ++i) {
        const PairNode& p = b.pair_nodes[i];
        for (int c = 0; c < 2; ++c) {
            uint32_t ref; memcpy(&ref, &p.q[12 + c], 4);
            uint32_t l[3], h[3];
            for (int a = 0; a < 3; ++a) {
                if (ref == kNoChild) { l[a] = 65535u; h[a] = 0u; }
                else { l[a] = q_lo(p.q[4 * a + c], a); h[a] = q_hi(p.q[4 * a + 2 + c], a); }
            }
            uint32_t* w = (c == 0 ? b.qnode_a.data() : b.qnode_b.data()) + 4 * i;
            w[0] = l[0] | (l[1] << 16); w[1] = l[2] | (h[0] << 16); w[2] = h[1] | (h[2] << 16); w[3] = ref;
        }
    }
}

// Quad nodes: the canonical BVH2 with every other level folded away. Starting from the two children of an inner node, the inner child
// with the largest surface area is replaced by its own two children until the node holds four children or only leaves: 2 to 4 children
// per node, the same leaves (and triangle slots) as the pair tree, about half as many dependent node fetches per ray. Layout (32 floats):
// lo.x[4], hi.x[4], lo.y[4], hi.y[4], lo.z[4], hi.z[4], reference[4], unused[4]; a reference is a quad node index or a leaf
// (kLeafFlag | count << 24 | first slot); an empty slot holds a far-away degenerate box (misses every ray) and kNoChild. Nodes are numbered
// breadth-first (the top of the tree is contiguous).
// Traversal stack: a node with k children hit pushes k - 1 references, so a folded tree can need more entries than the binary one. A fold is
// therefore only made while every inner child could still be finished as a plain binary subtree within kStackDepth entries (`used` = what the
// ancestors may already have pushed): the quad tree of any scene the builder accepts (depth <= kMaxBvhDepth) fits the stack by construction.
// Which binary nodes fold into one quad node: the surface-area dynamic programme of frt_bvh8.hpp (WideCollapse, 4 slots) when `dp`, else the greedy fold
// described above. The programme fills the nodes better (Cornell Box: 3.5 instead of 3.0 children per node, 331 instead of 390 nodes) but knows nothing of the
// traversal stack: a tree it makes too deep for the stack (the 82k- and 246k-triangle stand-ins: 36 and 39 entries) is built again with the greedy fold,
// which stays inside the budget by construction. Measured (profiles/r4_experiments/quad_dp.md): Cornell Box 1.540 -> 1.435 ms per frame (3840x2160: 5.89 ->
// 5.46), ReSTIR scene 1.269 -> 1.260; the mixed form — the programme only at the nodes where the greedy fold's worst-case bound (`fits`) lets it — gives
// the deep scenes MORE nodes than the greedy fold (colonnade 81,964 vs 71,172: 20.9 vs 19.7 ms) and is not used.
#ifndef FRT_QUAD_DP
#define FRT_QUAD_DP 1
#endif
static void build_quad_nodes(SceneBuilder& b, int dp = FRT_QUAD_DP ? 2 : 0) {      // 2: the programme everywhere; 0: the greedy fold (1: the programme where the stack bound below lets it, the greedy fold elsewhere: measured, not used)
    b.quad_nodes.clear(); b.quad_stack_need = 0;
    const std::vector<frt_bvh2_node>& t = b.bvh2;
    const uint32_t budget = (uint32_t)kStackDepth - 1u;
    std::vector<uint32_t> height(t.size(), 1u);          // levels of the binary subtree (a leaf = 1): children follow their parent in `t`
    for (size_t i = t.size(); i-- > 0;)
        if (t[i].count == 0) height[i] = 1u + std::max(height[t[i].left_first], height[t[i].left_first + 1]);
    auto half_area = [&](uint32_t ni) {
        const float dx = t[ni].bmax[0] - t[ni].bmin[0], dy = t[ni].bmax[1] - t[ni].bmin[1], dz = t[ni].bmax[2] - t[ni].bmin[2];
        return dx * dy + dy * dz + dz * dx;
    };
    struct Kids { uint32_t c[4]; int n; };
    auto fits = [&](const uint32_t* c, int n, uint32_t used) {
        if (used + (uint32_t)(n - 1) > budget) return false;      // (children that are all leaves; implied for a set the greedy fold reaches step by step)
        for (int i = 0; i < n; ++i)
            if (t[c[i]].count == 0 && used + (uint32_t)(n - 1) + (height[c[i]] - 1u) > budget) return false;
        return true;
    };
    std::unique_ptr<WideCollapse> collapse;
    if (dp) collapse.reset(new WideCollapse(t, 4));
    auto children_of = [&](uint32_t ni, uint32_t used) -> Kids {
        Kids k{};
        if (dp) { k.n = collapse->children_of(ni, k.c); if (dp == 2 || fits(k.c, k.n, used)) return k; k = Kids{}; }
        if (t[ni].count > 0) { k.c[0] = ni; k.n = 1; return k; }       // a lone leaf root
        k.n = 2; k.c[0] = t[ni].left_first; k.c[1] = t[ni].left_first + 1;
        while (k.n < 4) {
            int pick = -1; float best = -1.0f;
            for (int i = 0; i < k.n; ++i) if (t[k.c[i]].count == 0 && half_area(k.c[i]) > best) { best = half_area(k.c[i]); pick = i; }
            if (pick < 0) break;
            Kids w = k;
            for (int i = w.n; i > pick + 1; --i) w.c[i] = w.c[i - 1];
            w.c[pick] = t[k.c[pick]].left_first; w.c[pick + 1] = t[k.c[pick]].left_first + 1;
            ++w.n;
            if (!fits(w.c, w.n, used)) break;
            k = w;
        }
        return k;
    };
    std::vector<uint32_t> order(1, 0u), used(1, 0u), quad_of(t.size(), kNoChild);
    std::vector<Kids> kids;
    quad_of[0] = 0u;
    for (size_t h = 0; h < order.size(); ++h) {
        const Kids k = children_of(order[h], used[h]);
        kids.push_back(k);
        for (int i = 0; i < k.n; ++i)
            if (t[k.c[i]].count == 0) { quad_of[k.c[i]] = (uint32_t)order.size(); order.push_back(k.c[i]); used.push_back(used[h] + (uint32_t)(k.n - 1)); }
    }
    b.quad_nodes.resize(order.size());
    std::vector<uint32_t> need(order.size(), 0u);
    for (size_t h = order.size(); h-- > 0;) {      // children have larger indices than their parent: bottom-up
        const Kids& k = kids[h];
        QuadNode q{};
        uint32_t deepest = 0;
        for (int i = 0; i < 4; ++i) {
            uint32_t ref = kNoChild;
            for (int a = 0; a < 3; ++a) q.q[8 * a + i] = q.q[8 * a + 4 + i] = 1.0e30f;   // empty slot: a far-away point, misses every ray (|1 / d| >= 1)
            if (i < k.n) {
                const frt_bvh2_node& c = t[k.c[i]];
                for (int a = 0; a < 3; ++a) { q.q[8 * a + i] = c.bmin[a]; q.q[8 * a + 4 + i] = c.bmax[a]; }
                if (c.count > 0) ref = kLeafFlag | (c.count << 24) | c.left_first;
                else { ref = quad_of[k.c[i]]; deepest = std::max(deepest, need[ref]); }
            }
            memcpy(&q.q[24 + i], &ref, 4);
        }
        need[h] = (uint32_t)(k.n - 1) + deepest;
        b.quad_nodes[h] = q;
    }
    b.quad_stack_need = need[0];
    b.quad_fold = (uint32_t)dp;
    if (getenv("FRT_DEBUG_FOLD")) fprintf(stderr, "quad fold mode %d: %zu nodes, stack need %u (budget %u)\n", dp, order.size(), b.quad_stack_need, budget);
    if (dp && b.quad_stack_need > budget) { build_quad_nodes(b, 0); return; }
    if (b.quad_stack_need > (uint32_t)kStackDepth - 1u) b.error = "quad tree exceeds the traversal stack";   // (cannot happen: see above; the last row of the LDS array is not a stack entry, frt_kernels.hip: kMiscRow)
}

void SceneBuilder::ensure_wide8() const {
    if (wide8_built || !built) return;
    build_wide8(bvh2, wide8);
    tri_slots8.clear();
    if (wide8.ok) { tri_slots8.resize(wide8.tri_order.size()); for (size_t i = 0; i < tri_slots8.size(); ++i) tri_slots8[i] = tri_slots[wide8.tri_order[i]]; }
    wide8_built = true;
}

void SceneBuilder::build_gpu_layout() {
    pair_nodes.clear(); tri_slots.clear(); instances_dev.clear(); shade_tris.clear();
    if (!error.empty()) return;
    // triangle slots in leaf (bvh2_tri_index) order
    tri_slots.resize(tris.size());
    for (size_t s = 0; s < tris.size(); ++s) {
        uint32_t id = bvh2_tri_index[s];
        const TriRec& t = tris[id];
        TriSlot& o = tri_slots[s];
        uint32_t inst = tri_instance[id];
        float idf, instf;
        memcpy(&idf, &id, 4); memcpy(&instf, &inst, 4);
        o.q[0] = t.v0[0]; o.q[1] = t.v0[1]; o.q[2] = t.v0[2]; o.q[3] = idf;
        o.q[4] = t.e1[0]; o.q[5] = t.e1[1]; o.q[6] = t.e1[2]; o.q[7] = instf;
        o.q[8] = t.e2[0]; o.q[9] = t.e2[1]; o.q[10] = t.e2[2]; o.q[11] = 0.0f;
    }
    // pair nodes: one per BVH2 inner node (a lone leaf root becomes a pair with an absent second child)
    std::vector<uint32_t> pair_of(bvh2.size(), kNoChild);
    auto ref_of = [&](uint32_t ni) -> uint32_t {
        const frt_bvh2_node& n = bvh2[ni];
        if (n.count > 0) return kLeafFlag | (n.count << 24) | n.left_first;
        return pair_of[ni];
    };
    // assign pair indices in BFS order so the top of the tree is contiguous at the front
    std::vector<uint32_t> order;
    if (bvh2[0].count == 0) {
        order.push_back(0);
        for (size_t h = 0; h < order.size(); ++h) {
            const frt_bvh2_node& n = bvh2[order[h]];
            for (uint32_t c = n.left_first; c < n.left_first + 2; ++c)
                if (bvh2[c].count == 0) order.push_back(c);
        }
        for (size_t i = 0; i < order.size(); ++i) pair_of[order[i]] = (uint32_t)i;
    }
    auto put_box = [](PairNode& p, int child, const frt_bvh2_node* n) {
        float lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
            lo[a] = n ? n->bmin[a] : std::numeric_limits<float>::infinity();
            hi[a] = n ? n->bmax[a] : -std::numeric_limits<float>::infinity();
        }
        // one float4 per axis: (lo of child 0, lo of child 1, hi of child 0, hi of child 1) — frt_trace.hpp: slab2
        for (int a = 0; a < 3; ++a) { p.q[4 * a + child] = lo[a]; p.q[4 * a + 2 + child] = hi[a]; }
    };
    auto put_ref = [](PairNode& p, int child, uint32_t ref) { memcpy(&p.q[12 + child], &ref, 4); };
    if (order.empty()) {
        PairNode p{};
        put_box(p, 0, &bvh2[0]); put_ref(p, 0, ref_of(0));
        put_box(p, 1, nullptr); put_ref(p, 1, kNoChild);
        pair_nodes.push_back(p);
    } else {
        pair_nodes.resize(order.size());
        for (size_t i = 0; i < order.size(); ++i) {
            const frt_bvh2_node& n = bvh2[order[i]];
            PairNode p{};
            for (int c = 0; c < 2; ++c) { put_box(p, c, &bvh2[n.left_first + c]); put_ref(p, c, ref_of(n.left_first + c)); }
            pair_nodes[i] = p;
        }
    }
    quantize_pair_nodes(*this);
    build_quad_nodes(*this);
    wide8 = Wide8{}; tri_slots8.clear(); wide8_built = false;      // (built on first use: ensure_wide8)
    // shading records: the instance -> mesh -> index -> attribute chain of gbuffer.wgsl:129-145, flattened per triangle
    shade_tris.assign(tris.size(), ShadeTri{});
    for (size_t id = 0; id < tris.size(); ++id) {
        const InstanceRec& in = instances[tri_instance[id]];
        const MeshInfo& mi = mesh_infos[in.mesh_id];
        uint32_t prim = (uint32_t)id - in.first_tri;
        ShadeTri& o = shade_tris[id];
        for (int k = 0; k < 3; ++k) {
            const frt_vertex_attr& a = attributes[indices[mi.index_offset + prim * 3u + k] + mi.vertex_offset];
            f3 n = decode_octahedral_normal(a.normal[0], a.normal[1]);   // same function the kernels would run per hit
            o.q[4 * k] = n.x; o.q[4 * k + 1] = n.y; o.q[4 * k + 2] = n.z;
            o.q[12 + 4 * k] = a.tangent[0]; o.q[12 + 4 * k + 1] = a.tangent[1]; o.q[12 + 4 * k + 2] = a.tangent[2];
            // uvs: (uv0.x, uv0.y, uv1.x, uv1.y, uv2.x, uv2.y) in the .w lanes of q0..q5
            o.q[4 * (2 * k) + 3] = a.uv[0]; o.q[4 * (2 * k + 1) + 3] = a.uv[1];
            if (k == 0) o.q[24] = a.tangent[3];
        }
        uint32_t mat = in.mat_id;
        memcpy(&o.q[25], &mat, 4);
    }
    instances_dev.resize(instances.size());
    for (size_t i = 0; i < instances.size(); ++i) {
        InstanceDev& d = instances_dev[i];
        memset(&d, 0, sizeof(d));
        d.mesh_id = instances[i].mesh_id; d.mat_id = instances[i].mat_id; d.first_tri = instances[i].first_tri; d.flip = instances[i].flip;
        memcpy(d.w2o, instances[i].w2o, sizeof(d.w2o));
    }
}

} // namespace frt
