// bvh_quality.cpp — offline measure of what a BVH costs the traced kernels, on the host (no GPU).
//
//   g++ -O2 -std=c++17 tools/bvh_quality.cpp -Iinclude -Lfast-raytracing-wgpu_amd/lib -lfrt -Wl,-rpath,$PWD/fast-raytracing-wgpu_amd/lib -o tools/_build/bvh_quality
//   tools/_build/bvh_quality [cornell|restir] [tiles] [insertion passes] [policy 0|1|2] [threshold] [presence 0..1] [split 0|1] [chain 0|1]
//
// Takes the canonical BVH2 the product built (frt_scene_get), folds it into quad nodes the way frt_bvh.cpp: build_quad_nodes does, and walks it
// with the rays of the workload — 8x8 pixel tiles of primary rays from the benchmark camera, then from every primary hit a cosine-distributed
// bounce ray, a shadow ray to a point on a light, and a second bounce — in LOCKSTEP per tile, exactly like a wave of trace4 (frt_trace.hpp): an
// inner loop in which every lane that holds a node takes a node step until no lane holds a node, then one leaf step for the lanes that hold a
// leaf. Reported per ray kind: node steps and triangle tests per lane-ray, and node steps / leaf steps per WAVE-ray (what the SIMD executes).
// With [insertion passes] > 0 the BVH2 first goes through the insertion-based optimisation of csrc/frt_bvh_opt.hpp so that the
// two trees can be compared before anything runs on the GPU.
#include "frt.h"
#include "../fast-raytracing-wgpu_amd/csrc/frt_bvh_opt.hpp"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct V3 { float x, y, z; };
static V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static V3 norm(V3 a) { float l = std::sqrt(dot(a, a)); return l > 0 ? a * (1.0f / l) : a; }

struct Tri { V3 v0, e1, e2; };
static uint64_t g_dbg[2][4];      // any-hit rays: [quad | 8-wide][unoccluded rays, their triangle tests, occluded rays, theirs]
struct Quad { float lo[3][4], hi[3][4]; uint32_t ref[4]; int n; };
static const uint32_t kLeaf = 0x80000000u, kNone = 0xFFFFFFFFu;

struct Tree {
    std::vector<frt_bvh2_node> t;
    std::vector<uint32_t> tri_index;
    std::vector<Tri> tris;          // by flattened id
    std::vector<Quad> quads;
    uint32_t stack_need = 0;
};

static float half_area(const frt_bvh2_node& n) {
    float dx = n.bmax[0] - n.bmin[0], dy = n.bmax[1] - n.bmin[1], dz = n.bmax[2] - n.bmin[2];
    return dx * dy + dy * dz + dz * dx;
}
static double sah_cost(const std::vector<frt_bvh2_node>& t) {
    double inner = 0, leaf = 0, ra = half_area(t[0]);
    for (const auto& n : t) { if (n.count) leaf += half_area(n) * n.count; else inner += half_area(n); }
    return (inner + leaf) / ra;
}
static void build_quads(Tree& T) {        // frt_bvh.cpp: build_quad_nodes without the stack-budget guard (reported instead)
    const auto& t = T.t;
    T.quads.clear();
    std::vector<uint32_t> order(1, 0u), quad_of(t.size(), kNone);
    struct Kids { uint32_t c[4]; int n; };
    std::vector<Kids> kids;
    quad_of[0] = 0;
    for (size_t h = 0; h < order.size(); ++h) {
        Kids k{};
        uint32_t ni = order[h];
        if (t[ni].count > 0) { k.c[0] = ni; k.n = 1; }
        else {
            k.n = 2; k.c[0] = t[ni].left_first; k.c[1] = t[ni].left_first + 1;
            while (k.n < 4) {
                int pick = -1; float best = -1;
                for (int i = 0; i < k.n; ++i) if (t[k.c[i]].count == 0 && half_area(t[k.c[i]]) > best) { best = half_area(t[k.c[i]]); pick = i; }
                if (pick < 0) break;
                for (int i = k.n; i > pick + 1; --i) k.c[i] = k.c[i - 1];
                uint32_t l = t[k.c[pick]].left_first;
                k.c[pick] = l; k.c[pick + 1] = l + 1; ++k.n;
            }
        }
        kids.push_back(k);
        for (int i = 0; i < k.n; ++i) if (t[k.c[i]].count == 0) { quad_of[k.c[i]] = (uint32_t)order.size(); order.push_back(k.c[i]); }
    }
    T.quads.resize(order.size());
    std::vector<uint32_t> need(order.size(), 0);
    for (size_t h = order.size(); h-- > 0;) {
        const Kids& k = kids[h];
        Quad q{}; q.n = k.n;
        uint32_t deepest = 0;
        for (int i = 0; i < 4; ++i) {
            q.ref[i] = kNone;
            for (int a = 0; a < 3; ++a) q.lo[a][i] = q.hi[a][i] = 1e30f;
            if (i < k.n) {
                const auto& c = t[k.c[i]];
                for (int a = 0; a < 3; ++a) { q.lo[a][i] = c.bmin[a]; q.hi[a][i] = c.bmax[a]; }
                if (c.count) q.ref[i] = kLeaf | (c.count << 24) | c.left_first;
                else { q.ref[i] = quad_of[k.c[i]]; deepest = std::max(deepest, need[q.ref[i]]); }
            }
        }
        need[h] = (uint32_t)(k.n - 1) + deepest;
        T.quads[h] = q;
    }
    T.stack_need = need[0];
}

// ---- one lane of trace4 as a state machine, stepped by the wave loop below
struct Lane {
    V3 o, d, inv; float tmin, tmax, best; bool any, done; uint32_t cur, hit;
    std::vector<uint32_t> stk;
    uint64_t node_steps = 0, tri_tests = 0;
    int grp_size = 1, grp_rank = 0;      // ray splitting: this lane walks every grp_size-th hit child of the root, starting with the grp_rank-th nearest
    float* shared_best = nullptr;        // closest-hit rays: the group's common tmax (a cross-lane min per step on the GPU)
    bool has_next = false; V3 no, nd; float ntmin = 0, ntmax = 0; bool nany = false;      // chained walk: the lane's NEXT ray starts when this one ends, without waiting for the wave
    bool first_occluded = false; uint64_t node_steps_total = 0, tri_tests_total = 0;
    void start(V3 o_, V3 d_, float tmin_, float tmax_, bool any_) {
        o = o_; d = d_; tmin = tmin_; tmax = tmax_; any = any_; best = tmax_; done = false; cur = 0; hit = kNone; stk.clear(); grp_size = 1; grp_rank = 0; shared_best = nullptr; has_next = false;
        auto rc = [](float x) { const float k = 8.271806125530277e-25f; return 1.0f / (std::fabs(x) > k ? x : std::copysign(k, x)); };
        inv = {rc(d.x), rc(d.y), rc(d.z)};
        node_steps = tri_tests = 0;
    }
    void chain(V3 o_, V3 d_, float tmin_, float tmax_, bool any_) { has_next = true; no = o_; nd = d_; ntmin = tmin_; ntmax = tmax_; nany = any_; }
    void finish() {      // this ray is over: start the chained one, if any
        if (!has_next) { done = true; return; }
        first_occluded = hit != kNone;
        const uint64_t ns = node_steps, ts = tri_tests;
        start(no, nd, ntmin, ntmax, nany);
        node_steps = ns; tri_tests = ts;
    }
    bool at_node() const { return !done && !(cur & kLeaf); }
    bool at_leaf() const { return !done && (cur & kLeaf) && cur != kNone; }
    void pop() { if (stk.empty()) finish(); else { cur = stk.back(); stk.pop_back(); } }
    void node_step(const Tree& T) {
        ++node_steps;
        const Quad& q = T.quads[cur];
        float key[4]; uint32_t r[4];
        const float o3[3] = {o.x, o.y, o.z}, i3[3] = {inv.x, inv.y, inv.z};
        for (int c = 0; c < 4; ++c) {
            float tn = tmin, tf = any ? tmax : (shared_best ? std::min(best, *shared_best) : best);
            for (int a = 0; a < 3; ++a) {
                float t0 = (q.lo[a][c] - o3[a]) * i3[a], t1 = (q.hi[a][c] - o3[a]) * i3[a];
                tn = std::max(tn, std::min(t0, t1)); tf = std::min(tf, std::max(t0, t1));
            }
            key[c] = tn <= tf ? tn : 3e38f; r[c] = q.ref[c];
        }
        auto ce = [&](int a, int b) { if (key[b] < key[a]) { std::swap(key[a], key[b]); std::swap(r[a], r[b]); } };
        ce(0, 1); ce(2, 3); ce(0, 2); ce(1, 3); ce(1, 2);
        if (cur == 0 && grp_size > 1) {      // the root's hit children, near to far, dealt round-robin to the lanes of the group
            int nh = 0; while (nh < 4 && key[nh] < 3e38f) ++nh;
            std::vector<uint32_t> mine;
            for (int c = grp_rank; c < nh; c += grp_size) mine.push_back(r[c]);
            for (size_t c = mine.size(); c-- > 1;) stk.push_back(mine[c]);
            if (!mine.empty()) cur = mine[0]; else pop();
            return;
        }
        for (int c = 3; c >= 1; --c) if (key[c] < 3e38f) stk.push_back(r[c]);
        if (key[0] < 3e38f) cur = r[0]; else pop();
    }
    void leaf_step(const Tree& T) {
        uint32_t first = cur & 0xFFFFFFu, count = (cur >> 24) & 0x7F;
        for (uint32_t k = 0; k < count; ++k) {
            ++tri_tests;
            uint32_t id = T.tri_index[first + k];
            const Tri& tr = T.tris[id];
            V3 p = cross(d, tr.e2); float det = dot(tr.e1, p);
            if (det == 0) continue;
            float iv = 1.0f / det; V3 s = o - tr.v0; float u = dot(s, p) * iv;
            if (!(u >= 0 && u <= 1)) continue;
            V3 qq = cross(s, tr.e1); float v = dot(d, qq) * iv;
            if (!(v >= 0 && u + v <= 1)) continue;
            float t = dot(tr.e2, qq) * iv;
            if (!(t > tmin && t < tmax)) continue;
            if (getenv("NOCULL") && any) continue;
            if (any) { hit = id; finish(); return; }
            if (t < best || (t == best && id < hit)) { best = t; hit = id; if (shared_best && t < *shared_best) *shared_best = t; }
        }
        pop();
    }
};
struct WaveCost { uint64_t checksum = 0, rays = 0, lane_nodes = 0, lane_tris = 0, wave_rays = 0, wave_nodes = 0, wave_leaves = 0, wave_tri_tests = 0, max_lane_nodes = 0; };
static uint32_t leaf_count(uint32_t ref) { return (ref >> 24) & 0x7F; }
static int g_policy = 0;      // 0: while-while (trace4); 1: majority (the step more lanes wait for); 2: leaf step as soon as fewer than g_thresh lanes hold a node
static int g_thresh = 16;
static void run_wave(const Tree& T, std::vector<Lane>& L, const std::vector<char>& act, WaveCost& w) {
    bool anyact = false;
    for (size_t i = 0; i < L.size(); ++i) if (act[i]) anyact = true; else L[i].done = true;
    if (!anyact) return;
    ++w.wave_rays;
    if (g_policy == 0) {
        for (;;) {
            for (;;) {
                bool stepped = false;
                for (auto& l : L) if (l.at_node()) { l.node_step(T); stepped = true; }
                if (!stepped) break;
                ++w.wave_nodes;
            }
            bool leaf = false; uint32_t mc = 0;
            for (auto& l : L) if (l.at_leaf()) { mc = std::max(mc, leaf_count(l.cur)); l.leaf_step(T); leaf = true; }
            if (!leaf) break;
            ++w.wave_leaves; w.wave_tri_tests += mc;
        }
    } else {
        for (;;) {
            int nn = 0, nl = 0;
            for (auto& l : L) { if (l.at_node()) ++nn; else if (l.at_leaf()) ++nl; }
            if (nn == 0 && nl == 0) break;
            bool do_node;
            if (g_policy == 1) do_node = nn * g_thresh >= nl * 16;      // (threshold 16 = plain majority; 24 = a node step counts 1.5 x)
            else do_node = nn >= g_thresh || nl == 0;
            if (nn == 0) do_node = false;
            if (nl == 0) do_node = true;
            if (do_node) { for (auto& l : L) if (l.at_node()) l.node_step(T); ++w.wave_nodes; }
            else { uint32_t mc = 0; for (auto& l : L) if (l.at_leaf()) { mc = std::max(mc, leaf_count(l.cur)); l.leaf_step(T); } ++w.wave_leaves; w.wave_tri_tests += mc; }
        }
    }
    uint64_t mx = 0;
    for (size_t i = 0; i < L.size(); ++i) if (act[i]) { ++w.rays; w.lane_nodes += L[i].node_steps; w.lane_tris += L[i].tri_tests; mx = std::max(mx, L[i].node_steps);
        if (L[i].any) { const int k = L[i].hit != kNone; g_dbg[0][2 * k] += 1; g_dbg[0][2 * k + 1] += L[i].tri_tests; } }
    w.max_lane_nodes += mx;
}


// ---- the 8-wide compressed tree (csrc/frt_bvh8.hpp) walked the way frt_trace.hpp: trace8 walks it ---------------------------------------------
// Lane state: G = the group of inner children still to visit of the node stepped last (base, imask, hit mask), T = the hit LEAF children of that node,
// a stack of G words. A node step takes the next child of G (order: increasing slot ^ octant; g_order = 1: by entry distance, the ideal a sort would
// give), pushes what is left of G, fetches the child and intersects its eight boxes. A leaf step tests the triangles (<= 2) of ONE leaf child of T.
struct Tri8 { V3 v0, e1, e2; uint32_t id; };
struct WideTree { std::vector<uint32_t> w; std::vector<Tri8> tris; std::vector<float> fb; uint32_t stack_need = 0, depth = 0; };
static int g_precise = 0;     // 1: the children's float boxes instead of the 8-bit grid boxes (what a finer grid could give at best)
static int g_order = 0;       // 0: octant order (what the kernels can afford); 1: exact near-to-far order
static int g_leafloop = 0;    // 0: one leaf step, then back to the node loop (lanes with leaf children left sit the node steps out); 1: leaf steps until no lane has any
struct Group { uint32_t base = 0, imask = 0, hits = 0; float tn[8]; bool fresh = true; };      // fresh: straight from its node step (g_order = 2: the nearest child first, exactly; later picks by octant)
struct Lane8 {
    V3 o, d, inv; float tmin, tmax, best; bool any, done; uint32_t hit, oct;
    Group G; uint32_t T = 0, tbase = 0; uint8_t meta[8]; float ttn[8];
    std::vector<Group> stk;
    uint64_t node_steps = 0, tri_tests = 0, leaf_steps = 0; size_t max_stack = 0;
    void start(V3 o_, V3 d_, float tmin_, float tmax_, bool any_) {
        o = o_; d = d_; tmin = tmin_; tmax = tmax_; any = any_; best = tmax_; done = false; hit = kNone; stk.clear(); T = 0;
        auto rc = [](float x) { const float k = 8.271806125530277e-25f; return 1.0f / (std::fabs(x) > k ? x : std::copysign(k, x)); };
        inv = {rc(d.x), rc(d.y), rc(d.z)};
        oct = (d.x < 0 ? 1u : 0u) | (d.y < 0 ? 2u : 0u) | (d.z < 0 ? 4u : 0u);
        G = Group{}; G.base = 0; G.imask = 1; G.hits = 1; G.tn[0] = 0;      // the root: a group of one
        node_steps = tri_tests = leaf_steps = 0; max_stack = 0;
    }
    bool wants_leaf() const { return !done && T != 0; }
    bool wants_node() const { return !done && T == 0 && (G.hits != 0 || !stk.empty()); }
    void settle() { if (!done && T == 0 && G.hits == 0 && stk.empty()) done = true; }
    static int pick(const Group& g, uint32_t oct) {
        int bs = -1; float bt = 0; uint32_t bp = 99;
        for (int s = 0; s < 8; ++s) if (g.hits >> s & 1) {
            if (g_order == 1 || (g_order == 2 && g.fresh)) { if (bs < 0 || g.tn[s] < bt) { bs = s; bt = g.tn[s]; } }
            else { const uint32_t p = (uint32_t)s ^ oct; if (p < bp) { bp = p; bs = s; } }
        }
        return bs;
    }
    void node_step(const WideTree& W) {
        ++node_steps;
        if (G.hits == 0) { G = stk.back(); stk.pop_back(); }
        const int s = pick(G, oct);
        G.fresh = false;
        const uint32_t n = G.base + (uint32_t)__builtin_popcount(G.imask & ((1u << s) - 1u));
        G.hits &= ~(1u << s);
        if (G.hits) { stk.push_back(G); max_stack = std::max(max_stack, stk.size()); }
        const uint32_t* w = &W.w[(size_t)n * 32];
        float p[3]; memcpy(p, w, 12);
        const uint32_t imask = w[3] >> 24, leafmask = w[5] >> 24;
        const float o3[3] = {o.x, o.y, o.z}, i3[3] = {inv.x, inv.y, inv.z};
        Group ng; ng.base = w[4]; ng.imask = imask; ng.hits = 0;
        T = 0; tbase = w[5] & 0xFFFFFFu; memcpy(meta, &w[6], 8);
        const uint16_t* q = reinterpret_cast<const uint16_t*>(&w[8]);      // [axis][lo[8], hi[8]]
        for (int c = 0; c < 8; ++c) {
            if (!((imask | leafmask) >> c & 1)) continue;
            float tn = tmin, tf = any ? tmax : best;
            for (int a = 0; a < 3; ++a) {
                uint32_t eb = ((w[3] >> (8 * a)) & 0xFFu) << 23; float step; memcpy(&step, &eb, 4);
                float lo = p[a] + (float)q[16 * a + c] * step, hi = p[a] + (float)q[16 * a + 8 + c] * step;
                if (g_precise) { lo = W.fb[(size_t)n * 48 + c * 6 + a]; hi = W.fb[(size_t)n * 48 + c * 6 + 3 + a]; }
                const float t0 = (lo - o3[a]) * i3[a], t1 = (hi - o3[a]) * i3[a];
                tn = std::max(tn, std::min(t0, t1)); tf = std::min(tf, std::max(t0, t1));
            }
            if (tn <= tf) { if (imask >> c & 1) { ng.hits |= 1u << c; ng.tn[c] = tn; } else { T |= 1u << c; ttn[c] = tn; } }
        }
        G = ng;
        settle();
    }
    void leaf_step(const WideTree& W) {
        ++leaf_steps;
        int s = -1;
        if (g_order == 1) { for (int c = 0; c < 8; ++c) if ((T >> c & 1) && (s < 0 || ttn[c] < ttn[s])) s = c; }
        else { uint32_t bp = 99; for (int c = 0; c < 8; ++c) if (T >> c & 1) { const uint32_t pr = (uint32_t)c ^ oct; if (pr < bp) { bp = pr; s = c; } } }
        T &= ~(1u << s);
        const uint32_t first = tbase + (meta[s] & 31u), count = meta[s] >> 5;
        for (uint32_t k = 0; k < count; ++k) {
            ++tri_tests;
            const Tri8& tr = W.tris[first + k];
            V3 pv = cross(d, tr.e2); float det = dot(tr.e1, pv);
            if (det == 0) continue;
            float iv = 1.0f / det; V3 sv = o - tr.v0; float u = dot(sv, pv) * iv;
            if (!(u >= 0 && u <= 1)) continue;
            V3 qq = cross(sv, tr.e1); float v = dot(d, qq) * iv;
            if (!(v >= 0 && u + v <= 1)) continue;
            float t = dot(tr.e2, qq) * iv;
            if (!(t > tmin && t < tmax)) continue;
            if (getenv("NOCULL") && any) continue;
            if (any) { hit = tr.id; done = true; return; }
            if (t < best || (t == best && tr.id < hit)) { best = t; hit = tr.id; }
        }
        settle();
    }
};
struct WaveCost8 { uint64_t checksum = 0, rays = 0, lane_nodes = 0, lane_tris = 0, lane_leaves = 0, wave_rays = 0, wave_nodes = 0, wave_leaves = 0, max_lane_nodes = 0, max_stack = 0; };
static void run_wave8(const WideTree& W, std::vector<Lane8>& L, const std::vector<char>& act, WaveCost8& w) {
    bool anyact = false;
    for (size_t i = 0; i < L.size(); ++i) if (act[i]) anyact = true; else L[i].done = true;
    if (!anyact) return;
    ++w.wave_rays;
    for (;;) {
        for (;;) {
            bool stepped = false;
            for (auto& l : L) if (l.wants_node()) { l.node_step(W); stepped = true; }
            if (!stepped) break;
            ++w.wave_nodes;
        }
        bool leaf = false;
        for (;;) {
            bool stepped = false;
            for (auto& l : L) if (l.wants_leaf()) { l.leaf_step(W); stepped = true; }
            if (!stepped) break;
            leaf = true; ++w.wave_leaves;
            if (!g_leafloop) break;
        }
        if (!leaf) break;
    }
    uint64_t mx = 0;
    for (size_t i = 0; i < L.size(); ++i) if (act[i]) {
        ++w.rays; w.lane_nodes += L[i].node_steps; w.lane_tris += L[i].tri_tests; w.lane_leaves += L[i].leaf_steps; mx = std::max(mx, L[i].node_steps);
        w.max_stack = std::max<uint64_t>(w.max_stack, L[i].max_stack);
        w.checksum += L[i].any ? (L[i].hit == kNone ? 0u : 1u) : (L[i].hit == kNone ? 0u : L[i].hit + 1u);
    }
    w.max_lane_nodes += mx;
}
static void report8(const char* name, const WaveCost8& w) {
    if (!w.rays) return;
    printf("  %-10s rays %8llu  per lane-ray: nodes %6.2f leaf steps %5.2f tris %5.2f | per wave-ray: node steps %6.2f leaf steps %5.2f (slowest lane %6.2f)  deepest stack %llu  cost~ %7.1f\n", name,
           (unsigned long long)w.rays, (double)w.lane_nodes / w.rays, (double)w.lane_leaves / w.rays, (double)w.lane_tris / w.rays, (double)w.wave_nodes / w.wave_rays,
           (double)w.wave_leaves / w.wave_rays, (double)w.max_lane_nodes / w.wave_rays, (unsigned long long)w.max_stack,
           (200.0 * w.wave_nodes + 170.0 * w.wave_leaves) / w.wave_rays);      // VALU + SALU per step: an 8-wide node step ~ 200, a two-triangle leaf step ~ 170
}


// Unified order (g_unified): leaf and inner children of a node take their turn in ONE order; a leaf child behind an inner one waits on the stack with
// the rest of its node's hit mask (stack word = node index | hits) and the node's header is fetched again when the walk comes back to it ("resume").
static int g_unified = 0;      // 1: one order; 2: + children whose entry distance lies behind the closest hit found meanwhile are dropped when their turn comes (needs the
                               // entry distances: the ideal); 3: + the same, but only when a group is RESUMED from the stack (its node intersected again with the shorter ray)
struct Lane8U {
    V3 o, d, inv; float tmin, tmax, best; bool any, done; uint32_t hit, oct;
    uint32_t node = 0, hits = 0; float tn[8];
    struct E { uint32_t node, hits; float tn[8]; };
    std::vector<E> stk;
    uint64_t node_steps = 0, tri_tests = 0, leaf_steps = 0, resumes = 0; size_t max_stack = 0;
    void start(V3 o_, V3 d_, float tmin_, float tmax_, bool any_) {
        o = o_; d = d_; tmin = tmin_; tmax = tmax_; any = any_; best = tmax_; done = false; hit = kNone; stk.clear();
        auto rc = [](float x) { const float k = 8.271806125530277e-25f; return 1.0f / (std::fabs(x) > k ? x : std::copysign(k, x)); };
        inv = {rc(d.x), rc(d.y), rc(d.z)};
        oct = (d.x < 0 ? 1u : 0u) | (d.y < 0 ? 2u : 0u) | (d.z < 0 ? 4u : 0u);
        node = 0xFFFFFFFFu; hits = 1;      // pseudo group: the root
        node_steps = tri_tests = leaf_steps = resumes = 0; max_stack = 0;
    }
    bool fresh = true;      // the group in hand comes straight from its node step (not from the stack)
    int pick() const {
        int bs = -1; float bt = 0; uint32_t bp = 99;
        for (int s = 0; s < 8; ++s) if (hits >> s & 1) {
            if (g_order == 1 || (g_order == 2 && fresh)) { if (bs < 0 || tn[s] < bt) { bs = s; bt = tn[s]; } }
            else { const uint32_t p = (uint32_t)s ^ oct; if (p < bp) { bp = p; bs = s; } }
        }
        return bs;
    }
    bool next_is_inner(const WideTree& W) const { if (node == 0xFFFFFFFFu) return true; const int s = pick(); return (W.w[(size_t)node * 32 + 3] >> 24) >> s & 1; }
    bool wants_node(const WideTree& W) const { return !done && next_is_inner(W); }
    bool wants_leaf(const WideTree& W) const { return !done && !next_is_inner(W); }
    void cull() { if (!any) for (int c = 0; c < 8; ++c) if ((hits >> c & 1) && tn[c] > best) hits &= ~(1u << c); }
    void after() {
        if (done) return;
        if (g_unified == 2 && hits) cull();
        while (!hits) {
            if (stk.empty()) { done = true; return; }
            node = stk.back().node; hits = stk.back().hits; memcpy(tn, stk.back().tn, sizeof(tn)); stk.pop_back(); ++resumes; fresh = false;
            if (g_unified >= 2) { cull(); if (g_unified == 3 && !any) ++node_steps; }      // (3: the re-intersection costs a node step)
        }
    }
    void node_step(const WideTree& W) {
        ++node_steps;
        const int s = pick();
        uint32_t n;
        if (node == 0xFFFFFFFFu) n = 0;
        else { const uint32_t* pw = &W.w[(size_t)node * 32]; n = pw[4] + (uint32_t)__builtin_popcount((pw[3] >> 24) & ((1u << s) - 1u)); }
        hits &= ~(1u << s); fresh = false;
        if (hits) { E e; e.node = node; e.hits = hits; memcpy(e.tn, tn, sizeof(tn)); stk.push_back(e); max_stack = std::max(max_stack, stk.size()); }
        const uint32_t* w = &W.w[(size_t)n * 32];
        float p[3]; memcpy(p, w, 12);
        const uint32_t valid = (w[3] >> 24) | (w[5] >> 24);
        const float o3[3] = {o.x, o.y, o.z}, i3[3] = {inv.x, inv.y, inv.z};
        const uint16_t* q = reinterpret_cast<const uint16_t*>(&w[8]);      // [axis][lo[8], hi[8]]
        node = n; hits = 0; fresh = true;
        for (int c = 0; c < 8; ++c) {
            if (!(valid >> c & 1)) continue;
            float t0n = tmin, tf = any ? tmax : best;
            for (int a = 0; a < 3; ++a) {
                uint32_t eb = ((w[3] >> (8 * a)) & 0xFFu) << 23; float step; memcpy(&step, &eb, 4);
                float lo = p[a] + (float)q[16 * a + c] * step, hi = p[a] + (float)q[16 * a + 8 + c] * step;
                if (g_precise) { lo = W.fb[(size_t)n * 48 + c * 6 + a]; hi = W.fb[(size_t)n * 48 + c * 6 + 3 + a]; }
                const float t0 = (lo - o3[a]) * i3[a], t1 = (hi - o3[a]) * i3[a];
                t0n = std::max(t0n, std::min(t0, t1)); tf = std::min(tf, std::max(t0, t1));
            }
            if (t0n <= tf) { hits |= 1u << c; tn[c] = t0n; }
        }
        after();
    }
    void leaf_step(const WideTree& W) {
        ++leaf_steps;
        const int s = pick();
        hits &= ~(1u << s); fresh = false;
        const uint32_t* w = &W.w[(size_t)node * 32];
        const uint8_t* meta = reinterpret_cast<const uint8_t*>(&w[6]);
        const uint32_t first = (w[5] & 0xFFFFFFu) + (meta[s] & 31u), count = meta[s] >> 5;
        for (uint32_t k = 0; k < count; ++k) {
            ++tri_tests;
            const Tri8& tr = W.tris[first + k];
            V3 pv = cross(d, tr.e2); float det = dot(tr.e1, pv);
            if (det == 0) continue;
            float iv = 1.0f / det; V3 sv = o - tr.v0; float u = dot(sv, pv) * iv;
            if (!(u >= 0 && u <= 1)) continue;
            V3 qq = cross(sv, tr.e1); float v = dot(d, qq) * iv;
            if (!(v >= 0 && u + v <= 1)) continue;
            float t = dot(tr.e2, qq) * iv;
            if (!(t > tmin && t < tmax)) continue;
            if (any) { hit = tr.id; done = true; return; }
            if (t < best || (t == best && tr.id < hit)) { best = t; hit = tr.id; }
        }
        after();
    }
};
static uint64_t g_resumes = 0;
static void run_wave8u(const WideTree& W, std::vector<Lane8U>& L, const std::vector<char>& act, WaveCost8& w) {
    bool anyact = false;
    for (size_t i = 0; i < L.size(); ++i) if (act[i]) anyact = true; else L[i].done = true;
    if (!anyact) return;
    ++w.wave_rays;
    for (;;) {
        for (;;) {
            bool stepped = false;
            for (auto& l : L) if (l.wants_node(W)) { l.node_step(W); stepped = true; }
            if (!stepped) break;
            ++w.wave_nodes;
        }
        bool leaf = false;
        for (;;) {
            bool stepped = false;
            for (auto& l : L) if (l.wants_leaf(W)) { l.leaf_step(W); stepped = true; }
            if (!stepped) break;
            leaf = true; ++w.wave_leaves;
            if (!g_leafloop) break;
        }
        if (!leaf) break;
    }
    uint64_t mx = 0;
    for (size_t i = 0; i < L.size(); ++i) if (act[i]) {
        ++w.rays; w.lane_nodes += L[i].node_steps; w.lane_tris += L[i].tri_tests; w.lane_leaves += L[i].leaf_steps; mx = std::max(mx, L[i].node_steps);
        w.max_stack = std::max<uint64_t>(w.max_stack, L[i].max_stack); g_resumes += L[i].resumes;
        if (L[i].any) { const int k = L[i].hit != kNone; g_dbg[1][2 * k] += 1; g_dbg[1][2 * k + 1] += L[i].tri_tests; }
        w.checksum += L[i].any ? (L[i].hit == kNone ? 0u : 1u) : (L[i].hit == kNone ? 0u : L[i].hit + 1u);
    }
    w.max_lane_nodes += mx;
}

static float g_presence = 1.0f;      // fraction of the lanes that bring a ray to a walk (the renderer: 0.5 - 0.65)
static int g_chain = 0;              // 1: the shadow ray and the bounce ray of a lane walked back to back in ONE wave loop (no wait for the wave in between)
static int g_split = 0;              // 1: idle lanes help — a ray's root children are dealt to 2 or 4 lanes
static uint32_t rng_state = 12345u;
static float rnd() { rng_state = rng_state * 747796405u + 2891336453u; uint32_t w = ((rng_state >> ((rng_state >> 28) + 4)) ^ rng_state) * 277803737u; return (float)((w >> 22) ^ w) / 4294967296.0f; }
static V3 cosine_dir(V3 n) {
    float z = rnd() * 2 - 1, a = rnd() * 6.2831853f, r = std::sqrt(std::max(0.0f, 1 - z * z));
    return norm(n + V3{r * std::cos(a), r * std::sin(a), z});
}

// A walk of the rays in `L` (act = lane brings a ray; thinned here to the presence asked for) and, optionally, ray splitting over the idle lanes.
// Results (hit, best) end up in L as if every ray had walked alone.
static void run_walk(const Tree& T, std::vector<Lane>& L, std::vector<char>& act, WaveCost& w) {
    for (size_t i = 0; i < L.size(); ++i) if (act[i] && rnd() > g_presence) { act[i] = 0; L[i].hit = kNone; }
    if (!g_split) { run_wave(T, L, act, w); for (size_t i = 0; i < L.size(); ++i) if (act[i]) w.checksum += L[i].any ? (L[i].hit == kNone ? 0u : 1u) : (L[i].hit == kNone ? 0u : L[i].hit + 1u); return; }
    std::vector<int> rays;
    for (size_t i = 0; i < L.size(); ++i) if (act[i]) rays.push_back((int)i);
    const int n = (int)rays.size();
    if (n == 0) return;
    // lanes per ray: 1, 2 or 4, as many as fit; the first rays get the larger share
    std::vector<int> share(n, 1);
    int freel = 64 - n;
    for (int pass = 0; pass < 2; ++pass) for (int k = 0; k < n; ++k) { const int add = share[k]; if (freel >= add) { share[k] += add; freel -= add; } }
    std::vector<Lane> G; std::vector<char> gact; std::vector<int> owner;
    std::vector<float> shared(n, 0.0f);
    for (int k = 0; k < n; ++k) {
        const Lane& src = L[rays[k]];
        shared[k] = src.tmax;
        for (int j = 0; j < share[k]; ++j) { Lane l = src; l.grp_size = share[k]; l.grp_rank = j; l.shared_best = src.any ? nullptr : &shared[k]; G.push_back(l); gact.push_back(1); owner.push_back(k); }
    }
    while (G.size() < 64) { G.push_back(L[rays[0]]); gact.push_back(0); owner.push_back(-1); }
    const uint64_t rays_before = w.rays;
    run_wave(T, G, gact, w);
    w.rays = rays_before + (uint64_t)n;      // (run_wave counted lanes)
    for (int k = 0; k < n; ++k) { L[rays[k]].hit = kNone; L[rays[k]].best = L[rays[k]].tmax; }
    for (size_t g = 0; g < G.size(); ++g) if (gact[g] && G[g].hit != kNone) {
        Lane& dst = L[rays[owner[g]]];
        if (dst.hit == kNone || G[g].best < dst.best || (G[g].best == dst.best && G[g].hit < dst.hit)) { dst.hit = G[g].hit; dst.best = G[g].best; }
    }
    for (int k = 0; k < n; ++k) w.checksum += L[rays[k]].any ? (L[rays[k]].hit == kNone ? 0u : 1u) : (L[rays[k]].hit == kNone ? 0u : L[rays[k]].hit + 1u);
}

static void report(const char* name, const WaveCost& w) {
    if (!w.rays) return;
    printf("  %-10s rays %8llu  per lane-ray: nodes %6.2f tris %5.2f | per wave-ray: node steps %6.2f leaf steps %5.2f  tri tests %5.2f (slowest lane %6.2f)  cost~ %7.1f  per 64 rays %7.1f\n", name,
           (unsigned long long)w.rays, (double)w.lane_nodes / w.rays, (double)w.lane_tris / w.rays, (double)w.wave_nodes / w.wave_rays,
           (double)w.wave_leaves / w.wave_rays, (double)w.wave_tri_tests / w.wave_rays, (double)w.max_lane_nodes / w.wave_rays,
           (110.0 * w.wave_nodes + 20.0 * w.wave_leaves + 75.0 * w.wave_tri_tests) / w.wave_rays,
           (110.0 * w.wave_nodes + 20.0 * w.wave_leaves + 75.0 * w.wave_tri_tests) / w.rays * 64.0);      // VALU + SALU per step, from the ISA of trace4
}

int main(int argc, char** argv) {
    std::string which = argc > 1 ? argv[1] : "cornell";
    int tiles = argc > 2 ? atoi(argv[2]) : 600;
    frt_scene* s = which == "restir" ? frt_scene_create_restir_scene() : frt_scene_create_cornell_box();
    if (!s) { fprintf(stderr, "scene: %s\n", frt_last_error()); return 1; }
    uint32_t cnt[8]; frt_scene_counts(s, cnt);
    Tree T;
    T.t.resize(cnt[7]); T.tri_index.resize(cnt[0]);
    std::vector<float> tr(cnt[0] * 9);
    frt_scene_get(s, 0, tr.data()); frt_scene_get(s, 8, T.t.data()); frt_scene_get(s, 9, T.tri_index.data());
    T.tris.resize(cnt[0]);
    for (uint32_t i = 0; i < cnt[0]; ++i) T.tris[i] = {{tr[9 * i], tr[9 * i + 1], tr[9 * i + 2]}, {tr[9 * i + 3], tr[9 * i + 4], tr[9 * i + 5]}, {tr[9 * i + 6], tr[9 * i + 7], tr[9 * i + 8]}};
    std::vector<frt_light> lights(cnt[3]);
    frt_scene_get(s, 3, lights.data());
    uint32_t st[4]; frt_scene_bvh_stats(s, st);
    const int passes = argc > 3 ? atoi(argv[3]) : 0;
    g_policy = argc > 4 ? atoi(argv[4]) : 0;
    g_thresh = argc > 5 ? atoi(argv[5]) : 16;
    g_presence = argc > 6 ? (float)atof(argv[6]) : 1.0f;
    g_split = argc > 7 ? atoi(argv[7]) : 0;
    g_chain = argc > 8 ? atoi(argv[8]) : 0;
    g_order = argc > 9 ? atoi(argv[9]) : 0;
    g_leafloop = argc > 10 ? atoi(argv[10]) : 0;
    g_unified = argc > 11 ? atoi(argv[11]) : 0;
    g_precise = argc > 12 ? atoi(argv[12]) : 0;
    const int sort256 = argc > 13 ? atoi(argv[13]) : 0;
    // the 8-wide compressed tree as the product built it (csrc/frt_bvh8.hpp), decoded from its device form
    WideTree WT;
    {
        uint32_t ts[8]; frt_scene_tree_stats(s, ts);
        WT.w.resize((size_t)ts[2] * 32); WT.stack_need = ts[3]; WT.depth = ts[4];
        std::vector<float> sl((size_t)ts[6] * 12);
        WT.fb.resize((size_t)ts[2] * 48);
        if (ts[2]) { frt_scene_get(s, 11, WT.w.data()); frt_scene_get(s, 12, sl.data()); frt_scene_get(s, 14, WT.fb.data()); }
        WT.tris.resize(ts[6]);
        for (uint32_t i = 0; i < ts[6]; ++i) { const float* q = &sl[(size_t)i * 12]; uint32_t id; memcpy(&id, &q[3], 4); WT.tris[i] = {{q[0], q[1], q[2]}, {q[4], q[5], q[6]}, {q[8], q[9], q[10]}, id}; }
        printf("8-wide tree: %u nodes (%.2f children each), %u levels, stack need %u, %u triangle slots\n", ts[2], ts[2] ? (double)ts[5] / ts[2] : 0.0, ts[4], ts[3], ts[6]);
    }
    const bool wide = !WT.w.empty() && passes == 0 && !g_split && !g_chain;
    WaveCost8 primary8, bounce18, shadow8, bounce28;
    std::vector<Lane8> L8(64);
    std::vector<Lane8U> L8U(64);
    auto walk8 = [&](const std::vector<Lane>& Lq, const std::vector<char>& a, WaveCost8& w) {
        if (!wide) return;
        if (g_unified) { for (int i = 0; i < 64; ++i) if (a[i]) L8U[i].start(Lq[i].o, Lq[i].d, Lq[i].tmin, Lq[i].tmax, Lq[i].any); run_wave8u(WT, L8U, a, w); return; }
        for (int i = 0; i < 64; ++i) if (a[i]) L8[i].start(Lq[i].o, Lq[i].d, Lq[i].tmin, Lq[i].tmax, Lq[i].any);
        run_wave8(WT, L8, a, w);
    };
    if (passes > 0) {
        printf("as built: SAH cost %.3f depth %u; ", sah_cost(T.t), st[0]);
        st[0] = frt::optimize_bvh2(T.t, T.tri_index, passes, 30u, st[0]);
        printf("after %d insertion passes: SAH cost %.3f depth %u\n", passes, sah_cost(T.t), st[0]);
    }
    if (passes > 0) build_quads(T);      // (the tree was changed here: fold it here, greedily)
    else {                               // the quad nodes as the product built them (frt_scene_get 10)
        uint32_t ts[8]; frt_scene_tree_stats(s, ts);
        std::vector<float> qn((size_t)ts[0] * 32);
        frt_scene_get(s, 10, qn.data());
        T.quads.resize(ts[0]); T.stack_need = ts[1];
        for (uint32_t i = 0; i < ts[0]; ++i) {
            Quad q{}; const float* f = &qn[(size_t)i * 32];
            for (int a = 0; a < 3; ++a) for (int c = 0; c < 4; ++c) { q.lo[a][c] = f[8 * a + c]; q.hi[a][c] = f[8 * a + 4 + c]; }
            for (int c = 0; c < 4; ++c) { memcpy(&q.ref[c], &f[24 + c], 4); if (q.ref[c] != kNone) q.n = c + 1; }
            T.quads[i] = q;
        }
    }
    printf("%s: %u triangles, BVH2 %zu nodes depth %u, %zu quad nodes, stack need %u, SAH cost %.3f\n", which.c_str(), cnt[0], T.t.size(), st[0], T.quads.size(), T.stack_need, sah_cost(T.t));

    // benchmark camera: (0, 0, 3) looking down -z, 45 degrees vertical, 16:9 (camera.rs:40-42, :218-222)
    const int W = 1920, H = 1080;
    const float th = std::tan(0.5f * 45.0f * 3.14159265f / 180.0f), aspect = (float)W / H;

    // Experiment (argument 13): would re-dealing a 16x16 workgroup's bounce rays to its four waves by direction pay? Blocks of 2x2 tiles; the 256
    // bounce rays of a block walked (a) tile by tile, as the kernels do, (b) sorted by direction octant (1) or by octant + origin cell (2) and cut into
    // four waves. Reports wave-level node + leaf steps of both.
    if (sort256) {
        WaveCost byTile, sorted;
        std::vector<Lane> B(256), Wv(64);
        std::vector<char> bact(256), wact(64);
        for (int blk = 0; blk < tiles / 4; ++blk) {
            int bx = (int)(rnd() * (W / 16)), by = (int)(rnd() * (H / 16));
            V3 eye{0, 0, 3};
            for (int q = 0; q < 4; ++q) {
                for (int i = 0; i < 64; ++i) {
                    float px = bx * 16 + (q & 1) * 8 + (i & 7) + 0.5f, py = by * 16 + (q >> 1) * 8 + (i >> 3) + 0.5f;
                    float nx = px / W * 2 - 1, ny = 1 - py / H * 2;
                    Wv[i].start(eye, norm({nx * th * aspect, ny * th, -1}), 0.001f, 1000.0f, false); wact[i] = 1;
                }
                WaveCost dummy; run_wave(T, Wv, wact, dummy);
                for (int i = 0; i < 64; ++i) {
                    const int k = q * 64 + i;
                    bact[k] = Wv[i].hit != kNone && rnd() <= g_presence;      // (presence: the share of lanes that bring a ray to this walk)
                    if (!bact[k]) continue;
                    const Tri& t = T.tris[Wv[i].hit];
                    V3 n = norm(cross(t.e1, t.e2));
                    if (dot(n, Wv[i].d) > 0) n = n * -1.0f;
                    V3 pp = Wv[i].o + Wv[i].d * Wv[i].best;
                    B[k].start(pp + n * 0.001f, cosine_dir(n), 0.001f, 100.0f, false);
                }
            }
            for (int q = 0; q < 4; ++q) {      // (a) tile by tile
                for (int i = 0; i < 64; ++i) { wact[i] = bact[q * 64 + i]; if (wact[i]) Wv[i].start(B[q * 64 + i].o, B[q * 64 + i].d, 0.001f, 100.0f, false); }
                run_wave(T, Wv, wact, byTile);
            }
            std::vector<int> idx;
            for (int k = 0; k < 256; ++k) if (bact[k]) idx.push_back(k);
            auto key = [&](int k) {
                const V3 d = B[k].d;
                uint32_t oct = (d.x < 0 ? 1u : 0u) | (d.y < 0 ? 2u : 0u) | (d.z < 0 ? 4u : 0u);
                uint32_t sub = 0;
                if (sort256 >= 2) { const float ax = std::fabs(d.x), ay = std::fabs(d.y), az = std::fabs(d.z); sub = ax > ay ? (ax > az ? 0u : 2u) : (ay > az ? 1u : 2u); }
                return oct * 4u + sub;
            };
            if (sort256 != 3) std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return key(a) < key(b); });      // (3: dense waves, arrival order)
            for (size_t w0 = 0; w0 < idx.size(); w0 += 64) {      // (b) dense waves of the sorted rays
                for (int i = 0; i < 64; ++i) { wact[i] = w0 + i < idx.size(); if (wact[i]) { const Lane& r = B[idx[w0 + i]]; Wv[i].start(r.o, r.d, 0.001f, 100.0f, false); } }
                run_wave(T, Wv, wact, sorted);
            }
        }
        printf("bounce rays of 16x16 blocks, walked tile by tile vs re-dealt to dense waves by direction (%s):\n", sort256 >= 2 ? "octant + major axis" : "octant");
        report("by tile", byTile); report("sorted", sorted);
        printf("  total wave steps: by tile %llu node + %llu leaf in %llu waves; sorted %llu + %llu in %llu waves; checksum %llu %llu\n", (unsigned long long)byTile.wave_nodes, (unsigned long long)byTile.wave_leaves,
               (unsigned long long)byTile.wave_rays, (unsigned long long)sorted.wave_nodes, (unsigned long long)sorted.wave_leaves, (unsigned long long)sorted.wave_rays, (unsigned long long)byTile.checksum, (unsigned long long)sorted.checksum);
        frt_scene_destroy(s);
        return 0;
    }
    WaveCost primary, bounce1, shadow, bounce2, chained;
    std::vector<Lane> L(64);
    std::vector<char> act(64), act2(64);
    std::vector<V3> P(64), N(64);
    for (int tile = 0; tile < tiles; ++tile) {
        int tx = (int)(rnd() * (W / 8)), ty = (int)(rnd() * (H / 8));
        V3 eye{0, 0, 3};
        for (int i = 0; i < 64; ++i) {
            float px = tx * 8 + (i & 7) + 0.5f, py = ty * 8 + (i >> 3) + 0.5f;
            float nx = px / W * 2 - 1, ny = 1 - py / H * 2;
            L[i].start(eye, norm({nx * th * aspect, ny * th, -1}), 0.001f, 1000.0f, false);
            act[i] = 1;
        }
        run_wave(T, L, act, primary);
        walk8(L, act, primary8);
        for (int i = 0; i < 64; ++i) {
            act[i] = L[i].hit != kNone;
            if (!act[i]) continue;
            const Tri& t = T.tris[L[i].hit];
            V3 n = norm(cross(t.e1, t.e2));
            if (dot(n, L[i].d) > 0) n = n * -1.0f;
            P[i] = L[i].o + L[i].d * L[i].best; N[i] = n;
        }
        // shadow rays to a random point of a random light (restir.wgsl:219-245), from the offset hit point
        for (int i = 0; i < 64; ++i) {
            act2[i] = 0;
            if (!act[i] || lights.empty()) continue;
            const frt_light& l = lights[std::min((size_t)(rnd() * lights.size()), lights.size() - 1)];
            V3 lp{l.position[0], l.position[1], l.position[2]};
            if (l.type_ == 0) lp = lp + V3{l.u[0], l.u[1], l.u[2]} * (rnd() * 2 - 1) + V3{l.v[0], l.v[1], l.v[2]} * (rnd() * 2 - 1);
            else lp = lp + cosine_dir({0, 0, 0}) * l.v[0];
            V3 op = P[i] + N[i] * 0.001f, d = lp - op; float dist = std::sqrt(dot(d, d));
            if (dot(N[i], d) <= 0) continue;
            L[i].start(op, d * (1.0f / dist), 0.001f, dist * 0.999f, true); act2[i] = 1;
        }
        if (g_chain) {
            // one walk: lanes with a shadow ray walk it first and go straight on to their bounce ray
            std::vector<char> actc(64);
            for (int i = 0; i < 64; ++i) {
                actc[i] = act[i];
                if (!act[i]) continue;
                const V3 bo = P[i] + N[i] * 0.001f, bd = cosine_dir(N[i]);
                if (act2[i]) L[i].chain(bo, bd, 0.001f, 100.0f, false);      // (L[i] already holds the shadow ray)
                else L[i].start(bo, bd, 0.001f, 100.0f, false);
            }
            run_walk(T, L, actc, chained);
            act = actc;
        } else {
        run_walk(T, L, act2, shadow);
        walk8(L, act2, shadow8);
        for (int i = 0; i < 64; ++i) if (act[i]) L[i].start(P[i] + N[i] * 0.001f, cosine_dir(N[i]), 0.001f, 100.0f, false);
        run_walk(T, L, act, bounce1);
        walk8(L, act, bounce18);
        }
        for (int i = 0; i < 64; ++i) {
            bool a = act[i] && L[i].hit != kNone;
            if (a) {
                const Tri& t = T.tris[L[i].hit];
                V3 n = norm(cross(t.e1, t.e2));
                if (dot(n, L[i].d) > 0) n = n * -1.0f;
                V3 p = L[i].o + L[i].d * L[i].best;
                L[i].start(p + n * 0.001f, cosine_dir(n), 0.001f, 100.0f, false);
            }
            act[i] = a;
        }
        run_walk(T, L, act, bounce2);
        walk8(L, act, bounce28);
    }
    report("primary", primary); report("shadow", shadow); report("bounce 1", bounce1); report("shadow+b1", chained); report("bounce 2", bounce2);
    if (!g_chain) { WaveCost sum = shadow; sum.wave_nodes += bounce1.wave_nodes; sum.wave_leaves += bounce1.wave_leaves; sum.wave_tri_tests += bounce1.wave_tri_tests; sum.rays += bounce1.rays;
                    sum.lane_nodes += bounce1.lane_nodes; sum.lane_tris += bounce1.lane_tris; sum.max_lane_nodes += bounce1.max_lane_nodes; sum.wave_rays = bounce1.wave_rays; report("sh + b1", sum); }
    WaveCost all;
    for (const WaveCost* w : {&shadow, &bounce1, &bounce2}) { all.rays += w->rays; all.lane_nodes += w->lane_nodes; all.lane_tris += w->lane_tris; all.wave_rays += w->wave_rays; all.wave_nodes += w->wave_nodes; all.wave_leaves += w->wave_leaves; all.wave_tri_tests += w->wave_tri_tests; all.max_lane_nodes += w->max_lane_nodes; }
    report("incoherent", all);
    // what the rays hit does not depend on the schedule, the helpers or the tree: occluded shadow rays + sum of (hit triangle id + 1) over the bounce rays
    printf("hits checksum %llu %llu %llu\n", (unsigned long long)shadow.checksum, (unsigned long long)bounce1.checksum, (unsigned long long)bounce2.checksum);
    if (wide) {
        if (g_precise) printf("(float child boxes instead of the grid)\n");
        printf("8-wide tree, %s order, %s, %s:\n", g_order == 1 ? "exact near-to-far" : (g_order == 2 ? "nearest child first, then octant" : "octant"), g_leafloop ? "leaf steps until no lane has a leaf child left" : "one leaf step per round",
               g_unified ? "leaf and inner children in one order (resume = header fetched again)" : "a node's hit leaves first");
        report8("primary", primary8); report8("shadow", shadow8); report8("bounce 1", bounce18); report8("bounce 2", bounce28);
        WaveCost8 all8;
        for (const WaveCost8* w : {&shadow8, &bounce18, &bounce28}) { all8.rays += w->rays; all8.lane_nodes += w->lane_nodes; all8.lane_tris += w->lane_tris; all8.lane_leaves += w->lane_leaves; all8.wave_rays += w->wave_rays; all8.wave_nodes += w->wave_nodes; all8.wave_leaves += w->wave_leaves; all8.max_lane_nodes += w->max_lane_nodes; all8.max_stack = std::max(all8.max_stack, w->max_stack); }
        report8("incoherent", all8);
        if (g_unified) printf("  resumes per incoherent lane-ray %.2f\n", (double)g_resumes / (double)(primary8.rays + all8.rays));
        printf("hits checksum (8-wide) %llu %llu %llu\n", (unsigned long long)shadow8.checksum, (unsigned long long)bounce18.checksum, (unsigned long long)bounce28.checksum);
    }
    for (int m = 0; m < 2; ++m) printf("any-hit rays, %s: unoccluded %llu rays %.2f tris each; occluded %llu rays %.2f tris each\n", m ? "8-wide" : "quad", (unsigned long long)g_dbg[m][0], (double)g_dbg[m][1] / std::max<uint64_t>(g_dbg[m][0], 1), (unsigned long long)g_dbg[m][2], (double)g_dbg[m][3] / std::max<uint64_t>(g_dbg[m][2], 1));
    frt_scene_destroy(s);
    return 0;
}
