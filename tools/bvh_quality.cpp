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
    for (size_t i = 0; i < L.size(); ++i) if (act[i]) { ++w.rays; w.lane_nodes += L[i].node_steps; w.lane_tris += L[i].tri_tests; mx = std::max(mx, L[i].node_steps); }
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
    if (passes > 0) {
        printf("as built: SAH cost %.3f depth %u; ", sah_cost(T.t), st[0]);
        st[0] = frt::optimize_bvh2(T.t, T.tri_index, passes, 30u, st[0]);
        printf("after %d insertion passes: SAH cost %.3f depth %u\n", passes, sah_cost(T.t), st[0]);
    }
    build_quads(T);
    printf("%s: %u triangles, BVH2 %zu nodes depth %u, %zu quad nodes, stack need %u, SAH cost %.3f\n", which.c_str(), cnt[0], T.t.size(), st[0], T.quads.size(), T.stack_need, sah_cost(T.t));

    // benchmark camera: (0, 0, 3) looking down -z, 45 degrees vertical, 16:9 (camera.rs:40-42, :218-222)
    const int W = 1920, H = 1080;
    const float th = std::tan(0.5f * 45.0f * 3.14159265f / 180.0f), aspect = (float)W / H;
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
        for (int i = 0; i < 64; ++i) if (act[i]) L[i].start(P[i] + N[i] * 0.001f, cosine_dir(N[i]), 0.001f, 100.0f, false);
        run_walk(T, L, act, bounce1);
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
    }
    report("primary", primary); report("shadow", shadow); report("bounce 1", bounce1); report("shadow+b1", chained); report("bounce 2", bounce2);
    if (!g_chain) { WaveCost sum = shadow; sum.wave_nodes += bounce1.wave_nodes; sum.wave_leaves += bounce1.wave_leaves; sum.wave_tri_tests += bounce1.wave_tri_tests; sum.rays += bounce1.rays;
                    sum.lane_nodes += bounce1.lane_nodes; sum.lane_tris += bounce1.lane_tris; sum.max_lane_nodes += bounce1.max_lane_nodes; sum.wave_rays = bounce1.wave_rays; report("sh + b1", sum); }
    WaveCost all;
    for (const WaveCost* w : {&shadow, &bounce1, &bounce2}) { all.rays += w->rays; all.lane_nodes += w->lane_nodes; all.lane_tris += w->lane_tris; all.wave_rays += w->wave_rays; all.wave_nodes += w->wave_nodes; all.wave_leaves += w->wave_leaves; all.wave_tri_tests += w->wave_tri_tests; all.max_lane_nodes += w->max_lane_nodes; }
    report("incoherent", all);
    // what the rays hit does not depend on the schedule, the helpers or the tree: occluded shadow rays + sum of (hit triangle id + 1) over the bounce rays
    printf("hits checksum %llu %llu %llu\n", (unsigned long long)shadow.checksum, (unsigned long long)bounce1.checksum, (unsigned long long)bounce2.checksum);
    frt_scene_destroy(s);
    return 0;
}
