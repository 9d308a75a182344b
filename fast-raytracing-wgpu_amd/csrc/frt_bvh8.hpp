// frt_bvh8.hpp — the canonical BVH2 collapsed into 8-WIDE NODES WITH GRID BOXES (host only, deterministic). What frt_trace.hpp: trace8 walks.
//
// Why (round 4): a walk is a chain of dependent round trips — node -> node -> leaf — that four waves per SIMD cannot hide, and the quad
// node's step carries a sorting network and up to three stack pushes. An 8-wide node decides three levels of the binary tree per round trip; its
// children are visited in an order that depends only on the ray's direction signs (no sort; any-hit rays enter the NEAREST inner child first); a node
// pushes ONE stack word — the mask of its inner children still to visit — so the stack is as deep as the tree (4 entries for the Cornell Box, 7 for
// the 246k-triangle colonnade; the quad tree: 22 and 31), which is what frees a traced workgroup's LDS. The layout follows Ylitie, Karras and Laine,
// "Efficient incoherent ray traversal on GPUs through compressed wide BVHs" (HPG 2017), re-cut for a 64-lane wave, per-lane LDS stacks and this
// build's leaf format — and for rays that START AND END ON SURFACES: on their 8-bit grid (a step of 1/128 .. 1/255 of the node) the flat box of a
// wall is inflated to a slab two orders of magnitude thicker than the 0.001 offset of a bounce ray's origin, every ray then tests the triangles it
// starts on and ends at (tools/bvh_quality.cpp: 4.6 instead of 2.2 triangle tests per incoherent ray on the Cornell Box). The grid here has 16 bits: its
// step (<= 1/32768 of the node) is below the builder's box padding and the wide tree tests the triangles the binary tree tests (2.2).
//
// Node, 32 words = 128 bytes (eight uint4; the plane blocks are read with per-lane offsets chosen by the ray's direction signs, like the quad node's):
//   w0..w2   origin p (f32 x, y, z): the low corner of the node's grid (= of the union of its children's padded boxes)
//   w3       ex | ey << 8 | ez << 16 | imask << 24     e*: biased f32 exponent of the per-axis power-of-two grid step (step = bits(e << 23));
//                                                     imask: slots that hold an INNER child
//   w4       child_base: index of the first inner child; a node's inner children are contiguous, in slot order (child = base + rank among imask)
//   w5       tri_base (24 bits) | leafmask << 24       leafmask: slots that hold a LEAF child; tri_base: first triangle slot (tris8) of the node's leaves
//   w6, w7   meta[8] bytes: for a leaf slot  offset | count << 5  (its triangles: tris8[tri_base + offset .. + count)); 0 otherwise
//   bytes 32..47  lo.x[8] (u16)   48..63  hi.x[8]   64..79  lo.y[8]   80..95  hi.y[8]   96..111  lo.z[8]   112..127  hi.z[8]
//            child boxes on the node's grid: lo rounded down, hi rounded up, so a child's grid box contains its (padded) float box: the wide tree
//            prunes a little less than the binary one and never more (hits do not depend on the tree, frt_trace.hpp)
// An empty slot has neither mask bit set (trace8 ANDs the slab results with imask | leafmask); its grid box is 65535 .. 0.
// Slot assignment: a child's slot s encodes where it lies in the node: bit a of s set = on the high side along axis a. A ray whose direction has sign
// octant `oct` (bit a set = negative component) visits the hit children in increasing s ^ oct: near side first on every axis — Ylitie et al.'s
// ordering; the assignment that maximises sum(dot(centroid_c - centre, dir(s_c))) is found exactly (an 8 x 8 assignment problem, Kuhn–Munkres).
// Numbering: breadth-first (the top of the tree is contiguous: the first nodes are what a traced workgroup keeps in LDS).
// Triangles: their own slot array `tris8` (same 48-byte slots as tri_slots, other order): the triangles of a node's leaf children are contiguous.
#pragma once
#include "../../include/frt.h"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace frt {

static const uint32_t kWide8Words = 32;
static const uint32_t kWide8MaxNodes = 65536;      // trace8's stack word packs child_base into 16 bits (larger trees keep the quad-node walk)

struct Wide8 {
    std::vector<uint32_t> words;       // kWide8Words per node
    std::vector<uint32_t> tri_order;   // tris8[i] = tri_slots[tri_order[i]]
    std::vector<float> child_boxes;    // host only (tools/bvh_quality.cpp prices the grid against them): per node 8 x (lo.xyz, hi.xyz), the children's float boxes
    uint32_t stack_need = 0;           // entries of the deepest stack a ray can need (one per level with two or more inner children)
    uint32_t depth = 0;                // levels of wide nodes
    uint32_t children = 0;             // sum of child counts (statistics)
    bool ok = false;                   // false: more than kWide8MaxNodes nodes, or a node whose leaves hold more than 31 + 7 triangles
};

namespace wide8_detail {
// Kuhn–Munkres for an n x n cost matrix (minimisation), n <= 8. a[i][j]: cost of giving row i column j. Returns col_of_row.
inline void assign_min(const double a[8][8], int n, int col_of_row[8]) {
    const double INF = 1e300;
    double u[9] = {0}, v[9] = {0};
    int p[9] = {0}, way[9] = {0};
    for (int i = 1; i <= n; ++i) {
        p[0] = i;
        int j0 = 0;
        double minv[9]; bool used[9];
        for (int j = 0; j <= n; ++j) { minv[j] = INF; used[j] = false; }
        do {
            used[j0] = true;
            const int i0 = p[j0];
            double delta = INF; int j1 = 0;
            for (int j = 1; j <= n; ++j) if (!used[j]) {
                const double cur = a[i0 - 1][j - 1] - u[i0] - v[j];
                if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
                if (minv[j] < delta) { delta = minv[j]; j1 = j; }
            }
            for (int j = 0; j <= n; ++j) {
                if (used[j]) { u[p[j]] += delta; v[j] -= delta; }
                else minv[j] -= delta;
            }
            j0 = j1;
        } while (p[j0] != 0);
        do { const int j1 = way[j0]; p[j0] = p[j1]; j0 = j1; } while (j0);
    }
    for (int j = 1; j <= n; ++j) if (p[j]) col_of_row[p[j] - 1] = j - 1;
}
}   // namespace wide8_detail

// Which nodes of the binary tree become wide nodes of `width` (4 or 8) slots: the dynamic programme of Ylitie et al. (section 3.1). C(n, i) = the least
// surface-area cost of representing the subtree under n by at most i roots (wide nodes or leaves) —
//     C(n, 1) = area(n) + min over the ways of dealing `width` slots to n's two children of C(l, k) + C(r, width - k)      (n becomes a wide node)
//     C(n, i) = min(C(n, i - 1), min over 0 < k < i of C(l, k) + C(r, i - k))                                               (n dissolves into its parent)
// with a leaf of the binary tree costing 0.3 x its own area x triangles at every i. Minimising the summed area of the wide nodes is what fills them: the
// greedy fold (open the largest child first) leaves 4.7 of 8 slots used on the Cornell Box, this 5.6; of the quad tree's 4 slots 3.0 / 3.5.
struct WideCollapse {
    const std::vector<frt_bvh2_node>& t;
    int width;
    std::vector<float> C;             // C[n * width + i], i = 1 .. width - 1 roots (index 0 unused)
    std::vector<uint8_t> split;       // split[n * (width + 1) + i]: slots given to the left child when n's subtree gets i (i = 2 .. width; width = n is a wide node; 0 = n stays a root)
    static float half_area(const frt_bvh2_node& n) {
        const float dx = n.bmax[0] - n.bmin[0], dy = n.bmax[1] - n.bmin[1], dz = n.bmax[2] - n.bmin[2];
        return dx * dy + dy * dz + dz * dx;
    }
    WideCollapse(const std::vector<frt_bvh2_node>& tree, int w) : t(tree), width(w), C(tree.size() * (size_t)w, 0.0f), split(tree.size() * (size_t)(w + 1), 0) {
        const size_t W = (size_t)width;
        for (size_t n = t.size(); n-- > 0;) {                     // children follow their parent: bottom-up
            const float area = half_area(t[n]);
            if (t[n].count > 0) { for (size_t i = 1; i < W; ++i) C[n * W + i] = area * (float)t[n].count * 0.3f; continue; }
            const size_t l = t[n].left_first, r = l + 1;
            auto deal = [&](int i, uint8_t& k_best) {             // best way to give i >= 2 slots to the two children
                float best = 3.0e38f; k_best = 1;
                for (int k = 1; k < i; ++k) {
                    const float c = C[l * W + (size_t)std::min(k, width - 1)] + C[r * W + (size_t)std::min(i - k, width - 1)];
                    if (c < best) { best = c; k_best = (uint8_t)k; }
                }
                return best;
            };
            uint8_t kw; const float wide = area + deal(width, kw);
            split[n * (W + 1) + W] = kw;
            C[n * W + 1] = wide;
            for (int i = 2; i < width; ++i) {
                uint8_t k; const float d = deal(i, k);
                if (d < C[n * W + (size_t)i - 1]) { C[n * W + (size_t)i] = d; split[n * (W + 1) + (size_t)i] = k; }
                else { C[n * W + (size_t)i] = C[n * W + (size_t)i - 1]; split[n * (W + 1) + (size_t)i] = split[n * (W + 1) + (size_t)i - 1]; }      // (fewer roots were cheaper: the decision of i - 1)
            }
        }
    }
    // the children of wide node ni, in the binary tree's left-to-right order; returns their number (<= width)
    int children_of(uint32_t ni, uint32_t* out) const {
        if (t[ni].count > 0) { out[0] = ni; return 1; }       // a lone leaf root
        const size_t W = (size_t)width;
        struct Item { uint32_t n; int i; };
        Item stack[32]; int sp = 0, n_out = 0;
        const uint32_t l0 = t[ni].left_first; const int k0 = split[(size_t)ni * (W + 1) + W];
        stack[sp++] = Item{l0 + 1, width - k0}; stack[sp++] = Item{l0, k0};
        while (sp > 0) {
            const Item it = stack[--sp];
            const int i = std::min(it.i, width - 1);
            const int ks = t[it.n].count > 0 ? 0 : split[(size_t)it.n * (W + 1) + (size_t)i];
            if (t[it.n].count > 0 || i == 1 || ks == 0) { out[n_out++] = it.n; continue; }      // a leaf, or a wide node of its own
            const uint32_t l = t[it.n].left_first;
            stack[sp++] = Item{l + 1, i - ks}; stack[sp++] = Item{l, ks};
        }
        return n_out;
    }
};

// t: canonical BVH2 (children adjacent, behind their parent; boxes padded). Leaves of t index triangle slots [left_first, left_first + count).
inline void build_wide8(const std::vector<frt_bvh2_node>& t, Wide8& out) {
    out = Wide8{};
    if (t.empty()) return;
    struct Kids { uint32_t c[8]; int n; int slot[8]; };
    // 1. collapse: WideCollapse above
    WideCollapse dp(t, 8);
    auto children_of = [&](uint32_t ni) -> Kids {
        Kids k{};
        k.n = dp.children_of(ni, k.c);
        return k;
    };
    // 2. slots: maximise sum(dot(centroid_c - centre, dir(slot))), dir(s).a = +1 where bit a of s is set, else -1
    auto assign_slots = [&](uint32_t ni, Kids& k) {
        double cen[3];
        for (int a = 0; a < 3; ++a) cen[a] = 0.5 * ((double)t[ni].bmin[a] + (double)t[ni].bmax[a]);
        double cost[8][8];
        for (int i = 0; i < 8; ++i)
            for (int s = 0; s < 8; ++s) {
                double v = 0.0;
                if (i < k.n) for (int a = 0; a < 3; ++a) {
                    const double c = 0.5 * ((double)t[k.c[i]].bmin[a] + (double)t[k.c[i]].bmax[a]) - cen[a];
                    v += ((s >> a) & 1) ? c : -c;
                }
                cost[i][s] = -v;      // (dummy rows cost nothing anywhere)
            }
        int col[8];
        wide8_detail::assign_min(cost, 8, col);
        for (int i = 0; i < k.n; ++i) k.slot[i] = col[i];
    };
    std::vector<uint32_t> order(1, 0u), level(1, 1u);
    std::vector<Kids> kids;
    std::vector<uint32_t> base;        // child_base per wide node
    for (size_t h = 0; h < order.size(); ++h) {
        Kids k = children_of(order[h]);
        assign_slots(order[h], k);
        base.push_back((uint32_t)order.size());
        // inner children get consecutive numbers in SLOT order
        for (int s = 0; s < 8; ++s)
            for (int i = 0; i < k.n; ++i)
                if (k.slot[i] == s && t[k.c[i]].count == 0) { order.push_back(k.c[i]); level.push_back(level[h] + 1u); }
        kids.push_back(k);
        out.depth = std::max(out.depth, level[h]);
        out.children += (uint32_t)k.n;
        if (order.size() > kWide8MaxNodes) return;      // out.ok stays false
    }
    const size_t N = order.size();
    out.words.assign(N * kWide8Words, 0u);
    out.child_boxes.assign(N * 48, 0.0f);
    std::vector<uint32_t> need(N, 0u);
    // 3. triangles of the leaf children, node by node, slot by slot; 4. boxes on the node's grid
    for (size_t h = 0; h < N; ++h) {
        const Kids& k = kids[h];
        uint32_t* w = &out.words[h * kWide8Words];
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int i = 0; i < k.n; ++i)
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], (double)t[k.c[i]].bmin[a]); hi[a] = std::max(hi[a], (double)t[k.c[i]].bmax[a]); }
        float p[3]; uint32_t e[3]; double step[3];
        for (int a = 0; a < 3; ++a) {
            p[a] = (float)lo[a];
            if ((double)p[a] > lo[a]) p[a] = std::nextafterf(p[a], -INFINITY);      // (lo is a float already: never taken; kept for boxes that are not)
            const double ext = std::max(hi[a] - (double)p[a], 1e-30);
            int ex = (int)std::ceil(std::log2(ext / 65535.0));
            while (std::ldexp(65535.0, ex) < ext) ++ex;                             // 65535 steps must span the extent
            ex = std::max(-126, std::min(127, ex));
            e[a] = (uint32_t)(ex + 127);
            step[a] = std::ldexp(1.0, ex);
        }
        memcpy(&w[0], p, 12);
        uint32_t imask = 0, leafmask = 0, tri_off = 0;
        const uint32_t tri_base = (uint32_t)out.tri_order.size();
        uint8_t meta[8] = {0};
        uint16_t qlo[3][8], qhi[3][8];
        for (int s = 0; s < 8; ++s) for (int a = 0; a < 3; ++a) { qlo[a][s] = 65535; qhi[a][s] = 0; }
        for (int s = 0; s < 8; ++s)
            for (int i = 0; i < k.n; ++i) {
                if (k.slot[i] != s) continue;
                const frt_bvh2_node& c = t[k.c[i]];
                for (int a = 0; a < 3; ++a) { out.child_boxes[h * 48 + s * 6 + a] = c.bmin[a]; out.child_boxes[h * 48 + s * 6 + 3 + a] = c.bmax[a]; }
                for (int a = 0; a < 3; ++a) {
                    double gl = std::floor(((double)c.bmin[a] - (double)p[a]) / step[a]), gh = std::ceil(((double)c.bmax[a] - (double)p[a]) / step[a]);
                    gl = std::max(0.0, std::min(65535.0, gl)); gh = std::max(0.0, std::min(65535.0, gh));
                    // the planes behind trace8's distances are p + q * step (q * step is exact in f32: 16 bits times a power of two): they must bracket the float box
                    while (gl > 0.0 && (float)((double)p[a] + gl * step[a]) > c.bmin[a]) gl -= 1.0;
                    while (gh < 65535.0 && (float)((double)p[a] + gh * step[a]) < c.bmax[a]) gh += 1.0;
                    qlo[a][s] = (uint16_t)gl; qhi[a][s] = (uint16_t)gh;
                }
                if (c.count > 0) {
                    if (tri_off > 31u || c.count > 7u) return;      // (meta byte: 5 bits of offset, 3 of count; never with leaves of <= 4 triangles)
                    leafmask |= 1u << s;
                    meta[s] = (uint8_t)(tri_off | (c.count << 5));
                    for (uint32_t j = 0; j < c.count; ++j) out.tri_order.push_back(c.left_first + j);
                    tri_off += c.count;
                } else {
                    imask |= 1u << s;
                }
            }
        w[3] = e[0] | (e[1] << 8) | (e[2] << 16) | (imask << 24);
        w[4] = base[h];
        if (tri_base >= (1u << 24)) return;
        w[5] = tri_base | (leafmask << 24);
        memcpy(&w[6], meta, 8);
        for (int a = 0; a < 3; ++a) { memcpy(&w[8 + 8 * a], qlo[a], 16); memcpy(&w[12 + 8 * a], qhi[a], 16); }
    }
    // stack need, bottom-up (children have larger indices): a node with two or more inner children leaves one word on the stack while a child is walked
    for (size_t h = N; h-- > 0;) {
        const uint32_t* w = &out.words[h * kWide8Words];
        const uint32_t imask = w[3] >> 24, n_in = (uint32_t)__builtin_popcount(imask);
        uint32_t deepest = 0;
        for (uint32_t r = 0; r < n_in; ++r) deepest = std::max(deepest, need[w[4] + r]);
        need[h] = (n_in >= 2u ? 1u : 0u) + deepest;
    }
    out.stack_need = need[0];
    out.ok = true;
}

} // namespace frt
