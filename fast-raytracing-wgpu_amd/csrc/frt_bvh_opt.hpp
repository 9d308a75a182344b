// frt_bvh_opt.hpp — insertion-based optimisation of the canonical BVH2 (host only, deterministic). Used by frt_bvh.cpp for scenes of kBvhOptMinTris
// triangles and more, where it pays (one pass: 32k-triangle ReSTIR scene 1.336 -> 1.295 ms per frame, 82k-triangle blob 3.37 -> 3.24 ms, 246k-triangle
// colonnade 21.2 -> 21.0 ms); on the Cornell Box it lowers the tree's surface-area cost by 6 % and RAISES what a wave pays per incoherent ray by 8 %
// (deeper tree: 14 -> 25 levels; tools/bvh_quality.cpp, profiles/r3_experiments/traversal_in_situ.md) — measured 1.613 -> 1.618 ms: small scenes
// keep the tree as built.
//
// The binned-SAH build (frt_bvh.cpp: Builder) decides every split from centroids alone and never revisits it. This pass lowers the tree's
// surface-area cost afterwards, the way Bittner, Hapala and Havran describe ("Fast insertion-based optimization of bounding volume
// hierarchies", CGF 2013): take a subtree L out of the tree (its parent P goes with it, the sibling moves up), search the tree for the node X
// next to which L costs least — a branch-and-bound walk ordered by the area the ancestors of X would grow by — and put P back as the parent
// of (X, L). Subtrees are visited in order of decreasing area; a few passes converge. Leaves and their triangles are untouched, so which
// triangles a ray hits — and therefore every pixel — is unchanged (hit semantics do not depend on the tree); only the number of
// node steps a ray takes changes. The result is re-emitted in the canonical layout (children adjacent, after their parent, depth-first).
#pragma once
#include "../../include/frt.h"
#include <algorithm>
#include <cstdint>
#include <queue>
#include <utility>
#include <vector>

namespace frt {

struct BvhOpt {
    struct N { float lo[3], hi[3]; int parent, c[2]; uint32_t first, count; };
    std::vector<N> n;
    int root = 0;
    static const int kMaxVisits = 512;

    static float area(const float lo[3], const float hi[3]) {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
    float area(int i) const { return area(n[i].lo, n[i].hi); }
    float union_area(int i, const N& l) const {
        float lo[3], hi[3];
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(n[i].lo[a], l.lo[a]); hi[a] = std::max(n[i].hi[a], l.hi[a]); }
        return area(lo, hi);
    }
    void refit_up(int i) {
        for (; i >= 0; i = n[i].parent) {
            const N &a = n[n[i].c[0]], &b = n[n[i].c[1]];
            bool same = true;
            for (int k = 0; k < 3; ++k) {
                const float lo = std::min(a.lo[k], b.lo[k]), hi = std::max(a.hi[k], b.hi[k]);
                if (lo != n[i].lo[k] || hi != n[i].hi[k]) same = false;
                n[i].lo[k] = lo; n[i].hi[k] = hi;
            }
            if (same) break;
        }
    }
    void load(const std::vector<frt_bvh2_node>& t) {
        n.resize(t.size());
        for (size_t i = 0; i < t.size(); ++i) {
            N& x = n[i];
            // (the boxes carry the builder's padding: the union of two padded boxes IS the padded union — rounding is monotonic — so refitting
            // padded boxes gives exactly the boxes that padding the refitted ones would)
            for (int a = 0; a < 3; ++a) { x.lo[a] = t[i].bmin[a]; x.hi[a] = t[i].bmax[a]; }
            x.first = t[i].left_first; x.count = t[i].count;
            x.c[0] = x.c[1] = -1;
            if (i == 0) x.parent = -1;
            if (t[i].count == 0) { x.c[0] = (int)t[i].left_first; x.c[1] = (int)t[i].left_first + 1; n[x.c[0]].parent = n[x.c[1]].parent = (int)i; }
        }
        root = 0;
    }
    // Re-emits the tree depth-first in the canonical layout; leaves' triangle ranges are rewritten in that order too.
    void store(std::vector<frt_bvh2_node>& t, std::vector<uint32_t>& tri_index, uint32_t& max_depth) const {
        std::vector<frt_bvh2_node> out; out.reserve(n.size());
        std::vector<uint32_t> idx; idx.reserve(tri_index.size());
        out.push_back(frt_bvh2_node{});
        struct It { int src; uint32_t dst, depth; };
        std::vector<It> st(1, {root, 0u, 1u});
        max_depth = 0;
        while (!st.empty()) {
            const It it = st.back(); st.pop_back();
            const N& x = n[it.src];
            max_depth = std::max(max_depth, it.depth);
            // (out[...] is addressed by index after every push_back: the vector may have moved)
            for (int a = 0; a < 3; ++a) { out[it.dst].bmin[a] = x.lo[a]; out[it.dst].bmax[a] = x.hi[a]; }
            if (x.count) {
                out[it.dst].left_first = (uint32_t)idx.size(); out[it.dst].count = x.count;
                for (uint32_t k = 0; k < x.count; ++k) idx.push_back(tri_index[x.first + k]);
            } else {
                const uint32_t l = (uint32_t)out.size();
                out.push_back(frt_bvh2_node{}); out.push_back(frt_bvh2_node{});
                out[it.dst].left_first = l; out[it.dst].count = 0;
                st.push_back({x.c[1], l + 1, it.depth + 1});      // left child is expanded first: its subtree follows directly
                st.push_back({x.c[0], l, it.depth + 1});
            }
        }
        t.swap(out); tri_index.swap(idx);
    }

    double cost() const {
        double c = 0;
        for (const N& x : n) c += area(x.lo, x.hi) * (x.count ? (double)x.count : 1.0);
        return c / area(root);
    }

    // One pass over all subtrees in order of decreasing area. Returns the number of subtrees that moved.
    size_t pass() {
        std::vector<int> order;
        for (int i = 0; i < (int)n.size(); ++i) if (n[i].parent >= 0 && n[n[i].parent].parent >= 0) order.push_back(i);
        std::vector<float> a0(n.size());
        for (size_t i = 0; i < n.size(); ++i) a0[i] = area((int)i);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return a0[a] > a0[b]; });
        size_t moved = 0;
        struct Cand { float induced; int node; bool operator<(const Cand& o) const { return induced > o.induced || (induced == o.induced && node > o.node); } };
        std::priority_queue<Cand> pq;
        for (int L : order) {
            const int P = n[L].parent;
            if (P < 0) continue;
            const int G = n[P].parent;
            if (G < 0) continue;      // (a child of the root after earlier moves)
            const int S = n[P].c[0] == L ? n[P].c[1] : n[P].c[0];
            // take L (and P) out: S moves up
            n[G].c[n[G].c[0] == P ? 0 : 1] = S; n[S].parent = G;
            refit_up(G);
            // best position: minimise  area(X u L) + sum over the ancestors A of X of (area(A u L) - area(A))
            const N l = n[L];
            const float la = area(l.lo, l.hi);
            float best = 3.0e38f; int bx = -1;
            while (!pq.empty()) pq.pop();
            pq.push({0.0f, root});
            // (the search is cut after kMaxVisits candidates: where boxes overlap heavily — coincident geometry — the bound prunes nothing and a full
            // search would make the pass quadratic; the best position found so far is a valid one)
            int visits = 0;
            while (!pq.empty()) {
                const Cand c = pq.top(); pq.pop();
                if (c.induced + la >= best || ++visits > kMaxVisits) break;
                const float direct = union_area(c.node, l);
                const float total = c.induced + direct;
                if (total < best) { best = total; bx = c.node; }
                if (n[c.node].count == 0) {
                    const float ind = c.induced + (direct - area(c.node));
                    if (ind + la < best) { pq.push({ind, n[c.node].c[0]}); pq.push({ind, n[c.node].c[1]}); }
                }
            }
            int X = bx;
            if (X < 0) X = S;      // put it back where it was
            if (X != S) ++moved;
            // P becomes the parent of (X, L) in X's place
            const int XP = n[X].parent;
            n[P].parent = XP;
            if (XP >= 0) n[XP].c[n[XP].c[0] == X ? 0 : 1] = P; else root = P;
            n[P].c[0] = X; n[P].c[1] = L; n[X].parent = P; n[L].parent = P;
            for (int a = 0; a < 3; ++a) { n[P].lo[a] = std::min(n[X].lo[a], n[L].lo[a]); n[P].hi[a] = std::max(n[X].hi[a], n[L].hi[a]); }
            if (XP >= 0) refit_up(XP);
        }
        return moved;
    }
};

// Optimises `t` / `tri_index` in place (boxes as stored, padding included). A result deeper than `max_depth` levels — the traversal stack is
// sized from that bound — is not accepted: fewer passes are tried, down to none (the tree as built). Returns the depth of the tree kept.
inline uint32_t optimize_bvh2(std::vector<frt_bvh2_node>& t, std::vector<uint32_t>& tri_index, int passes, uint32_t max_depth, uint32_t depth_as_built) {
    for (; passes > 0; passes /= 2) {
        BvhOpt o;
        o.load(t);
        for (int p = 0; p < passes; ++p) if (o.pass() == 0) break;
        std::vector<frt_bvh2_node> t2 = t;
        std::vector<uint32_t> idx2 = tri_index;
        uint32_t depth = 0;
        o.store(t2, idx2, depth);
        if (depth <= max_depth) { t.swap(t2); tri_index.swap(idx2); return depth; }
    }
    return depth_as_built;
}

} // namespace frt
