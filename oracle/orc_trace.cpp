// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path. PARITY UNPINNED (see orc_math.hpp).
// Closest-hit / any-hit over the flattened triangle list: brute force (BASELINE.json configs[0]:
// "scalar loop over src/geometry.rs triangles") or a scalar walk of an imported canonical BVH2.
#include "orc_trace.hpp"

namespace orc {

static inline bool better(float t, uint32_t id, float bt, uint32_t bid) { return t < bt || (t == bt && id < bid); }

// Conservative slab test. fminf/fmaxf drop NaNs (0 * inf), which keeps the test conservative.
static inline bool box_test(const BvhNode& n, vec3 o, vec3 inv, float tmin, float tmax, float* tnear) {
    float tx1 = (n.bmin[0] - o.x) * inv.x, tx2 = (n.bmax[0] - o.x) * inv.x;
    float ty1 = (n.bmin[1] - o.y) * inv.y, ty2 = (n.bmax[1] - o.y) * inv.y;
    float tz1 = (n.bmin[2] - o.z) * inv.z, tz2 = (n.bmax[2] - o.z) * inv.z;
    float t0 = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fmaxf(fminf(tz1, tz2), tmin));
    float t1 = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fminf(fmaxf(tz1, tz2), tmax));
    *tnear = t0;
    return t0 <= t1 * 1.0000004f;
}

Hit Tracer::closest(vec3 o, vec3 d, float tmin, float tmax, TraceStats& st) const {
    st.rays_closest++;
    Hit best; best.t = tmax; best.tri = 0xffffffffu;
    float det_best = 0;
    auto test = [&](uint32_t id) {
        st.tris++;
        float t, u, v, det;
        // tmax is kept at the ray's tmax (not shrunk) so that equal-t candidates are seen; selection is by better()
        if (intersect_tri(sc.tris[id], o, d, tmin, tmax, &t, &u, &v, &det)) {
            if (!best.hit || better(t, id, best.t, best.tri)) {
                best.hit = true; best.t = t; best.u = u; best.v = v; best.tri = id; det_best = det;
            }
        }
    };
    if (!bvh) {
        for (uint32_t i = 0; i < (uint32_t)sc.tris.size(); ++i) test(i);
    } else {
        vec3 inv = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        uint32_t stack[128]; int sp = 0;
        uint32_t node = 0;
        const BvhNode* N = sc.bvh_nodes.data();
        float tn;
        st.nodes++;
        if (!box_test(N[0], o, inv, tmin, tmax, &tn)) node = 0xffffffffu;
        while (node != 0xffffffffu) {
            const BvhNode& n = N[node];
            if (n.count > 0) {
                for (uint32_t k = 0; k < n.count; ++k) test(sc.bvh_tri_index[n.left_first + k]);
                node = sp ? stack[--sp] : 0xffffffffu;
                continue;
            }
            uint32_t c0 = n.left_first, c1 = n.left_first + 1;
            float lim = best.hit ? best.t : tmax;
            float t0, t1;
            st.nodes += 2;
            bool h0 = box_test(N[c0], o, inv, tmin, lim, &t0);
            bool h1 = box_test(N[c1], o, inv, tmin, lim, &t1);
            if (h0 && h1) {
                if (t1 < t0) { uint32_t tmp = c0; c0 = c1; c1 = tmp; }
                stack[sp++] = c1; node = c0;
            } else if (h0) node = c0;
            else if (h1) node = c1;
            else node = sp ? stack[--sp] : 0xffffffffu;
        }
    }
    if (best.hit) {
        bool front = det_best > 0.0f;
        if (sc.instances[sc.tri_instance[best.tri]].flip) front = !front;
        best.front = front;
    }
    return best;
}

bool Tracer::any(vec3 o, vec3 d, float tmin, float tmax, TraceStats& st) const {
    st.rays_any++;
    float t, u, v, det;
    if (!bvh) {
        for (uint32_t i = 0; i < (uint32_t)sc.tris.size(); ++i) {
            st.tris++;
            if (intersect_tri(sc.tris[i], o, d, tmin, tmax, &t, &u, &v, &det)) return true;
        }
        return false;
    }
    vec3 inv = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    uint32_t stack[128]; int sp = 0;
    const BvhNode* N = sc.bvh_nodes.data();
    float tn;
    st.nodes++;
    if (!box_test(N[0], o, inv, tmin, tmax, &tn)) return false;
    uint32_t node = 0;
    while (node != 0xffffffffu) {
        const BvhNode& n = N[node];
        if (n.count > 0) {
            for (uint32_t k = 0; k < n.count; ++k) {
                st.tris++;
                if (intersect_tri(sc.tris[sc.bvh_tri_index[n.left_first + k]], o, d, tmin, tmax, &t, &u, &v, &det)) return true;
            }
            node = sp ? stack[--sp] : 0xffffffffu;
            continue;
        }
        uint32_t c0 = n.left_first, c1 = n.left_first + 1;
        float t0, t1;
        st.nodes += 2;
        bool h0 = box_test(N[c0], o, inv, tmin, tmax, &t0);
        bool h1 = box_test(N[c1], o, inv, tmin, tmax, &t1);
        if (h0 && h1) { stack[sp++] = c1; node = c0; }
        else if (h0) node = c0;
        else if (h1) node = c1;
        else node = sp ? stack[--sp] : 0xffffffffu;
    }
    return false;
}

} // namespace orc
