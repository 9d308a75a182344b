// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path. PARITY UNPINNED (see orc_math.hpp).
//
// Replaces the Vulkan ray query behind wgpu 28.0.0 EXPERIMENTAL_RAY_QUERY (Cargo.lock:2821), whose BVH and
// ray/triangle arithmetic are not in /root/reference (call sites: gbuffer.wgsl:108-112, restir.wgsl:376-380,
// :601-607, restir_spatial.wgsl:397-399, :627-633). Published algorithm restated: Möller–Trumbore (1997).
//
// Hit semantics fixed by this build — independent of any acceleration structure:
//   * a triangle is hit iff det != 0, 0 <= u <= 1, v >= 0, u + v <= 1, tmin < t < tmax (f32, ops as below);
//   * closest hit = minimum t; ties -> smallest flattened triangle id;
//   * any hit (flag 0x4, terminate on first hit) = "some triangle is hit";
//   * front face <=> det > 0 (xor instance flip) — equals dot(cross(v1-v0, v2-v0), dir) < 0.
#pragma once
#include "orc_scene.hpp"

namespace orc {

struct Hit {
    bool hit = false;
    float t = 0, u = 0, v = 0;   // barycentrics: u = weight of v1, v = weight of v2
    uint32_t tri = 0xffffffffu;  // flattened world triangle id
    bool front = false;
};

struct TraceStats {
    uint64_t rays_closest = 0, rays_any = 0, nodes = 0, tris = 0;
    void add(const TraceStats& o) { rays_closest += o.rays_closest; rays_any += o.rays_any; nodes += o.nodes; tris += o.tris; }
};

inline bool intersect_tri(const Tri& tr, vec3 o, vec3 d, float tmin, float tmax, float* t, float* u, float* v, float* det_out) {
    vec3 p = cross(d, tr.e2);
    float det = dot(tr.e1, p);
    if (det == 0.0f) return false;
    float inv = 1.0f / det;
    vec3 s = o - tr.v0;
    float uu = dot(s, p) * inv;
    if (!(uu >= 0.0f && uu <= 1.0f)) return false;
    vec3 q = cross(s, tr.e1);
    float vv = dot(d, q) * inv;
    if (!(vv >= 0.0f && uu + vv <= 1.0f)) return false;
    float tt = dot(tr.e2, q) * inv;
    if (!(tt > tmin && tt < tmax)) return false;
    *t = tt; *u = uu; *v = vv; *det_out = det;
    return true;
}

class Tracer {
public:
    Tracer(const Scene& s, bool use_bvh) : sc(s), bvh(use_bvh && !s.bvh_nodes.empty()) {}
    Hit closest(vec3 o, vec3 d, float tmin, float tmax, TraceStats& st) const;
    bool any(vec3 o, vec3 d, float tmin, float tmax, TraceStats& st) const;   // true = occluded
private:
    const Scene& sc;
    bool bvh;
};

} // namespace orc
