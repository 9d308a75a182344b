// frt_trace.hpp — software ray traversal: replaces rayQueryInitialize / rayQueryProceed /
// rayQueryGetCommittedIntersection (gbuffer.wgsl:108-112, restir.wgsl:376-380, :601-607,
// restir_spatial.wgsl:397-399, :627-633), i.e. the Vulkan driver's BVH walk and ray/triangle test.
//
// Hit semantics (they do not depend on the tree):
//   triangle hit  <=>  det != 0, 0 <= u <= 1, v >= 0, u + v <= 1, tmin < t < tmax   (Möller–Trumbore, f32, no FMA)
//   closest hit    =   min t, ties -> smallest flattened triangle id
//   any hit        =   exists a triangle hit (RayDesc flag 0x4)
//   front face    <=>  det > 0 (xor instance flip)
// Box tests may use fmaf: they only prune, and boxes are padded on the host (frt_bvh.cpp).
#pragma once
#include "frt_math.hpp"

namespace frt {

static const int kLdsTopNodes = 5;      // quad nodes 0 .. 4 (the root and, numbered breadth-first, its children) are copied into a workgroup's LDS (trace4)
#ifndef FRT_STACK
#define FRT_STACK 32      // (A/B builds only: a shallower LDS stack for scenes whose quad tree needs no more)
#endif
static const int kStackDepth = FRT_STACK;

struct InstanceView { uint32_t mesh_id, mat_id, first_tri, flip; float w2o[9]; float pad[3]; };   // = InstanceDev (64 B)
struct MeshInfoView { uint32_t vertex_offset, index_offset, pad0, pad1; };
struct VertexAttrView { float normal[2]; float uv[2]; float tangent[4]; };
struct MaterialView {
    float base_color[4]; float emissive_factor[3]; float roughness; float metallic, transmission, ior; int32_t light_index;
    uint32_t tex_info_0, tex_info_1, tex_info_2, pad_final;
};
struct LightView { float position[3]; uint32_t type_; float u[3]; float area; float v[3]; uint32_t pad; float emission[4]; };

// Read-only scene replica resident in HBM (one per GPU). All arrays are 16-byte aligned.
struct SceneView {
    const float4* nodes;        // pair nodes, 4 x float4 each
    const float4* tris;         // triangle slots, 3 x float4 each
    const float4* shade_tris;   // shading records, 8 x float4 per flattened triangle id (frt_shade.hpp: fetch_hit_geometry)
    const InstanceView* instances;
    const MeshInfoView* mesh_infos;
    const VertexAttrView* attributes;
    const uint32_t* indices;
    const MaterialView* materials;
    const LightView* lights;
    const uint8_t* color_tex;   // layers of 1024*1024*4 bytes (sRGB8)
    const uint8_t* data_tex;    // layers of 1024*1024*4 bytes (unorm8)
    const float* srgb_lut;      // 256 floats
    uint32_t num_materials, num_lights, num_nodes, num_tris;
    // quantized pair nodes (frt_bvh.cpp: quantize_pair_nodes): two 16-byte halves per node, SoA
    const uint4* qnode_a; const uint4* qnode_b;
    float qmin[3], qstep[3];
    uint32_t bvh_depth;
    uint32_t num_nodes4;        // quad nodes (in the padding before the pointer below)
    const float4* nodes4;       // quad nodes, 8 x float4 each (frt_bvh.cpp: build_quad_nodes): what the default kernels walk (trace4)
    const uint4* nodes8;        // 8-wide nodes with 16-bit grid boxes, 8 x uint4 each (frt_bvh8.hpp; trace8), or null: the scene has no such tree
    const float4* tris8;        // the triangle slots in that tree's order
    uint32_t num_nodes8, stack_need8;
};

struct HitRec {
    float t, u, v;
    uint32_t tri;      // flattened triangle id, 0xFFFFFFFF = miss
    uint32_t inst;
    bool front;
};

struct RayCounters { uint32_t closest, any; };

// Möller–Trumbore, contract arithmetic (identical operation order in the CPU checker).
FRT_HD bool intersect_tri(f3 v0, f3 e1, f3 e2, f3 o, f3 d, float tmin, float tmax, float& t, float& u, float& v, float& det_out) {
    f3 p = cross(d, e2);
    float det = dot(e1, p);
    if (det == 0.0f) return false;
    float inv = 1.0f / det;
    f3 s = o - v0;
    float uu = dot(s, p) * inv;
    if (!(uu >= 0.0f && uu <= 1.0f)) return false;
    f3 q = cross(s, e1);
    float vv = dot(d, q) * inv;
    if (!(vv >= 0.0f && uu + vv <= 1.0f)) return false;
    float tt = dot(e2, q) * inv;
    if (!(tt > tmin && tt < tmax)) return false;
    t = tt; u = uu; v = vv; det_out = det;
    return true;
}

// Slab test of one child box against the ray interval [tmin, tlim]. Not contract code (fmaf allowed).
FRT_HD bool slab(float lox, float loy, float loz, float hix, float hiy, float hiz, f3 inv, f3 oinv, float tmin, float tlim, float& tnear) {
    float x0 = __builtin_fmaf(lox, inv.x, oinv.x), x1 = __builtin_fmaf(hix, inv.x, oinv.x);
    float y0 = __builtin_fmaf(loy, inv.y, oinv.y), y1 = __builtin_fmaf(hiy, inv.y, oinv.y);
    float z0 = __builtin_fmaf(loz, inv.z, oinv.z), z1 = __builtin_fmaf(hiz, inv.z, oinv.z);
    float tn = fmaxn(fmaxn(fminn(x0, x1), fminn(y0, y1)), fmaxn(fminn(z0, z1), tmin));
    float tf = fminn(fminn(fmaxn(x0, x1), fmaxn(y0, y1)), fminn(fmaxn(z0, z1), tlim));
    tnear = tn;
    return tn <= tf * 1.0000004f;
}

// Both child boxes of a pair node at once. Node layout (frt_bvh.cpp: put_box): one float4 per axis = (lo0, lo1, hi0, hi1), so that the
// six ray-plane distances of an axis pair up as two packed fma (v_pk_fma_f32: two results per issue slot) straight out of the load.
typedef float frt_v2f __attribute__((ext_vector_type(2)));
FRT_HD void slab2(float4 qx, float4 qy, float4 qz, f3 inv, f3 oinv, float tmin, float tlim, float& t0, float& t1, bool& h0, bool& h1) {
    const frt_v2f ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
    const frt_v2f ox = {oinv.x, oinv.x}, oy = {oinv.y, oinv.y}, oz = {oinv.z, oinv.z};
    const frt_v2f xl = __builtin_elementwise_fma(frt_v2f{qx.x, qx.y}, ix, ox), xh = __builtin_elementwise_fma(frt_v2f{qx.z, qx.w}, ix, ox);
    const frt_v2f yl = __builtin_elementwise_fma(frt_v2f{qy.x, qy.y}, iy, oy), yh = __builtin_elementwise_fma(frt_v2f{qy.z, qy.w}, iy, oy);
    const frt_v2f zl = __builtin_elementwise_fma(frt_v2f{qz.x, qz.y}, iz, oz), zh = __builtin_elementwise_fma(frt_v2f{qz.z, qz.w}, iz, oz);
    float tn0 = fmaxn(fmaxn(fminn(xl.x, xh.x), fminn(yl.x, yh.x)), fmaxn(fminn(zl.x, zh.x), tmin));
    float tf0 = fminn(fminn(fmaxn(xl.x, xh.x), fmaxn(yl.x, yh.x)), fminn(fmaxn(zl.x, zh.x), tlim));
    float tn1 = fmaxn(fmaxn(fminn(xl.y, xh.y), fminn(yl.y, yh.y)), fmaxn(fminn(zl.y, zh.y), tmin));
    float tf1 = fminn(fminn(fmaxn(xl.y, xh.y), fmaxn(yl.y, yh.y)), fminn(fmaxn(zl.y, zh.y), tlim));
    t0 = tn0; t1 = tn1;
    h0 = tn0 <= tf0 * 1.0000004f;
    h1 = tn1 <= tf1 * 1.0000004f;
}
// 1 / d for the box tests with |d| clamped to 2^-80: keeps inv finite so that fma(b, inv, -o*inv) never evaluates inf - inf (a ray lying
// in an axis plane, d.y == 0, is common here: reconnection rays along a wall). Hardware reciprocal (1 ulp) on the device: pruning only,
// not contract arithmetic — the boxes are padded and the test has its own slack.
FRT_HD float prune_rcp(float x) {
    const float kTiny = 8.271806125530277e-25f;
    const float c = fabsf_(x) > kTiny ? x : __builtin_copysignf(kTiny, x);
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(c);
#else
    return 1.0f / c;
#endif
}

// ANY = true: terminate on the first accepted hit (shadow / visibility rays); returns hit.tri != miss.
// `stk` is this lane's traversal stack: kStackDepth entries, `stride` words apart (LDS column on the device).
template <bool ANY>
FRT_HD void trace(const SceneView& sc, f3 o, f3 d, float tmin, float tmax, uint32_t* stk, uint32_t stride, HitRec& hit) {
    hit.t = tmax; hit.tri = 0xFFFFFFFFu; hit.u = 0.0f; hit.v = 0.0f; hit.inst = 0u; hit.front = false;
    float best_det = 0.0f;
    f3 inv = mk3(prune_rcp(d.x), prune_rcp(d.y), prune_rcp(d.z));
    f3 oinv = mk3(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z);
    // "while-while" traversal: an inner loop walks pair nodes until THIS lane holds a leaf (lanes that already do wait for
    // the rest of the wave), then the leaf's triangles are tested together (measured 3 % faster than one loop with a branch).
    const uint32_t kDone = 0xFFFFFFFFu;   // leaf flag set, so it also ends the node loop
    int sp = 0;
    uint32_t cur = 0u;   // pair node 0 is the root
    for (;;) {
        while (!(cur & 0x80000000u)) {
            const float4* n = sc.nodes + (size_t)cur * 4u;
            float4 q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3];
            float tlim = ANY ? tmax : hit.t;
            float t0, t1;
            bool h0, h1;
            slab2(q0, q1, q2, inv, oinv, tmin, tlim, t0, t1, h0, h1);
            uint32_t r0 = f2u(q3.x), r1 = f2u(q3.y);
            h0 = h0 && (r0 != kDone);   // absent child (single-leaf scenes)
            h1 = h1 && (r1 != kDone);
            if (h0 && h1) {
                bool swap = t1 < t0;
                uint32_t nearr = swap ? r1 : r0, farr = swap ? r0 : r1;
                stk[(uint32_t)sp * stride] = farr; ++sp;
                cur = nearr;
            } else if (h0) cur = r0;
            else if (h1) cur = r1;
            else if (sp == 0) cur = kDone;
            else { --sp; cur = stk[(uint32_t)sp * stride]; }
        }
        if (cur == kDone) break;
        uint32_t first = cur & 0x00FFFFFFu, count = (cur >> 24) & 0x7Fu;
        for (uint32_t k = 0; k < count; ++k) {
            const float4* tp = sc.tris + (size_t)(first + k) * 3u;
            float4 a = tp[0], b = tp[1], c = tp[2];
            float t, u, v, det;
            if (intersect_tri(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), o, d, tmin, tmax, t, u, v, det)) {
                uint32_t id = f2u(a.w);
                if (ANY) { hit.tri = id; hit.t = t; return; }
                if (t < hit.t || (t == hit.t && id < hit.tri)) {
                    hit.t = t; hit.u = u; hit.v = v; hit.tri = id; hit.inst = f2u(b.w); best_det = det;
                }
            }
        }
        if (sp == 0) break;
        --sp; cur = stk[(uint32_t)sp * stride];
    }
    if (!ANY && hit.tri != 0xFFFFFFFFu) {
        bool front = best_det > 0.0f;
        if (sc.instances[hit.inst].flip) front = !front;
        hit.front = front;
    }
}

// ---- quad nodes -----------------------------------------------------------------------------------------------------------------
// Four child boxes per node (frt_bvh.cpp: build_quad_nodes): one dependent fetch (7 x 16 B) decides two levels of the binary tree, so a
// ray takes about half as many node steps, each with four independent slab tests. Closest-hit rays visit the hit children near to far
// (a five-exchange sorting network on (entry distance, reference)) — any-hit rays too: which triangles are hit does not depend on the
// order (hit semantics above), but near-first finds an occluder sooner (slot order measured 10 % slower per traced stage).
// Near / far planes are picked by the sign of the ray direction: the near planes of a child are its lo planes for a positive direction
// component and its hi planes for a negative one, so no min / max per axis is needed (4 instead of 10 instructions per child; fma is
// monotonic, so these are the values min / max would select). `n` is the uniform base of the node array and sx, sy, sz the 32-bit byte
// offsets of this lane's near planes (node offset | axis offset 0 / 32 / 64 | 16 where the direction component is negative; the far planes
// are at that offset ^ 16): uniform base + 32-bit lane offset keeps the seven loads of a
// step in the scalar-base addressing form (one offset register each instead of a 64-bit address).
// The 24 plane distances are plain v_fma_f32: the packed form (12 v_pk_fma_f32, as in slab2) measured 1.8 % slower per frame here — a packed
// fma issues no faster than two scalar ones on this chip and ties its operands to aligned register pairs.
FRT_HD void slab4(const char* n, uint32_t sx, uint32_t sy, uint32_t sz, f3 inv, f3 oinv, float tmin, float tlim, float t[4], bool h[4]) {
    const char* base = n;
    const float4 nx = *reinterpret_cast<const float4*>(base + sx), fx = *reinterpret_cast<const float4*>(base + (sx ^ 16u));
    const float4 ny = *reinterpret_cast<const float4*>(base + sy), fy = *reinterpret_cast<const float4*>(base + (sy ^ 16u));
    const float4 nz = *reinterpret_cast<const float4*>(base + sz), fz = *reinterpret_cast<const float4*>(base + (sz ^ 16u));
    const float xn[4] = {__builtin_fmaf(nx.x, inv.x, oinv.x), __builtin_fmaf(nx.y, inv.x, oinv.x), __builtin_fmaf(nx.z, inv.x, oinv.x), __builtin_fmaf(nx.w, inv.x, oinv.x)};
    const float xf[4] = {__builtin_fmaf(fx.x, inv.x, oinv.x), __builtin_fmaf(fx.y, inv.x, oinv.x), __builtin_fmaf(fx.z, inv.x, oinv.x), __builtin_fmaf(fx.w, inv.x, oinv.x)};
    const float yn[4] = {__builtin_fmaf(ny.x, inv.y, oinv.y), __builtin_fmaf(ny.y, inv.y, oinv.y), __builtin_fmaf(ny.z, inv.y, oinv.y), __builtin_fmaf(ny.w, inv.y, oinv.y)};
    const float yf[4] = {__builtin_fmaf(fy.x, inv.y, oinv.y), __builtin_fmaf(fy.y, inv.y, oinv.y), __builtin_fmaf(fy.z, inv.y, oinv.y), __builtin_fmaf(fy.w, inv.y, oinv.y)};
    const float zn[4] = {__builtin_fmaf(nz.x, inv.z, oinv.z), __builtin_fmaf(nz.y, inv.z, oinv.z), __builtin_fmaf(nz.z, inv.z, oinv.z), __builtin_fmaf(nz.w, inv.z, oinv.z)};
    const float zf[4] = {__builtin_fmaf(fz.x, inv.z, oinv.z), __builtin_fmaf(fz.y, inv.z, oinv.z), __builtin_fmaf(fz.z, inv.z, oinv.z), __builtin_fmaf(fz.w, inv.z, oinv.z)};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float tn = fmaxn(fmaxn(xn[c], yn[c]), fmaxn(zn[c], tmin));
        const float tf = fminn(fminn(xf[c], yf[c]), fminn(zf[c], tlim));
        t[c] = tn;
        // No slack factor here (slab2 has one): the host pads every box by 1e-4 of the scene extent (frt_bvh.cpp), i.e. the entry / exit
        // distances of a ray that hits a triangle inside are at least 1e-4 |1/d| apart from the hit on either side, four hundred times the
        // rounding of the two fma and of the 1-ulp reciprocal (~2.4e-7 |1/d| for coordinates of the order of the extent).
        h[c] = tn <= tf;
    }
}
// VOTE: the node loop ends when the wave VOTES for a leaf step instead of when its last lane has reached a leaf. A lane's own sequence of node and
// leaf steps is fixed by its ray; the wave executes a common supersequence of its lanes' sequences, and "while-while" (node steps until no lane holds
// a node) is one heuristic for building it: every lane that has reached a leaf waits for the slowest node walker of the round. With VOTE every trip
// of the node loop counts the lanes that wait for a node step and those that wait for a leaf step (two ballots) and leaves for a leaf step as soon as
// the latter are the majority; lanes still holding a node sit that leaf step out. Same hits (hit semantics above do not depend on the order of the
// steps). Pays on DEEP trees, where walks are long and their phases drift apart, and costs a little on shallow ones — its own instructions against
// the steps it saves: 246k-triangle colonnade 21.0 -> 19.8 ms per 4K frame, 82k-triangle blob 2.95 -> 2.99 ms, 32k-triangle ReSTIR scene 1.28 ->
// 1.34 ms, Cornell Box 1.57 -> 1.61 ms (profiles/r3_experiments/traversal_in_situ.md). The renderer picks the kernels by the size of the scene's
// quad tree (frt_renderer.hip: kVoteMinQuadNodes). On the host (tests/hostcheck) a "wave" is one lane: the loop degenerates to the lane's own sequence.
FRT_HD uint32_t wave_count(bool b) {      // lanes of the wave for which b holds
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popcll(__ballot(b));
#else
    return b ? 1u : 0u;
#endif
}
template <bool ANY, bool VOTE = false>
FRT_HD void trace4(const SceneView& sc, f3 o, f3 d, float tmin, float tmax, uint32_t* stk, uint32_t stride, HitRec& hit, const uint32_t* lds_top = nullptr) {
    hit.t = tmax; hit.tri = 0xFFFFFFFFu; hit.u = 0.0f; hit.v = 0.0f; hit.inst = 0u; hit.front = false;
    float best_det = 0.0f;
    f3 inv = mk3(prune_rcp(d.x), prune_rcp(d.y), prune_rcp(d.z));
    f3 oinv = mk3(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z);
    const uint32_t sx = (f2u(d.x) >> 31) << 4, sy = 32u | ((f2u(d.y) >> 31) << 4), sz = 64u | ((f2u(d.z) >> 31) << 4);
    const uint32_t kDone = 0xFFFFFFFFu;
    const float kFar = 3.0e38f;
    uint32_t* top = stk;      // next free stack entry (entries are `stride` words apart)
    uint32_t cur = 0u;   // quad node 0 is the root
    // node loop goes on: this lane holds a node (while-while) / the wave's vote says "node step" (VOTE; wave-uniform)
    auto node_phase = [&]() -> bool {
        const bool at_node = !(cur & 0x80000000u);
        if (!VOTE) return at_node;
        // (a lane that is done has left the outer loop, or holds kDone until this loop ends: it waits for neither step)
        const uint32_t nn = wave_count(at_node), nl = wave_count(!at_node && cur != kDone);
        return nn != 0u && nn >= nl;
    };
#if defined(__HIP_DEVICE_COMPILE__)
    // The first steps of every walk — the root, then one of its children (quad nodes 1 .. 4: the tree is numbered breadth-first) — read their node from
    // the workgroup's LDS copy of nodes 0 .. 4 while EVERY lane of the wave is still up there (wave-uniform test): two of a walk's ten node steps
    // leave the L1 alone and see the LDS's latency instead: Cornell Box 1.560 -> 1.537 ms per frame (8 more VGPRs, still four waves per SIMD).
    if (lds_top) {
        while (__ballot(cur >= (uint32_t)kLdsTopNodes) == 0ull) {
            const uint32_t noff = cur << 7;
            const char* nb = reinterpret_cast<const char*>(lds_top);
            const float4 rf = *reinterpret_cast<const float4*>(nb + (noff + 96u));
            float t[4]; bool h[4];
            slab4(nb, noff | sx, noff | sy, noff | sz, inv, oinv, tmin, ANY ? tmax : hit.t, t, h);
            uint32_t r[4] = {f2u(rf.x), f2u(rf.y), f2u(rf.z), f2u(rf.w)};
            float k[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) k[c] = h[c] ? t[c] : kFar;
#define FRT_CE(a, b) { const bool s_ = k[b] < k[a]; const float ka_ = s_ ? k[b] : k[a], kb_ = s_ ? k[a] : k[b]; \
                       const uint32_t ra_ = s_ ? r[b] : r[a], rb_ = s_ ? r[a] : r[b]; k[a] = ka_; k[b] = kb_; r[a] = ra_; r[b] = rb_; }
            FRT_CE(0, 1) FRT_CE(2, 3) FRT_CE(0, 2) FRT_CE(1, 3) FRT_CE(1, 2)
#undef FRT_CE
            if (k[3] < kFar) { *top = r[3]; top += stride; }
            if (k[2] < kFar) { *top = r[2]; top += stride; }
            if (k[1] < kFar) { *top = r[1]; top += stride; }
            if (k[0] < kFar) cur = r[0];
            else if (top == stk) cur = kDone;
            else { top -= stride; cur = *top; }
        }
    }
#endif
    for (;;) {
        while (node_phase()) {
            if (VOTE && (cur & 0x80000000u)) continue;      // this lane holds a leaf (or is done): it sits the node step out
            const uint32_t noff = cur << 7;
            const char* nb = reinterpret_cast<const char*>(sc.nodes4);
            const float4 rf = *reinterpret_cast<const float4*>(nb + (noff + 96u));
            float t[4]; bool h[4];
            slab4(nb, noff | sx, noff | sy, noff | sz, inv, oinv, tmin, ANY ? tmax : hit.t, t, h);
            uint32_t r[4] = {f2u(rf.x), f2u(rf.y), f2u(rf.z), f2u(rf.w)};
            // (an empty slot holds a far-away degenerate box: it never passes the slab test, no reference check needed)
            float k[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) k[c] = h[c] ? t[c] : kFar;
#define FRT_CE(a, b) { const bool s_ = k[b] < k[a]; const float ka_ = s_ ? k[b] : k[a], kb_ = s_ ? k[a] : k[b]; \
                       const uint32_t ra_ = s_ ? r[b] : r[a], rb_ = s_ ? r[a] : r[b]; k[a] = ka_; k[b] = kb_; r[a] = ra_; r[b] = rb_; }
            FRT_CE(0, 1) FRT_CE(2, 3) FRT_CE(0, 2) FRT_CE(1, 3) FRT_CE(1, 2)
#undef FRT_CE
            if (k[3] < kFar) { *top = r[3]; top += stride; }
            if (k[2] < kFar) { *top = r[2]; top += stride; }
            if (k[1] < kFar) { *top = r[1]; top += stride; }
            if (k[0] < kFar) cur = r[0];
            else if (top == stk) cur = kDone;
            else { top -= stride; cur = *top; }
        }
        if (cur == kDone) break;
        if (VOTE && !(cur & 0x80000000u)) continue;      // this lane still holds a node: it sits the leaf step out
        uint32_t first = cur & 0x00FFFFFFu, count = (cur >> 24) & 0x7Fu;
        // leaves hold one or two triangles (frt_bvh.cpp; up to four under FRT_BVH_LEAF): the first two are tested in line, without a loop
        auto test3 = [&](float4 a, float4 b, float4 c) -> bool {
            float t, u, v, det;
            if (intersect_tri(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), o, d, tmin, tmax, t, u, v, det)) {
                uint32_t id = f2u(a.w);
                if (ANY) { hit.tri = id; hit.t = t; return true; }
                if (t < hit.t || (t == hit.t && id < hit.tri)) {
                    hit.t = t; hit.u = u; hit.v = v; hit.tri = id; hit.inst = f2u(b.w); best_det = det;
                }
            }
            return false;
        };
        auto test = [&](uint32_t slot) -> bool { const float4* tp = sc.tris + (size_t)slot * 3u; return test3(tp[0], tp[1], tp[2]); };
        // The walk is a chain of dependent round trips to the L1 / LDS, so what a leaf step fetches is fetched TOGETHER, up front (round 3):
        // * both triangles of a two-triangle leaf (consecutive slots) — one round trip per leaf step instead of two: Cornell Box 1.614 -> 1.574 ms per
        //   frame, ReSTIR scene +3.3 % Mrays/s, 82k-triangle blob 3.30 -> 3.09 ms, 246k-triangle colonnade 21.5 -> 19.9 ms (4 - 8 more VGPRs);
        // * the stack entry the lane continues with (it does not depend on the tests): 1.571 -> 1.555 ms, blob 3.08 -> 3.02 ms.
        // (Fetching the NODE behind that entry as well and stepping it at once costs 10 VGPRs the kernels do not have: scratch, 1.555 -> 1.742 ms.)
        const float4* tp = sc.tris + (size_t)first * 3u;
        const float4 a0 = tp[0], b0 = tp[1], c0 = tp[2];
        float4 a1 = a0, b1 = b0, c1 = c0;      // (left uninitialised the allocator needs 5 - 8 MORE registers)
        if (count > 1u) { a1 = tp[3]; b1 = tp[4]; c1 = tp[5]; }
        const bool more = top != stk;
        uint32_t next = kDone;
        if (more) next = *(top - stride);
        if (test3(a0, b0, c0)) return;
        if (count > 1u && test3(a1, b1, c1)) return;
        for (uint32_t kk = 2u; kk < count; ++kk) if (test(first + kk)) return;
        if (!more) break;
        top -= stride; cur = next;
    }
    if (!ANY && hit.tri != 0xFFFFFFFFu) {
        bool front = best_det > 0.0f;
        if (sc.instances[hit.inst].flip) front = !front;
        hit.front = front;
    }
}


// ---- 8-wide nodes with grid boxes (frt_bvh8.hpp) -------------------------------------------------------------------------------------------
// One node step decides three levels of the binary tree: eight slab tests on planes reconstructed from 16-bit grid coordinates (t = q * (step / d) +
// (p - o) / d: one conversion and one fma per plane, the near / far plane blocks picked by the ray's direction signs at load time like the quad
// node's), the hit children split into LEAF children — tested next, one leaf child (<= 2 triangles, fetched together) per leaf step — and INNER
// children, of which one is entered and the others wait on the stack as ONE word, (child base << 16) | (inner mask << 8) | (mask still to visit).
// Order of the inner children: increasing slot ^ octant (the builder put a child into the slot that says on which side of the node it lies, so this
// is near side first on every axis; no sort); an any-hit ray enters the NEAREST hit inner child first, exactly (it only has to find an occluder: in the
// host model of a wave's lockstep walk, tools/bvh_quality.cpp, 2.9 + 1.8 node + leaf steps per shadow wave-ray instead of 6.6 + 3.3 in octant order —
// the quad tree: 6.8 + 1.8 — while a bounce ray's 5.7 + 4.6 — quad tree 10.7 + 3.9 — do not depend on it). Same hits as every other walk (hit
// semantics at the top of this file; the grid boxes contain the float boxes).
// The stack holds one word per level with two or more inner children hit: kStack8 entries cover every tree the renderer hands to this walk
// (SceneView::stack_need8; a deeper tree keeps the quad walk, frt_renderer.hip).
static const int kStack8 = 8;
FRT_HD uint32_t ctz32(uint32_t x) { return (uint32_t)__builtin_ctz(x); }
FRT_HD uint32_t popc32(uint32_t x) { return (uint32_t)__builtin_popcount(x); }
// the set bit s of the 8-bit mask m (non-zero) with the smallest s ^ octant; om = A0 | A1 << 8 | A2 << 16, A_k = the slots whose bit k equals the octant's
FRT_HD uint32_t pick_octant(uint32_t m, uint32_t om) {
    uint32_t t = m & (om >> 16); m = t ? t : m;
    t = m & (om >> 8); m = t ? t : m;
    t = m & om; m = t ? t : m;
    return ctz32(m);
}
FRT_HD float half_lo(uint32_t w) { return (float)(w & 0xFFFFu); }      // (v_cvt_f32_u32 with an SDWA word select: one instruction each)
FRT_HD float half_hi(uint32_t w) { return (float)(w >> 16); }
// NODES: the node array's address space is the caller's business — `nb` points at node 0 (HBM, or a workgroup's LDS copy of the whole tree).
template <bool ANY>
FRT_HD void trace8(const SceneView& sc, const char* nb, f3 o, f3 d, float tmin, float tmax, uint32_t* stk, uint32_t stride, HitRec& hit) {
    hit.t = tmax; hit.tri = 0xFFFFFFFFu; hit.u = 0.0f; hit.v = 0.0f; hit.inst = 0u; hit.front = false;
    float best_det = 0.0f;
    const f3 inv = mk3(prune_rcp(d.x), prune_rcp(d.y), prune_rcp(d.z));
    const uint32_t nxo = 32u | ((f2u(d.x) >> 31) << 4), nyo = 64u | ((f2u(d.y) >> 31) << 4), nzo = 96u | ((f2u(d.z) >> 31) << 4);      // near plane blocks
    const uint32_t fxo = nxo ^ 16u, fyo = nyo ^ 16u, fzo = nzo ^ 16u;                                                                         // far plane blocks
    const uint32_t oct = (f2u(d.x) >> 31) | ((f2u(d.y) >> 31) << 1) | ((f2u(d.z) >> 31) << 2);
    const uint32_t om = (0x55u << (oct & 1u)) | ((0x33u << (oct & 2u)) << 8) | ((0x0Fu << (oct & 4u)) << 16);
    const uint32_t kNone = 0xFFFFFFFFu;
    uint32_t* top = stk;           // next free stack entry
    uint32_t cur = 0u;             // the node to step next (node 0 is the root), or kNone
    uint32_t T = 0u, meta0 = 0u, meta1 = 0u;      // leaf children still to test: (tri_base << 8) | mask, and the node's meta bytes
    // the next node from the stack: the topmost word's next child in octant order; the word stays while it has children left
    auto from_stack = [&]() {
        if (top == stk) { cur = kNone; return; }
        uint32_t g = *(top - stride);
        const uint32_t s = pick_octant(g & 0xFFu, om);
        cur = (g >> 16) + popc32((g >> 8) & ((1u << s) - 1u) & 0xFFu);
        g &= ~(1u << s);
        if (g & 0xFFu) *(top - stride) = g; else top -= stride;
    };
    for (;;) {
        while (T == 0u && cur != kNone) {
            const uint32_t noff = cur << 7;
            const uint4 h0 = *reinterpret_cast<const uint4*>(nb + noff), h1 = *reinterpret_cast<const uint4*>(nb + (noff + 16u));
            const uint4 qnx = *reinterpret_cast<const uint4*>(nb + (noff | nxo)), qfx = *reinterpret_cast<const uint4*>(nb + (noff | fxo));
            const uint4 qny = *reinterpret_cast<const uint4*>(nb + (noff | nyo)), qfy = *reinterpret_cast<const uint4*>(nb + (noff | fyo));
            const uint4 qnz = *reinterpret_cast<const uint4*>(nb + (noff | nzo)), qfz = *reinterpret_cast<const uint4*>(nb + (noff | fzo));
            const float sx = u2f((h0.w & 0xFFu) << 23) * inv.x, sy = u2f(((h0.w >> 8) & 0xFFu) << 23) * inv.y, sz = u2f(((h0.w >> 16) & 0xFFu) << 23) * inv.z;
            const float bx = (u2f(h0.x) - o.x) * inv.x, by = (u2f(h0.y) - o.y) * inv.y, bz = (u2f(h0.z) - o.z) * inv.z;
            const float tlim = ANY ? tmax : hit.t;
            const uint32_t imask = h0.w >> 24;
            const uint32_t wnx[4] = {qnx.x, qnx.y, qnx.z, qnx.w}, wfx[4] = {qfx.x, qfx.y, qfx.z, qfx.w};
            const uint32_t wny[4] = {qny.x, qny.y, qny.z, qny.w}, wfy[4] = {qfy.x, qfy.y, qfy.z, qfy.w};
            const uint32_t wnz[4] = {qnz.x, qnz.y, qnz.z, qnz.w}, wfz[4] = {qfz.x, qfz.y, qfz.z, qfz.w};
            uint32_t hits = 0u, kmin = 0xFFFFFFFFu;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int w = c >> 1;
                const float nx = (c & 1) ? half_hi(wnx[w]) : half_lo(wnx[w]), fx = (c & 1) ? half_hi(wfx[w]) : half_lo(wfx[w]);
                const float ny = (c & 1) ? half_hi(wny[w]) : half_lo(wny[w]), fy = (c & 1) ? half_hi(wfy[w]) : half_lo(wfy[w]);
                const float nz = (c & 1) ? half_hi(wnz[w]) : half_lo(wnz[w]), fz = (c & 1) ? half_hi(wfz[w]) : half_lo(wfz[w]);
                const float tn = fmaxn(fmaxn(__builtin_fmaf(nx, sx, bx), __builtin_fmaf(ny, sy, by)), fmaxn(__builtin_fmaf(nz, sz, bz), tmin));
                const float tf = fminn(fminn(__builtin_fmaf(fx, sx, bx), __builtin_fmaf(fy, sy, by)), fminn(__builtin_fmaf(fz, sz, bz), tlim));
                // (no slack factor: the host pads every box by 1e-4 of the scene extent, four hundred times the rounding of these fma; slab4)
                const bool h = tn <= tf;
                hits |= h ? (1u << c) : 0u;
                if (ANY) {      // the nearest hit INNER child: entry distance (> 0, so its bits order like the float) with the slot in the three lowest bits
                    const uint32_t key = (h && ((imask >> c) & 1u)) ? ((f2u(tn) & ~7u) | (uint32_t)c) : 0xFFFFFFFFu;
                    kmin = key < kmin ? key : kmin;
                }
            }
            const uint32_t ih = hits & imask;
            // (the header's second half is used under conditions only: left alone, the compiler sinks the load of child_base into the branch below —
            // a second dependent round trip in every node step. Pinning the words here keeps all eight loads of a step in flight together.)
            uint32_t child_base = h1.x, tri_word = h1.y, m0 = h1.z, m1 = h1.w;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(child_base), "+v"(tri_word), "+v"(m0), "+v"(m1));
#endif
            T = hits & (tri_word >> 24);
            if (T) { T |= tri_word << 8; meta0 = m0; meta1 = m1; }
            if (ih) {
                const uint32_t s = ANY ? (kmin & 7u) : pick_octant(ih, om);
                cur = child_base + popc32(imask & ((1u << s) - 1u));
                const uint32_t rem = ih & ~(1u << s);
                if (rem) { *top = (child_base << 16) | (imask << 8) | rem; top += stride; }
            } else if (T) cur = kNone;      // (the stack is looked at after this node's leaves)
            else from_stack();
        }
        if (T == 0u) break;      // nothing to step, nothing on the stack
        // one leaf child of the node stepped last: its (one or two) triangles fetched together, like trace4's leaf step
        {
            const uint32_t lh = T & 0xFFu;
            const uint32_t s = ANY ? pick_octant(lh, om) : ctz32(lh);
            T &= ~(1u << s);
            const uint32_t meta = ((s & 4u) ? meta1 : meta0) >> ((s & 3u) << 3);
            const uint32_t first = (T >> 8) + (meta & 31u), count = (meta >> 5) & 7u;
            auto test3 = [&](float4 a, float4 b, float4 c) -> bool {
                float t, u, v, det;
                if (intersect_tri(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), o, d, tmin, tmax, t, u, v, det)) {
                    uint32_t id = f2u(a.w);
                    if (ANY) { hit.tri = id; hit.t = t; return true; }
                    if (t < hit.t || (t == hit.t && id < hit.tri)) {
                        hit.t = t; hit.u = u; hit.v = v; hit.tri = id; hit.inst = f2u(b.w); best_det = det;
                    }
                }
                return false;
            };
            const float4* tp = sc.tris8 + (size_t)first * 3u;
            const float4 a0 = tp[0], b0 = tp[1], c0 = tp[2];
            float4 a1 = a0, b1 = b0, c1 = c0;
            if (count > 1u) { a1 = tp[3]; b1 = tp[4]; c1 = tp[5]; }
            if (test3(a0, b0, c0)) return;
            if (count > 1u && test3(a1, b1, c1)) return;
            for (uint32_t kk = 2u; kk < count; ++kk) { const float4* tq = tp + (size_t)kk * 3u; if (test3(tq[0], tq[1], tq[2])) return; }
            if ((T & 0xFFu) == 0u) { T = 0u; if (cur == kNone) from_stack(); }
        }
    }
    if (!ANY && hit.tri != 0xFFFFFFFFu) {
        bool front = best_det > 0.0f;
        if (sc.instances[hit.inst].flip) front = !front;
        hit.front = front;
    }
}

// ---- quantized pair nodes, cacheable in LDS -----------------------------------------------------------------------------------
// A pair node holds the boxes of both children on a 16-bit grid over the scene box: 2 x (6 x u16 + child reference) = 32 bytes, half the
// float form, and 25 KB for the whole Cornell Box tree: it fits in LDS next to the triangle slots. Boxes are rounded OUTWARD by one
// extra quantum on the host, so the quantized tree prunes a little less than the float tree and never more; which triangles are hit is
// decided by the exact triangle test alone (hit semantics above), so results are bit-identical to trace().
//   half A: x = lo0.x | lo0.y << 16, y = lo0.z | hi0.x << 16, z = hi0.y | hi0.z << 16, w = reference of child 0;   half B: child 1
// The ray is moved into grid units once (inv_q = qstep * inv, oinv_q = (qmin - o) * inv); a box test then costs 6 conversions + 6 fma.
// Accessor over plain memory (HBM on the device, host arrays in tests/hostcheck). The resident kernels use their own accessor with
// LDS-typed pointers (frt_kernels.hip: LdsBvh); trace_q is a template on the accessor so that each gets the right load instructions.
struct QBvh {
    const uint4* a; const uint4* b;     // the two halves of every quantized pair node
    const float4* tris;                 // triangle slots, 3 x float4 each
    f3 qmin, qstep;
    FRT_HD void node(uint32_t i, uint4& qa, uint4& qb) const { qa = a[i]; qb = b[i]; }
    FRT_HD void tri(uint32_t slot, float4& t0, float4& t1, float4& t2) const { const float4* p = tris + (size_t)slot * 3u; t0 = p[0]; t1 = p[1]; t2 = p[2]; }
};
FRT_HD bool slab_q(uint32_t w0, uint32_t w1, uint32_t w2, f3 inv, f3 oinv, float tmin, float tlim, float& tnear) {
    float lox = (float)(w0 & 0xFFFFu), loy = (float)(w0 >> 16), loz = (float)(w1 & 0xFFFFu);
    float hix = (float)(w1 >> 16), hiy = (float)(w2 & 0xFFFFu), hiz = (float)(w2 >> 16);
    return slab(lox, loy, loz, hix, hiy, hiz, inv, oinv, tmin, tlim, tnear);
}
template <bool ANY, class Bvh>
FRT_HD void trace_q(const SceneView& sc, const Bvh& bv, f3 o, f3 d, float tmin, float tmax, uint32_t* stk, uint32_t stride, HitRec& hit) {
    hit.t = tmax; hit.tri = 0xFFFFFFFFu; hit.u = 0.0f; hit.v = 0.0f; hit.inst = 0u; hit.front = false;
    float best_det = 0.0f;
    f3 inv = mk3(prune_rcp(d.x), prune_rcp(d.y), prune_rcp(d.z));
    f3 oinv = mk3((bv.qmin.x - o.x) * inv.x, (bv.qmin.y - o.y) * inv.y, (bv.qmin.z - o.z) * inv.z);
    inv = mk3(inv.x * bv.qstep.x, inv.y * bv.qstep.y, inv.z * bv.qstep.z);
    const uint32_t kDone = 0xFFFFFFFFu;
    int sp = 0;
    uint32_t cur = 0u;
    for (;;) {
        while (!(cur & 0x80000000u)) {
            uint4 qa, qb;
            bv.node(cur, qa, qb);
            float tlim = ANY ? tmax : hit.t;
            float t0, t1;
            bool h0 = slab_q(qa.x, qa.y, qa.z, inv, oinv, tmin, tlim, t0);
            bool h1 = slab_q(qb.x, qb.y, qb.z, inv, oinv, tmin, tlim, t1);
            uint32_t r0 = qa.w, r1 = qb.w;
            h0 = h0 && (r0 != kDone);
            h1 = h1 && (r1 != kDone);
            if (h0 && h1) {
                bool swap = t1 < t0;
                uint32_t nearr = swap ? r1 : r0, farr = swap ? r0 : r1;
                stk[(uint32_t)sp * stride] = farr; ++sp;
                cur = nearr;
            } else if (h0) cur = r0;
            else if (h1) cur = r1;
            else if (sp == 0) cur = kDone;
            else { --sp; cur = stk[(uint32_t)sp * stride]; }
        }
        if (cur == kDone) break;
        uint32_t first = cur & 0x00FFFFFFu, count = (cur >> 24) & 0x7Fu;
        for (uint32_t k = 0; k < count; ++k) {
            float4 a, b, c;
            bv.tri(first + k, a, b, c);
            float t, u, v, det;
            if (intersect_tri(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), o, d, tmin, tmax, t, u, v, det)) {
                uint32_t id = f2u(a.w);
                if (ANY) { hit.tri = id; hit.t = t; return; }
                if (t < hit.t || (t == hit.t && id < hit.tri)) {
                    hit.t = t; hit.u = u; hit.v = v; hit.tri = id; hit.inst = f2u(b.w); best_det = det;
                }
            }
        }
        if (sp == 0) break;
        --sp; cur = stk[(uint32_t)sp * stride];
    }
    if (!ANY && hit.tri != 0xFFFFFFFFu) {
        bool front = best_det > 0.0f;
        if (sc.instances[hit.inst].flip) front = !front;
        hit.front = front;
    }
}

} // namespace frt
