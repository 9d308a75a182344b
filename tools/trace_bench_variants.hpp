// trace_bench_variants.hpp — experimental traversal loops measured by tools/trace_bench.hip (not part of the product library).
#pragma once
typedef float tb_v4f __attribute__((ext_vector_type(4)));

// Loop shape: the seven loads of a node are issued as soon as the node is CHOSEN — right after the three exchanges of the sorting
// network that settle the nearest hit child (or after the pop) — and not at the top of the next trip: the rest of the sort, the pushes and
// the loop bookkeeping (~30 instructions of a lone wave's dependent stream) then run under the load latency instead of behind it.
template <bool ANY>
__device__ __forceinline__ void trace4_early(const SceneView& sc, f3 o, f3 d, float tmin, float tmax, uint32_t* stk, uint32_t stride, HitRec& hit) {
    hit.t = tmax; hit.tri = 0xFFFFFFFFu; hit.u = 0.0f; hit.v = 0.0f; hit.inst = 0u; hit.front = false;
    float best_det = 0.0f;
    f3 inv = mk3(prune_rcp(d.x), prune_rcp(d.y), prune_rcp(d.z));
    f3 oinv = mk3(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z);
    const uint32_t sx = (f2u(d.x) >> 31) << 4, sy = 32u | ((f2u(d.y) >> 31) << 4), sz = 64u | ((f2u(d.z) >> 31) << 4);
    const uint32_t kDone = 0xFFFFFFFFu;
    const float kFar = 3.0e38f;
    uint32_t* top = stk;      // next free stack entry (entries are `stride` words apart)
    const char* nb = reinterpret_cast<const char*>(sc.nodes4);
    tb_v4f nx, fx, ny, fy, nz, fz, rf;     // the node in flight: near / far planes per axis (slab4's layout), child references
#define FRT_FETCH(ref) { const uint32_t n_ = (ref) << 7; \
        nx = *reinterpret_cast<const tb_v4f*>(nb + (n_ | sx)); fx = *reinterpret_cast<const tb_v4f*>(nb + ((n_ | sx) ^ 16u)); \
        ny = *reinterpret_cast<const tb_v4f*>(nb + (n_ | sy)); fy = *reinterpret_cast<const tb_v4f*>(nb + ((n_ | sy) ^ 16u)); \
        nz = *reinterpret_cast<const tb_v4f*>(nb + (n_ | sz)); fz = *reinterpret_cast<const tb_v4f*>(nb + ((n_ | sz) ^ 16u)); \
        rf = *reinterpret_cast<const tb_v4f*>(nb + (n_ + 96u)); }
    // a lane that holds a leaf (or is done) has no node in flight: its seven registers hold nothing (an unspecified value keeps the old node's
    // data from staying live across the leaf phase)
#if defined(__HIP_DEVICE_COMPILE__)
#define FRT_NO_NODE { nx = __builtin_nondeterministic_value(nx); fx = __builtin_nondeterministic_value(fx); ny = __builtin_nondeterministic_value(ny); \
        fy = __builtin_nondeterministic_value(fy); nz = __builtin_nondeterministic_value(nz); fz = __builtin_nondeterministic_value(fz); rf = __builtin_nondeterministic_value(rf); }
#else
#define FRT_NO_NODE { nx = fx = ny = fy = nz = fz = rf = tb_v4f{0.0f, 0.0f, 0.0f, 0.0f}; }
#endif
    uint32_t cur = 0u;   // quad node 0 is the root
    FRT_FETCH(0u)
    for (;;) {
        while (!(cur & 0x80000000u)) {
            const float tlim = ANY ? tmax : hit.t;
            const float xn[4] = {__builtin_fmaf(nx.x, inv.x, oinv.x), __builtin_fmaf(nx.y, inv.x, oinv.x), __builtin_fmaf(nx.z, inv.x, oinv.x), __builtin_fmaf(nx.w, inv.x, oinv.x)};
            const float xf[4] = {__builtin_fmaf(fx.x, inv.x, oinv.x), __builtin_fmaf(fx.y, inv.x, oinv.x), __builtin_fmaf(fx.z, inv.x, oinv.x), __builtin_fmaf(fx.w, inv.x, oinv.x)};
            const float yn[4] = {__builtin_fmaf(ny.x, inv.y, oinv.y), __builtin_fmaf(ny.y, inv.y, oinv.y), __builtin_fmaf(ny.z, inv.y, oinv.y), __builtin_fmaf(ny.w, inv.y, oinv.y)};
            const float yf[4] = {__builtin_fmaf(fy.x, inv.y, oinv.y), __builtin_fmaf(fy.y, inv.y, oinv.y), __builtin_fmaf(fy.z, inv.y, oinv.y), __builtin_fmaf(fy.w, inv.y, oinv.y)};
            const float zn[4] = {__builtin_fmaf(nz.x, inv.z, oinv.z), __builtin_fmaf(nz.y, inv.z, oinv.z), __builtin_fmaf(nz.z, inv.z, oinv.z), __builtin_fmaf(nz.w, inv.z, oinv.z)};
            const float zf[4] = {__builtin_fmaf(fz.x, inv.z, oinv.z), __builtin_fmaf(fz.y, inv.z, oinv.z), __builtin_fmaf(fz.z, inv.z, oinv.z), __builtin_fmaf(fz.w, inv.z, oinv.z)};
            uint32_t r[4] = {f2u(rf.x), f2u(rf.y), f2u(rf.z), f2u(rf.w)};
            // (an empty slot holds a far-away degenerate box: it never passes the slab test, no reference check needed)
            float k[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float tn = fmaxn(fmaxn(xn[c], yn[c]), fmaxn(zn[c], tmin));
                const float tf = fminn(fminn(xf[c], yf[c]), fminn(zf[c], tlim));
                k[c] = tn <= tf ? tn : kFar;      // no slack factor: see slab4
            }
#define FRT_CE(a, b) { const bool s_ = k[b] < k[a]; const float ka_ = s_ ? k[b] : k[a], kb_ = s_ ? k[a] : k[b]; \
                       const uint32_t ra_ = s_ ? r[b] : r[a], rb_ = s_ ? r[a] : r[b]; k[a] = ka_; k[b] = kb_; r[a] = ra_; r[b] = rb_; }
            FRT_CE(0, 1) FRT_CE(2, 3) FRT_CE(0, 2)      // k[0] / r[0]: the nearest hit child, if any child is hit at all
            uint32_t next;
            if (k[0] < kFar) next = r[0];
            else if (top == stk) next = kDone;      // (nothing is hit: nothing will be pushed below either)
            else { top -= stride; next = *top; }
            if (!(next & 0x80000000u)) FRT_FETCH(next) else FRT_NO_NODE
            FRT_CE(1, 3) FRT_CE(1, 2)
#undef FRT_CE
            if (k[3] < kFar) { *top = r[3]; top += stride; }
            if (k[2] < kFar) { *top = r[2]; top += stride; }
            if (k[1] < kFar) { *top = r[1]; top += stride; }
            cur = next;
        }
        if (cur == kDone) break;
        uint32_t first = cur & 0x00FFFFFFu, count = (cur >> 24) & 0x7Fu;
        // leaves hold one or two triangles (frt_bvh.cpp; up to four under FRT_BVH_LEAF): the first two are tested in line, without a loop
        auto test = [&](uint32_t slot) -> bool {
            const float4* tp = sc.tris + (size_t)slot * 3u;
            float4 a = tp[0], b = tp[1], c = tp[2];
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
        if (test(first)) return;
        if (count > 1u && test(first + 1u)) return;
        for (uint32_t kk = 2u; kk < count; ++kk) if (test(first + kk)) return;
        if (top == stk) break;
        top -= stride; cur = *top;
        if (!(cur & 0x80000000u)) FRT_FETCH(cur) else FRT_NO_NODE
    }
#undef FRT_FETCH
#undef FRT_NO_NODE
    if (!ANY && hit.tri != 0xFFFFFFFFu) {
        bool front = best_det > 0.0f;
        if (sc.instances[hit.inst].flip) front = !front;
        hit.front = front;
    }
}


template <bool ANY>
__device__ __forceinline__ void trace4_xload(const SceneView& sc, f3 o, f3 d, float tmin, float tmax, uint32_t* stk, uint32_t stride, HitRec& hit) {
    hit.t = tmax; hit.tri = 0xFFFFFFFFu; hit.u = 0.0f; hit.v = 0.0f; hit.inst = 0u; hit.front = false;
    float best_det = 0.0f;
    tb_v4f dummy = {0, 0, 0, 0};
    f3 inv = mk3(prune_rcp(d.x), prune_rcp(d.y), prune_rcp(d.z));
    f3 oinv = mk3(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z);
    const uint32_t sx = (f2u(d.x) >> 31) << 4, sy = 32u | ((f2u(d.y) >> 31) << 4), sz = 64u | ((f2u(d.z) >> 31) << 4);
    const uint32_t kDone = 0xFFFFFFFFu;
    const float kFar = 3.0e38f;
    uint32_t* top = stk;      // next free stack entry (entries are `stride` words apart)
    const char* nb = reinterpret_cast<const char*>(sc.nodes4);
    tb_v4f nx, fx, ny, fy, nz, fz, rf;     // the node in flight: near / far planes per axis (slab4's layout), child references
#define FRT_FETCH(ref) { const uint32_t n_ = (ref) << 7; \
        nx = *reinterpret_cast<const tb_v4f*>(nb + (n_ | sx)); fx = *reinterpret_cast<const tb_v4f*>(nb + ((n_ | sx) ^ 16u)); \
        ny = *reinterpret_cast<const tb_v4f*>(nb + (n_ | sy)); fy = *reinterpret_cast<const tb_v4f*>(nb + ((n_ | sy) ^ 16u)); \
        nz = *reinterpret_cast<const tb_v4f*>(nb + (n_ | sz)); fz = *reinterpret_cast<const tb_v4f*>(nb + ((n_ | sz) ^ 16u)); \
        rf = *reinterpret_cast<const tb_v4f*>(nb + (n_ + 96u)); }
    // a lane that holds a leaf (or is done) has no node in flight: its seven registers hold nothing (an unspecified value keeps the old node's
    // data from staying live across the leaf phase)
#if defined(__HIP_DEVICE_COMPILE__)
#define FRT_NO_NODE { nx = __builtin_nondeterministic_value(nx); fx = __builtin_nondeterministic_value(fx); ny = __builtin_nondeterministic_value(ny); \
        fy = __builtin_nondeterministic_value(fy); nz = __builtin_nondeterministic_value(nz); fz = __builtin_nondeterministic_value(fz); rf = __builtin_nondeterministic_value(rf); }
#else
#define FRT_NO_NODE { nx = fx = ny = fy = nz = fz = rf = tb_v4f{0.0f, 0.0f, 0.0f, 0.0f}; }
#endif
    uint32_t cur = 0u;   // quad node 0 is the root
    FRT_FETCH(0u)
    for (;;) {
        while (!(cur & 0x80000000u)) {
            const float tlim = ANY ? tmax : hit.t;
            { const uint32_t j_ = (cur & 7u) << 7; dummy += *reinterpret_cast<const tb_v4f*>(nb + j_) + *reinterpret_cast<const tb_v4f*>(nb + j_ + 16u) + *reinterpret_cast<const tb_v4f*>(nb + j_ + 32u) + *reinterpret_cast<const tb_v4f*>(nb + j_ + 48u); }
            const float xn[4] = {__builtin_fmaf(nx.x, inv.x, oinv.x), __builtin_fmaf(nx.y, inv.x, oinv.x), __builtin_fmaf(nx.z, inv.x, oinv.x), __builtin_fmaf(nx.w, inv.x, oinv.x)};
            const float xf[4] = {__builtin_fmaf(fx.x, inv.x, oinv.x), __builtin_fmaf(fx.y, inv.x, oinv.x), __builtin_fmaf(fx.z, inv.x, oinv.x), __builtin_fmaf(fx.w, inv.x, oinv.x)};
            const float yn[4] = {__builtin_fmaf(ny.x, inv.y, oinv.y), __builtin_fmaf(ny.y, inv.y, oinv.y), __builtin_fmaf(ny.z, inv.y, oinv.y), __builtin_fmaf(ny.w, inv.y, oinv.y)};
            const float yf[4] = {__builtin_fmaf(fy.x, inv.y, oinv.y), __builtin_fmaf(fy.y, inv.y, oinv.y), __builtin_fmaf(fy.z, inv.y, oinv.y), __builtin_fmaf(fy.w, inv.y, oinv.y)};
            const float zn[4] = {__builtin_fmaf(nz.x, inv.z, oinv.z), __builtin_fmaf(nz.y, inv.z, oinv.z), __builtin_fmaf(nz.z, inv.z, oinv.z), __builtin_fmaf(nz.w, inv.z, oinv.z)};
            const float zf[4] = {__builtin_fmaf(fz.x, inv.z, oinv.z), __builtin_fmaf(fz.y, inv.z, oinv.z), __builtin_fmaf(fz.z, inv.z, oinv.z), __builtin_fmaf(fz.w, inv.z, oinv.z)};
            uint32_t r[4] = {f2u(rf.x), f2u(rf.y), f2u(rf.z), f2u(rf.w)};
            // (an empty slot holds a far-away degenerate box: it never passes the slab test, no reference check needed)
            float k[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float tn = fmaxn(fmaxn(xn[c], yn[c]), fmaxn(zn[c], tmin));
                const float tf = fminn(fminn(xf[c], yf[c]), fminn(zf[c], tlim));
                k[c] = tn <= tf ? tn : kFar;      // no slack factor: see slab4
            }
#define FRT_CE(a, b) { const bool s_ = k[b] < k[a]; const float ka_ = s_ ? k[b] : k[a], kb_ = s_ ? k[a] : k[b]; \
                       const uint32_t ra_ = s_ ? r[b] : r[a], rb_ = s_ ? r[a] : r[b]; k[a] = ka_; k[b] = kb_; r[a] = ra_; r[b] = rb_; }
            FRT_CE(0, 1) FRT_CE(2, 3) FRT_CE(0, 2)      // k[0] / r[0]: the nearest hit child, if any child is hit at all
            uint32_t next;
            if (k[0] < kFar) next = r[0];
            else if (top == stk) next = kDone;      // (nothing is hit: nothing will be pushed below either)
            else { top -= stride; next = *top; }
            if (!(next & 0x80000000u)) FRT_FETCH(next) else FRT_NO_NODE
            FRT_CE(1, 3) FRT_CE(1, 2)
#undef FRT_CE
            if (k[3] < kFar) { *top = r[3]; top += stride; }
            if (k[2] < kFar) { *top = r[2]; top += stride; }
            if (k[1] < kFar) { *top = r[1]; top += stride; }
            cur = next;
        }
        if (cur == kDone) break;
        uint32_t first = cur & 0x00FFFFFFu, count = (cur >> 24) & 0x7Fu;
        // leaves hold one or two triangles (frt_bvh.cpp; up to four under FRT_BVH_LEAF): the first two are tested in line, without a loop
        auto test = [&](uint32_t slot) -> bool {
            const float4* tp = sc.tris + (size_t)slot * 3u;
            float4 a = tp[0], b = tp[1], c = tp[2];
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
        if (test(first)) return;
        if (count > 1u && test(first + 1u)) return;
        for (uint32_t kk = 2u; kk < count; ++kk) if (test(first + kk)) return;
        if (top == stk) break;
        top -= stride; cur = *top;
        if (!(cur & 0x80000000u)) FRT_FETCH(cur) else FRT_NO_NODE
    }
#undef FRT_FETCH
#undef FRT_NO_NODE
    if (dummy.x + dummy.y + dummy.z + dummy.w == 1.2345e-33f) hit.u += 1.0f;
    if (!ANY && hit.tri != 0xFFFFFFFFu) {
        bool front = best_det > 0.0f;
        if (sc.instances[hit.inst].flip) front = !front;
        hit.front = front;
    }
}



// ---- two rays of one lane in one loop -------------------------------------------------------------------------------------------------
// A = any-hit (shadow) ray, B = closest-hit ray. Every trip of the node loop steps BOTH rays of the lane (a ray that is not at an inner node
// loads the root, harmlessly, and its results are dropped), so that the two dependency chains interleave in one instruction stream: the
// loads of both nodes are in flight together and a lone wave issues from two independent chains. The leaf phase tests one triangle of
// each ray per trip the same way. The stack column is shared: A's entries grow up from the bottom, B's down from the top.
// Möller–Trumbore without early-outs (same operations in the same order as intersect_tri; the comparisons reject NaN).
__device__ __forceinline__ bool tri_eval(f3 v0, f3 e1, f3 e2, f3 o, f3 d, float tmin, float tmax, float& t, float& u, float& v, float& det_out) {
    f3 p = cross(d, e2);
    float det = dot(e1, p);
    float inv = 1.0f / det;
    f3 s = o - v0;
    float uu = dot(s, p) * inv;
    f3 q = cross(s, e1);
    float vv = dot(d, q) * inv;
    float tt = dot(e2, q) * inv;
    t = tt; u = uu; v = vv; det_out = det;
    return (det != 0.0f) & (uu >= 0.0f) & (uu <= 1.0f) & (vv >= 0.0f) & (uu + vv <= 1.0f) & (tt > tmin) & (tt < tmax);
}

struct RaySetup { f3 inv, oinv; uint32_t sx, sy, sz; };
__device__ __forceinline__ RaySetup ray_setup(f3 o, f3 d) {
    RaySetup s;
    s.inv = mk3(prune_rcp(d.x), prune_rcp(d.y), prune_rcp(d.z));
    s.oinv = mk3(-o.x * s.inv.x, -o.y * s.inv.y, -o.z * s.inv.z);
    s.sx = (f2u(d.x) >> 31) << 4; s.sy = 32u | ((f2u(d.y) >> 31) << 4); s.sz = 64u | ((f2u(d.z) >> 31) << 4);
    return s;
}
struct NodeRegs { tb_v4f nx, fx, ny, fy, nz, fz, rf; };
__device__ __forceinline__ NodeRegs node_fetch(const char* nb, uint32_t noff, const RaySetup& s) {
    NodeRegs n;
    n.nx = *reinterpret_cast<const tb_v4f*>(nb + (noff | s.sx)); n.fx = *reinterpret_cast<const tb_v4f*>(nb + ((noff | s.sx) ^ 16u));
    n.ny = *reinterpret_cast<const tb_v4f*>(nb + (noff | s.sy)); n.fy = *reinterpret_cast<const tb_v4f*>(nb + ((noff | s.sy) ^ 16u));
    n.nz = *reinterpret_cast<const tb_v4f*>(nb + (noff | s.sz)); n.fz = *reinterpret_cast<const tb_v4f*>(nb + ((noff | s.sz) ^ 16u));
    n.rf = *reinterpret_cast<const tb_v4f*>(nb + (noff + 96u));
    return n;
}
// slab tests of the four children + the sorting network: k[0..3] ascending entry distances (kFar = not hit), r[] the references in that order
__device__ __forceinline__ void node_eval(const NodeRegs& n, const RaySetup& s, float tmin, float tlim, float k[4], uint32_t r[4]) {
    const float kFar = 3.0e38f;
    const float xn[4] = {__builtin_fmaf(n.nx.x, s.inv.x, s.oinv.x), __builtin_fmaf(n.nx.y, s.inv.x, s.oinv.x), __builtin_fmaf(n.nx.z, s.inv.x, s.oinv.x), __builtin_fmaf(n.nx.w, s.inv.x, s.oinv.x)};
    const float xf[4] = {__builtin_fmaf(n.fx.x, s.inv.x, s.oinv.x), __builtin_fmaf(n.fx.y, s.inv.x, s.oinv.x), __builtin_fmaf(n.fx.z, s.inv.x, s.oinv.x), __builtin_fmaf(n.fx.w, s.inv.x, s.oinv.x)};
    const float yn[4] = {__builtin_fmaf(n.ny.x, s.inv.y, s.oinv.y), __builtin_fmaf(n.ny.y, s.inv.y, s.oinv.y), __builtin_fmaf(n.ny.z, s.inv.y, s.oinv.y), __builtin_fmaf(n.ny.w, s.inv.y, s.oinv.y)};
    const float yf[4] = {__builtin_fmaf(n.fy.x, s.inv.y, s.oinv.y), __builtin_fmaf(n.fy.y, s.inv.y, s.oinv.y), __builtin_fmaf(n.fy.z, s.inv.y, s.oinv.y), __builtin_fmaf(n.fy.w, s.inv.y, s.oinv.y)};
    const float zn[4] = {__builtin_fmaf(n.nz.x, s.inv.z, s.oinv.z), __builtin_fmaf(n.nz.y, s.inv.z, s.oinv.z), __builtin_fmaf(n.nz.z, s.inv.z, s.oinv.z), __builtin_fmaf(n.nz.w, s.inv.z, s.oinv.z)};
    const float zf[4] = {__builtin_fmaf(n.fz.x, s.inv.z, s.oinv.z), __builtin_fmaf(n.fz.y, s.inv.z, s.oinv.z), __builtin_fmaf(n.fz.z, s.inv.z, s.oinv.z), __builtin_fmaf(n.fz.w, s.inv.z, s.oinv.z)};
    r[0] = f2u(n.rf.x); r[1] = f2u(n.rf.y); r[2] = f2u(n.rf.z); r[3] = f2u(n.rf.w);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float tn = fmaxn(fmaxn(xn[c], yn[c]), fmaxn(zn[c], tmin));
        const float tf = fminn(fminn(xf[c], yf[c]), fminn(zf[c], tlim));
        k[c] = tn <= tf ? tn : kFar;
    }
#define TB_CE(a, b) { const bool s_ = k[b] < k[a]; const float ka_ = s_ ? k[b] : k[a], kb_ = s_ ? k[a] : k[b]; \
                      const uint32_t ra_ = s_ ? r[b] : r[a], rb_ = s_ ? r[a] : r[b]; k[a] = ka_; k[b] = kb_; r[a] = ra_; r[b] = rb_; }
    TB_CE(0, 1) TB_CE(2, 3) TB_CE(0, 2) TB_CE(1, 3) TB_CE(1, 2)
#undef TB_CE
}

template <int V>
__device__ __forceinline__ void trace_pair(const SceneView& sc, bool haveA, f3 oA, f3 dA, float tminA, float tmaxA, bool haveB, f3 oB, f3 dB, float tminB, float tmaxB,
                                           uint32_t* stk, uint32_t stride, bool& occluded, HitRec& hit) {
    hit.t = tmaxB; hit.tri = 0xFFFFFFFFu; hit.u = 0.0f; hit.v = 0.0f; hit.inst = 0u; hit.front = false;
    float best_det = 0.0f;
    occluded = false;
    const RaySetup sA = ray_setup(oA, dA), sB = ray_setup(oB, dB);
    const uint32_t kDone = 0xFFFFFFFFu;
    const float kFar = 3.0e38f;
    uint32_t* topA = stk;                                           // A: next free entry, ascending
    uint32_t* const botB = stk + (uint32_t)(kStackDepth - 1) * stride;
    uint32_t* topB = botB;                                          // B: next free entry, descending
    uint32_t curA = haveA ? 0u : kDone, curB = haveB ? 0u : kDone;
    const char* nb = reinterpret_cast<const char*>(sc.nodes4);
    for (;;) {
        for (;;) {
            const bool nA = !(curA & 0x80000000u), nB = !(curB & 0x80000000u);
            if (!(nA | nB)) break;
            if (V == 3) {
                // wave-uniform shortcuts: one kind has no lane at an inner node -> step the other kind alone
                if (__ballot(nA) == 0ull) {
                    const NodeRegs n = node_fetch(nb, curB << 7, sB);
                    float k[4]; uint32_t r[4];
                    node_eval(n, sB, tminB, hit.t, k, r);
                    if (k[3] < kFar) { *topB = r[3]; topB -= stride; }
                    if (k[2] < kFar) { *topB = r[2]; topB -= stride; }
                    if (k[1] < kFar) { *topB = r[1]; topB -= stride; }
                    if (k[0] < kFar) curB = r[0];
                    else if (topB == botB) curB = kDone;
                    else { topB += stride; curB = *topB; }
                    continue;
                }
                if (__ballot(nB) == 0ull) {
                    const NodeRegs n = node_fetch(nb, curA << 7, sA);
                    float k[4]; uint32_t r[4];
                    node_eval(n, sA, tminA, tmaxA, k, r);
                    if (k[3] < kFar) { *topA = r[3]; topA += stride; }
                    if (k[2] < kFar) { *topA = r[2]; topA += stride; }
                    if (k[1] < kFar) { *topA = r[1]; topA += stride; }
                    if (k[0] < kFar) curA = r[0];
                    else if (topA == stk) curA = kDone;
                    else { topA -= stride; curA = *topA; }
                    continue;
                }
            }
            const NodeRegs na = node_fetch(nb, nA ? curA << 7 : 0u, sA);
            const NodeRegs nbq = node_fetch(nb, nB ? curB << 7 : 0u, sB);
            float ka[4], kb[4]; uint32_t ra[4], rb[4];
            node_eval(na, sA, tminA, tmaxA, ka, ra);
            node_eval(nbq, sB, tminB, hit.t, kb, rb);
            if (nA) {
                if (ka[3] < kFar) { *topA = ra[3]; topA += stride; }
                if (ka[2] < kFar) { *topA = ra[2]; topA += stride; }
                if (ka[1] < kFar) { *topA = ra[1]; topA += stride; }
                if (ka[0] < kFar) curA = ra[0];
                else if (topA == stk) curA = kDone;
                else { topA -= stride; curA = *topA; }
            }
            if (nB) {
                if (kb[3] < kFar) { *topB = rb[3]; topB -= stride; }
                if (kb[2] < kFar) { *topB = rb[2]; topB -= stride; }
                if (kb[1] < kFar) { *topB = rb[1]; topB -= stride; }
                if (kb[0] < kFar) curB = rb[0];
                else if (topB == botB) curB = kDone;
                else { topB += stride; curB = *topB; }
            }
        }
        if (curA == kDone && curB == kDone) break;
        // leaf phase: one triangle of each ray per trip
        const bool lA = curA != kDone, lB = curB != kDone;
        const uint32_t firstA = curA & 0x00FFFFFFu, countA = lA ? (curA >> 24) & 0x7Fu : 0u;
        const uint32_t firstB = curB & 0x00FFFFFFu, countB = lB ? (curB >> 24) & 0x7Fu : 0u;
        for (uint32_t kk = 0u;; ++kk) {
            const bool tA = kk < countA && !occluded, tB = kk < countB;
            if (!(tA | tB)) break;
            const float4* pa = sc.tris + (size_t)(tA ? firstA + kk : 0u) * 3u;
            const float4* pb = sc.tris + (size_t)(tB ? firstB + kk : 0u) * 3u;
            const float4 a0 = pa[0], a1 = pa[1], a2 = pa[2], b0 = pb[0], b1 = pb[1], b2 = pb[2];
            float ta, ua, va, da, tb, ub, vb, db;
            const bool ha = tri_eval(mk3(a0.x, a0.y, a0.z), mk3(a1.x, a1.y, a1.z), mk3(a2.x, a2.y, a2.z), oA, dA, tminA, tmaxA, ta, ua, va, da);
            const bool hb = tri_eval(mk3(b0.x, b0.y, b0.z), mk3(b1.x, b1.y, b1.z), mk3(b2.x, b2.y, b2.z), oB, dB, tminB, tmaxB, tb, ub, vb, db);
            if (tA && ha) occluded = true;
            if (tB && hb) {
                const uint32_t id = f2u(b0.w);
                if (tb < hit.t || (tb == hit.t && id < hit.tri)) { hit.t = tb; hit.u = ub; hit.v = vb; hit.tri = id; hit.inst = f2u(b1.w); best_det = db; }
            }
        }
        if (lA) {
            if (occluded || topA == stk) { curA = kDone; topA = stk; }
            else { topA -= stride; curA = *topA; }
        }
        if (lB) {
            if (topB == botB) curB = kDone;
            else { topB += stride; curB = *topB; }
        }
    }
    if (hit.tri != 0xFFFFFFFFu) {
        bool front = best_det > 0.0f;
        if (sc.instances[hit.inst].flip) front = !front;
        hit.front = front;
    }
}


// ---- variant 5: quad nodes on the 16-bit grid: 64 bytes = FOUR loads per step instead of seven ---------------------------------------------
// chunk a (a = x, y, z): lo[4] as u16 x 4 | hi[4] as u16 x 4 of the four children; chunk 3: the four references. Boxes rounded outward on the
// host (tools/trace_bench.hip: build_qquad), so the tree prunes a little less than the float tree and never more: same hits.
struct QQuadView { const uint4* nodes; f3 qmin, qstep; };
template <bool ANY>
__device__ __forceinline__ void trace4_q16(const SceneView& sc, const QQuadView& qv, f3 o, f3 d, float tmin, float tmax, uint32_t* stk, uint32_t stride, HitRec& hit) {
    hit.t = tmax; hit.tri = 0xFFFFFFFFu; hit.u = 0.0f; hit.v = 0.0f; hit.inst = 0u; hit.front = false;
    float best_det = 0.0f;
    f3 inv = mk3(prune_rcp(d.x), prune_rcp(d.y), prune_rcp(d.z));
    const f3 oinv = mk3((qv.qmin.x - o.x) * inv.x, (qv.qmin.y - o.y) * inv.y, (qv.qmin.z - o.z) * inv.z);
    inv = mk3(inv.x * qv.qstep.x, inv.y * qv.qstep.y, inv.z * qv.qstep.z);
    const bool negx = d.x < 0.0f, negy = d.y < 0.0f, negz = d.z < 0.0f;
    const uint32_t kDone = 0xFFFFFFFFu;
    const float kFar = 3.0e38f;
    uint32_t* top = stk;
    uint32_t cur = 0u;
    for (;;) {
        while (!(cur & 0x80000000u)) {
            const uint4* n = qv.nodes + (size_t)cur * 4u;
            const uint4 qx = n[0], qy = n[1], qz = n[2], qr = n[3];
            const float tlim = ANY ? tmax : hit.t;
            // near / far plane words by the sign of the direction (two children per word)
            const uint32_t nx0 = negx ? qx.z : qx.x, nx1 = negx ? qx.w : qx.y, fx0 = negx ? qx.x : qx.z, fx1 = negx ? qx.y : qx.w;
            const uint32_t ny0 = negy ? qy.z : qy.x, ny1 = negy ? qy.w : qy.y, fy0 = negy ? qy.x : qy.z, fy1 = negy ? qy.y : qy.w;
            const uint32_t nz0 = negz ? qz.z : qz.x, nz1 = negz ? qz.w : qz.y, fz0 = negz ? qz.x : qz.z, fz1 = negz ? qz.y : qz.w;
            uint32_t r[4] = {qr.x, qr.y, qr.z, qr.w};
            float k[4];
#define TBQ(w0, w1, c) ((float)((c) < 2 ? ((c) == 0 ? (w0) & 0xFFFFu : (w0) >> 16) : ((c) == 2 ? (w1) & 0xFFFFu : (w1) >> 16)))
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float xn = __builtin_fmaf(TBQ(nx0, nx1, c), inv.x, oinv.x), xf = __builtin_fmaf(TBQ(fx0, fx1, c), inv.x, oinv.x);
                const float yn = __builtin_fmaf(TBQ(ny0, ny1, c), inv.y, oinv.y), yf = __builtin_fmaf(TBQ(fy0, fy1, c), inv.y, oinv.y);
                const float zn = __builtin_fmaf(TBQ(nz0, nz1, c), inv.z, oinv.z), zf = __builtin_fmaf(TBQ(fz0, fz1, c), inv.z, oinv.z);
                const float tn = fmaxn(fmaxn(xn, yn), fmaxn(zn, tmin));
                const float tf = fminn(fminn(xf, yf), fminn(zf, tlim));
                k[c] = tn <= tf ? tn : kFar;
            }
#undef TBQ
#define TB_CE(a, b) { const bool s_ = k[b] < k[a]; const float ka_ = s_ ? k[b] : k[a], kb_ = s_ ? k[a] : k[b]; \
                      const uint32_t ra_ = s_ ? r[b] : r[a], rb_ = s_ ? r[a] : r[b]; k[a] = ka_; k[b] = kb_; r[a] = ra_; r[b] = rb_; }
            TB_CE(0, 1) TB_CE(2, 3) TB_CE(0, 2) TB_CE(1, 3) TB_CE(1, 2)
#undef TB_CE
            if (k[3] < kFar) { *top = r[3]; top += stride; }
            if (k[2] < kFar) { *top = r[2]; top += stride; }
            if (k[1] < kFar) { *top = r[1]; top += stride; }
            if (k[0] < kFar) cur = r[0];
            else if (top == stk) cur = kDone;
            else { top -= stride; cur = *top; }
        }
        if (cur == kDone) break;
        uint32_t first = cur & 0x00FFFFFFu, count = (cur >> 24) & 0x7Fu;
        auto test = [&](uint32_t slot) -> bool {
            const float4* tp = sc.tris + (size_t)slot * 3u;
            float4 a = tp[0], b = tp[1], c = tp[2];
            float t, u, v, det;
            if (intersect_tri(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), o, d, tmin, tmax, t, u, v, det)) {
                uint32_t id = f2u(a.w);
                if (ANY) { hit.tri = id; hit.t = t; return true; }
                if (t < hit.t || (t == hit.t && id < hit.tri)) { hit.t = t; hit.u = u; hit.v = v; hit.tri = id; hit.inst = f2u(b.w); best_det = det; }
            }
            return false;
        };
        if (test(first)) return;
        if (count > 1u && test(first + 1u)) return;
        for (uint32_t kk = 2u; kk < count; ++kk) if (test(first + kk)) return;
        if (top == stk) break;
        top -= stride; cur = *top;
    }
    if (!ANY && hit.tri != 0xFFFFFFFFu) {
        bool front = best_det > 0.0f;
        if (sc.instances[hit.inst].flip) front = !front;
        hit.front = front;
    }
}


// ---- variant 6: a lane that reaches a leaf while the rest of its wave still walks nodes POSTPONES the leaf and keeps walking --------------
// In trace4 a lane that holds a leaf waits until every lane of the wave holds one (the node loop runs at ~50 % lane efficiency: 9.4 wave-level
// node steps for ~5 per lane). Here it puts the leaf aside (one register), pops its next node and stays in the node loop; the leaf phase then
// tests the postponed leaf and the current one. Closest-hit rays lose a little pruning (the postponed leaf's hit would have shortened the
// interval earlier), any-hit rays may walk past their occluder: same hits either way (hit semantics), more work per lane, fewer idle lanes.
template <bool ANY>
__device__ __forceinline__ void trace4_postpone(const SceneView& sc, f3 o, f3 d, float tmin, float tmax, uint32_t* stk, uint32_t stride, HitRec& hit) {
    hit.t = tmax; hit.tri = 0xFFFFFFFFu; hit.u = 0.0f; hit.v = 0.0f; hit.inst = 0u; hit.front = false;
    float best_det = 0.0f;
    const RaySetup rs = ray_setup(o, d);
    const uint32_t kDone = 0xFFFFFFFFu;
    const float kFar = 3.0e38f;
    uint32_t* top = stk;
    const char* nb = reinterpret_cast<const char*>(sc.nodes4);
    uint32_t cur = 0u, pend = kDone;      // pend: a postponed leaf, or kDone
    bool done = false;
    auto test_leaf = [&](uint32_t leaf) -> bool {      // true: an any-hit ray found its occluder
        const uint32_t first = leaf & 0x00FFFFFFu, count = (leaf >> 24) & 0x7Fu;
        for (uint32_t kk = 0u; kk < count; ++kk) {
            const float4* tp = sc.tris + (size_t)(first + kk) * 3u;
            const float4 a = tp[0], b = tp[1], c = tp[2];
            float t, u, v, det;
            if (intersect_tri(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), o, d, tmin, tmax, t, u, v, det)) {
                const uint32_t id = f2u(a.w);
                if (ANY) { hit.tri = id; hit.t = t; return true; }
                if (t < hit.t || (t == hit.t && id < hit.tri)) { hit.t = t; hit.u = u; hit.v = v; hit.tri = id; hit.inst = f2u(b.w); best_det = det; }
            }
        }
        return false;
    };
    for (;;) {
        // node loop: while any lane of the wave is at an inner node
        for (;;) {
            const bool at_node = !done && !(cur & 0x80000000u);
            if (__ballot(at_node) == 0ull) break;
            if (at_node) {
                const NodeRegs n = node_fetch(nb, cur << 7, rs);
                float k[4]; uint32_t r[4];
                node_eval(n, rs, tmin, ANY ? tmax : hit.t, k, r);
                if (k[3] < kFar) { *top = r[3]; top += stride; }
                if (k[2] < kFar) { *top = r[2]; top += stride; }
                if (k[1] < kFar) { *top = r[1]; top += stride; }
                if (k[0] < kFar) cur = r[0];
                else if (top == stk) cur = kDone;
                else { top -= stride; cur = *top; }
            } else if (!done && cur != kDone && pend == kDone && top != stk) {
                pend = cur;                      // postpone this leaf, keep walking
                top -= stride; cur = *top;
            }
        }
        if (!done) {
            if (pend != kDone) { if (test_leaf(pend)) done = true; pend = kDone; }
            if (!done && cur != kDone) { if (test_leaf(cur)) done = true; }
            if (!done) {
                if (top == stk) done = true;
                else { top -= stride; cur = *top; }
            }
        }
        if (__ballot(!done) == 0ull) break;
    }
    if (!ANY && hit.tri != 0xFFFFFFFFu) {
        bool front = best_det > 0.0f;
        if (sc.instances[hit.inst].flip) front = !front;
        hit.front = front;
    }
}

// ---- variant 7 / 8: the wave VOTES for its next step --------------------------------------------------------------------------------------
// trace4 is "while-while": node steps until no lane of the wave holds a node, then one leaf step. A lane's own sequence of node and leaf steps
// does not depend on how the wave interleaves them, so the wave's schedule is a common supersequence of its lanes' sequences, and while-while
// is only one way to build it: tools/bvh_quality.cpp (host model of the lockstep walk) finds 9.9 node + 3.3 leaf steps per wave-ray for
// incoherent rays where the slowest lane needs 7.0 node steps. Here every trip of ONE loop takes the step that more lanes wait for (weights WN :
// WL), which the model puts at 8.0 + 4.0.
template <bool ANY, int WN, int WL>
__device__ __forceinline__ void trace4_vote(const SceneView& sc, f3 o, f3 d, float tmin, float tmax, uint32_t* stk, uint32_t stride, HitRec& hit) {
    hit.t = tmax; hit.tri = 0xFFFFFFFFu; hit.u = 0.0f; hit.v = 0.0f; hit.inst = 0u; hit.front = false;
    float best_det = 0.0f;
    const RaySetup rs = ray_setup(o, d);
    const uint32_t kDone = 0xFFFFFFFFu;
    const float kFar = 3.0e38f;
    uint32_t* top = stk;
    const char* nb = reinterpret_cast<const char*>(sc.nodes4);
    uint32_t cur = 0u;
    auto test = [&](uint32_t slot) -> bool {
        const float4* tp = sc.tris + (size_t)slot * 3u;
        const float4 a = tp[0], b = tp[1], c = tp[2];
        float t, u, v, det;
        if (intersect_tri(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), o, d, tmin, tmax, t, u, v, det)) {
            const uint32_t id = f2u(a.w);
            if (ANY) { hit.tri = id; hit.t = t; return true; }
            if (t < hit.t || (t == hit.t && id < hit.tri)) { hit.t = t; hit.u = u; hit.v = v; hit.tri = id; hit.inst = f2u(b.w); best_det = det; }
        }
        return false;
    };
    for (;;) {
        const bool at_node = !(cur & 0x80000000u);
        const bool at_leaf = !at_node && cur != kDone;
        const unsigned long long mn = __ballot(at_node), ml = __ballot(at_leaf);
        if ((mn | ml) == 0ull) break;
        const bool do_node = ml == 0ull || (mn != 0ull && __popcll(mn) * WN >= __popcll(ml) * WL);      // wave-uniform
        if (do_node) {
            if (at_node) {
                const NodeRegs n = node_fetch(nb, cur << 7, rs);
                float k[4]; uint32_t r[4];
                node_eval(n, rs, tmin, ANY ? tmax : hit.t, k, r);
                if (k[3] < kFar) { *top = r[3]; top += stride; }
                if (k[2] < kFar) { *top = r[2]; top += stride; }
                if (k[1] < kFar) { *top = r[1]; top += stride; }
                if (k[0] < kFar) cur = r[0];
                else if (top == stk) cur = kDone;
                else { top -= stride; cur = *top; }
            }
        } else if (at_leaf) {
            const uint32_t first = cur & 0x00FFFFFFu, count = (cur >> 24) & 0x7Fu;
            bool found = test(first);
            if (!found && count > 1u) found = test(first + 1u);
            for (uint32_t kk = 2u; !found && kk < count; ++kk) found = test(first + kk);
            if (found || top == stk) cur = kDone;
            else { top -= stride; cur = *top; }
        }
    }
    if (!ANY && hit.tri != 0xFFFFFFFFu) {
        bool front = best_det > 0.0f;
        if (sc.instances[hit.inst].flip) front = !front;
        hit.front = front;
    }
}
