// TEST INFRASTRUCTURE ONLY — not linked into libfrt.so and never used to produce a result the product returns.
// Host instantiation of the product's __host__ __device__ stage functions (csrc/frt_shade.hpp, frt_trace.hpp) so that the
// CPU test suite (-m "not gpu") can check the kernel bodies against the oracle before they ever run on an MI355X.
// The kernels themselves (LDS stack columns, wave tiling, ray-counter reduction) are only exercised by the -m gpu tests.
#include "../../fast-raytracing-wgpu_amd/csrc/frt_scene.hpp"
#include "../../fast-raytracing-wgpu_amd/csrc/frt_mono.hpp"
#include <vector>
#include <cstring>
#include <thread>

using namespace frt;

namespace {
struct HostCheck {
    const SceneBuilder* b;
    std::vector<uint8_t> color_tex, data_tex;
    SceneView sv{};
    uint32_t W, H, max_depth, frame_count = 0;
    std::vector<float4> gpos[2], gnormal[2], accum[2];
    std::vector<uint32_t> galbedo[2], display;
    std::vector<float2> gmotion;
    std::vector<float4> cand;
    float jitter[2] = {0.0f, 0.0f};
    std::vector<ReservoirView> res[2];
    std::vector<uint2> raw;
    unsigned long long rays[2] = {0, 0};
    int nthreads = 8;
};
}

// One stage through the cut / park / resume protocol of the continuation kernels, sequentially: every pixel runs head + bounces
// [1, cut), survivors are parked in a host-side ContQueue, then each parked path is resumed for [cut, cut2) and [cut2, max).
// split (stage 1 only): the form the default kernels run — T-trace ends with the candidate record, a separate T-merge pass follows.
template <int STAGE>
static void finish_on_host(PathCtx& c, const FrameView& fv, uint32_t pix, const ReservoirView& r, const LoopState& s, bool split) {
    if (STAGE == 1) {
        if (split) fv.cand[pix] = temporal_candidate(s.accumulated, s.v1_pos);
        else { PathState st; make_path_state(st, pix, s.accumulated, s.v1_pos); temporal_finalize(c, st); }
    } else spatial_tail(c, pix, r, s.accumulated, s.v1_pos);
}
template <int STAGE>
static void run_stage_cut(HostCheck* h, FrameView& fv, uint32_t cut, unsigned long long rc[2], bool split, bool pulled_apart = false) {
    const uint32_t npix = h->W * h->H;
    std::vector<uint32_t> wa((size_t)kContWordsSpatial * npix), wb((size_t)kContWordsSpatial * npix);
    uint32_t ca = 0, cb = 0;
    ContQueue qa{wa.data(), &ca, npix, nullptr, 1u}, qb{wb.data(), &cb, npix, nullptr, 1u};
    uint32_t stack[kStackDepth];
    constexpr int V = STAGE == 1 ? 0 : 1;
    for (uint32_t pix = 0; pix < npix; ++pix) {
        PathCtx c(h->sv, fv, stack, 1u);
        ReservoirView r = zero_reservoir();
        uint32_t seed;
        if (STAGE == 1) {
            if (fv.gpos[pix].w < 0.0f) { if (!split) fv.res_temporal[pix] = zero_reservoir(); continue; }
            seed = temporal_seed(fv, pix);
        } else {
            if (!spatial_neighbors(c, pix, r)) { rc[1] += c.n_any; continue; }
            seed = r.y;
        }
        LoopState s;
        if (pulled_apart) {      // the form the collective kernels run (frt_kernels.hip: pixel_kernel_wg): the head's shadow ray handed back, then pulled-apart bounces
            ShadowReq req;
            path_head<V>(c, pix, seed, s, &req);
            bool lit = req.add_now;
            if (req.want) lit = !c.any(req.o, req.d, req.tmin, req.tmax);
            s.accumulated = s.accumulated + (lit ? req.contrib : req.dark);
            if (s.alive) path_loop_split<V>(c, s, 1u, cut < fv.max_depth ? cut : fv.max_depth);
        } else {
            path_head<V>(c, pix, seed, s);
            if (s.alive) path_loop<V>(c, s, 1u, cut < fv.max_depth ? cut : fv.max_depth);
        }
        rc[0] += c.n_closest; rc[1] += c.n_any;
        if (s.alive) cont_store(qa, ca++, pix, c.rng, true, s, STAGE == 2 ? &r : nullptr);
        else finish_on_host<STAGE>(c, fv, pix, r, s, split);
    }
    uint32_t d0 = cut;
    ContQueue* qin = &qa; ContQueue* qout = &qb;
    while (*qin->count > 0) {
        uint32_t d1 = d0 + 2u < fv.max_depth ? d0 + 2u : fv.max_depth;
        *qout->count = 0;
        for (uint32_t slot = 0; slot < *qin->count; ++slot) {
            PathCtx c(h->sv, fv, stack, 1u);
            LoopState s; ReservoirView r = zero_reservoir(); uint32_t pix; bool owned;
            cont_load(*qin, slot, pix, c.rng, owned, s, STAGE == 2 ? &r : nullptr);
            if (pulled_apart) path_loop_split<V>(c, s, d0, d1); else path_loop<V>(c, s, d0, d1);
            rc[0] += c.n_closest; rc[1] += c.n_any;
            if (s.alive) { cont_store(*qout, (*qout->count)++, pix, c.rng, owned, s, STAGE == 2 ? &r : nullptr); }
            else finish_on_host<STAGE>(c, fv, pix, r, s, split);
        }
        std::swap(qin, qout);
        d0 = d1;
    }
    if (STAGE == 1 && split)
        for (uint32_t pix = 0; pix < npix; ++pix) temporal_merge_pixel(h->sv, fv, pix);
}

extern "C" {

void* hc_create(const frt_scene* s, uint32_t W, uint32_t H, uint32_t max_depth, int nthreads) {
    HostCheck* h = new HostCheck();
    const SceneBuilder& b = s->b;
    h->b = &b; h->W = W; h->H = H; h->max_depth = max_depth; h->nthreads = nthreads < 1 ? 1 : nthreads;
    for (auto& l : b.color_textures) h->color_tex.insert(h->color_tex.end(), l.begin(), l.end());
    for (auto& l : b.data_textures) h->data_tex.insert(h->data_tex.end(), l.begin(), l.end());
    SceneView& sv = h->sv;
    sv.nodes = reinterpret_cast<const float4*>(b.pair_nodes.data());
    sv.nodes4 = reinterpret_cast<const float4*>(b.quad_nodes.data());
    sv.tris = reinterpret_cast<const float4*>(b.tri_slots.data());
    b.ensure_wide8();
    sv.nodes8 = b.wide8.ok ? reinterpret_cast<const uint4*>(b.wide8.words.data()) : nullptr;
    sv.tris8 = reinterpret_cast<const float4*>(b.tri_slots8.data());
    sv.num_nodes8 = b.wide8.ok ? (uint32_t)(b.wide8.words.size() / kWide8Words) : 0u; sv.stack_need8 = b.wide8.stack_need;
    sv.shade_tris = reinterpret_cast<const float4*>(b.shade_tris.data());
    sv.instances = reinterpret_cast<const InstanceView*>(b.instances_dev.data());
    sv.mesh_infos = reinterpret_cast<const MeshInfoView*>(b.mesh_infos.data());
    sv.attributes = reinterpret_cast<const VertexAttrView*>(b.attributes.data());
    sv.indices = b.indices.data();
    sv.materials = reinterpret_cast<const MaterialView*>(b.materials.data());
    sv.lights = reinterpret_cast<const LightView*>(b.lights.data());
    sv.color_tex = h->color_tex.data(); sv.data_tex = h->data_tex.data(); sv.srgb_lut = b.srgb_lut;
    sv.num_materials = (uint32_t)b.materials.size(); sv.num_lights = (uint32_t)b.lights.size();
    sv.num_nodes = (uint32_t)b.pair_nodes.size(); sv.num_tris = (uint32_t)b.tri_slots.size();
    size_t n = (size_t)W * H;
    for (int i = 0; i < 2; ++i) {
        h->gpos[i].assign(n, make_float4(0, 0, 0, 0)); h->gnormal[i].assign(n, make_float4(0, 0, 0, 0)); h->accum[i].assign(n, make_float4(0, 0, 0, 0));
        h->galbedo[i].assign(n, 0u); h->res[i].assign(n, zero_reservoir());
    }
    h->display.assign(n, 0u); h->gmotion.assign(n, make_float2(0, 0)); h->raw.assign(n, make_uint2(0, 0));
    h->cand.assign(n, make_float4(0, 0, 0, 0));
    return h;
}
void hc_destroy(void* p) { delete (HostCheck*)p; }

void hc_set_jitter(void* p, float jx, float jy) { HostCheck* h = (HostCheck*)p; h->jitter[0] = jx; h->jitter[1] = jy; }

// sm == 1: drive the resumable state machine (frt_path.hpp) instead of the straight-line functions (frt_mono.hpp);
// sm >= 2: straight-line functions cut at bounce depth `sm` with the continuation-queue protocol (run_stage_cut);
// sm >= 1000: cut at sm - 1000 AND the temporal stage split into T-trace (candidate record) + T-merge, as the default kernels run it
// sm >= 2000: cut at sm - 2000, split, and every resumed bounce in the pulled-apart form of the stream kernel (path_loop_split)
void hc_render(void* p, const frt_camera_uniform* cam, int sm) {
    HostCheck* h = (HostCheck*)p;
    uint32_t cur = h->frame_count & 1u, prv = cur ^ 1u;
    FrameView fv{};
    fv.gpos = h->gpos[cur].data(); fv.gnormal = h->gnormal[cur].data(); fv.galbedo = h->galbedo[cur].data();
    fv.gpos_prev = h->gpos[prv].data(); fv.gnormal_prev = h->gnormal[prv].data(); fv.galbedo_prev = h->galbedo[prv].data();
    fv.gmotion = h->gmotion.data(); fv.res_temporal = h->res[0].data(); fv.res_spatial = h->res[1].data();
    fv.cand = h->cand.data(); fv.jitter_x = h->jitter[0]; fv.jitter_y = h->jitter[1];
    fv.raw = h->raw.data(); fv.display = h->display.data(); fv.history = h->accum[prv].data(); fv.accum = h->accum[cur].data();
    fv.ray_counters = nullptr; fv.W = h->W; fv.H = h->H; fv.frame_count = h->frame_count; fv.max_depth = h->max_depth;
    fv.y0 = 0; fv.y1 = h->H; fv.own_y0 = 0; fv.own_y1 = h->H; fv.prev_y0 = 0; fv.prev_y1 = h->H; fv.overflow = nullptr;
    memcpy(&fv.cam, cam, sizeof(CameraView));
    int nt = h->nthreads;
    std::vector<unsigned long long> rc((size_t)nt * 2, 0ull);
    for (int stage = 0; stage < 4; ++stage) {
        if (sm >= 2 && (stage == 1 || stage == 2)) {
            unsigned long long r2[2] = {0, 0};
            const bool pulled = sm >= 2000, split = sm >= 1000; const uint32_t cutd = (uint32_t)(pulled ? sm - 2000 : (split ? sm - 1000 : sm));
            if (stage == 1) run_stage_cut<1>(h, fv, cutd, r2, split, pulled); else run_stage_cut<2>(h, fv, cutd, r2, false, pulled);
            h->rays[0] += r2[0]; h->rays[1] += r2[1];
            continue;
        }
        auto work = [&](int tid) {
            uint32_t stack[kStackDepth];
            for (uint32_t y = (uint32_t)tid; y < h->H; y += (uint32_t)nt)
                for (uint32_t x = 0; x < h->W; ++x) {
                    PathCtx c(h->sv, fv, stack, 1u);
                    if (stage == 0) gbuffer_pixel(c, x, y);
                    else if (stage == 1) { if (sm == 1) temporal_pixel_sm(c, x, y); else temporal_pixel(c, x, y); }
                    else if (stage == 2) { if (sm == 1) spatial_pixel_sm(c, x, y); else spatial_pixel(c, x, y); }
                    else post_pixel(fv, x, y);
                    rc[2 * tid] += c.n_closest; rc[2 * tid + 1] += c.n_any;
                }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    for (int t = 0; t < nt; ++t) { h->rays[0] += rc[2 * t]; h->rays[1] += rc[2 * t + 1]; }
    h->frame_count += 1;
}

// same selectors as FRT_BUF_* in include/frt.h
int hc_read(void* p, int buf, int index, void* out) {
    HostCheck* h = (HostCheck*)p;
    size_t n = (size_t)h->W * h->H;
    int i = index & 1;
    switch (buf) {
    case FRT_BUF_GPOS: memcpy(out, h->gpos[i].data(), n * 16); break;
    case FRT_BUF_GNORMAL: memcpy(out, h->gnormal[i].data(), n * 16); break;
    case FRT_BUF_GALBEDO: memcpy(out, h->galbedo[i].data(), n * 4); break;
    case FRT_BUF_GMOTION: memcpy(out, h->gmotion.data(), n * 8); break;
    case FRT_BUF_RESERVOIR: memcpy(out, h->res[i].data(), n * 32); break;
    case FRT_BUF_RAW: memcpy(out, h->raw.data(), n * 8); break;
    case FRT_BUF_DISPLAY: memcpy(out, h->display.data(), n * 4); break;
    case FRT_BUF_ACCUM: memcpy(out, h->accum[i].data(), n * 16); break;
    default: return -1;
    }
    return 0;
}
void hc_rays(void* p, unsigned long long out[2]) { out[0] = ((HostCheck*)p)->rays[0]; out[1] = ((HostCheck*)p)->rays[1]; }

// probe: closest / any hits through the product traversal. quantized = 0: float pair nodes (trace), 1: the 16-bit pair nodes the
// resident kernels cache in LDS (trace_q), here over host arrays, 2: quad nodes (trace4)
void hc_trace(const frt_scene* s, int any, uint32_t n, const float* o, const float* d, float tmin, const float* tmax,
              float* t_out, uint32_t* tri_out, float* uv_out, uint8_t* front_out, int quantized) {
    const SceneBuilder& b = s->b;
    SceneView sv{};
    sv.nodes = reinterpret_cast<const float4*>(b.pair_nodes.data());
    sv.nodes4 = reinterpret_cast<const float4*>(b.quad_nodes.data());
    sv.tris = reinterpret_cast<const float4*>(b.tri_slots.data());
    b.ensure_wide8();
    sv.nodes8 = b.wide8.ok ? reinterpret_cast<const uint4*>(b.wide8.words.data()) : nullptr;
    sv.tris8 = reinterpret_cast<const float4*>(b.tri_slots8.data());
    sv.num_nodes8 = b.wide8.ok ? (uint32_t)(b.wide8.words.size() / kWide8Words) : 0u; sv.stack_need8 = b.wide8.stack_need;
    sv.instances = reinterpret_cast<const InstanceView*>(b.instances_dev.data());
    QBvh qb;
    qb.a = reinterpret_cast<const uint4*>(b.qnode_a.data()); qb.b = reinterpret_cast<const uint4*>(b.qnode_b.data()); qb.tris = sv.tris;
    qb.qmin = mk3(b.qmin[0], b.qmin[1], b.qmin[2]); qb.qstep = mk3(b.qstep[0], b.qstep[1], b.qstep[2]);
    uint32_t stack[kStackDepth];
    for (uint32_t i = 0; i < n; ++i) {
        HitRec h;
        f3 oo = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), dd = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        if (quantized == 3) {      // the 8-wide nodes with grid boxes (frt_trace.hpp: trace8)
            const char* nb = reinterpret_cast<const char*>(sv.nodes8);
            if (any) trace8<true>(sv, nb, oo, dd, tmin, tmax[i], stack, 1u, h); else trace8<false>(sv, nb, oo, dd, tmin, tmax[i], stack, 1u, h);
        }
        else if (quantized == 2) { if (any) trace4<true>(sv, oo, dd, tmin, tmax[i], stack, 1u, h); else trace4<false>(sv, oo, dd, tmin, tmax[i], stack, 1u, h); }
        else if (quantized) { if (any) trace_q<true>(sv, qb, oo, dd, tmin, tmax[i], stack, 1u, h); else trace_q<false>(sv, qb, oo, dd, tmin, tmax[i], stack, 1u, h); }
        else if (any) trace<true>(sv, oo, dd, tmin, tmax[i], stack, 1u, h);
        else trace<false>(sv, oo, dd, tmin, tmax[i], stack, 1u, h);
        t_out[i] = h.tri != 0xFFFFFFFFu ? h.t : -1.0f;
        tri_out[i] = h.tri;
        if (uv_out) { uv_out[2 * i] = h.u; uv_out[2 * i + 1] = h.v; }
        if (front_out) front_out[i] = h.front;
    }
}

// probe: the quad tree the default kernels walk. out = {quad nodes, stack need reported by the builder, stack need found by walking every
// root-to-leaf path here, leaves reached, triangle slots covered by those leaves, children per node x 100}
void hc_quad_stats(const frt_scene* s, uint32_t out[6]) {
    const SceneBuilder& b = s->b;
    out[0] = (uint32_t)b.quad_nodes.size(); out[1] = b.quad_stack_need; out[2] = out[3] = out[4] = out[5] = 0;
    if (b.quad_nodes.empty()) return;
    struct Item { uint32_t node, used; };
    std::vector<Item> todo(1, Item{0u, 0u});
    uint64_t kids = 0;
    while (!todo.empty()) {
        const Item it = todo.back(); todo.pop_back();
        const QuadNode& q = b.quad_nodes[it.node];
        uint32_t refs[4]; int n = 0;
        for (int i = 0; i < 4; ++i) { uint32_t r; memcpy(&r, &q.q[24 + i], 4); if (r != 0xFFFFFFFFu) refs[n++] = r; }
        kids += (uint64_t)n;
        const uint32_t used = it.used + (uint32_t)(n - 1);       // all children hit: n - 1 pushed while the nearest is entered
        out[2] = std::max(out[2], used);
        for (int i = 0; i < n; ++i) {
            if (refs[i] & 0x80000000u) { out[3] += 1; out[4] += (refs[i] >> 24) & 0x7Fu; }
            else todo.push_back(Item{refs[i], used});
        }
    }
    out[5] = (uint32_t)(kids * 100u / b.quad_nodes.size());
}

// probe: the 8-wide tree (frt_bvh8.hpp). out = {nodes, stack need reported by the builder, stack need found by walking every root-to-leaf path with all
// inner children hit, leaf children reached, triangle slots covered by them, children per node x 100, grid boxes that do not contain their child's float box, levels}
void hc_wide8_stats(const frt_scene* s, uint32_t out[8]) {
    const SceneBuilder& b = s->b;
    b.ensure_wide8();
    for (int i = 0; i < 8; ++i) out[i] = 0;
    if (!b.wide8.ok) return;
    const std::vector<uint32_t>& w = b.wide8.words;
    out[0] = (uint32_t)(w.size() / kWide8Words); out[1] = b.wide8.stack_need;
    struct Item { uint32_t node, used, level; };
    std::vector<Item> todo(1, Item{0u, 0u, 1u});
    std::vector<uint8_t> seen(b.tri_slots8.size(), 0);
    uint64_t kids = 0;
    while (!todo.empty()) {
        const Item it = todo.back(); todo.pop_back();
        const uint32_t* n = &w[(size_t)it.node * kWide8Words];
        const uint32_t imask = n[3] >> 24, leafmask = n[5] >> 24, tri_base = n[5] & 0xFFFFFFu;
        if (imask & leafmask) out[6] += 1000000u;      // a slot cannot be both
        const uint32_t n_in = (uint32_t)__builtin_popcount(imask);
        kids += n_in + (uint32_t)__builtin_popcount(leafmask);
        out[7] = std::max(out[7], it.level);
        float p[3]; memcpy(p, n, 12);
        const uint16_t* q = reinterpret_cast<const uint16_t*>(&n[8]);
        const uint8_t* meta = reinterpret_cast<const uint8_t*>(&n[6]);
        const uint32_t used = it.used + (n_in >= 2u ? 1u : 0u);
        out[2] = std::max(out[2], used);
        uint32_t rank = 0;
        for (int c = 0; c < 8; ++c) {
            if (!(((imask | leafmask) >> c) & 1u)) continue;
            for (int a = 0; a < 3; ++a) {      // the planes trace8 stands on bracket the child's float box
                const float step = u2f(((n[3] >> (8 * a)) & 0xFFu) << 23);
                const float lo = p[a] + (float)q[16 * a + c] * step, hi = p[a] + (float)q[16 * a + 8 + c] * step;
                if (!(lo <= b.wide8.child_boxes[(size_t)it.node * 48 + c * 6 + a] && hi >= b.wide8.child_boxes[(size_t)it.node * 48 + c * 6 + 3 + a])) out[6] += 1u;
            }
            if ((imask >> c) & 1u) { todo.push_back(Item{n[4] + rank, used, it.level + 1u}); ++rank; }
            else {
                out[3] += 1u;
                const uint32_t first = tri_base + (meta[c] & 31u), cnt = meta[c] >> 5;
                for (uint32_t k = 0; k < cnt; ++k) { if (first + k < seen.size() && !seen[first + k]) { seen[first + k] = 1; out[4] += 1u; } else out[6] += 1000u; }
            }
        }
    }
    out[5] = (uint32_t)(kids * 100u / std::max<size_t>(w.size() / kWide8Words, 1));
}

} // extern "C"
