// frt_renderer.hip — C ABI (include/frt.h) + the renderer object: per-pixel buffers in HBM, scene replica upload,
// per-frame stage launches. Mirrors Renderer / RenderTargets of src/renderer.rs:26-170, :206-336, :349-518 and the
// ping-pong wiring of src/passes/{gbuffer,restir,restir_spatial,post}.rs. There is no CPU path in this file.
#include "frt_scene.hpp"
#include "frt_kernels.hpp"
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <string>
#include <vector>
#include <cstdlib>

#ifndef FRT_EXPERIMENTS
#define FRT_EXPERIMENTS 0      // 1: lib/libfrt_exp.so (`make experiments`): the measured-and-not-kept kernel designs and their FRT_* environment knobs
#endif

using namespace frt;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
namespace frt { int set_error(int code, const std::string& msg) { return fail(code, msg); } }   // for the other translation units of the ABI
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(FRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Every entry point that touches the device makes the renderer's device current and gives the caller's back on return (a host that
// drives several devices from one thread — frt_multi_renderer does — must not find its current device changed by a call).
struct DeviceGuard {
    int prev = -1; bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define FRT_DEVICE(r) DeviceGuard guard_((r)->device); if (!guard_.ok) return fail(FRT_ERR_HIP, "hipSetDevice failed")

static_assert(sizeof(frt_vertex_attr) == 32 && sizeof(frt_material) == 64 && sizeof(frt_light) == 64, "ABI struct sizes");
static_assert(sizeof(frt_camera_uniform) == 288 && sizeof(frt_reservoir) == 32 && sizeof(frt_bvh2_node) == 32, "ABI struct sizes");
static_assert(sizeof(CameraView) == 288 && sizeof(ReservoirView) == 32 && sizeof(InstanceView) == 64 && sizeof(InstanceDev) == 64, "view sizes");
static_assert(sizeof(MaterialView) == 64 && sizeof(LightView) == 64 && sizeof(VertexAttrView) == 32 && sizeof(MeshInfoView) == 16, "view sizes");
static_assert(sizeof(PairNode) == 64 && sizeof(TriSlot) == 48 && sizeof(ShadeTri) == 128, "GPU layout sizes");

// ------------------------------------------------------------------------------------------------ renderer object
static const uint32_t kHaloGbuffer = 12;   // spatial reuse radius 10 (restir_spatial.wgsl:903, :921) + spatial halo 2
static const uint32_t kHaloSpatial = 2;    // post reads raw radiance within +-2 rows (post.wgsl:93)
static const uint32_t kReuseRadius = 10;   // rows of temporal reservoirs a spatial pixel may read above / below itself

// G-buffer, motion and candidate targets exist kGSets = kSpecDepth + 1 times. With one frame running ahead (the default) that is the
// reference's two ping-pong slots (gbuffer.rs:299): G-buffer(f+1) overwrites the set of frame f-1, whose last readers are done by then.
// Two frames ahead (kSpecDepth = 2, a third set: +60 B per pixel) was built and measured: 2.204 vs 2.208 ms — the ahead stream is in
// order, so frame f+2 cannot start before f+1's latency-bound tail has drained — and is therefore not compiled in. Which physical set
// holds which of the reference's two logical slots is tracked per frame (GSlots below), so any kSpecDepth works.
static const int kSpecDepth = 1;             // frames whose G-buffer + T-trace may run ahead
static const int kGSets = 3;                 // physical G-buffer sets addressable; a renderer owns `gsets` of them (2, strips under the pipeline 3)
enum { B_GPOS0 = 0, B_GNRM0 = B_GPOS0 + kGSets, B_GALB0 = B_GNRM0 + kGSets, B_GMOT0 = B_GALB0 + kGSets, B_CAND0 = B_GMOT0 + kGSets,
       B_RES0 = B_CAND0 + kGSets, B_RES1, B_RAW, B_DISP, B_ACC0, B_ACC1, B_COUNT };
// Buffers outside the arena (frt_renderer_arena_bytes stays 228 B per pixel): the third G-buffer set. Only strip renderers under the pipeline
// allocate them (frt_renderer::extras): with a third set the next frame's G-buffer + T-trace need not wait for this frame's T-merge.
static bool is_extra(int b) { return b < B_RES0 && (b % kGSets) == 2; }
static uint32_t bpp_of(int b) {
    if (b < B_GALB0) return 16u;        // gpos, gnormal
    if (b < B_GMOT0) return 4u;         // galbedo
    if (b < B_CAND0) return 8u;         // gmotion
    if (b < B_RES0) return 16u;         // candidate
    if (b <= B_RES1) return 32u;
    if (b == B_RAW) return 8u;
    if (b == B_DISP) return 4u;
    return 16u;                         // accumulation
}
struct GSlots { uint32_t g, gprev, aux; };   // physical sets: this frame's G-buffer, the previous logical slot's, and motion / candidate (= g under the pipeline)

// Device counters (unsigned long long each): [0..7] committed rays per stage {closest, any}; [8] halo overflow;
// [9..12] PENDING rays of a G-buffer + T-trace pair that ran ahead of its frame (committed by its T-merge, dropped with a discarded speculation)
static const int kPending = 2;               // pending ray-count sets / T-trace events: consecutive speculated frames alternate (with three G-buffer sets
                                             // T-trace(f+1) may start before T-merge(f) has committed the counts of T-trace(f))
enum { C_STAGE = 0, C_HALO = 8, C_PENDING = 9, C_COUNT = 9 + 4 * kPending };   // a pending set of four per speculated frame in flight
#if FRT_EXPERIMENTS
static const int kTileStateWords = 8;      // per traced stage (experiments/frt_experiment_kernels.hpp: TileOrder uses 6)
#endif

#if FRT_EXPERIMENTS
// Everything the measured-and-not-kept kernel designs (csrc/experiments/frt_experiment_kernels.hpp) need in a renderer: their state, their FRT_*
// environment knobs, their allocations. Compiled into lib/libfrt_exp.so only (`make experiments`); the product library has none of it.
struct ExpState {
    int spec_depth = kSpecDepth;           // frames speculated ahead (FRT_SPEC_DEPTH, 0 .. kSpecDepth)
    long cont_grid = -1;                   // FRT_CONT_GRID: slots the spatial continuation grids cover at least (-1: half the stage's pixels)
    uint32_t* d_tiles = nullptr;           // per traced stage kTileStateWords words of sweep-direction state (FRT_TILE_ORDER=1), or null
    bool wg_park = true;                   // pixel kernels reserve queue slots once per workgroup (FRT_WG_PARK=0: once per wave)
    bool wavefront = false;                // ray-level wavefront (FRT_WAVEFRONT=1): per bounce depth a trace launch and a shade launch
    uint32_t* d_wf_words[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; uint32_t* d_wf_items[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    uint32_t* d_wf_hits[2] = {nullptr, nullptr}; uint32_t* d_wf_counts = nullptr;   // per stage: records x 2, item lists x 2, hit buffer; 2 x 96 counters
    bool stream_mode = false; uint32_t shade_min = 32, stream_slice = 8;   // stream kernel (resumable traversal) instead of continuation launches
    bool refill = false; uint32_t refill_min = 16;   // bounce kernel with lane refill (single cut) instead of continuation launches
    bool resident = false;                 // traced stages through the resident kernels (BVH cached in LDS, persistent workgroups)
    uint32_t res_nodes = 0; bool res_tris = false; uint32_t num_cus = 0, res_batch = 0;
    uint32_t* d_work = nullptr;            // work counters of the resident launches: [stage 1|2][launch slot][2]
};
#endif

struct frt_renderer {
    int device = 0;
    hipStream_t stream = nullptr;          // the chain: T-merge -> spatial pixels -> spatial continuations (and everything, without FRT_FLAG_PIPELINE)
    bool own_stream = false;
    hipStream_t ahead = nullptr;           // FRT_FLAG_PIPELINE: G-buffer(f+1), T-trace(f+1)
    hipStream_t edge = nullptr;            // FRT_FLAG_PIPELINE, strips: the spatial pixel launches of the halo-dependent edge rows (beside the interior launch)
    hipStream_t edge2 = nullptr;           // ... the second edge of a middle strip: its launch runs beside the first one's instead of behind it
    hipEvent_t ev_spix = nullptr, ev_tt[kPending] = {}, ev_tail = nullptr, ev_tm[2] = {nullptr, nullptr}, ev_edge = nullptr, ev_edge2 = nullptr, ev_edge_ready = nullptr;
    bool tail_pending = false;             // work enqueued on `ahead` that the main stream has not been ordered behind yet
    bool edge_in_flight = false, edge2_in_flight = false;
    uint32_t W = 0, H = 0, max_depth = 8, rb = 0, re = 0, flags = 0, motion_halo = 0;
    uint32_t frame_count = 0;
    float jitter[2] = {0.0f, 0.0f};        // PostParams.jitter of the next post stage
    SceneView sv{};
    std::vector<void*> scene_allocs;
    uint8_t* arena = nullptr;
    bool own_arena = false;
    size_t arena_bytes = 0;
    size_t off[B_COUNT] = {};
    unsigned long long* d_counters = nullptr;
    // continuation queues: per traced stage one word buffer per path segment parity (the second only with two cuts or more)
    uint32_t* d_qwords[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    uint32_t qcap = 0, qcap_max = 0;       // slots per queue; upper bound = every traced pixel parks
    uint32_t qslots[2][2] = {{0, 0}, {0, 0}};   // slots of each word buffer [stage][first | second buffer], derived from qcap (alloc_queues)
    uint64_t qbytes = 0;                   // device bytes of the word buffers
    bool qcap_fixed = false;               // capacity given by the caller: never grown
    uint32_t* d_qcount = nullptr;          // [stage 1|2][launch parity 0|1][kMaxCuts + 1] counters, then [stage] overflow counters
    uint32_t* h_qseen = nullptr; uint32_t* d_qseen = nullptr;   // one word of mapped host memory: set by a wave that found its queue full (ContQueue::seen)
    uint32_t qparity[2] = {0, 0};
    uint32_t ncuts = 2, cuts[kMaxCuts] = {3, 4, 0, 0};   // measured best on the Cornell Box
    bool vote = false;                     // traced kernels with the voting BVH walk (set from the size of the scene's quad tree, upload_scene)
    uint32_t walk = kWalkQuad, wide_lds_bytes = 0, wg_rows = 0;      // which tree the traced kernels walk and how (frt_kernels.hpp: kWalk*; upload_scene)
#if FRT_EXPERIMENTS
    ExpState x;                            // lib/libfrt_exp.so only: state and FRT_* knobs of the measured-and-not-kept kernel designs (csrc/experiments/)
#endif
    frt_stats stats{};
    struct Timed { hipEvent_t a, b; int slot; };
    std::vector<Timed> pending;
    std::vector<hipEvent_t> event_pool;
    // frame in progress
    bool failed = false;                   // a HIP call failed in the middle of a frame: the stage flags and stream order are no longer trustworthy; render calls
                                           // return FRT_ERR_STATE until frt_renderer_clear
    bool frame_open = false, g_done = false, tt_done = false, tm_done = false, s_started = false, s_inner_done = false, s_edge_done = false;
    bool from_speculation = false;
    Timed s_timer{};
    bool s_timed = false;
    // which physical G set holds the reference's logical slot 0 / 1 (frame_count % 2) for the NEXT G-buffer launch, the sets of the frame in
    // progress and of the last finished frames (what reads through the ABI see)
    uint32_t logical_phys[2] = {0, 1};
    GSlots cur_slots{0, 1, 0}, last_slots{0, 1, 0}, before_last_slots{1, 0, 0};
    // speculation: G-buffer + T-trace of the next frames, enqueued on `ahead` under the cameras a static scene will present
    struct Spec { frt_camera_uniform cam; uint32_t frame; GSlots slots; uint32_t logical_before[2]; int idx; };
    std::vector<Spec> specs;               // oldest first, at most kSpecDepth
    int spec_next_idx = 0, cur_spec_idx = 0;
    bool camera_static = false;
    frt_camera_uniform last_cam{}, cur_cam{};
    bool have_last_cam = false;
    uint8_t* extras = nullptr; size_t extras_bytes = 0;      // the buffers of is_extra(), when this renderer has them
    uint32_t gsets = 2;                    // G-buffer sets in use
    uint64_t serial = 0;                   // frames finished since creation (never reset: parity of the per-frame events)
    void* buf(int b) const { return is_extra(b) ? extras + off[b] : arena + off[b]; }
    bool pipeline() const { return ahead != nullptr; }
};

static size_t arena_layout(uint32_t W, uint32_t H, size_t off[B_COUNT], bool extras = false) {
    size_t n = (size_t)W * H, cur = 0;
    for (int b = 0; b < B_COUNT; ++b) {
        if (is_extra(b) != extras) continue;
        if (off) off[b] = cur;
        cur += (n * bpp_of(b) + 255u) & ~(size_t)255u;
    }
    return cur;
}

template <class T, class D>
static int upload(frt_renderer* r, const std::vector<T>& v, const D** out) {
    void* d = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    HIP_TRY(hipMalloc(&d, bytes));
    r->scene_allocs.push_back(d);
    if (!v.empty()) HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = reinterpret_cast<const D*>(d);
    return FRT_OK;
}

#ifndef FRT_VOTE_MIN_NODES
#define FRT_VOTE_MIN_NODES 32768      // (4 MiB of quad nodes; A/B builds: 0 = every scene votes, a huge value = none does)
#endif
static const size_t kVoteMinQuadNodes = FRT_VOTE_MIN_NODES;
#if FRT_EXPERIMENTS
#ifndef FRT_WIDE_LDS_MAX
#define FRT_WIDE_LDS_MAX 28672      // bytes of 8-wide nodes a traced workgroup may hold in LDS beside its 9 KiB of stack words: 4 workgroups per CU (A/B builds)
#endif
static const size_t kWideLdsMaxBytes = FRT_WIDE_LDS_MAX;
#endif
static int upload_scene(frt_renderer* r, const SceneBuilder& b) {
    SceneView& sv = r->sv;
    int rc;
    if ((rc = upload(r, b.pair_nodes, &sv.nodes))) return rc;
    if ((rc = upload(r, b.quad_nodes, &sv.nodes4))) return rc;
    // Which walk the traced kernels use (frt_trace.hpp: trace4<ANY, VOTE>). The voting loop pays where walks are long and costs its own instructions
    // where they are short. Per frame, never / always voting, with the leaf step that fetches both triangles together (before it the 25k-node scene
    // still gained 2 %): 74k quad nodes (246k triangles, 4K, 16 bounces) 20.98 / 19.79 ms; 25k (82k triangles) 2.948 / 2.987 ms; 9.5k (32k triangles)
    // 1.279 / 1.340 ms; 390 (the Cornell Box) 1.565 / 1.611 ms.
    r->vote = b.quad_nodes.size() >= kVoteMinQuadNodes;
    if ((rc = upload(r, b.tri_slots, &sv.tris))) return rc;
    sv.nodes8 = nullptr; sv.tris8 = nullptr; sv.num_nodes8 = 0u; sv.stack_need8 = 0u;
    r->walk = kWalkQuad; r->wide_lds_bytes = 0u;
    r->wg_rows = b.quad_stack_need + 1u;
#if FRT_EXPERIMENTS
    // Round 4's two measured-and-not-kept walks (lib/libfrt_exp.so only; profiles/r4_experiments/wide8.md, collective_walks.md):
    // FRT_FLAG_WALK_WIDE / _HBM: the 8-wide tree with grid boxes (frt_bvh8.hpp; frt_trace.hpp: trace8) when the scene has one whose stack fits trace8's
    // kStack8 words; a tree of at most kWideLdsMaxBytes is copied into every traced workgroup's LDS (kWalkWideLds; the Cornell Box: 174 nodes, 22 KiB).
    // Cornell Box 1.57 (LDS) / 1.60 (HBM) vs 1.54 ms per frame, ReSTIR scene 1.52 vs 1.27, configs[3] stand-in 3.50 vs 2.99, configs[4] stand-in 26.0 vs 19.6.
    if (r->flags & (FRT_FLAG_WALK_WIDE | FRT_FLAG_WALK_WIDE_HBM)) {
        b.ensure_wide8();
        if (b.wide8.ok && b.wide8.stack_need <= (uint32_t)kStack8) {
            if ((rc = upload(r, b.wide8.words, &sv.nodes8))) return rc;
            if ((rc = upload(r, b.tri_slots8, &sv.tris8))) return rc;
            sv.num_nodes8 = (uint32_t)(b.wide8.words.size() / kWide8Words); sv.stack_need8 = b.wide8.stack_need;
            const size_t bytes = b.wide8.words.size() * sizeof(uint32_t);
            r->walk = (bytes <= kWideLdsMaxBytes && !(r->flags & FRT_FLAG_WALK_WIDE_HBM)) ? kWalkWideLds : kWalkWide;
            r->wide_lds_bytes = r->walk == kWalkWideLds ? (uint32_t)bytes : 0u;
            r->vote = false;
        }
    }
    // FRT_FLAG_WG_TRACE: collective walks over the quad tree (frt_kernels.hip: wg_trace): a workgroup's rays re-dealt to dense, direction-sorted waves.
    // Cornell Box 1.65 (octant-sorted) / 1.55 (dense only) vs 1.43 ms per frame.
    if (r->walk == kWalkQuad && (r->flags & FRT_FLAG_WG_TRACE)) r->walk = kWalkQuadWg;
#endif
    if ((rc = upload(r, b.qnode_a, &sv.qnode_a))) return rc;
    if ((rc = upload(r, b.qnode_b, &sv.qnode_b))) return rc;
    for (int a = 0; a < 3; ++a) { sv.qmin[a] = b.qmin[a]; sv.qstep[a] = b.qstep[a]; }
    sv.bvh_depth = b.bvh_depth;
    sv.num_nodes4 = (uint32_t)b.quad_nodes.size();
    if ((rc = upload(r, b.shade_tris, &sv.shade_tris))) return rc;
    if ((rc = upload(r, b.instances_dev, &sv.instances))) return rc;
    if ((rc = upload(r, b.mesh_infos, &sv.mesh_infos))) return rc;
    if ((rc = upload(r, b.attributes, &sv.attributes))) return rc;
    if ((rc = upload(r, b.indices, &sv.indices))) return rc;
    if ((rc = upload(r, b.materials, &sv.materials))) return rc;
    if ((rc = upload(r, b.lights, &sv.lights))) return rc;
    auto pack_layers = [](const std::vector<std::vector<uint8_t>>& layers) {
        std::vector<uint8_t> all;
        for (auto& l : layers) all.insert(all.end(), l.begin(), l.end());
        return all;
    };
    std::vector<uint8_t> ct = pack_layers(b.color_textures), dt = pack_layers(b.data_textures);
    if ((rc = upload(r, ct, &sv.color_tex))) return rc;
    if ((rc = upload(r, dt, &sv.data_tex))) return rc;
    std::vector<float> lut(b.srgb_lut, b.srgb_lut + 256);
    if ((rc = upload(r, lut, &sv.srgb_lut))) return rc;
    sv.num_materials = (uint32_t)b.materials.size();
    sv.num_lights = (uint32_t)b.lights.size();
    sv.num_nodes = (uint32_t)b.pair_nodes.size();
    sv.num_tris = (uint32_t)b.tri_slots.size();
    return FRT_OK;
}

static void phase_rows(const frt_renderer* r, uint32_t out[8]) {
    bool whole = (r->rb == 0 && r->re == r->H);
    auto lo = [&](uint32_t h) { return whole ? 0u : (r->rb > h ? r->rb - h : 0u); };
    auto hi = [&](uint32_t h) { return whole ? r->H : std::min(r->H, r->re + h); };
    const uint32_t hg = std::max(kHaloGbuffer, r->motion_halo);   // temporal(f+1) reprojects into the previous G-buffer within the motion halo
    out[0] = lo(hg); out[1] = hi(hg);
    out[2] = r->rb; out[3] = r->re;
    out[4] = lo(kHaloSpatial); out[5] = hi(kHaloSpatial);
    out[6] = r->rb; out[7] = r->re;
}
// Spatial rows that need nothing from a neighbour strip: [ia, ib) on the stage's 16-row tile grid (the rows whose +-10-row reuse
// neighbourhood lies inside the strip's own temporal rows). The rest, [y0, ia) and [ib, y1), wait for the halo exchange.
static void spatial_inner_rows(const frt_renderer* r, uint32_t y0, uint32_t y1, uint32_t& ia, uint32_t& ib) {
    ia = y0; ib = y1;
    if (r->rb > 0) { const uint32_t need = r->rb + kReuseRadius; ia = y0 + ((need - y0 + 15u) / 16u) * 16u; }
    if (r->re < r->H) { const uint32_t lim = r->re > kReuseRadius ? r->re - kReuseRadius : 0u; ib = lim > y0 ? y0 + ((lim - y0) / 16u) * 16u : y0; }
    if (ia > y1) ia = y1;
    if (ib < ia) ib = ia;
}

static void fill_frame_view(const frt_renderer* r, const frt_camera_uniform* cam, uint32_t frame_count, const GSlots& gs, FrameView& fv) {
    uint32_t cur = frame_count & 1u, prv = cur ^ 1u;   // gbuffer.rs:299, restir.rs:543, post.rs:244 (accumulation; the G-buffer's two slots are renamed: gs)
    fv.gpos = (float4*)r->buf(B_GPOS0 + gs.g); fv.gnormal = (float4*)r->buf(B_GNRM0 + gs.g); fv.galbedo = (uint32_t*)r->buf(B_GALB0 + gs.g);
    fv.gpos_prev = (const float4*)r->buf(B_GPOS0 + gs.gprev); fv.gnormal_prev = (const float4*)r->buf(B_GNRM0 + gs.gprev);
    fv.galbedo_prev = (const uint32_t*)r->buf(B_GALB0 + gs.gprev);
    fv.gmotion = (float2*)r->buf(B_GMOT0 + gs.aux);
    fv.res_temporal = (ReservoirView*)r->buf(B_RES0);   // restir.rs:362-378: reads buffers[1], writes buffers[0]
    fv.res_spatial = (ReservoirView*)r->buf(B_RES1);    // renderer.rs:292-293: spatial buffers[0] -> buffers[1]
    fv.cand = (float4*)r->buf(B_CAND0 + gs.aux);
    fv.raw = (uint2*)r->buf(B_RAW); fv.display = (uint32_t*)r->buf(B_DISP);
    fv.history = (const float4*)r->buf(B_ACC0 + prv);   // post.rs:209-224: BG0 history = accum[1], out = accum[0]
    fv.accum = (float4*)r->buf(B_ACC0 + cur);
    fv.ray_counters = r->d_counters;
    fv.W = r->W; fv.H = r->H; fv.frame_count = frame_count; fv.max_depth = r->max_depth;
    fv.y0 = 0; fv.y1 = 0;
    fv.own_y0 = r->rb; fv.own_y1 = r->re;
    bool whole = (r->rb == 0 && r->re == r->H);
    fv.prev_y0 = whole ? 0u : (r->rb > r->motion_halo ? r->rb - r->motion_halo : 0u);
    fv.prev_y1 = whole ? r->H : std::min(r->H, r->re + r->motion_halo);
    fv.overflow = whole ? nullptr : r->d_counters + C_HALO;
    fv.jitter_x = r->jitter[0]; fv.jitter_y = r->jitter[1];
    memcpy(&fv.cam, cam, sizeof(CameraView));
}

// Stream-level fence: the main stream waits for everything enqueued on the `ahead` stream (no host wait).
static int fence_ahead(frt_renderer* r) {
    if (r->ahead && r->tail_pending) {
        HIP_TRY(hipEventRecord(r->ev_tail, r->ahead));
        HIP_TRY(hipStreamWaitEvent(r->stream, r->ev_tail, 0));
        r->tail_pending = false;
    }
    return FRT_OK;
}
static int sync_all(frt_renderer* r) {
    HIP_TRY(hipStreamSynchronize(r->stream));
    if (r->edge) HIP_TRY(hipStreamSynchronize(r->edge));
    if (r->edge2) HIP_TRY(hipStreamSynchronize(r->edge2));
    if (r->ahead) { HIP_TRY(hipStreamSynchronize(r->ahead)); r->tail_pending = false; }
    return FRT_OK;
}

// Per-stage timing (FRT_FLAG_TIMING): event pairs from a pool (no create / destroy per frame), resolved at the next stats call.
static int timer_begin(frt_renderer* r, frt_renderer::Timed& t, int slot, hipStream_t q) {
    for (hipEvent_t* e : {&t.a, &t.b}) {
        if (r->event_pool.empty()) { HIP_TRY(hipEventCreate(e)); }
        else { *e = r->event_pool.back(); r->event_pool.pop_back(); }
    }
    t.slot = slot;
    hipError_t e_ = hipEventRecord(t.a, q);
    if (e_ != hipSuccess) { r->event_pool.push_back(t.a); r->event_pool.push_back(t.b); return fail(FRT_ERR_HIP, std::string("hipEventRecord: ") + hipGetErrorString(e_)); }
    return FRT_OK;
}
static int timer_end(frt_renderer* r, frt_renderer::Timed& t, hipStream_t q) {
    hipError_t e_ = hipEventRecord(t.b, q);
    if (e_ != hipSuccess) { r->event_pool.push_back(t.a); r->event_pool.push_back(t.b); return fail(FRT_ERR_HIP, std::string("hipEventRecord: ") + hipGetErrorString(e_)); }
    r->pending.push_back(t);
    return FRT_OK;
}
static int resolve_timing(frt_renderer* r) {
    for (auto& t : r->pending) {
        float ms = 0.0f;
        HIP_TRY(hipEventSynchronize(t.b));
        HIP_TRY(hipEventElapsedTime(&ms, t.a, t.b));
        if (t.slot < 4) r->stats.ms_stage[t.slot] += ms;
        else r->stats.ms_merge += ms;
        r->event_pool.push_back(t.a); r->event_pool.push_back(t.b);
    }
    r->pending.clear();
    return FRT_OK;
}

extern "C" {

const char* frt_last_error(void) { return g_err.c_str(); }

int frt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------------------------ geometry / materials
int frt_geometry_create(int which, uint32_t subdiv, uint32_t* nverts, uint32_t* nidx, float* pos4, frt_vertex_attr* attrs, uint32_t* idx) {
    Geometry g;
    switch (which) {
    case 0: g = geometry::create_plane(); break;
    case 1: g = geometry::create_cube(); break;
    case 2: if (subdiv > 8) return fail(FRT_ERR_INVALID_ARG, "icosphere subdivisions > 8"); g = geometry::create_sphere(subdiv); break;
    case 3: g = geometry::create_crystal(); break;
    default: return fail(FRT_ERR_INVALID_ARG, "unknown geometry kind");
    }
    if (nverts) *nverts = (uint32_t)g.attributes.size();
    if (nidx) *nidx = (uint32_t)g.indices.size();
    if (pos4) memcpy(pos4, g.positions.data(), g.positions.size() * 4);
    if (attrs) memcpy(attrs, g.attributes.data(), g.attributes.size() * sizeof(frt_vertex_attr));
    if (idx) memcpy(idx, g.indices.data(), g.indices.size() * 4);
    return FRT_OK;
}
void frt_encode_octahedral_normal(const float n[3], float out[2]) { geometry::encode_octahedral_normal(n, out); }
void frt_material_default(const float c[4], frt_material* out) { *out = MaterialBuilder(c[0], c[1], c[2], c[3]); }

// ------------------------------------------------------------------------------------------------ scene
frt_scene* frt_scene_create(void) { return new frt_scene(); }
void frt_scene_destroy(frt_scene* s) { delete s; }

int frt_scene_add_mesh(frt_scene* s, const float* pos4, uint32_t nverts, const frt_vertex_attr* attrs, const uint32_t* idx, uint32_t nidx) {
    if (!s || !pos4 || !attrs || !idx || nverts == 0 || nidx == 0 || nidx % 3 != 0) return fail(FRT_ERR_INVALID_ARG, "add_mesh: bad arguments");
    for (uint32_t i = 0; i < nidx; ++i) if (idx[i] >= nverts) return fail(FRT_ERR_INVALID_ARG, "add_mesh: index out of range");
    Geometry g;
    g.positions.assign(pos4, pos4 + (size_t)nverts * 4);
    g.attributes.assign(attrs, attrs + nverts);
    g.indices.assign(idx, idx + nidx);
    return (int)s->b.add_mesh(g);
}
int frt_scene_add_material(frt_scene* s, const frt_material* m) {
    if (!s || !m) return fail(FRT_ERR_INVALID_ARG, "add_material: null");
    if (s->b.materials.size() >= 0xFFFFu) return fail(FRT_ERR_LIMIT, "more than 65535 materials (custom index packs 16 bits, builder.rs:184)");
    return (int)s->b.add_material(*m);
}
static int check_instance(frt_scene* s, uint32_t mesh_id, uint32_t mat_id, const float* m) {
    if (!s || !m) return fail(FRT_ERR_INVALID_ARG, "instance: null");
    if (mesh_id >= s->b.mesh_infos.size()) return fail(FRT_ERR_INVALID_ARG, "instance: unknown mesh id");
    if (mat_id != 0xFFFFFFFFu && mat_id >= s->b.materials.size()) return fail(FRT_ERR_INVALID_ARG, "instance: unknown material id");
    return FRT_OK;
}
int frt_scene_add_instance(frt_scene* s, uint32_t mesh_id, uint32_t mat_id, const float m[16]) {
    int rc = check_instance(s, mesh_id, mat_id, m);
    if (rc) return rc;
    Mat4 t; memcpy(t.m, m, 64);
    s->b.add_instance(mesh_id, mat_id, t);
    return FRT_OK;
}
int frt_scene_add_light(frt_scene* s, const frt_light* l) {
    if (!s || !l) return fail(FRT_ERR_INVALID_ARG, "add_light: null");
    return (int)s->b.add_light(*l);
}
int frt_scene_register_quad_light(frt_scene* s, uint32_t mesh_id, const float m[16], const float color[3], float intensity) {
    int rc = check_instance(s, mesh_id, 0xFFFFFFFFu, m);
    if (rc) return rc;
    Mat4 t; memcpy(t.m, m, 64);
    s->b.register_quad_light(mesh_id, t, color, intensity);
    return FRT_OK;
}
int frt_scene_register_sphere_light(frt_scene* s, uint32_t mesh_id, const float m[16], const float color[3], float intensity) {
    int rc = check_instance(s, mesh_id, 0xFFFFFFFFu, m);
    if (rc) return rc;
    Mat4 t; memcpy(t.m, m, 64);
    s->b.register_sphere_light(mesh_id, t, color, intensity);
    return FRT_OK;
}
int frt_scene_add_texture(frt_scene* s, int kind, const uint8_t* rgba8) {
    if (!s || !rgba8 || (kind != 0 && kind != 1)) return fail(FRT_ERR_INVALID_ARG, "add_texture: bad arguments");
    auto& v = kind == 0 ? s->b.color_textures : s->b.data_textures;
    if (v.size() >= 0xFFFFu) return fail(FRT_ERR_LIMIT, "too many texture layers");
    return (int)(kind == 0 ? s->b.add_color_texture(rgba8) : s->b.add_data_texture(rgba8));
}
int frt_scene_build(frt_scene* s) {
    if (!s) return fail(FRT_ERR_INVALID_ARG, "build: null");
    s->b.build();
    if (!s->b.built) return fail(FRT_ERR_LIMIT, "build: " + s->b.error);
    return FRT_OK;
}
frt_scene* frt_scene_create_cornell_box(void) {
    frt_scene* s = new frt_scene();
    scenes::create_cornell_box(s->b);
    if (!s->b.built) { g_err = s->b.error; delete s; return nullptr; }
    return s;
}
frt_scene* frt_scene_create_restir_scene(void) {
    frt_scene* s = new frt_scene();
    scenes::create_restir_scene(s->b);
    if (!s->b.built) { g_err = s->b.error; delete s; return nullptr; }
    return s;
}
int frt_scene_counts(const frt_scene* s, uint32_t c[8]) {
    if (!s || !c) return fail(FRT_ERR_INVALID_ARG, "counts: null");
    const SceneBuilder& b = s->b;
    c[0] = (uint32_t)b.tris.size(); c[1] = (uint32_t)b.instances.size(); c[2] = (uint32_t)b.materials.size(); c[3] = (uint32_t)b.lights.size();
    c[4] = (uint32_t)b.mesh_infos.size(); c[5] = (uint32_t)b.attributes.size(); c[6] = (uint32_t)b.indices.size(); c[7] = (uint32_t)b.bvh2.size();
    return FRT_OK;
}
int frt_scene_get(const frt_scene* s, int which, void* out) {
    if (!s || !out) return fail(FRT_ERR_INVALID_ARG, "get: null");
    const SceneBuilder& b = s->b;
    switch (which) {
    case 0: memcpy(out, b.tris.data(), b.tris.size() * sizeof(TriRec)); break;
    case 1: memcpy(out, b.tri_instance.data(), b.tri_instance.size() * 4); break;
    case 2: memcpy(out, b.materials.data(), b.materials.size() * 64); break;
    case 3: memcpy(out, b.lights.data(), b.lights.size() * 64); break;
    case 4: memcpy(out, b.attributes.data(), b.attributes.size() * 32); break;
    case 5: memcpy(out, b.indices.data(), b.indices.size() * 4); break;
    case 6: memcpy(out, b.mesh_infos.data(), b.mesh_infos.size() * 16); break;
    case 7: {
        uint8_t* p = (uint8_t*)out;
        for (const InstanceRec& in : b.instances) {
            const uint32_t h[5] = {in.mesh_id, in.mat_id, in.first_tri, in.tri_count, in.flip};
            memcpy(p, h, 20); memcpy(p + 20, in.m, 64); memcpy(p + 84, in.w2o, 36); p += 120;
        }
    } break;
    case 8: memcpy(out, b.bvh2.data(), b.bvh2.size() * sizeof(frt_bvh2_node)); break;
    case 9: memcpy(out, b.bvh2_tri_index.data(), b.bvh2_tri_index.size() * 4); break;
    case 10: memcpy(out, b.quad_nodes.data(), b.quad_nodes.size() * sizeof(QuadNode)); break;
    case 11: b.ensure_wide8(); memcpy(out, b.wide8.words.data(), b.wide8.words.size() * 4); break;
    case 12: b.ensure_wide8(); memcpy(out, b.tri_slots8.data(), b.tri_slots8.size() * sizeof(TriSlot)); break;
    case 13: memcpy(out, b.tri_slots.data(), b.tri_slots.size() * sizeof(TriSlot)); break;
    case 14: b.ensure_wide8(); memcpy(out, b.wide8.child_boxes.data(), b.wide8.child_boxes.size() * 4); break;
    default: return fail(FRT_ERR_INVALID_ARG, "get: unknown selector");
    }
    return FRT_OK;
}
int frt_scene_bvh_stats(const frt_scene* s, uint32_t st[4]) {
    if (!s || !st) return fail(FRT_ERR_INVALID_ARG, "bvh_stats: null");
    st[0] = s->b.bvh_depth; st[1] = s->b.bvh_leaves; st[2] = s->b.bvh_max_leaf; st[3] = (uint32_t)s->b.pair_nodes.size();
    return FRT_OK;
}
int frt_scene_tree_stats(const frt_scene* s, uint32_t st[8]) {
    if (!s || !st) return fail(FRT_ERR_INVALID_ARG, "tree_stats: null");
    const SceneBuilder& b = s->b;
    b.ensure_wide8();
    st[0] = (uint32_t)b.quad_nodes.size(); st[1] = b.quad_stack_need;
    st[2] = b.wide8.ok ? (uint32_t)(b.wide8.words.size() / kWide8Words) : 0u; st[3] = b.wide8.stack_need; st[4] = b.wide8.depth; st[5] = b.wide8.children;
    st[6] = (uint32_t)b.tri_slots8.size(); st[7] = b.quad_fold;
    return FRT_OK;
}
void frt_camera_default(float aspect, uint32_t frame_count, uint32_t num_lights, frt_camera_uniform* out) {
    camera_default(aspect, frame_count, num_lights, out);
}
int frt_camera_build_uniform(const float position[3], float yaw, float pitch, const float* prev_view_proj, float aspect, uint32_t frame_count,
                             uint32_t num_lights, const float jitter[2], frt_camera_uniform* out, float* unjittered_view_proj) {
    if (!position || !out || !(aspect > 0.0f)) return fail(FRT_ERR_INVALID_ARG, "camera_build_uniform: bad arguments");
    camera_build_uniform(position, yaw, pitch, prev_view_proj, aspect, frame_count, num_lights, jitter ? jitter[0] : 0.0f, jitter ? jitter[1] : 0.0f, out, unjittered_view_proj);
    return FRT_OK;
}
void frt_camera_halton_jitter(uint32_t index, uint32_t width, uint32_t height, float scale, float out[2]) { camera_halton_jitter(index, width, height, scale, out); }

// ------------------------------------------------------------------------------------------------ renderer
uint64_t frt_renderer_arena_bytes(uint32_t width, uint32_t height) { return arena_layout(width, height, nullptr); }

static void free_queues(frt_renderer* r) {
    for (auto& st : r->d_qwords) for (uint32_t*& p : st) { if (p) (void)hipFree(p); p = nullptr; }
#if FRT_EXPERIMENTS
    for (auto& st : r->x.d_wf_words) for (uint32_t*& p : st) { if (p) (void)hipFree(p); p = nullptr; }
    for (auto& st : r->x.d_wf_items) for (uint32_t*& p : st) { if (p) (void)hipFree(p); p = nullptr; }
    for (uint32_t*& p : r->x.d_wf_hits) { if (p) (void)hipFree(p); p = nullptr; }
#endif
}
void frt_renderer_destroy(frt_renderer* r) {
    if (!r) return;
    DeviceGuard guard_(r->device);
    if (r->stream || r->own_stream) (void)hipStreamSynchronize(r->stream);
    if (r->ahead) { (void)hipStreamSynchronize(r->ahead); (void)hipStreamDestroy(r->ahead); }
    if (r->edge) { (void)hipStreamSynchronize(r->edge); (void)hipStreamDestroy(r->edge); }
    if (r->edge2) { (void)hipStreamSynchronize(r->edge2); (void)hipStreamDestroy(r->edge2); }
    for (hipEvent_t e : {r->ev_spix, r->ev_tail, r->ev_tm[0], r->ev_tm[1], r->ev_edge, r->ev_edge2, r->ev_edge_ready}) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : r->ev_tt) if (e) (void)hipEventDestroy(e);
    for (auto& t : r->pending) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (hipEvent_t e : r->event_pool) (void)hipEventDestroy(e);
    for (void* p : r->scene_allocs) (void)hipFree(p);
    if (r->own_arena && r->arena) (void)hipFree(r->arena);
    if (r->extras) (void)hipFree(r->extras);
    if (r->d_counters) (void)hipFree(r->d_counters);
    free_queues(r);
    if (r->d_qcount) (void)hipFree(r->d_qcount);
    if (r->h_qseen) (void)hipHostFree(r->h_qseen);
#if FRT_EXPERIMENTS
    if (r->x.d_wf_counts) (void)hipFree(r->x.d_wf_counts);
    if (r->x.d_tiles) (void)hipFree(r->x.d_tiles);
    if (r->x.d_work) (void)hipFree(r->x.d_work);
#endif
    if (r->own_stream && r->stream) (void)hipStreamDestroy(r->stream);
    delete r;
}

#if FRT_EXPERIMENTS
static const size_t kWfCounterWords = 1024;                        // per stage: record counts per pass and region, item counts at +512
static const int kWorkSlots = 3 + kMaxCuts;                       // pixel launch (interior / whole), the two edge launches, one per continuation launch
static const size_t kWorkWords = 2 * kWorkSlots * 2;               // [stage][slot]{next, ticket}
#endif
static const size_t kQoverflowAt = 2 * 2 * (kMaxCuts + 1);      // (even: the pointers in the overflow blocks are 8-byte aligned)
static const size_t kQcountWords = kQoverflowAt + 2 * kOverflowBlockWords;   // [stage][parity][segment] counters + [stage] overflow blocks {count, pad, pointer to the mapped flag}
static bool stage_is_cut(const frt_renderer* r) { return r->ncuts > 0 && r->cuts[0] < r->max_depth && !(r->flags & FRT_FLAG_COMPACTION); }

// Continuation queues for `cap` parked paths per segment. T-trace parks 22 words per path, spatial 30 (its merged reservoir rides along).
// Zeroes the queue counters and (re)writes the overflow blocks' pointers to the mapped flag (ContQueue, frt_mono.hpp). On r->stream.
static int clear_queue_counters(frt_renderer* r) {
    HIP_TRY(hipMemsetAsync(r->d_qcount, 0, kQcountWords * sizeof(uint32_t), r->stream));
    if (r->d_qseen)
        for (int st = 0; st < 2; ++st)
            HIP_TRY(hipMemcpyAsync(r->d_qcount + kQoverflowAt + (size_t)st * kOverflowBlockWords + 2, &r->d_qseen, sizeof(uint32_t*), hipMemcpyHostToDevice, r->stream));
    if (r->h_qseen) *reinterpret_cast<volatile uint32_t*>(r->h_qseen) = 0u;
    return FRT_OK;
}
static int alloc_queues(frt_renderer* r, uint32_t cap) {
    free_queues(r);
    r->qcap = cap;
    if (!stage_is_cut(r) || cap == 0) return FRT_OK;
    r->qbytes = 0;
#if FRT_EXPERIMENTS
    if (r->x.wavefront) {
        r->qbytes = (uint64_t)2 * (2 * 44 + 2 * 2 + 7) * cap * sizeof(uint32_t);
        for (auto& q : r->qslots) q[0] = q[1] = cap;
        for (int st = 0; st < 2; ++st) {
            for (int k = 0; k < 2; ++k) {
                HIP_TRY(hipMalloc((void**)&r->x.d_wf_words[st][k], (size_t)44 * cap * sizeof(uint32_t)));
                HIP_TRY(hipMalloc((void**)&r->x.d_wf_items[st][k], (size_t)2 * cap * sizeof(uint32_t)));
            }
            HIP_TRY(hipMalloc((void**)&r->x.d_wf_hits[st], (size_t)7 * cap * sizeof(uint32_t)));
        }
        return FRT_OK;
    }
#endif
    // `cap` is the first buffer of the spatial stage. Measured per pixel on the Cornell Box (FRT_DEBUG_QUEUES; cuts 3 and 4): the spatial stage
    // parks 0.165 paths at the first cut and 0.10 at the second, T-trace 0.108 and 0.049. The other three buffers are sized in that
    // proportion (x 1.0 / 0.6 / 0.65 / 0.3 of `cap`, itself 0.25 per pixel by default); any of them may overflow (paths finish in place) and
    // they are grown together. A capacity given by the caller applies to all four.
    const int nbuf = r->ncuts >= 2 ? 2 : 1;
    static const double kShare[2][2] = {{0.65, 0.3}, {1.0, 0.6}};
    for (int st = 0; st < 2; ++st)
        for (int k = 0; k < 2; ++k) {
            r->qslots[st][k] = (r->qcap_fixed || cap >= r->qcap_max) ? cap : std::min(cap, std::max(4096u, (uint32_t)(kShare[st][k] * cap)));
            if (k >= nbuf) continue;
            const size_t bytes = (size_t)(st == 0 ? kContWordsPath : kContWordsSpatial) * r->qslots[st][k] * sizeof(uint32_t);
            HIP_TRY(hipMalloc((void**)&r->d_qwords[st][k], bytes));
            r->qbytes += bytes;
        }
    return FRT_OK;
}
#if FRT_EXPERIMENTS
static int init_tile_state(frt_renderer* r) {
    if (!r->x.d_tiles) return FRT_OK;
    uint32_t init[2 * kTileStateWords] = {0};
    init[0] = 1u; init[kTileStateWords] = 1u;   // first launch: bottom tile row first (floors cost more than ceilings and skies)
    HIP_TRY(hipMemcpyAsync(r->x.d_tiles, init, sizeof(init), hipMemcpyHostToDevice, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return FRT_OK;
}
// The experiment knobs and allocations of a renderer, in the two places renderer_init needs them: phase 0 inside the block that decides the cuts
// (before the queues are sized), phase 1 behind the scene upload.
static int exp_init(frt_renderer* r, int phase) {
    ExpState& x = r->x;
    if (phase == 0) {
        if (const char* e = getenv("FRT_CUTS")) {   // comma-separated ascending depths, "0" = never cut
            r->ncuts = 0;
            uint32_t last = 0;
            for (const char* p = e; *p && r->ncuts < (uint32_t)kMaxCuts;) {
                char* end = nullptr;
                const unsigned long v = strtoul(p, &end, 10);
                if (end == p) break;                                   // not a number: stop parsing (never loops on "abc" or "3;5")
                if (v >= 1 && v > last && v < 0xFFFFu) { r->cuts[r->ncuts++] = (uint32_t)v; last = (uint32_t)v; }   // ascending only; others are skipped
                p = end;
                if (*p == ',') ++p; else break;
            }
        }
        if (const char* e = getenv("FRT_WAVEFRONT")) {      // the ray-level wavefront needs the cut at depth 1
            x.wavefront = atoi(e) != 0 && !(r->flags & FRT_FLAG_COMPACTION) && r->max_depth > 1 && r->max_depth < 31;
            if (x.wavefront) { r->ncuts = 1; r->cuts[0] = 1; }
        }
        HIP_TRY(hipMalloc((void**)&x.d_wf_counts, 2 * kWfCounterWords * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(x.d_wf_counts, 0, 2 * kWfCounterWords * sizeof(uint32_t), r->stream));
        // Sweep direction of the tile rows (experiments/frt_experiment_kernels.hpp: TileOrder; resident pixel kernels only): FRT_TILE_ORDER=1
        if (const char* e = getenv("FRT_TILE_ORDER"); e && atoi(e) != 0) {
            HIP_TRY(hipMalloc((void**)&x.d_tiles, 2 * kTileStateWords * sizeof(uint32_t)));
            const int rc = init_tile_state(r);
            if (rc) return rc;
        }
        return FRT_OK;
    }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, r->device));
    x.num_cus = (uint32_t)prop.multiProcessorCount;
    resident_plan(r->sv, x.res_nodes, x.res_tris);
    if (const char* e = getenv("FRT_RESIDENT")) x.resident = atoi(e) != 0 && x.res_nodes > 0 && !(r->flags & FRT_FLAG_COMPACTION);
    if (const char* e = getenv("FRT_STREAM")) { x.stream_mode = atoi(e) != 0; if (atoi(e) > 1) x.shade_min = (uint32_t)std::min(64, atoi(e)); }   // n > 1 = shade_min
    if (const char* e = getenv("FRT_WG_PARK")) x.wg_park = atoi(e) != 0;
    if (const char* e = getenv("FRT_CONT_GRID")) x.cont_grid = std::max(0l, atol(e));
    if (const char* e = getenv("FRT_STREAM_SLICE")) x.stream_slice = (uint32_t)std::max(1, atoi(e));
    if (const char* e = getenv("FRT_REFILL")) { x.refill = atoi(e) != 0; if (atoi(e) > 1) x.refill_min = (uint32_t)std::min(64, atoi(e)); }   // 0 off, 1 on, n > 1: refill when >= n lanes are free
    if ((x.refill || x.stream_mode) && !getenv("FRT_CUTS") && r->ncuts > 1) r->ncuts = 1;   // (those kernels replace the continuation launches of a single cut)
    if (const char* e = getenv("FRT_RES_BATCH")) x.res_batch = (uint32_t)atoi(e);      // tiles per fetch (1, 2, 4)
    if (const char* e = getenv("FRT_RES_TRIS")) { if (atoi(e) == 0) x.res_tris = false; }   // triangles from L2
    HIP_TRY(hipMalloc((void**)&x.d_work, kWorkWords * sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(x.d_work, 0, kWorkWords * sizeof(uint32_t), r->stream));
    return FRT_OK;
}
// The experiment fields of a stage's launch description (frt_kernels.hpp: TraceLaunch).
static void exp_fill_launch(frt_renderer* r, int stage, bool with_tile_state, TraceLaunch& L, int work_slot, bool cut) {
    const ExpState& x = r->x;
    L.wg_park = x.wg_park;
    L.refill = x.refill; L.refill_min = x.refill_min; L.stream = x.stream_mode; L.shade_min = x.shade_min; L.slice = x.stream_slice;
    L.resident = x.resident; L.res_nodes = x.res_nodes; L.res_tris = x.res_tris; L.num_cus = x.num_cus; L.res_batch = x.res_batch;
    L.work = x.d_work + ((size_t)(stage - 1) * kWorkSlots + (size_t)work_slot) * 2;
    if (x.cont_grid >= 0) L.grid_min_slots = (uint32_t)x.cont_grid;   // (FRT_CONT_GRID, read at creation)
    L.tile_state = (with_tile_state && x.d_tiles) ? x.d_tiles + (size_t)(stage - 1) * kTileStateWords : nullptr;
    if (x.wavefront && cut) {
        L.wavefront = true;
        for (int k = 0; k < 2; ++k) { L.wf_words[k] = x.d_wf_words[stage - 1][k]; L.wf_items[k] = x.d_wf_items[stage - 1][k]; }
        L.wf_hits = x.d_wf_hits[stage - 1];
        L.qwords[0] = L.wf_words[0]; L.qwords[1] = nullptr;
        L.counts = x.d_wf_counts + (size_t)(stage - 1) * kWfCounterWords;      // cleared by the host before the stage's pixel launch
        L.zero_counts = nullptr;
        L.slice = x.stream_slice;
    }
}
#endif

static int renderer_init(frt_renderer* r, const frt_scene* s, const frt_render_opts* o) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(FRT_ERR_NO_DEVICE, "no HIP device: this library has no CPU rendering path");
    if (r->device < 0 || r->device >= ndev) return fail(FRT_ERR_INVALID_ARG, "device ordinal out of range");
    FRT_DEVICE(r);
    if (o && (o->stream || (o->flags & FRT_FLAG_USE_STREAM))) { r->stream = (hipStream_t)o->stream; r->own_stream = false; }
    else { HIP_TRY(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking)); r->own_stream = true; }
    if ((r->flags & FRT_FLAG_PIPELINE) && !(r->flags & FRT_FLAG_COMPACTION)) {   // (the compacting kernels keep the fused temporal stage: nothing to run ahead)
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        int prio = (lo + hi) / 2;
#if FRT_EXPERIMENTS
        if (const char* e = getenv("FRT_AHEAD_PRIO")) prio = !strcmp(e, "low") ? lo : (!strcmp(e, "high") ? hi : prio);
#endif
        HIP_TRY(hipStreamCreateWithPriority(&r->ahead, hipStreamNonBlocking, prio));
        HIP_TRY(hipStreamCreateWithPriority(&r->edge, hipStreamNonBlocking, prio));
        if (r->rb > 0 && r->re < r->H) HIP_TRY(hipStreamCreateWithPriority(&r->edge2, hipStreamNonBlocking, prio));   // a middle strip has two edges
        for (hipEvent_t* e : {&r->ev_spix, &r->ev_tail, &r->ev_tm[0], &r->ev_tm[1], &r->ev_edge, &r->ev_edge2, &r->ev_edge_ready}) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
        for (hipEvent_t& e : r->ev_tt) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
#if FRT_EXPERIMENTS
        if (const char* e = getenv("FRT_SPEC_DEPTH")) r->x.spec_depth = std::max(0, std::min(kSpecDepth, atoi(e)));
#endif
    }
    r->arena_bytes = arena_layout(r->W, r->H, r->off);
    // strips own a third G-buffer set; a whole-frame renderer only under FRT_FLAG_THIRD_GSET (include/frt.h)
    bool third = r->pipeline() && (!(r->rb == 0 && r->re == r->H) || (r->flags & FRT_FLAG_THIRD_GSET));
#if FRT_EXPERIMENTS
    if (getenv("FRT_NO_EXTRAS")) third = false;      // two sets for strips too
#endif
    if (third) {
        r->extras_bytes = arena_layout(r->W, r->H, r->off, true);
        HIP_TRY(hipMalloc((void**)&r->extras, r->extras_bytes));
        r->gsets = 3;
    }
    if (o && o->device_arena) {
        if (o->arena_bytes < r->arena_bytes) return fail(FRT_ERR_INVALID_ARG, "device_arena smaller than frt_renderer_arena_bytes");
        if (((uintptr_t)o->device_arena & 255u) != 0) return fail(FRT_ERR_INVALID_ARG, "device_arena must be 256-byte aligned");
        r->arena = (uint8_t*)o->device_arena; r->own_arena = false;
    } else {
        HIP_TRY(hipMalloc((void**)&r->arena, r->arena_bytes)); r->own_arena = true;
    }
    HIP_TRY(hipMalloc((void**)&r->d_counters, C_COUNT * sizeof(unsigned long long)));
    {   // the bounce depths at which paths are cut, and the continuation queues
        // Measured with the quad-tree kernels (tools/cut_sweep.sh, tools/strip_cuts.py): a 1080p frame 2.57 ms uncut, 2.08 cut at depth 3,
        // 1.97 at 3 and 5, 1.92 at 3 and 4; half a frame 1.41 uncut, 1.13 cut at 3, 1.09 at 3 and 4; a quarter 0.76 / 0.66 / 0.69; an
        // eighth 0.61 / 0.57 / 0.61. The second cut pays once the launch fills the chip several times over; a thin strip gets the first only.
        if ((size_t)r->W * (r->re - r->rb) < 800000u) r->ncuts = 1;
        // deep paths (MAX_DEPTH above the reference's 8): a third cut at 6 — configs[4]'s stand-in (MAX_DEPTH 16, 3840x2160) 21.7 -> 21.3 ms with cuts at
        // 3, 4, 6 (3, 4, 5, 7: 21.2; 3, 4, 6, 9: 21.4; 3, 6, 10: 22.0; tools/cuts_big.py)
        else if (r->max_depth > 8u) { r->ncuts = 3; r->cuts[2] = 6u; }
        // frt_render_opts.cut_depths: ascending depths (others are skipped); all zero = the choice above; first entry 0xFFFFFFFF = never cut
        if (o && o->cut_depths[0] != 0u) {
            r->ncuts = 0;
            uint32_t last = 0;
            for (int k = 0; k < kMaxCuts && o->cut_depths[0] != 0xFFFFFFFFu; ++k) {
                const uint32_t v = o->cut_depths[k];
                if (v >= 1u && v > last && v < 0xFFFFu) { r->cuts[r->ncuts++] = v; last = v; }
            }
        }
#if FRT_EXPERIMENTS
        { const int rc_ = exp_init(r, 0); if (rc_) return rc_; }      // FRT_CUTS, FRT_WAVEFRONT, FRT_TILE_ORDER: before the queues are sized
#endif
        r->qcap_max = r->W * std::min(r->H, (r->re - r->rb) + 2u * kHaloSpatial);   // every traced pixel parks
        // Default capacity from the share of paths that reach the first cut (Cornell Box, oracle counts per pixel: 0.65 / 0.50 / 0.11
        // alive at depth 1 / 2 / 3): generous, but not the worst case — a full queue is not an error (run_segment_and_park), and
        // frt_renderer_stats grows a queue that overflowed.
        const uint32_t c0 = r->ncuts ? r->cuts[0] : 0u;
        const double share = c0 >= 3 ? 0.25 : (c0 == 2 ? 0.6 : 0.8);
        uint32_t cap = (uint32_t)std::min<double>(r->qcap_max, std::max(4096.0, share * r->qcap_max));
        if (o && o->queue_capacity) { cap = std::min(o->queue_capacity, r->qcap_max); r->qcap_fixed = true; }
#if FRT_EXPERIMENTS
        if (const char* e = getenv("FRT_QUEUE_CAP")) { const long v = atol(e); if (v > 0) { cap = std::min<uint32_t>((uint32_t)v, r->qcap_max); r->qcap_fixed = true; } }
#endif
        int rc = alloc_queues(r, cap);
        if (rc) return rc;
        HIP_TRY(hipMalloc((void**)&r->d_qcount, kQcountWords * sizeof(uint32_t)));
        // (no mapped host memory -> no flag: the queues are then grown by frt_renderer_stats alone, as before)
        if (hipHostMalloc((void**)&r->h_qseen, sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
            *r->h_qseen = 0u;
            if (hipHostGetDevicePointer((void**)&r->d_qseen, r->h_qseen, 0) != hipSuccess) { (void)hipHostFree(r->h_qseen); r->h_qseen = nullptr; r->d_qseen = nullptr; }
        } else { r->h_qseen = nullptr; (void)hipGetLastError(); }
        { int rc_ = clear_queue_counters(r); if (rc_) return rc_; }
    }
    HIP_TRY(hipMemsetAsync(r->arena, 0, r->arena_bytes, r->stream));   // wgpu zero-initialises textures and buffers
    if (r->extras) HIP_TRY(hipMemsetAsync(r->extras, 0, r->extras_bytes, r->stream));
    HIP_TRY(hipMemsetAsync(r->d_counters, 0, C_COUNT * sizeof(unsigned long long), r->stream));
    int rc = upload_scene(r, s->b);
    if (rc) return rc;
#if FRT_EXPERIMENTS
    { const int rc_ = exp_init(r, 1); if (rc_) return rc_; }      // the experimental kernel forms' knobs and work counters
#endif
    HIP_TRY(hipStreamSynchronize(r->stream));
    return FRT_OK;
}

frt_renderer* frt_renderer_create(const frt_scene* s, uint32_t width, uint32_t height, const frt_render_opts* o) {
    if (!s || !s->b.built) { fail(FRT_ERR_STATE, "renderer_create: scene is not built"); return nullptr; }
    if (width == 0 || height == 0 || (uint64_t)width * height > 0x7FFFFFFFull) { fail(FRT_ERR_INVALID_ARG, "renderer_create: bad size"); return nullptr; }
    frt_renderer* r = new frt_renderer();
    r->W = width; r->H = height;
    r->max_depth = (o && o->max_depth) ? o->max_depth : 8u;
    r->device = o ? o->device : 0;
    r->flags = o ? o->flags : 0u;
    r->rb = 0; r->re = height;
    if (o && !(o->row_begin == 0 && o->row_end == 0)) {
        if (o->row_begin >= o->row_end || o->row_end > height) { fail(FRT_ERR_INVALID_ARG, "renderer_create: bad row range"); delete r; return nullptr; }
        r->rb = o->row_begin; r->re = o->row_end;
        r->motion_halo = o->motion_halo_rows;
    }
#if !FRT_EXPERIMENTS
    if (r->flags & (FRT_FLAG_COMPACTION | FRT_FLAG_WALK_WIDE | FRT_FLAG_WALK_WIDE_HBM | FRT_FLAG_WG_TRACE)) {
        fail(FRT_ERR_INVALID_ARG, "renderer_create: FRT_FLAG_COMPACTION / _WALK_WIDE / _WALK_WIDE_HBM / _WG_TRACE select experimental kernel families that live in lib/libfrt_exp.so (make experiments)");
        delete r; return nullptr;
    }
#endif
    if (renderer_init(r, s, o) != FRT_OK) { std::string keep = g_err; frt_renderer_destroy(r); g_err = keep; return nullptr; }
    return r;
}

// ------------------------------------------------------------------------------------------------ frame schedule
// Without FRT_FLAG_PIPELINE: one stream, G-buffer -> T-trace -> T-merge -> spatial -> post, in order.
// With it, two streams (and a third for a strip's edge rows):
//   main  : T-merge(f) | spatial pixels(f)               | spatial continuations(f) | post(f) | T-merge(f+1) ...
//   ahead : (behind T-merge(f))  G-buffer(f+1) | T-trace(f+1) pixels | T-trace(f+1) continuations
// T-trace depends on nothing of the previous frame (frt_path.hpp), so G-buffer + T-trace of frame f+1 are enqueued behind T-merge(f)
// and share the chip with the spatial stage of frame f: the two pixel kernels take turns on the wave slots and finish together, and
// the two latency-bound continuation tails, which leave most of the chip idle, then run side by side instead of one after the other
// (measured: 2.35 -> 2.20 ms per 1080p frame; started behind the spatial PIXEL kernel instead: 2.43; profiles/r2_schedule_notes.md).
// The next frame's camera is not known yet: the renderer SPECULATES that it is this frame's camera with frame_count + 1 and
// prev_view_proj = view_proj — what build_uniform produces for a camera that did not move (camera.rs:207-256, state.rs:172) — and
// only while the camera has in fact been static for a frame. At the next render call the speculated 288 bytes are compared with the
// real uniform: equal -> the work is adopted (its ray counts are committed by T-merge); different -> it is dropped (its buffers are
// simply overwritten, its counts cleared) and the stages run in order on the main stream. Same pixels either way.
// Buffer hazards: the speculated G-buffer goes to the physical set neither logical slot uses (three sets, alloc_g), whose last readers —
// spatial and post of the frame before last, and T-merge(f) through `prev` — are all on the main stream before the event the ahead
// stream waits for; the candidate and motion targets travel with the set; everything else is touched by the main stream only.
// work_slot: which pair of work counters a resident PIXEL launch uses (0 interior / whole stage, 1 and 2 the edge launches, which may
// run beside the interior one); the continuation launches use the pairs behind them.
static void trace_launch_of(frt_renderer* r, int stage, bool with_tile_state, TraceLaunch& L, int work_slot = 0) {
    memset(&L, 0, sizeof(L));
    L.wg_park = true;
    L.vote = r->vote;
    L.walk = r->walk; L.wide_lds_bytes = r->wide_lds_bytes; L.wg_rows = r->wg_rows;
    const bool cut = stage_is_cut(r) && r->qcap > 0;
    L.ncuts = cut ? r->ncuts : 0u;
    for (int k = 0; k < kMaxCuts; ++k) L.cuts[k] = r->cuts[k];
    L.qwords[0] = r->d_qwords[stage - 1][0]; L.qwords[1] = r->d_qwords[stage - 1][1];
    const uint32_t par = r->qparity[stage - 1];
    uint32_t* base = r->d_qcount + (size_t)(stage - 1) * 2 * (kMaxCuts + 1);
    L.counts = base + (size_t)par * (kMaxCuts + 1);
    L.zero_counts = cut ? base + (size_t)(par ^ 1u) * (kMaxCuts + 1) : nullptr;
    L.capacity = r->qslots[stage - 1][0]; L.capacity_odd = r->qslots[stage - 1][1];
    // Two streams: the spatial continuation launches (main stream) get a grid over half the stage's pixels, whatever the queue holds; the
    // surplus workgroups retire at once. Measured 1.97 -> 1.92 ms per 1080p frame (grid of 0.6 / 0.8 / 1.0 / 2 / 4 M slots: 1.98 / 1.95 /
    // 1.92 / 1.92 / 1.92; no effect on one stream, none for the T-trace launches on the ahead stream): while the grid is still being
    // dispatched the main stream keeps its turn at the dispatcher beside the next frame's T-trace pixel kernel (profiles/r2_schedule_notes.md).
    L.grid_min_slots = (stage == 2 && r->pipeline()) ? (uint32_t)std::min<uint64_t>(r->qcap_max, (uint64_t)r->W * (r->re - r->rb) / 2u) : 0u;
    L.overflow = r->d_qcount + kQoverflowAt + (size_t)(stage - 1) * kOverflowBlockWords;
#if FRT_EXPERIMENTS
    exp_fill_launch(r, stage, with_tile_state, L, work_slot, cut);
#else
    (void)with_tile_state; (void)work_slot;
#endif
}

// G-buffer + T-trace over their rows on stream `q`; pending_set >= 0: count the rays in that pending set (speculative work).
static int launch_g_and_trace(frt_renderer* r, FrameView fv, hipStream_t q, bool do_g, bool do_tt, int pending_set) {
    const bool pending = pending_set >= 0;
    uint32_t rows[8];
    phase_rows(r, rows);
    const bool timed = (r->flags & FRT_FLAG_TIMING) != 0;
    if (do_g) {
        fv.y0 = rows[0]; fv.y1 = rows[1];
        fv.ray_counters = r->d_counters + (pending ? C_PENDING + 4 * pending_set : C_STAGE);
        frt_renderer::Timed t{};
        if (timed) { int rc = timer_begin(r, t, 0, q); if (rc) return rc; }
        HIP_TRY(launch_gbuffer(r->sv, fv, q, r->walk == kWalkQuadWg ? (uint32_t)kWalkQuad : r->walk));      // (primary rays are coherent: plain walks)
        if (timed) { int rc = timer_end(r, t, q); if (rc) return rc; }
        r->stats.launches[0] += 1;
    }
    if (do_tt) {
        fv.y0 = rows[2]; fv.y1 = rows[3];
        fv.ray_counters = r->d_counters + (pending ? C_PENDING + 4 * pending_set + 2 : C_STAGE + 2);
        TraceLaunch L;
        trace_launch_of(r, 1, true, L);
        frt_renderer::Timed t{};
        if (timed) { int rc = timer_begin(r, t, 1, q); if (rc) return rc; }
#if FRT_EXPERIMENTS
        if (L.wavefront) HIP_TRY(hipMemsetAsync(L.counts, 0, kWfCounterWords * sizeof(uint32_t), q));
#endif
        HIP_TRY(launch_trace_pixels(1, r->sv, fv, q, L));
        if (trace_has_continuations(L, r->max_depth)) HIP_TRY(launch_trace_continuations(1, r->sv, fv, q, L));
        if (timed) { int rc = timer_end(r, t, q); if (rc) return rc; }
        r->qparity[0] ^= 1u;
        r->stats.launches[1] += 1;
    }
    return FRT_OK;
}

static bool same_camera(const frt_camera_uniform& a, const frt_camera_uniform& b) { return memcmp(&a, &b, sizeof(a)) == 0; }
// What build_uniform gives for the frame after `cam` when the camera does not move.
static frt_camera_uniform next_static_camera(const frt_camera_uniform& cam) {
    frt_camera_uniform n = cam;
    memcpy(n.prev_view_proj, cam.view_proj, sizeof(n.prev_view_proj));
    n.frame_count = cam.frame_count + 1u;
    return n;
}

// The physical sets of the G-buffer launch of frame `frame_count`: under the pipeline the set neither logical slot points at (three sets,
// two logical slots: exactly one is free) becomes the frame's slot; otherwise the reference's plain ping-pong.
static GSlots alloc_g(frt_renderer* r, uint32_t frame_count) {
    const uint32_t L = frame_count & 1u;
    GSlots gs;
    if (!r->pipeline()) { gs.g = L; gs.gprev = L ^ 1u; gs.aux = 0u; r->logical_phys[0] = 0u; r->logical_phys[1] = 1u; return gs; }
    gs.g = r->logical_phys[L];      // two sets: the logical slot's own set (its frame's readers are done); more: a set neither slot points at
    for (uint32_t p = 0; p < r->gsets; ++p)
        if (p != r->logical_phys[0] && p != r->logical_phys[1]) { gs.g = p; break; }
    gs.gprev = r->logical_phys[L ^ 1u];
    gs.aux = gs.g;
    r->logical_phys[L] = gs.g;
    return gs;
}

// Queues that overflowed since the last look (paths were finished in place, nothing was lost) are doubled. The device must be idle (sync_all):
// called from frt_renderer_stats, and from the first phase of a frame when a wave has raised the mapped flag (ContQueue::seen), so that a host
// that never asks for statistics still gets queues that fit its scene — the default shares are the Cornell Box's.
static int grow_queues_if_overflowed(frt_renderer* r) {
    if (!r->d_qcount) return FRT_OK;
    uint32_t blk[2 * kOverflowBlockWords] = {0};
    HIP_TRY(hipMemcpy(blk, r->d_qcount + kQoverflowAt, sizeof(blk), hipMemcpyDeviceToHost));
    const uint32_t ov[2] = {blk[0], blk[kOverflowBlockWords]};
#if FRT_EXPERIMENTS
    if (getenv("FRT_DEBUG_QUEUES")) {   // (debug: the queue counters of the last launches, [stage][parity][segment])
        uint32_t qc[kQcountWords];
        HIP_TRY(hipMemcpy(qc, r->d_qcount, sizeof(qc), hipMemcpyDeviceToHost));
        fprintf(stderr, "queues slots T %u %u S %u %u:", r->qslots[0][0], r->qslots[0][1], r->qslots[1][0], r->qslots[1][1]);
        for (size_t i = 0; i < kQcountWords; ++i) fprintf(stderr, " %u", qc[i]);
        fprintf(stderr, "\n");
    }
#endif
    if (r->h_qseen) *reinterpret_cast<volatile uint32_t*>(r->h_qseen) = 0u;
    if (ov[0] || ov[1]) {
        r->stats.queue_overflow += (uint64_t)ov[0] + ov[1];
        for (int st = 0; st < 2; ++st) HIP_TRY(hipMemset(r->d_qcount + kQoverflowAt + (size_t)st * kOverflowBlockWords, 0, sizeof(uint32_t)));
        if (!r->qcap_fixed && r->qcap < r->qcap_max) {
            const int rc = alloc_queues(r, (uint32_t)std::min<uint64_t>(r->qcap_max, (uint64_t)r->qcap * 2u));
            if (rc) return rc;
        }
    }
    return FRT_OK;
}
static int open_frame(frt_renderer* r, const frt_camera_uniform* cam) {
    // a wave of an earlier frame found its continuation queue full: grow the queues now, between two frames (one synchronisation, at most twice in
    // a renderer's life: 0.25 -> 0.5 -> 1 slot per pixel)
    if (r->h_qseen && *reinterpret_cast<volatile uint32_t*>(r->h_qseen) != 0u && !r->qcap_fixed && r->qcap < r->qcap_max) {
        int rc = sync_all(r);
        if (rc) return rc;
        rc = grow_queues_if_overflowed(r);
        if (rc) return rc;
    }
    r->frame_open = true;
    r->cur_cam = *cam;
    r->g_done = r->tt_done = r->tm_done = r->s_started = r->s_inner_done = r->s_edge_done = false;
    r->from_speculation = false;
    r->camera_static = r->have_last_cam && same_camera(next_static_camera(r->last_cam), *cam);
    if (!r->specs.empty()) {
        const frt_renderer::Spec sp = r->specs.front();
        if (same_camera(sp.cam, *cam) && sp.frame == r->frame_count) {
            r->specs.erase(r->specs.begin());
            r->g_done = r->tt_done = r->from_speculation = true;      // adopted: T-merge waits for its ev_tt and commits the ray counts
            r->cur_slots = sp.slots;
            r->cur_spec_idx = sp.idx;
            r->stats.speculated_frames += 1;
        } else {
            // dropped, with everything speculated behind it: order the main stream behind the work, clear its ray counts, give the physical
            // sets back; the buffers it wrote are simply overwritten by the stages that follow
            int rc = fence_ahead(r);
            if (rc) return rc;
            HIP_TRY(hipMemsetAsync(r->d_counters + C_PENDING, 0, 4 * kPending * sizeof(unsigned long long), r->stream));
            r->logical_phys[0] = sp.logical_before[0]; r->logical_phys[1] = sp.logical_before[1];
            r->stats.discarded_speculations += r->specs.size();
            r->specs.clear();
        }
    }
    return FRT_OK;
}

// G-buffer + T-trace of the NEXT frames on the ahead stream, under the cameras a static scene will present: up to kSpecDepth frames ahead,
// so that the frame two ahead — due only after a whole further frame — is the bulk work that fills the latency-bound tails of this frame's
// spatial stage and of the next frame's T-trace. Only for a camera that has been static for a frame (a moving camera never matches).
// Ordered behind T-merge of this frame, the last reader of the physical set the launch overwrites (the G-buffer two frames back: read
// by that frame's spatial and post stages and, as `prev`, by this T-merge — all on the main stream before the event).
static int launch_speculation(frt_renderer* r, const frt_camera_uniform* cam) {
    if (!r->pipeline() || !r->camera_static || !r->tm_done) return FRT_OK;
#if FRT_EXPERIMENTS
    const int spec_depth = r->x.spec_depth;
#else
    const int spec_depth = kSpecDepth;
#endif
    while ((int)r->specs.size() < spec_depth) {
        frt_renderer::Spec sp;
        sp.cam = next_static_camera(r->specs.empty() ? *cam : r->specs.back().cam);
        sp.frame = (r->specs.empty() ? r->frame_count : r->specs.back().frame) + 1u;
        sp.logical_before[0] = r->logical_phys[0]; sp.logical_before[1] = r->logical_phys[1];
        sp.slots = alloc_g(r, sp.frame);
        sp.idx = r->spec_next_idx; r->spec_next_idx = (r->spec_next_idx + 1) % kPending;
        FrameView fa;
        fill_frame_view(r, &sp.cam, sp.frame, sp.slots, fa);
        // two sets: the launch overwrites the set this frame's T-merge reads as `prev` -> behind this T-merge; three sets: it overwrites the set of
        // the frame before that, whose last reader is the PREVIOUS frame's T-merge (and post) -> a whole frame of slack for the ahead stream
        // (only when this frame's own T-trace ran on the ahead stream too: a T-trace on the main stream shares the stage's queues and counters
        // with the launch behind it and is ordered by this frame's T-merge event)
        const bool relaxed = r->gsets >= 3 && r->from_speculation;
        HIP_TRY(hipStreamWaitEvent(r->ahead, r->ev_tm[(r->serial + (relaxed ? 1u : 0u)) & 1u], 0));
        int rc = launch_g_and_trace(r, fa, r->ahead, true, true, sp.idx);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(r->ev_tt[sp.idx], r->ahead));
        r->tail_pending = true;
        r->specs.push_back(sp);
    }
    return FRT_OK;
}

static int render_phases_impl(frt_renderer* r, const frt_camera_uniform* cam, int phases);
int frt_renderer_render_phases(frt_renderer* r, const frt_camera_uniform* cam, int phases) {
    if (!r || !cam) return fail(FRT_ERR_INVALID_ARG, "render: null");
    if (r->failed) return fail(FRT_ERR_STATE, "render: an earlier frame failed in the middle of its stages; call frt_renderer_clear");
    const int rc = render_phases_impl(r, cam, phases);
    if (rc == FRT_ERR_HIP) { r->failed = true; r->frame_open = false; }      // (argument errors are raised before anything is enqueued)
    return rc;
}
static int render_phases_impl(frt_renderer* r, const frt_camera_uniform* cam, int phases) {
    if ((r->jitter[0] != 0.0f || r->jitter[1] != 0.0f) && !(r->rb == 0 && r->re == r->H) && (phases & FRT_PHASE_POST))
        return fail(FRT_ERR_INVALID_ARG, "render: a non-zero post jitter (bilinear taps with Repeat addressing) is not supported by strip renderers");
    FRT_DEVICE(r);
    if (!r->frame_open) { int rc = open_frame(r, cam); if (rc) return rc; }
    const bool timed = (r->flags & FRT_FLAG_TIMING) != 0;
    const bool compaction = (r->flags & FRT_FLAG_COMPACTION) != 0;
    uint32_t rows[8];
    phase_rows(r, rows);
    FrameView fv;

    // ---- G-buffer, T-trace: on the main stream unless they already ran ahead of the frame
    const bool want_g = (phases & FRT_PHASE_GBUFFER) && !r->g_done;
    const bool want_tt = (phases & FRT_PHASE_TEMPORAL) && !r->tt_done && !compaction;
    if (want_g || want_tt) {
        int rc = fence_ahead(r);      // (a dropped speculation may still be writing the slots these stages write)
        if (rc) return rc;
        if (want_g) r->cur_slots = alloc_g(r, r->frame_count);
        fill_frame_view(r, cam, r->frame_count, r->cur_slots, fv);
        rc = launch_g_and_trace(r, fv, r->stream, want_g, want_tt, -1);
        if (rc) return rc;
        if (want_g) r->g_done = true;
        if (want_tt) r->tt_done = true;
    }
    // ---- T-merge (or the fused temporal stage of the compacting kernels)
    if ((phases & FRT_PHASE_TEMPORAL) && !r->tm_done) {
        fill_frame_view(r, cam, r->frame_count, r->cur_slots, fv);
        fv.y0 = rows[2]; fv.y1 = rows[3];
        if (compaction) {
            fv.ray_counters = r->d_counters + C_STAGE + 2;
            frt_renderer::Timed t{};
            if (timed) { int rc = timer_begin(r, t, 1, r->stream); if (rc) return rc; }
            HIP_TRY(launch_compact(1, r->sv, fv, r->stream));
            if (timed) { int rc = timer_end(r, t, r->stream); if (rc) return rc; }
            r->stats.launches[1] += 1;
        } else {
            if (r->from_speculation) HIP_TRY(hipStreamWaitEvent(r->stream, r->ev_tt[r->cur_spec_idx], 0));
            frt_renderer::Timed t{};
            if (timed) { int rc = timer_begin(r, t, 4, r->stream); if (rc) return rc; }
            HIP_TRY(launch_merge(r->sv, fv, r->stream, r->from_speculation ? r->d_counters + C_PENDING + 4 * r->cur_spec_idx : nullptr, r->d_counters + C_STAGE));
            if (timed) { int rc = timer_end(r, t, r->stream); if (rc) return rc; }
            if (r->pipeline()) HIP_TRY(hipEventRecord(r->ev_tm[r->serial & 1u], r->stream));
        }
        r->tm_done = true;
    }
    // ---- spatial + shade: pixels (whole, or interior / edge rows separately), then the continuations
    const int sp = ((phases & FRT_PHASE_SPATIAL) ? (FRT_PHASE_SPATIAL_INNER | FRT_PHASE_SPATIAL_EDGE) : 0) | (phases & (FRT_PHASE_SPATIAL_INNER | FRT_PHASE_SPATIAL_EDGE));
    if (sp) {
        if (!r->s_started) {
            r->s_started = true;
            { int rc = launch_speculation(r, cam); if (rc) return rc; }
            r->s_timed = false;
            if (timed) { int rc = timer_begin(r, r->s_timer, 2, r->stream); if (rc) return rc; r->s_timed = true; }
        }
        fill_frame_view(r, cam, r->frame_count, r->cur_slots, fv);
        fv.ray_counters = r->d_counters + C_STAGE + 4;
        const uint32_t y0 = rows[4], y1 = rows[5];
        const bool was_complete = r->s_inner_done && r->s_edge_done;
        TraceLaunch L;
        memset(&L, 0, sizeof(L));
        if (compaction) {
            if (!r->s_inner_done) { fv.y0 = y0; fv.y1 = y1; HIP_TRY(launch_compact(2, r->sv, fv, r->stream)); }
            r->s_inner_done = r->s_edge_done = true;
        } else {
            uint32_t ia, ib;
            spatial_inner_rows(r, y0, y1, ia, ib);
            if ((sp & FRT_PHASE_SPATIAL_INNER) && !r->s_inner_done) {
                trace_launch_of(r, 2, true, L);
#if FRT_EXPERIMENTS
                if (L.wavefront) HIP_TRY(hipMemsetAsync(L.counts, 0, kWfCounterWords * sizeof(uint32_t), r->stream));   // (whole-frame renderers: the interior launch is the first of the stage)
#endif
                fv.y0 = ia; fv.y1 = ib;
                HIP_TRY(launch_trace_pixels(2, r->sv, fv, r->stream, L));
                r->s_inner_done = true;
            }
            if ((sp & FRT_PHASE_SPATIAL_EDGE) && !r->s_edge_done) {
                trace_launch_of(r, 2, false, L);
                uint32_t* const zc = L.zero_counts;
                bool need_clear = ia >= ib;        // otherwise the interior launch clears the next launch's counters
                const uint32_t edge[2][2] = {{y0, ia}, {ib, y1}};
                // Pipeline: on their own stream, beside the interior launch (two launches in one stream would run one after the other,
                // and a thin strip's launch lasts as long as its longest path whatever its size). The caller has ordered that stream behind
                // the halo exchange (frt_renderer_stream(r, 2)); here it is ordered behind T-merge.
                hipStream_t q = r->stream;
                const bool any_edge = (ia > y0) || (y1 > ib);
                // ALWAYS the edge stream when the renderer has one, also for a strip too thin to have interior rows: frt_renderer_stream(r, 2) names
                // that stream unconditionally, and it is the only one the caller orders behind the arrival of the halo rows.
                if (r->edge && any_edge) { q = r->edge; HIP_TRY(hipStreamWaitEvent(r->edge, r->ev_tm[r->serial & 1u], 0)); }
                // A middle strip has two edges: the second launch goes to a stream of its own, ordered behind everything the edge stream holds
                // so far (T-merge, the caller's exchange), so that the two run side by side (one after the other they cost a 1/8 strip 0.12 ms
                // of its 0.52 ms frame: the spatial continuation waits for both).
                const bool two_edges = q == r->edge && r->edge2 && ia > y0 && y1 > ib;
                if (two_edges) { HIP_TRY(hipEventRecord(r->ev_edge_ready, r->edge)); HIP_TRY(hipStreamWaitEvent(r->edge2, r->ev_edge_ready, 0)); }
                int slot = 1;
                for (const auto& e : edge) {
                    const int this_slot = slot++;
                    if (e[1] <= e[0]) continue;
                    L.zero_counts = need_clear ? zc : nullptr;
                    need_clear = false;
#if FRT_EXPERIMENTS
                    L.work = r->x.d_work + ((size_t)kWorkSlots + (size_t)this_slot) * 2;     // stage 2, its own counters (resident kernels)
#else
                    (void)this_slot;
#endif
                    fv.y0 = e[0]; fv.y1 = e[1];
                    HIP_TRY(launch_trace_pixels(2, r->sv, fv, (two_edges && this_slot == 2) ? r->edge2 : q, L));
                }
                if (q != r->stream) { HIP_TRY(hipEventRecord(r->ev_edge, r->edge)); r->edge_in_flight = true; }
                if (two_edges) { HIP_TRY(hipEventRecord(r->ev_edge2, r->edge2)); r->edge2_in_flight = true; }
                r->s_edge_done = true;
            }
        }
        if (r->s_inner_done && r->s_edge_done && !was_complete) {
            if (r->edge_in_flight) { HIP_TRY(hipStreamWaitEvent(r->stream, r->ev_edge, 0)); r->edge_in_flight = false; }
            if (r->edge2_in_flight) { HIP_TRY(hipStreamWaitEvent(r->stream, r->ev_edge2, 0)); r->edge2_in_flight = false; }
            if (!compaction) {
                trace_launch_of(r, 2, false, L);
                fv.y0 = y0; fv.y1 = y1;
                if (trace_has_continuations(L, r->max_depth)) HIP_TRY(launch_trace_continuations(2, r->sv, fv, r->stream, L));
                r->qparity[1] ^= 1u;
            }
            if (r->s_timed) { int rc = timer_end(r, r->s_timer, r->stream); if (rc) return rc; r->s_timed = false; }
            r->stats.launches[2] += 1;
        }
    }
    // ---- post / accumulate (main stream: it runs while the ahead stream is in the latency-bound tail of the next frame's T-trace)
    if (phases & FRT_PHASE_POST) {
        fill_frame_view(r, cam, r->frame_count, r->cur_slots, fv);
        fv.y0 = rows[6]; fv.y1 = rows[7];
        frt_renderer::Timed t{};
        if (timed) { int rc = timer_begin(r, t, 3, r->stream); if (rc) return rc; }
        HIP_TRY(launch_post(fv, r->stream));
        if (timed) { int rc = timer_end(r, t, r->stream); if (rc) return rc; }
        r->stats.launches[3] += 1;
    }
    return FRT_OK;
}
int frt_renderer_end_frame(frt_renderer* r) {
    if (!r) return fail(FRT_ERR_INVALID_ARG, "end_frame: null");
    r->frame_count += 1;   // renderer.rs:515
    r->serial += 1;
    r->stats.frames += 1;
    if (r->frame_open) { r->last_cam = r->cur_cam; r->have_last_cam = true; r->before_last_slots = r->last_slots; r->last_slots = r->cur_slots; }
    r->frame_open = false;
    return FRT_OK;
}
int frt_renderer_render(frt_renderer* r, const frt_camera_uniform* cam) {
    int rc = frt_renderer_render_phases(r, cam, FRT_PHASE_ALL);
    if (rc) return rc;
    return frt_renderer_end_frame(r);
}
int frt_renderer_set_jitter(frt_renderer* r, float jx, float jy) {
    if (!r) return fail(FRT_ERR_INVALID_ARG, "set_jitter: null");
    r->jitter[0] = jx; r->jitter[1] = jy;
    return FRT_OK;
}
int frt_renderer_render_jittered(frt_renderer* r, const frt_camera_uniform* cam, float jx, float jy) {
    int rc = frt_renderer_set_jitter(r, jx, jy);
    if (rc) return rc;
    return frt_renderer_render(r, cam);
}
int frt_renderer_sync(frt_renderer* r) {
    if (!r) return fail(FRT_ERR_INVALID_ARG, "sync: null");
    FRT_DEVICE(r);
    return sync_all(r);
}
int frt_renderer_fence(frt_renderer* r) {
    if (!r) return fail(FRT_ERR_INVALID_ARG, "fence: null");
    FRT_DEVICE(r);
    return fence_ahead(r);
}
int frt_renderer_order_edge_stream(frt_renderer* r) {
    if (!r) return fail(FRT_ERR_INVALID_ARG, "order_edge_stream: null");
    if (!r->edge) return FRT_OK;
    if (!r->frame_open || !r->tm_done) return fail(FRT_ERR_STATE, "order_edge_stream: call it after FRT_PHASE_TEMPORAL of the open frame");
    FRT_DEVICE(r);
    HIP_TRY(hipStreamWaitEvent(r->edge, r->ev_tm[r->serial & 1u], 0));
    return FRT_OK;
}
void* frt_renderer_stream(const frt_renderer* r, int which) {
    if (!r) return nullptr;
    if (which == 1 && r->ahead) return (void*)r->ahead;
    if (which == 2 && r->edge) return (void*)r->edge;
    return (void*)r->stream;
}
uint32_t frt_renderer_frame_count(const frt_renderer* r) { return r ? r->frame_count : 0u; }
int frt_renderer_reset(frt_renderer* r) {
    if (!r) return fail(FRT_ERR_INVALID_ARG, "reset: null");
    // Only the counter restarts (state.rs:152: every frame while the camera moves), asynchronously. A speculated next frame no longer
    // matches (its frame_count) and is dropped by the next render call, which also orders the main stream behind the ahead stream.
    r->frame_count = 0;
    r->frame_open = false;
    return FRT_OK;
}
int frt_renderer_clear(frt_renderer* r) {
    if (!r) return fail(FRT_ERR_INVALID_ARG, "clear: null");
    FRT_DEVICE(r);
    { int rc_ = sync_all(r); if (rc_) return rc_; }
    int rc = resolve_timing(r);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(r->arena, 0, r->arena_bytes, r->stream));
    if (r->extras) HIP_TRY(hipMemsetAsync(r->extras, 0, r->extras_bytes, r->stream));
    HIP_TRY(hipMemsetAsync(r->d_counters, 0, C_COUNT * sizeof(unsigned long long), r->stream));
    { int rc_ = clear_queue_counters(r); if (rc_) return rc_; }
#if FRT_EXPERIMENTS
    HIP_TRY(hipMemsetAsync(r->x.d_work, 0, kWorkWords * sizeof(uint32_t), r->stream));
#endif
    HIP_TRY(hipStreamSynchronize(r->stream));
#if FRT_EXPERIMENTS
    rc = init_tile_state(r);
    if (rc) return rc;
#endif
    r->frame_count = 0;
    r->failed = false;
    r->frame_open = false; r->specs.clear(); r->have_last_cam = false; r->camera_static = false;
    r->qparity[0] = r->qparity[1] = 0; r->logical_phys[0] = 0; r->logical_phys[1] = 1;
    r->cur_slots = r->last_slots = GSlots{0, 1, 0}; r->before_last_slots = GSlots{1, 0, 0};
    memset(&r->stats, 0, sizeof(r->stats));
    return FRT_OK;
}

// Reads through the ABI see the reference's two logical slots as of the last finished frame F: slot F % 2 is that frame's G-buffer, the other
// one what it read as `prev` (work running ahead writes a third physical set and is invisible here).
static int buf_index(const frt_renderer* r, int buf, int index) {
    const uint32_t last_parity = r->frame_count ? ((r->frame_count - 1u) & 1u) : 0u;
    const uint32_t g = ((uint32_t)index & 1u) == last_parity ? r->last_slots.g : r->last_slots.gprev;
    switch (buf) {
    case FRT_BUF_GPOS: return B_GPOS0 + (int)g;
    case FRT_BUF_GNORMAL: return B_GNRM0 + (int)g;
    case FRT_BUF_GALBEDO: return B_GALB0 + (int)g;
    case FRT_BUF_GMOTION: return B_GMOT0 + (int)((index & 1) ? r->before_last_slots.aux : r->last_slots.aux);   // 0 = the last rendered frame's (the reference has one)
    case FRT_BUF_RESERVOIR: return B_RES0 + (index & 1);
    case FRT_BUF_RAW: return B_RAW;
    case FRT_BUF_DISPLAY: return B_DISP;
    case FRT_BUF_ACCUM: return B_ACC0 + (index & 1);
    case FRT_BUF_CANDIDATE: return B_CAND0 + (int)r->last_slots.aux;
    }
    return -1;
}
int frt_renderer_buffer_info(const frt_renderer* r, int buf, int index, void** device_ptr, uint32_t* bpp) {
    int b = r ? buf_index(r, buf, index) : -1;
    if (b < 0) return fail(FRT_ERR_INVALID_ARG, "buffer_info: bad buffer");
    if (device_ptr) *device_ptr = r->buf(b);
    if (bpp) *bpp = bpp_of(b);
    return FRT_OK;
}
int frt_renderer_read_buffer(frt_renderer* r, int buf, int index, void* out) {
    int b = r ? buf_index(r, buf, index) : -1;
    if (b < 0 || !out) return fail(FRT_ERR_INVALID_ARG, "read_buffer: bad arguments");
    FRT_DEVICE(r);
    { int rc_ = sync_all(r); if (rc_) return rc_; }
    HIP_TRY(hipMemcpy(out, r->buf(b), (size_t)r->W * r->H * bpp_of(b), hipMemcpyDeviceToHost));
    return FRT_OK;
}
int frt_renderer_read_rows(frt_renderer* r, int buf, int index, uint32_t y0, uint32_t y1, void* out) {
    int b = r ? buf_index(r, buf, index) : -1;
    if (b < 0 || !out || y0 > y1 || y1 > r->H) return fail(FRT_ERR_INVALID_ARG, "read_rows: bad arguments");
    FRT_DEVICE(r);
    { int rc_ = sync_all(r); if (rc_) return rc_; }
    size_t pitch = (size_t)r->W * bpp_of(b);
    HIP_TRY(hipMemcpy(out, (uint8_t*)r->buf(b) + pitch * y0, pitch * (y1 - y0), hipMemcpyDeviceToHost));
    return FRT_OK;
}
int frt_renderer_write_rows(frt_renderer* r, int buf, int index, uint32_t y0, uint32_t y1, const void* in) {
    int b = r ? buf_index(r, buf, index) : -1;
    if (b < 0 || !in || y0 > y1 || y1 > r->H) return fail(FRT_ERR_INVALID_ARG, "write_rows: bad arguments");
    FRT_DEVICE(r);
    { int rc_ = sync_all(r); if (rc_) return rc_; }
    size_t pitch = (size_t)r->W * bpp_of(b);
    HIP_TRY(hipMemcpy((uint8_t*)r->buf(b) + pitch * y0, in, pitch * (y1 - y0), hipMemcpyHostToDevice));
    return FRT_OK;
}
int frt_renderer_read_display(frt_renderer* r, uint8_t* rgba8) { return frt_renderer_read_buffer(r, FRT_BUF_DISPLAY, 0, rgba8); }
int frt_renderer_read_accum(frt_renderer* r, float* rgba32f) {
    if (!r) return fail(FRT_ERR_INVALID_ARG, "read_accum: null");
    uint32_t last = r->frame_count ? (r->frame_count - 1u) & 1u : 0u;   // slot written by the last rendered frame
    return frt_renderer_read_buffer(r, FRT_BUF_ACCUM, (int)last, rgba32f);
}
int frt_renderer_phase_rows(const frt_renderer* r, uint32_t out[8]) {
    if (!r || !out) return fail(FRT_ERR_INVALID_ARG, "phase_rows: null");
    phase_rows(r, out);
    return FRT_OK;
}
int frt_renderer_stats(frt_renderer* r, frt_stats* out) {
    if (!r || !out) return fail(FRT_ERR_INVALID_ARG, "stats: null");
    FRT_DEVICE(r);
    { int rc_ = sync_all(r); if (rc_) return rc_; }
    int rc = resolve_timing(r);
    if (rc) return rc;
    unsigned long long c[C_COUNT] = {0};
    HIP_TRY(hipMemcpy(c, r->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    r->stats.halo_overflow = c[C_HALO];
    r->stats.rays_closest = 0; r->stats.rays_any = 0;
    for (int st = 0; st < 4; ++st) {
        r->stats.rays_stage[st][0] = c[2 * st]; r->stats.rays_stage[st][1] = c[2 * st + 1];
        r->stats.rays_closest += c[2 * st]; r->stats.rays_any += c[2 * st + 1];
    }
    rc = grow_queues_if_overflowed(r);
    if (rc) return rc;
    r->stats.queue_capacity = r->qcap;
    r->stats.queue_bytes = r->qbytes;
    *out = r->stats;
    return FRT_OK;
}

int frt_renderer_set_timing(frt_renderer* r, int on) {
    if (!r) return fail(FRT_ERR_INVALID_ARG, "set_timing: null");
    if (on) r->flags |= FRT_FLAG_TIMING; else r->flags &= ~FRT_FLAG_TIMING;
    return FRT_OK;
}

} // extern "C"
