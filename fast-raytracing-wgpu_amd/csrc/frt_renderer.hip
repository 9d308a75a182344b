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

using namespace frt;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
namespace frt { int set_error(int code, const std::string& msg) { return fail(code, msg); } }   // for the other translation units of the ABI
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(FRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

static_assert(sizeof(frt_vertex_attr) == 32 && sizeof(frt_material) == 64 && sizeof(frt_light) == 64, "ABI struct sizes");
static_assert(sizeof(frt_camera_uniform) == 288 && sizeof(frt_reservoir) == 32 && sizeof(frt_bvh2_node) == 32, "ABI struct sizes");
static_assert(sizeof(CameraView) == 288 && sizeof(ReservoirView) == 32 && sizeof(InstanceView) == 64 && sizeof(InstanceDev) == 64, "view sizes");
static_assert(sizeof(MaterialView) == 64 && sizeof(LightView) == 64 && sizeof(VertexAttrView) == 32 && sizeof(MeshInfoView) == 16, "view sizes");
static_assert(sizeof(PairNode) == 64 && sizeof(TriSlot) == 48 && sizeof(ShadeTri) == 128, "GPU layout sizes");

// ------------------------------------------------------------------------------------------------ renderer object
static const uint32_t kHaloGbuffer = 12;   // spatial reuse radius 10 (restir_spatial.wgsl:903, :921) + spatial halo 2
static const uint32_t kHaloSpatial = 2;    // post reads raw radiance within +-2 rows (post.wgsl:93)

enum { B_GPOS0, B_GPOS1, B_GNRM0, B_GNRM1, B_GALB0, B_GALB1, B_GMOT, B_RES0, B_RES1, B_RAW, B_DISP, B_ACC0, B_ACC1, B_GMOT1, B_COUNT };
static const uint32_t kBpp[B_COUNT] = {16, 16, 16, 16, 4, 4, 8, 32, 32, 8, 4, 16, 16, 8};

struct frt_renderer {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t side = nullptr;            // FRT_FLAG_OVERLAP_POST: the post stage runs here
    hipEvent_t ev_spatial = nullptr, ev_post = nullptr, ev_smain = nullptr, ev_scont = nullptr, ev_tmain = nullptr;
    bool post_in_flight = false, scont_in_flight = false;
    uint32_t motion_slot = 0;             // which motion buffer the last G-buffer stage wrote (ping-pong under the side-stream schedule)
    uint32_t W = 0, H = 0, max_depth = 8, rb = 0, re = 0, flags = 0, motion_halo = 0;
    uint32_t frame_count = 0;
    SceneView sv{};
    std::vector<void*> scene_allocs;
    uint8_t* arena = nullptr;
    bool own_arena = false;
    size_t arena_bytes = 0;
    size_t off[B_COUNT] = {};
    unsigned long long* d_counters = nullptr;
    uint32_t* d_qwords = nullptr;          // continuation queues: [stage 1|2][A|B] x kContWordsSpatial x qcap words
    uint32_t qcap = 0;                     // slots per queue = the pixels this renderer traces (its rows + the spatial halo)
    uint32_t* d_qcount = nullptr;          // per stage, one counter per path segment (kMaxCuts + 1)
    uint32_t ncuts = 1, cuts[kMaxCuts] = {3, 0, 0, 0};   // measured best on the Cornell Box (sweep over 0 / 2 / 3 / 4 / 5 and multi-cut sets)
    bool pair_tail = false;               // last path segment through the two-wave kernel (continue_pair_kernel)
    bool counts_clean[2] = {false, false};   // stage's queue counters already cleared by a kernel of the other stage (no memset needed)
    uint32_t* d_tiles = nullptr;           // per traced stage: tile-row order [n] + tile-row cost [n] (frt_kernels.hip: TileOrder)
    uint32_t ntiles[2] = {0, 0};           // n = tile rows of the stage
    frt_stats stats{};
    struct Timed { hipEvent_t a, b; int stage; };
    std::vector<Timed> pending;
    bool post_deferred = false;           // side-stream schedule: post(f) is launched behind the temporal pixel kernel of frame f+1
    FrameView post_fv;                    // (or at the next sync / read), so that it fills the latency-bound temporal continuation
    void* buf(int b) const { return arena + off[b]; }
};

static size_t arena_layout(uint32_t W, uint32_t H, size_t off[B_COUNT]) {
    size_t n = (size_t)W * H, cur = 0;
    for (int b = 0; b < B_COUNT; ++b) {
        if (off) off[b] = cur;
        cur += (n * kBpp[b] + 255u) & ~(size_t)255u;
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

static int upload_scene(frt_renderer* r, const SceneBuilder& b) {
    SceneView& sv = r->sv;
    int rc;
    if ((rc = upload(r, b.pair_nodes, &sv.nodes))) return rc;
    if ((rc = upload(r, b.tri_slots, &sv.tris))) return rc;
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

static void fill_frame_view(const frt_renderer* r, const frt_camera_uniform* cam, FrameView& fv) {
    uint32_t cur = r->frame_count & 1u, prv = cur ^ 1u;   // gbuffer.rs:299, restir.rs:543, post.rs:244
    fv.gpos = (float4*)r->buf(B_GPOS0 + cur); fv.gnormal = (float4*)r->buf(B_GNRM0 + cur); fv.galbedo = (uint32_t*)r->buf(B_GALB0 + cur);
    fv.gpos_prev = (const float4*)r->buf(B_GPOS0 + prv); fv.gnormal_prev = (const float4*)r->buf(B_GNRM0 + prv);
    fv.galbedo_prev = (const uint32_t*)r->buf(B_GALB0 + prv);
    fv.gmotion = (float2*)r->buf((r->side && cur) ? B_GMOT1 : B_GMOT);   // ping-pong only when post overlaps the next G-buffer
    fv.res_temporal = (ReservoirView*)r->buf(B_RES0);   // restir.rs:362-378: reads buffers[1], writes buffers[0]
    fv.res_spatial = (ReservoirView*)r->buf(B_RES1);    // renderer.rs:292-293: spatial buffers[0] -> buffers[1]
    fv.raw = (uint2*)r->buf(B_RAW); fv.display = (uint32_t*)r->buf(B_DISP);
    fv.history = (const float4*)r->buf(B_ACC0 + prv);   // post.rs:209-224: BG0 history = accum[1], out = accum[0]
    fv.accum = (float4*)r->buf(B_ACC0 + cur);
    fv.ray_counters = r->d_counters;
    fv.W = r->W; fv.H = r->H; fv.frame_count = r->frame_count; fv.max_depth = r->max_depth;
    fv.own_y0 = r->rb; fv.own_y1 = r->re;
    bool whole = (r->rb == 0 && r->re == r->H);
    fv.prev_y0 = whole ? 0u : (r->rb > r->motion_halo ? r->rb - r->motion_halo : 0u);
    fv.prev_y1 = whole ? r->H : std::min(r->H, r->re + r->motion_halo);
    fv.overflow = whole ? nullptr : r->d_counters + 8;
    memcpy(&fv.cam, cam, sizeof(CameraView));
}

static int launch_deferred_post(frt_renderer* r);
static int sync_all(frt_renderer* r) {
    if (r->post_deferred) { int rc_ = launch_deferred_post(r); if (rc_) return rc_; }
    HIP_TRY(hipStreamSynchronize(r->stream));
    if (r->side) HIP_TRY(hipStreamSynchronize(r->side));
    return FRT_OK;
}

static int resolve_timing(frt_renderer* r) {
    for (auto& t : r->pending) {
        float ms = 0.0f;
        HIP_TRY(hipEventSynchronize(t.b));
        HIP_TRY(hipEventElapsedTime(&ms, t.a, t.b));
        r->stats.ms_stage[t.stage] += ms;
        (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b);
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
    default: return fail(FRT_ERR_INVALID_ARG, "get: unknown selector");
    }
    return FRT_OK;
}
int frt_scene_bvh_stats(const frt_scene* s, uint32_t st[4]) {
    if (!s || !st) return fail(FRT_ERR_INVALID_ARG, "bvh_stats: null");
    st[0] = s->b.bvh_depth; st[1] = s->b.bvh_leaves; st[2] = s->b.bvh_max_leaf; st[3] = (uint32_t)s->b.pair_nodes.size();
    return FRT_OK;
}
void frt_camera_default(float aspect, uint32_t frame_count, uint32_t num_lights, frt_camera_uniform* out) {
    camera_default(aspect, frame_count, num_lights, out);
}

// ------------------------------------------------------------------------------------------------ renderer
uint64_t frt_renderer_arena_bytes(uint32_t width, uint32_t height) { return arena_layout(width, height, nullptr); }

void frt_renderer_destroy(frt_renderer* r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    (void)hipStreamSynchronize(r->stream);
    if (r->side) { (void)hipStreamSynchronize(r->side); (void)hipStreamDestroy(r->side); }
    if (r->ev_spatial) (void)hipEventDestroy(r->ev_spatial);
    if (r->ev_post) (void)hipEventDestroy(r->ev_post);
    if (r->ev_smain) (void)hipEventDestroy(r->ev_smain);
    if (r->ev_scont) (void)hipEventDestroy(r->ev_scont);
    if (r->ev_tmain) (void)hipEventDestroy(r->ev_tmain);
    for (auto& t : r->pending) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (void* p : r->scene_allocs) (void)hipFree(p);
    if (r->own_arena && r->arena) (void)hipFree(r->arena);
    if (r->d_counters) (void)hipFree(r->d_counters);
    if (r->d_qwords) (void)hipFree(r->d_qwords);
    if (r->d_qcount) (void)hipFree(r->d_qcount);
    if (r->d_tiles) (void)hipFree(r->d_tiles);
    if (r->own_stream && r->stream) (void)hipStreamDestroy(r->stream);
    delete r;
}

static int renderer_init(frt_renderer* r, const frt_scene* s, const frt_render_opts* o) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(FRT_ERR_NO_DEVICE, "no HIP device: this library has no CPU rendering path");
    if (r->device < 0 || r->device >= ndev) return fail(FRT_ERR_INVALID_ARG, "device ordinal out of range");
    HIP_TRY(hipSetDevice(r->device));
    if (o && (o->stream || (o->flags & FRT_FLAG_USE_STREAM))) { r->stream = (hipStream_t)o->stream; r->own_stream = false; }
    else { HIP_TRY(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking)); r->own_stream = true; }
    if (r->flags & FRT_FLAG_OVERLAP_POST) {
        HIP_TRY(hipStreamCreateWithFlags(&r->side, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&r->ev_spatial, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&r->ev_post, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&r->ev_smain, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&r->ev_scont, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&r->ev_tmain, hipEventDisableTiming));
    }
    r->arena_bytes = arena_layout(r->W, r->H, r->off);
    if (o && o->device_arena) {
        if (o->arena_bytes < r->arena_bytes) return fail(FRT_ERR_INVALID_ARG, "device_arena smaller than frt_renderer_arena_bytes");
        if (((uintptr_t)o->device_arena & 255u) != 0) return fail(FRT_ERR_INVALID_ARG, "device_arena must be 256-byte aligned");
        r->arena = (uint8_t*)o->device_arena; r->own_arena = false;
    } else {
        HIP_TRY(hipMalloc((void**)&r->arena, r->arena_bytes)); r->own_arena = true;
    }
    HIP_TRY(hipMalloc((void**)&r->d_counters, 9 * sizeof(unsigned long long)));   // 4 stages x {closest, any} + halo overflow
    {   // continuation queues (worst case: every pixel parks) and the bounce depths at which paths are cut
        r->qcap = r->W * std::min(r->H, (r->re - r->rb) + 2u * kHaloSpatial);   // worst case: every traced pixel parks
        HIP_TRY(hipMalloc((void**)&r->d_qwords, 4 * (size_t)kContWordsSpatial * r->qcap * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void**)&r->d_qcount, 2 * (kMaxCuts + 1) * sizeof(uint32_t)));
        // Parking pays when the launch saturates the chip (>= ~0.6 M pixels: +6 % at 1080p, +5 % at half a frame); a thin strip
        // is bound by the latency of its longest path and the extra launch only adds to it (tools/strip_time.py: 0.79 vs 0.70 ms
        // for 1/8 of a 1080p frame), so thin strips run uncut.
        if ((size_t)r->W * (r->re - r->rb) < 600000u) r->ncuts = 0;
        if (const char* e = getenv("FRT_PAIR")) r->pair_tail = atoi(e) != 0;   // experiment knob
        {   // tile-row orders of the two traced stages; frame 0 starts bottom row first (floors cost more than ceilings and skies)
            uint32_t rows[8];
            phase_rows(r, rows);
            for (int k = 0; k < 2; ++k) r->ntiles[k] = (rows[2 * (k + 1) + 1] - rows[2 * (k + 1)] + 15u) / 16u;
            if (r->ntiles[0] <= 1024u && r->ntiles[1] <= 1024u && !getenv("FRT_NO_TILE_ORDER")) {   // (the variable is an experiment knob: rows top to bottom)
                const size_t total = 2 * ((size_t)r->ntiles[0] + r->ntiles[1]);
                HIP_TRY(hipMalloc((void**)&r->d_tiles, total * sizeof(uint32_t)));
                std::vector<uint32_t> init(total, 0u);
                size_t o = 0;
                for (int k = 0; k < 2; ++k) { for (uint32_t i = 0; i < r->ntiles[k]; ++i) init[o + i] = r->ntiles[k] - 1u - i; o += 2 * (size_t)r->ntiles[k]; }
                HIP_TRY(hipMemcpy(r->d_tiles, init.data(), total * sizeof(uint32_t), hipMemcpyHostToDevice));
            }
        }
        if (const char* e = getenv("FRT_CUTS")) {   // experiment knob: comma-separated ascending depths, "0" = never cut
            r->ncuts = 0;
            for (const char* p = e; *p && r->ncuts < (uint32_t)kMaxCuts;) {
                uint32_t v = (uint32_t)strtoul(p, (char**)&p, 10);
                if (v >= 1) r->cuts[r->ncuts++] = v;
                if (*p == ',') ++p;
            }
        }
    }
    HIP_TRY(hipMemsetAsync(r->arena, 0, r->arena_bytes, r->stream));   // wgpu zero-initialises textures and buffers
    HIP_TRY(hipMemsetAsync(r->d_counters, 0, 9 * sizeof(unsigned long long), r->stream));
    int rc = upload_scene(r, s->b);
    if (rc) return rc;
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
    if (renderer_init(r, s, o) != FRT_OK) { std::string keep = g_err; frt_renderer_destroy(r); g_err = keep; return nullptr; }
    return r;
}

static int launch_deferred_post(frt_renderer* r) {
    r->post_deferred = false;
    frt_renderer::Timed t{};
    const bool timed = (r->flags & FRT_FLAG_TIMING) != 0;
    if (timed) {
        HIP_TRY(hipEventCreate(&t.a)); HIP_TRY(hipEventCreate(&t.b)); t.stage = 3;
        HIP_TRY(hipEventRecord(t.a, r->side));
    }
    StageLaunch L{};
    HIP_TRY(launch_stage(3, r->sv, r->post_fv, r->side, L));
    if (timed) { HIP_TRY(hipEventRecord(t.b, r->side)); r->pending.push_back(t); }
    HIP_TRY(hipEventRecord(r->ev_post, r->side));
    r->post_in_flight = true;
    r->stats.launches[3] += 1;
    return FRT_OK;
}

int frt_renderer_render_phases(frt_renderer* r, const frt_camera_uniform* cam, int phases) {
    if (!r || !cam) return fail(FRT_ERR_INVALID_ARG, "render: null");
    HIP_TRY(hipSetDevice(r->device));
    FrameView fv;
    fill_frame_view(r, cam, fv);
    if (phases & FRT_PHASE_GBUFFER) r->motion_slot = (r->side && (r->frame_count & 1u)) ? 1u : 0u;
    uint32_t rows[8];
    phase_rows(r, rows);
    for (int stage = 0; stage < 4; ++stage) {
        if (!(phases & (1 << stage))) continue;
        fv.y0 = rows[2 * stage]; fv.y1 = rows[2 * stage + 1];
        fv.ray_counters = r->d_counters + 2 * stage;
        hipStream_t q = r->stream;
        // Side-stream schedule (FRT_FLAG_OVERLAP_POST). The two latency-bound tails of a frame leave the GPU mostly idle, and the two
        // stages off the temporal -> spatial -> temporal chain are put there:
        //   main: G(f+1) | wait S-cont(f) | T-pixel(f+1) | T-cont(f+1)          | wait post(f) | S-pixel(f+1) ...
        //   side: S-cont(f) ............. |               | post(f), deferred    |              | S-cont(f+1) ...
        // G-buffer(f+1) depends on nothing of frame f; post(f) needs spatial(f) complete and must finish before S-pixel(f+1)
        // overwrites the radiance. post(f) is therefore not launched when it is requested but behind the temporal pixel kernel
        // of the next frame (or at the next sync / read / reset, whichever comes first).
        if (r->side && (stage == 2 || stage == 3) && r->post_deferred) { int rc_ = launch_deferred_post(r); if (rc_) return rc_; }
        if (r->side && stage == 3) {   // post(f) after spatial(f): its pixel kernel (main) and its continuation (side, in order)
            HIP_TRY(hipEventRecord(r->ev_spatial, r->stream));
            HIP_TRY(hipStreamWaitEvent(r->side, r->ev_spatial, 0));
            q = r->side;
            const bool traced_cut = r->ncuts > 0 && r->cuts[0] < r->max_depth && !(r->flags & FRT_FLAG_COMPACTION);
            if (traced_cut) { r->post_deferred = true; r->post_fv = fv; continue; }   // no temporal continuation to hide behind otherwise
        }
        if (r->side && (stage == 1 || stage == 2) && r->scont_in_flight) {
            HIP_TRY(hipStreamWaitEvent(r->stream, r->ev_scont, 0));
            r->scont_in_flight = false;
        }
        if (r->side && stage == 2 && r->post_in_flight) {   // spatial(f+1) overwrites the radiance post(f) reads
            HIP_TRY(hipStreamWaitEvent(r->stream, r->ev_post, 0));
            r->post_in_flight = false;
        }
        const bool traced_cut_stage = (stage == 1 || stage == 2) && r->ncuts > 0 && r->cuts[0] < r->max_depth && !(r->flags & FRT_FLAG_COMPACTION);
        if (traced_cut_stage) {   // this stage's segment counters must be zero (after the waits: the previous frame's tail used them)
            if (!r->counts_clean[stage - 1])
                HIP_TRY(hipMemsetAsync(r->d_qcount + (size_t)(stage - 1) * (kMaxCuts + 1), 0, (kMaxCuts + 1) * sizeof(uint32_t), r->stream));
            r->counts_clean[stage - 1] = false;   // about to be used
        }
        frt_renderer::Timed t{};
        bool timed = (r->flags & FRT_FLAG_TIMING) != 0;
        if (timed) {
            HIP_TRY(hipEventCreate(&t.a)); HIP_TRY(hipEventCreate(&t.b)); t.stage = stage;
            HIP_TRY(hipEventRecord(t.a, q));
        }
        StageLaunch L{};
        L.compaction = (r->flags & FRT_FLAG_COMPACTION) != 0;
        L.pair_tail = r->pair_tail;
        L.ncuts = r->ncuts;
        for (int k = 0; k < kMaxCuts; ++k) L.cuts[k] = r->cuts[k];
        if (stage == 1 || stage == 2) {
            size_t qsz = (size_t)kContWordsSpatial * r->qcap;
            for (int k = 0; k < 2; ++k) L.qwords[k] = r->d_qwords + (size_t)(2 * (stage - 1) + k) * qsz;
            L.counts = r->d_qcount + (size_t)(stage - 1) * (kMaxCuts + 1);
            L.capacity = r->qcap;
            if (r->d_tiles) {
                uint32_t* p = r->d_tiles;
                for (int k = 0; k < 2; ++k) { L.row_order[k] = p; L.row_cost[k] = p + r->ntiles[k]; L.nrows[k] = r->ntiles[k]; p += 2 * (size_t)r->ntiles[k]; }
            }
        }
        // Clearing the counters in passing saves the two memsets per frame (each sits between two dependent kernels): the temporal
        // pixel kernel clears the spatial stage's counters (the main stream has waited for the previous spatial continuation by
        // then), the row-order kernel behind the spatial pixel kernel clears the temporal stage's for the next frame.
        const bool sort_runs = stage == 2 && traced_cut_stage && L.row_cost[0] && L.row_cost[1];
        if (stage == 1 && traced_cut_stage) L.zero_in_pixel = r->d_qcount + (size_t)(kMaxCuts + 1);
        if (sort_runs) L.zero_in_sort = r->d_qcount;
        bool has_cont = false;
        const bool tail_on_side = r->side && stage == 2;
        hipEvent_t ev = tail_on_side ? r->ev_smain : ((r->side && stage == 1 && r->post_deferred) ? r->ev_tmain : nullptr);
        HIP_TRY(launch_stage(stage, r->sv, fv, q, L, tail_on_side ? r->side : nullptr, ev, &has_cont));
        if (stage == 1 && traced_cut_stage) r->counts_clean[1] = true;
        if (sort_runs) r->counts_clean[0] = true;
        const bool on_side = tail_on_side && has_cont;
        if (on_side) { HIP_TRY(hipEventRecord(r->ev_scont, r->side)); r->scont_in_flight = true; }
        if (timed) { HIP_TRY(hipEventRecord(t.b, on_side ? r->side : q)); r->pending.push_back(t); }
        if (r->side && stage == 3) { HIP_TRY(hipEventRecord(r->ev_post, r->side)); r->post_in_flight = true; }
        if (r->side && stage == 1 && r->post_deferred) {   // post(f) behind T-pixel(f+1): it runs while T-cont(f+1) leaves the GPU idle
            if (has_cont) HIP_TRY(hipStreamWaitEvent(r->side, r->ev_tmain, 0));
            int rc_ = launch_deferred_post(r);
            if (rc_) return rc_;
        }
        if (r->side && stage == 1 && r->post_in_flight) {
            // whatever the caller enqueues next on the main stream (the halo exchange of the previous accumulation rows, then
            // spatial) must see post(f) finished
            HIP_TRY(hipStreamWaitEvent(r->stream, r->ev_post, 0));
            r->post_in_flight = false;
        }
        r->stats.launches[stage] += 1;
    }
    return FRT_OK;
}
int frt_renderer_end_frame(frt_renderer* r) {
    if (!r) return fail(FRT_ERR_INVALID_ARG, "end_frame: null");
    r->frame_count += 1;   // renderer.rs:515
    r->stats.frames += 1;
    return FRT_OK;
}
int frt_renderer_render(frt_renderer* r, const frt_camera_uniform* cam) {
    int rc = frt_renderer_render_phases(r, cam, FRT_PHASE_ALL);
    if (rc) return rc;
    return frt_renderer_end_frame(r);
}
int frt_renderer_sync(frt_renderer* r) {
    if (!r) return fail(FRT_ERR_INVALID_ARG, "sync: null");
    HIP_TRY(hipSetDevice(r->device));
    return sync_all(r);
}
uint32_t frt_renderer_frame_count(const frt_renderer* r) { return r ? r->frame_count : 0u; }
int frt_renderer_reset(frt_renderer* r) {
    if (!r) return fail(FRT_ERR_INVALID_ARG, "reset: null");
    // The side-stream schedule relies on the ping-pong slots alternating from frame to frame; a reset breaks the alternation (the
    // next frame may write the slot the in-flight tail of the last frame still reads), so the main stream first waits for that tail.
    // Stream-level only: the reference resets every frame while the camera moves (state.rs:152), this must stay asynchronous.
    if (r->side) {
        HIP_TRY(hipSetDevice(r->device));
        if (r->post_deferred) { int rc_ = launch_deferred_post(r); if (rc_) return rc_; }
        if (r->scont_in_flight) { HIP_TRY(hipStreamWaitEvent(r->stream, r->ev_scont, 0)); r->scont_in_flight = false; }
        if (r->post_in_flight) { HIP_TRY(hipStreamWaitEvent(r->stream, r->ev_post, 0)); r->post_in_flight = false; }
    }
    r->frame_count = 0;
    return FRT_OK;
}
int frt_renderer_clear(frt_renderer* r) {
    if (!r) return fail(FRT_ERR_INVALID_ARG, "clear: null");
    HIP_TRY(hipSetDevice(r->device));
    { int rc_ = sync_all(r); if (rc_) return rc_; }
    int rc = resolve_timing(r);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(r->arena, 0, r->arena_bytes, r->stream));
    HIP_TRY(hipMemsetAsync(r->d_counters, 0, 9 * sizeof(unsigned long long), r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    r->frame_count = 0;
    memset(&r->stats, 0, sizeof(r->stats));
    return FRT_OK;
}

static int buf_index(const frt_renderer* r, int buf, int index) {
    switch (buf) {
    case FRT_BUF_GPOS: return B_GPOS0 + (index & 1);
    case FRT_BUF_GNORMAL: return B_GNRM0 + (index & 1);
    case FRT_BUF_GALBEDO: return B_GALB0 + (index & 1);
    case FRT_BUF_GMOTION: return ((r->motion_slot ^ (uint32_t)index) & 1u) ? B_GMOT1 : B_GMOT;   // 0 = the last rendered frame's (the reference has one)
    case FRT_BUF_RESERVOIR: return B_RES0 + (index & 1);
    case FRT_BUF_RAW: return B_RAW;
    case FRT_BUF_DISPLAY: return B_DISP;
    case FRT_BUF_ACCUM: return B_ACC0 + (index & 1);
    }
    return -1;
}
int frt_renderer_buffer_info(const frt_renderer* r, int buf, int index, void** device_ptr, uint32_t* bpp) {
    int b = r ? buf_index(r, buf, index) : -1;
    if (b < 0) return fail(FRT_ERR_INVALID_ARG, "buffer_info: bad buffer");
    if (device_ptr) *device_ptr = r->buf(b);
    if (bpp) *bpp = kBpp[b];
    return FRT_OK;
}
int frt_renderer_read_buffer(frt_renderer* r, int buf, int index, void* out) {
    int b = r ? buf_index(r, buf, index) : -1;
    if (b < 0 || !out) return fail(FRT_ERR_INVALID_ARG, "read_buffer: bad arguments");
    HIP_TRY(hipSetDevice(r->device));
    { int rc_ = sync_all(r); if (rc_) return rc_; }
    HIP_TRY(hipMemcpy(out, r->buf(b), (size_t)r->W * r->H * kBpp[b], hipMemcpyDeviceToHost));
    return FRT_OK;
}
int frt_renderer_read_rows(frt_renderer* r, int buf, int index, uint32_t y0, uint32_t y1, void* out) {
    int b = r ? buf_index(r, buf, index) : -1;
    if (b < 0 || !out || y0 > y1 || y1 > r->H) return fail(FRT_ERR_INVALID_ARG, "read_rows: bad arguments");
    HIP_TRY(hipSetDevice(r->device));
    { int rc_ = sync_all(r); if (rc_) return rc_; }
    size_t pitch = (size_t)r->W * kBpp[b];
    HIP_TRY(hipMemcpy(out, (uint8_t*)r->buf(b) + pitch * y0, pitch * (y1 - y0), hipMemcpyDeviceToHost));
    return FRT_OK;
}
int frt_renderer_write_rows(frt_renderer* r, int buf, int index, uint32_t y0, uint32_t y1, const void* in) {
    int b = r ? buf_index(r, buf, index) : -1;
    if (b < 0 || !in || y0 > y1 || y1 > r->H) return fail(FRT_ERR_INVALID_ARG, "write_rows: bad arguments");
    HIP_TRY(hipSetDevice(r->device));
    { int rc_ = sync_all(r); if (rc_) return rc_; }
    size_t pitch = (size_t)r->W * kBpp[b];
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
    HIP_TRY(hipSetDevice(r->device));
    { int rc_ = sync_all(r); if (rc_) return rc_; }
    int rc = resolve_timing(r);
    if (rc) return rc;
    unsigned long long c[9] = {0};
    HIP_TRY(hipMemcpy(c, r->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    r->stats.halo_overflow = c[8];
    r->stats.rays_closest = 0; r->stats.rays_any = 0;
    for (int st = 0; st < 4; ++st) {
        r->stats.rays_stage[st][0] = c[2 * st]; r->stats.rays_stage[st][1] = c[2 * st + 1];
        r->stats.rays_closest += c[2 * st]; r->stats.rays_any += c[2 * st + 1];
    }
    *out = r->stats;
    return FRT_OK;
}

int frt_renderer_set_timing(frt_renderer* r, int on) {
    if (!r) return fail(FRT_ERR_INVALID_ARG, "set_timing: null");
    if (on) r->flags |= FRT_FLAG_TIMING; else r->flags &= ~FRT_FLAG_TIMING;
    return FRT_OK;
}

} // extern "C"
