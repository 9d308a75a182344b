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
    std::vector<ReservoirView> res[2];
    std::vector<uint2> raw;
    unsigned long long rays[2] = {0, 0};
    int nthreads = 8;
};
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
    sv.tris = reinterpret_cast<const float4*>(b.tri_slots.data());
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
    return h;
}
void hc_destroy(void* p) { delete (HostCheck*)p; }

// sm != 0: drive the resumable state machine (frt_path.hpp) instead of the straight-line functions (frt_mono.hpp)
void hc_render(void* p, const frt_camera_uniform* cam, int sm) {
    HostCheck* h = (HostCheck*)p;
    uint32_t cur = h->frame_count & 1u, prv = cur ^ 1u;
    FrameView fv{};
    fv.gpos = h->gpos[cur].data(); fv.gnormal = h->gnormal[cur].data(); fv.galbedo = h->galbedo[cur].data();
    fv.gpos_prev = h->gpos[prv].data(); fv.gnormal_prev = h->gnormal[prv].data(); fv.galbedo_prev = h->galbedo[prv].data();
    fv.gmotion = h->gmotion.data(); fv.res_temporal = h->res[0].data(); fv.res_spatial = h->res[1].data();
    fv.raw = h->raw.data(); fv.display = h->display.data(); fv.history = h->accum[prv].data(); fv.accum = h->accum[cur].data();
    fv.ray_counters = nullptr; fv.W = h->W; fv.H = h->H; fv.frame_count = h->frame_count; fv.max_depth = h->max_depth;
    fv.y0 = 0; fv.y1 = h->H; fv.own_y0 = 0; fv.own_y1 = h->H;
    memcpy(&fv.cam, cam, sizeof(CameraView));
    int nt = h->nthreads;
    std::vector<unsigned long long> rc((size_t)nt * 2, 0ull);
    for (int stage = 0; stage < 4; ++stage) {
        auto work = [&](int tid) {
            uint32_t stack[kStackDepth];
            for (uint32_t y = (uint32_t)tid; y < h->H; y += (uint32_t)nt)
                for (uint32_t x = 0; x < h->W; ++x) {
                    PathCtx c(h->sv, fv, stack, 1u);
                    if (stage == 0) gbuffer_pixel(c, x, y);
                    else if (stage == 1) { if (sm) temporal_pixel_sm(c, x, y); else temporal_pixel(c, x, y); }
                    else if (stage == 2) { if (sm) spatial_pixel_sm(c, x, y); else spatial_pixel(c, x, y); }
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

// probe: closest / any hits through the product traversal (pair nodes + triangle slots)
void hc_trace(const frt_scene* s, int any, uint32_t n, const float* o, const float* d, float tmin, const float* tmax,
              float* t_out, uint32_t* tri_out, float* uv_out, uint8_t* front_out) {
    const SceneBuilder& b = s->b;
    SceneView sv{};
    sv.nodes = reinterpret_cast<const float4*>(b.pair_nodes.data());
    sv.tris = reinterpret_cast<const float4*>(b.tri_slots.data());
    sv.instances = reinterpret_cast<const InstanceView*>(b.instances_dev.data());
    uint32_t stack[kStackDepth];
    for (uint32_t i = 0; i < n; ++i) {
        HitRec h;
        f3 oo = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), dd = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        if (any) trace<true>(sv, oo, dd, tmin, tmax[i], stack, 1u, h);
        else trace<false>(sv, oo, dd, tmin, tmax[i], stack, 1u, h);
        t_out[i] = h.tri != 0xFFFFFFFFu ? h.t : -1.0f;
        tri_out[i] = h.tri;
        if (uv_out) { uv_out[2 * i] = h.u; uv_out[2 * i + 1] = h.v; }
        if (front_out) front_out[i] = h.front;
    }
}

} // extern "C"
