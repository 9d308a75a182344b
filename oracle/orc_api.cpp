// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path. PARITY UNPINNED (see orc_math.hpp).
// C API for ctypes. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include "orc_render.hpp"
#include <cstring>
#include <chrono>

using namespace orc;

extern "C" {

// ---- known-answer helpers (SURVEY.md §8c T0 pins)
uint32_t orc_pcg_hash(uint32_t x) {
    uint32_t state = x * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
void orc_encode_octahedral(const float n[3], float out[2]) {
    vec2 e = encode_octahedral_normal(V3(n[0], n[1], n[2])); out[0] = e.x; out[1] = e.y;
}
uint16_t orc_f32_to_f16(float f) { return f32_to_f16(f); }
float orc_f16_to_f32(uint16_t h) { return f16_to_f32(h); }
uint8_t orc_f32_to_unorm8(float f) { return f32_to_unorm8(f); }
float orc_sin(float x) { return sin_(x); }
float orc_cos(float x) { return cos_(x); }
float orc_exp2(float x) { return exp2_(x); }
float orc_log2(float x) { return log2_(x); }
float orc_pow(float x, float y) { return pow_(x, y); }
float orc_exp(float x) { return exp_(x); }
void orc_struct_sizes(uint32_t out[8]) {
    out[0] = sizeof(CameraUniform); out[1] = sizeof(VertexAttributes); out[2] = sizeof(Material); out[3] = sizeof(LightUniform);
    out[4] = sizeof(MeshInfo); out[5] = sizeof(Reservoir); out[6] = 16; out[7] = 32;
}
void orc_mesh_counts(int which, uint32_t subdiv, uint32_t out[2]) {
    Geometry g = which == 0 ? create_plane() : which == 1 ? create_cube() : which == 2 ? create_sphere(subdiv) : create_crystal();
    out[0] = (uint32_t)g.positions.size(); out[1] = (uint32_t)g.indices.size() / 3u;
}
// which: 0 plane, 1 cube, 2 sphere(subdiv), 3 crystal. Buffers may be null to query sizes only.
void orc_mesh_get(int which, uint32_t subdiv, float* pos4, void* attrs, uint32_t* idx) {
    Geometry g = which == 0 ? create_plane() : which == 1 ? create_cube() : which == 2 ? create_sphere(subdiv) : create_crystal();
    if (pos4) memcpy(pos4, g.positions.data(), g.positions.size() * 16);
    if (attrs) memcpy(attrs, g.attributes.data(), g.attributes.size() * 32);
    if (idx) memcpy(idx, g.indices.data(), g.indices.size() * 4);
}

// ---- scene
void* orc_scene_create() { return new Scene(); }
void orc_scene_destroy(void* s) { delete (Scene*)s; }
void* orc_scene_create_cornell_box() { Scene* s = new Scene(); create_cornell_box(*s); return s; }
void* orc_scene_create_restir_scene() { Scene* s = new Scene(); create_restir_scene(*s); return s; }
int orc_scene_add_mesh(void* sp, const float* pos4, uint32_t nverts, const void* attrs, const uint32_t* idx, uint32_t nidx) {
    Geometry g;
    g.positions.resize(nverts); memcpy(g.positions.data(), pos4, (size_t)nverts * 16);
    g.attributes.resize(nverts); memcpy(g.attributes.data(), attrs, (size_t)nverts * 32);
    g.indices.assign(idx, idx + nidx);
    return (int)((Scene*)sp)->add_mesh(g);
}
int orc_scene_add_material(void* sp, const void* mat64) { Material m; memcpy(&m, mat64, 64); return (int)((Scene*)sp)->add_material(m); }
int orc_scene_add_instance(void* sp, uint32_t mesh, uint32_t mat, const float m[16]) {
    mat4 t; for (int c = 0; c < 4; ++c) t.c[c] = V4(m[4 * c], m[4 * c + 1], m[4 * c + 2], m[4 * c + 3]);
    ((Scene*)sp)->add_instance(mesh, mat, t); return 0;
}
int orc_scene_add_light(void* sp, const void* light64) { LightUniform l; memcpy(&l, light64, 64); ((Scene*)sp)->lights.push_back(l); return 0; }
int orc_scene_add_texture(void* sp, int kind, const uint8_t* rgba8) {
    Scene* s = (Scene*)sp; return (int)(kind == 0 ? s->add_color_texture(rgba8) : s->add_data_texture(rgba8));
}
int orc_scene_build(void* sp) { ((Scene*)sp)->build(); return 0; }
// counts: tris, instances, materials, lights, meshes, attributes, indices, bvh nodes
void orc_scene_counts(void* sp, uint32_t out[8]) {
    Scene* s = (Scene*)sp;
    out[0] = (uint32_t)s->tris.size(); out[1] = (uint32_t)s->instances.size(); out[2] = (uint32_t)s->materials.size();
    out[3] = (uint32_t)s->lights.size(); out[4] = (uint32_t)s->mesh_infos.size(); out[5] = (uint32_t)s->attributes.size();
    out[6] = (uint32_t)s->indices.size(); out[7] = (uint32_t)s->bvh_nodes.size();
}
// which: 0 tris (9 f32 each: v0,e1,e2), 1 tri_instance (u32), 2 materials (64 B), 3 lights (64 B), 4 attributes (32 B),
// 5 indices (u32), 6 mesh_infos (16 B), 7 instances (mesh,mat,first_tri,tri_count,flip u32 + m[16] + w2o[9] f32 = 120 B)
void orc_scene_get(void* sp, int which, void* out) {
    Scene* s = (Scene*)sp;
    switch (which) {
    case 0: memcpy(out, s->tris.data(), s->tris.size() * 36); break;
    case 1: memcpy(out, s->tri_instance.data(), s->tri_instance.size() * 4); break;
    case 2: memcpy(out, s->materials.data(), s->materials.size() * 64); break;
    case 3: memcpy(out, s->lights.data(), s->lights.size() * 64); break;
    case 4: memcpy(out, s->attributes.data(), s->attributes.size() * 32); break;
    case 5: memcpy(out, s->indices.data(), s->indices.size() * 4); break;
    case 6: memcpy(out, s->mesh_infos.data(), s->mesh_infos.size() * 16); break;
    case 7: {
        uint8_t* p = (uint8_t*)out;
        for (const Instance& in : s->instances) {
            uint32_t h[5] = {in.mesh_id, in.mat_id, in.first_tri, in.tri_count, in.flip};
            memcpy(p, h, 20); memcpy(p + 20, in.m, 64); memcpy(p + 84, in.w2o, 36); p += 120;
        }
    } break;
    }
}
// Import a canonical BVH2 (32-byte nodes, see orc_scene.hpp) + triangle permutation.
int orc_scene_set_bvh(void* sp, const void* nodes, uint32_t nnodes, const uint32_t* tri_index, uint32_t ntris) {
    Scene* s = (Scene*)sp;
    if (ntris != s->tris.size()) return -1;
    s->bvh_nodes.resize(nnodes); memcpy(s->bvh_nodes.data(), nodes, (size_t)nnodes * 32);
    s->bvh_tri_index.assign(tri_index, tri_index + ntris);
    return 0;
}
void orc_camera_default(float aspect, uint32_t frame, uint32_t nlights, void* out288) {
    CameraUniform c = camera_default(aspect, frame, nlights); memcpy(out288, &c, 288);
}

void orc_camera_build(const float pos[3], float yaw, float pitch, const float* prev_vp, float aspect, uint32_t frame, uint32_t nlights,
                      float jx, float jy, void* out288, float* unjittered16) {
    CameraUniform c = camera_build(V3(pos[0], pos[1], pos[2]), yaw, pitch, prev_vp, aspect, frame, nlights, jx, jy, unjittered16); memcpy(out288, &c, 288);
}
void orc_camera_halton_jitter(uint32_t index, uint32_t w, uint32_t h, float scale, float out[2]) { camera_halton_jitter(index, w, h, scale, out); }

// Text mode (orc_math.hpp): process-wide; set it before rendering.
void orc_set_text_mode(int on) { text_mode() = on != 0; }
int orc_get_text_mode(void) { return text_mode() ? 1 : 0; }

// ---- tracing probes (T2): closest/any over n rays; o,d are n*3 floats
void orc_trace_closest(void* sp, int use_bvh, uint32_t n, const float* o, const float* d, float tmin, float tmax,
                       float* t_out, uint32_t* tri_out, float* uv_out, uint8_t* front_out, uint64_t stats[4]) {
    Scene* s = (Scene*)sp; Tracer tr(*s, use_bvh != 0); TraceStats st;
    for (uint32_t i = 0; i < n; ++i) {
        Hit h = tr.closest(V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmin, tmax, st);
        t_out[i] = h.hit ? h.t : -1.0f; tri_out[i] = h.tri; uv_out[2 * i] = h.u; uv_out[2 * i + 1] = h.v; front_out[i] = h.front;
    }
    if (stats) { stats[0] = st.rays_closest; stats[1] = st.rays_any; stats[2] = st.nodes; stats[3] = st.tris; }
}
void orc_trace_any(void* sp, int use_bvh, uint32_t n, const float* o, const float* d, float tmin, const float* tmax, uint8_t* occ_out) {
    Scene* s = (Scene*)sp; Tracer tr(*s, use_bvh != 0); TraceStats st;
    for (uint32_t i = 0; i < n; ++i)
        occ_out[i] = tr.any(V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmin, tmax[i], st);
}

// ---- renderer
void* orc_renderer_create(void* scene, uint32_t w, uint32_t h, uint32_t max_depth, int use_bvh, int nthreads) {
    return new Renderer((Scene*)scene, w, h, max_depth, use_bvh != 0, nthreads);
}
void orc_renderer_destroy(void* r) { delete (Renderer*)r; }
void orc_renderer_reset(void* r) { ((Renderer*)r)->reset(); }                   // zero every buffer and the counter
void orc_renderer_restart_counter(void* r) { ((Renderer*)r)->frame_count = 0; }   // src/state.rs:152: only renderer.frame_count = 0
void orc_renderer_render(void* r, const void* cam288) { CameraUniform c; memcpy(&c, cam288, 288); ((Renderer*)r)->render(c); }
void orc_renderer_render_phases(void* r, const void* cam288, int phases, uint32_t y0, uint32_t y1) {
    CameraUniform c; memcpy(&c, cam288, 288); ((Renderer*)r)->render_phases(c, phases, y0, y1);
}
void orc_renderer_end_frame(void* r) { ((Renderer*)r)->end_frame(); }
void orc_renderer_set_jitter(void* r, float jx, float jy) { ((Renderer*)r)->jitter[0] = jx; ((Renderer*)r)->jitter[1] = jy; }
uint32_t orc_renderer_frame_count(void* r) { return ((Renderer*)r)->frame_count; }
// Buffers. which: 0 gpos[i] (16 B/px), 1 gnormal[i] (16), 2 galbedo[i] (4), 3 gmotion (8), 4 reservoirs[i] (32),
// 5 raw rgba16f (8), 6 display rgba8 (4), 7 accum[i] (16)
static void* buf_ptr(Renderer* R, int which, int i, size_t* bpp) {
    switch (which) {
    case 0: *bpp = 16; return R->gpos[i & 1].data();
    case 1: *bpp = 16; return R->gnormal[i & 1].data();
    case 2: *bpp = 4; return R->galbedo[i & 1].data();
    case 3: *bpp = 8; return R->gmotion.data();
    case 4: *bpp = 32; return R->reservoirs[i & 1].data();
    case 5: *bpp = 8; return R->raw.data();
    case 6: *bpp = 4; return R->display.data();
    case 7: *bpp = 16; return R->accum[i & 1].data();
    }
    *bpp = 0; return nullptr;
}
int orc_renderer_read(void* r, int which, int i, void* out) {
    Renderer* R = (Renderer*)r; size_t bpp; void* p = buf_ptr(R, which, i, &bpp);
    if (!p) return -1;
    memcpy(out, p, bpp * R->W * R->H); return 0;
}
// Row-range write (halo exchange in the multi-rank CPU test): rows [y0, y1) of buffer `which`/`i` from `src` (tightly packed rows).
int orc_renderer_write_rows(void* r, int which, int i, uint32_t y0, uint32_t y1, const void* src) {
    Renderer* R = (Renderer*)r; size_t bpp; uint8_t* p = (uint8_t*)buf_ptr(R, which, i, &bpp);
    if (!p || y1 > R->H || y0 > y1) return -1;
    memcpy(p + (size_t)y0 * R->W * bpp, src, (size_t)(y1 - y0) * R->W * bpp); return 0;
}
int orc_renderer_read_rows(void* r, int which, int i, uint32_t y0, uint32_t y1, void* dst) {
    Renderer* R = (Renderer*)r; size_t bpp; uint8_t* p = (uint8_t*)buf_ptr(R, which, i, &bpp);
    if (!p || y1 > R->H || y0 > y1) return -1;
    memcpy(dst, p + (size_t)y0 * R->W * bpp, (size_t)(y1 - y0) * R->W * bpp); return 0;
}
// stats: total {closest, any, nodes, tris} then per stage g,t,s,p x 4
void orc_renderer_stats(void* r, uint64_t out[20]) {
    Renderer* R = (Renderer*)r;
    const TraceStats* all[5] = {&R->stats_total, &R->stats_stage[0], &R->stats_stage[1], &R->stats_stage[2], &R->stats_stage[3]};
    for (int k = 0; k < 5; ++k) { out[4 * k] = all[k]->rays_closest; out[4 * k + 1] = all[k]->rays_any; out[4 * k + 2] = all[k]->nodes; out[4 * k + 3] = all[k]->tris; }
}
// CPU baseline leg of bench.py: render `frames` frames, return wall seconds; rays via orc_renderer_stats.
double orc_renderer_time_frames(void* r, const void* cams288, uint32_t frames) {
    Renderer* R = (Renderer*)r;
    auto t0 = std::chrono::steady_clock::now();
    for (uint32_t f = 0; f < frames; ++f) { CameraUniform c; memcpy(&c, (const uint8_t*)cams288 + 288 * f, 288); R->render(c); }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

} // extern "C"
