// frt_shade.hpp — device functions of the shading path: BSDF (GGX-VNDF + Lambert + delta glass), light sampling,
// NEE with MIS, Russian roulette, trace_path in both of the reference's variants, and the per-pixel bodies of the four
// stages (gbuffer.wgsl, restir.wgsl, restir_spatial.wgsl, post.wgsl). Contract arithmetic throughout (frt_math.hpp).
// __host__ __device__ so that tests/hostcheck can instantiate the same functions on the CPU; the library only ever
// launches them from kernels (frt_kernels.hip).
#pragma once
#include "frt_trace.hpp"

namespace frt {

struct ReservoirView { uint32_t y; float w_sum; uint32_t M; float W; float sx, sy, sz; float p_hat; };   // restir.rs:5-14

struct CameraView {   // camera.rs:4-15 (288 B)
    float view_proj[16], view_inverse[16], proj_inverse[16], view_pos[4], prev_view_proj[16];
    uint32_t frame_count, num_lights, pad0, pad1;
};

// Per-frame view of the per-pixel buffers in HBM. Full-frame pitch; a launch covers rows [y0, y1).
struct FrameView {
    float4* gpos; float4* gnormal; uint32_t* galbedo;                     // current G-buffer slot (frame_count % 2)
    const float4* gpos_prev; const float4* gnormal_prev; const uint32_t* galbedo_prev;
    float2* gmotion;
    ReservoirView* res_temporal;     // reservoir_buffers[0]
    ReservoirView* res_spatial;      // reservoir_buffers[1]
    float4* cand;                    // temporal candidate record (v1_pos, p_hat): T-trace -> T-merge (frt_path.hpp)
    uint2* raw;                      // rgba16f
    uint32_t* display;               // rgba8
    const float4* history; float4* accum;
    unsigned long long* ray_counters;   // [0] closest, [1] any
    uint32_t W, H, frame_count, max_depth, y0, y1;
    uint32_t own_y0, own_y1;         // rows whose rays are counted (a strip's redundant halo rows are not)
    uint32_t prev_y0, prev_y1;       // rows whose previous-frame reservoirs / G-buffer are valid here (whole frame: 0, H)
    unsigned long long* overflow;    // counts previous-frame reads outside them (may be null)
    float jitter_x, jitter_y;        // PostParams.jitter (renderer.rs:14, :376); (0, 0) in the shipped reference (camera.rs:202-203)
    CameraView cam;
};

// A strip read previous-frame state it does not hold (camera moved further than the strip's motion halo): counted, not fatal.
FRT_HD void note_halo_overflow(const FrameView& fv) {
    if (!fv.overflow) return;
#if defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(fv.overflow, 1ull);
#else
    ++*fv.overflow;
#endif
}

static constexpr float kPI = 3.14159265359f;   // restir.wgsl:4
static constexpr float kInvPI = 1.0f / kPI;    // x / PI is evaluated as x * (1 / PI) (contract)

struct PathCtx {
    const SceneView& sc;
    const FrameView& fv;
    uint32_t* stk; uint32_t stride;
    const uint32_t* lds_top = nullptr;   // the workgroup's LDS copy of quad nodes 0 .. 4 (frt_kernels.hip: stage_top_nodes), or null: trace4 then reads them like any node
    uint32_t rng;               // var<private> rng_seed, restir.wgsl:130
    uint32_t n_closest, n_any;  // rays issued by this lane
    FRT_HD PathCtx(const SceneView& s, const FrameView& f, uint32_t* st, uint32_t sd) : sc(s), fv(f), stk(st), stride(sd), rng(0), n_closest(0), n_any(0) {}
    FRT_HD float rand() { rng = pcg_hash(rng); return (float)rng / 4294967296.0f; }   // restir.wgsl:138-141 (literal rounds to 2^32)
    // The two ray queries of the shaders (rayQueryInitialize ... rayQueryGetCommittedIntersection). Functions that trace are templates on
    // the context type and call these, so that a kernel can substitute its own traversal (frt_kernels.hip: ResidentCtx walks a BVH cached in LDS).
    // Default: the quad tree (frt_trace.hpp: trace4).
    // FRT_DBG_TWICE (timing builds only, tools/abrun.sh): bit 0 walks every closest-hit ray of the traced stages twice, bit 1 every any-hit ray; the
    // second walk's result is discarded, so the frame-time difference IS the time the renderer spends in traversal — measured in place, beside the
    // shading it shares the CU with (profiles/r3_experiments/traversal_in_situ.md: 0.52 + 0.49 ms of the 1.68 ms the traced stages take on one stream).
#ifndef FRT_DBG_TWICE
#define FRT_DBG_TWICE 0
#endif
    template <bool ANY, bool VOTE>
    FRT_HD void walk(f3 o, f3 d, float tmin, float tmax, HitRec& h) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (FRT_DBG_TWICE & (ANY ? 2 : 1)) { HitRec h2; f3 o2 = o; asm volatile("" : "+v"(o2.x)); trace4<ANY, VOTE>(sc, o2, d, tmin, tmax, stk, stride, h2, lds_top); asm volatile("" :: "v"(h2.t), "v"(h2.tri)); }
#endif
        trace4<ANY, VOTE>(sc, o, d, tmin, tmax, stk, stride, h, lds_top);
    }
    FRT_HD void closest(f3 o, f3 d, float tmin, float tmax, HitRec& h) { n_closest++; walk<false, false>(o, d, tmin, tmax, h); }
    FRT_HD bool any(f3 o, f3 d, float tmin, float tmax) { HitRec h; n_any++; walk<true, false>(o, d, tmin, tmax, h); return h.tri != 0xFFFFFFFFu; }
};
// The same context with the voting walk (frt_trace.hpp: trace4<ANY, VOTE = true>): what the traced kernels of scenes with a deep tree run.
struct VotePathCtx : PathCtx {
    FRT_HD VotePathCtx(const SceneView& s, const FrameView& f, uint32_t* st, uint32_t sd) : PathCtx(s, f, st, sd) {}
    FRT_HD void closest(f3 o, f3 d, float tmin, float tmax, HitRec& h) { n_closest++; walk<false, true>(o, d, tmin, tmax, h); }
    FRT_HD bool any(f3 o, f3 d, float tmin, float tmax) { HitRec h; n_any++; walk<true, true>(o, d, tmin, tmax, h); return h.tri != 0xFFFFFFFFu; }
};

// The context of the kernels that walk the 8-wide tree (frt_trace.hpp: trace8). `nb`: node 0 of the tree — in HBM (SceneView::nodes8), or the workgroup's
// LDS copy of the whole tree when it is small enough (frt_kernels.hip: stage_wide_nodes). The stack column has kStack8 entries.
struct Wide8PathCtx : PathCtx {
    const char* nb;
    FRT_HD Wide8PathCtx(const SceneView& s, const FrameView& f, uint32_t* st, uint32_t sd) : PathCtx(s, f, st, sd), nb(reinterpret_cast<const char*>(s.nodes8)) {}
    template <bool ANY>
    FRT_HD void walk8(f3 o, f3 d, float tmin, float tmax, HitRec& h) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (FRT_DBG_TWICE & (ANY ? 2 : 1)) { HitRec h2; f3 o2 = o; asm volatile("" : "+v"(o2.x)); trace8<ANY>(sc, nb, o2, d, tmin, tmax, stk, stride, h2); asm volatile("" :: "v"(h2.t), "v"(h2.tri)); }
#endif
        trace8<ANY>(sc, nb, o, d, tmin, tmax, stk, stride, h);
    }
    FRT_HD void closest(f3 o, f3 d, float tmin, float tmax, HitRec& h) { n_closest++; walk8<false>(o, d, tmin, tmax, h); }
    FRT_HD bool any(f3 o, f3 d, float tmin, float tmax) { HitRec h; n_any++; walk8<true>(o, d, tmin, tmax, h); return h.tri != 0xFFFFFFFFu; }
};

FRT_HD float rand_lcg(uint32_t& state) {   // restir.wgsl:781-786
    uint32_t old = state;
    state = old * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (float)((word >> 22u) ^ word) / 4294967296.0f;
}
FRT_HD float luminance(f3 c) { return c.x * 0.2126f + c.y * 0.7152f + c.z * 0.0722f; }   // restir.wgsl:742

// ------------------------------------------------------------------------------------------------ textures
// textureSampleLevel(level 0) under the sampler of renderer.rs:240-249 (Repeat / Linear): f32 bilinear on texel centres.
template <bool SRGB>
FRT_HD f4 fetch_texel(const SceneView& sc, const uint8_t* base, int x, int y) {
    uint32_t p = reinterpret_cast<const uint32_t*>(base)[(uint32_t)y * 1024u + (uint32_t)x];
    uint32_t r = p & 0xffu, g = (p >> 8) & 0xffu, b = (p >> 16) & 0xffu, a = p >> 24;
    if (SRGB) return mk4(sc.srgb_lut[r], sc.srgb_lut[g], sc.srgb_lut[b], (float)a / 255.0f);
    return mk4((float)r / 255.0f, (float)g / 255.0f, (float)b / 255.0f, (float)a / 255.0f);
}
template <bool SRGB>
FRT_HD f4 sample_layer(const SceneView& sc, uint32_t layer, f2 uv) {
    const uint8_t* base = (SRGB ? sc.color_tex : sc.data_tex) + (size_t)layer * (1024u * 1024u * 4u);
    float x = uv.x * 1024.0f - 0.5f, y = uv.y * 1024.0f - 0.5f;
    float fx = floorf_(x), fy = floorf_(y);
    float ax = x - fx, ay = y - fy;
    int x0 = (int)fx & 1023, y0 = (int)fy & 1023, x1 = (x0 + 1) & 1023, y1 = (y0 + 1) & 1023;
    f4 t00 = fetch_texel<SRGB>(sc, base, x0, y0), t10 = fetch_texel<SRGB>(sc, base, x1, y0);
    f4 t01 = fetch_texel<SRGB>(sc, base, x0, y1), t11 = fetch_texel<SRGB>(sc, base, x1, y1);
    f4 top = t00 * (1.0f - ax) + t10 * ax;
    f4 bot = t01 * (1.0f - ax) + t11 * ax;
    return top * (1.0f - ay) + bot * ay;
}

// ------------------------------------------------------------------------------------------------ helpers
FRT_HD f3 decode_octahedral_normal(float ex, float ey) {   // gbuffer.wgsl:38-44
    f3 n = mk3(ex, ey, 1.0f - fabsf_(ex) - fabsf_(ey));
    float t = fmaxn(-n.z, 0.0f);
    n.x += (n.x >= 0.0f) ? -t : t;
    n.y += (n.y >= 0.0f) ? -t : t;
    return normalize(n);
}
FRT_HD f2 encode_octahedral_normal(f3 n) {   // gbuffer.wgsl:46-62
    float l1 = fabsf_(n.x) + fabsf_(n.y) + fabsf_(n.z);
    float s = 1.0f / fmaxn(l1, 1e-6f);
    f2 res = l1 > 0.0f ? mk2(n.x * s, n.y * s) : mk2(0.0f, 0.0f);
    if (n.z < 0.0f) {
        float sx = res.x >= 0.0f ? 1.0f : -1.0f, sy = res.y >= 0.0f ? 1.0f : -1.0f;
        return mk2((1.0f - fabsf_(res.y)) * sx, (1.0f - fabsf_(res.x)) * sy);
    }
    return res;
}
FRT_HD void make_orthonormal_basis(f3 n, f3& tangent, f3& bitangent) {   // restir.wgsl:161-168
    float sign = n.z >= 0.0f ? 1.0f : -1.0f;
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    tangent = mk3(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    bitangent = mk3(b, sign + n.y * n.y * a, -n.y);
}
FRT_HD f3 fresnel_schlick(f3 f0, float v_dot_h) {   // :170
    return f0 + (1.0f - f0) * pow5_(clampf(1.0f - v_dot_h, 0.0f, 1.0f));
}
FRT_HD float reflectance(float cosine, float ref_idx) {   // :175
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * pow5_(1.0f - cosine);
}
FRT_HD float ndf_ggx(float n_dot_h, float roughness) {   // :182
    float a = roughness * roughness;
    float a2 = a * a;
    float d = n_dot_h * n_dot_h * (a2 - 1.0f) + 1.0f;
    return a2 / (kPI * d * d);
}
FRT_HD float geometry_schlick_ggx(float n_dot_v, float roughness) {   // :189
    float a2 = roughness * roughness;
    return 2.0f * n_dot_v / (n_dot_v + sqrtf_(a2 + (1.0f - a2) * n_dot_v * n_dot_v));
}
FRT_HD float geometry_smith(float n_dot_l, float n_dot_v, float roughness) {
    return geometry_schlick_ggx(n_dot_l, roughness) * geometry_schlick_ggx(n_dot_v, roughness);
}
FRT_HD f3 sample_ggx_vndf(f3 wo, float roughness, float ux, float uy) {   // :202-216
    float alpha = roughness * roughness;
    f3 Vh = normalize(mk3(alpha * wo.x, alpha * wo.y, wo.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    f3 T1 = lensq > 0.0f ? mk3(-Vh.y, Vh.x, 0.0f) * rsqrt_exact(lensq) : mk3(1.0f, 0.0f, 0.0f);
    f3 T2 = cross(Vh, T1);
    float r = sqrtf_(ux);
    float phi = 2.0f * kPI * uy;
    float sphi, cphi;
    sincosf_(phi, sphi, cphi);
    float t1 = r * cphi;
    float t2 = r * sphi;
    float s = 0.5f * (1.0f + Vh.z);
    float t2l = (1.0f - s) * sqrtf_(1.0f - t1 * t1) + s * t2;
    f3 Nh = t1 * T1 + t2l * T2 + sqrtf_(fmaxn(0.0f, 1.0f - t1 * t1 - t2l * t2l)) * Vh;
    return normalize(mk3(alpha * Nh.x, alpha * Nh.y, fmaxn(0.0f, Nh.z)));
}

struct MatParams {   // the fields of `mat` the BSDF code reads
    float roughness, metallic, transmission, ior;
};
struct LightSmp { f3 pos, normal; float pdf; f4 emission; };
struct BsdfSmp { f3 wi; float pdf; f3 weight; };

FRT_HD LightSmp sample_light(PathCtx& c, uint32_t light_idx) {   // :219-245
    const LightView& L = c.sc.lights[light_idx];
    LightSmp s;
    s.emission = mk4(L.emission[0], L.emission[1], L.emission[2], L.emission[3]);
    float r1 = c.rand();
    float r2 = c.rand();
    f3 lp = mk3(L.position[0], L.position[1], L.position[2]);
    if (L.type_ == 0u) {
        f3 lu = mk3(L.u[0], L.u[1], L.u[2]), lv = mk3(L.v[0], L.v[1], L.v[2]);
        float su = r1 * 2.0f - 1.0f;
        float sv = r2 * 2.0f - 1.0f;
        s.pos = lp + lu * su + lv * sv;
        s.normal = normalize(cross(lu, lv));
        s.pdf = 1.0f / L.area;
    } else {
        float z = 1.0f - 2.0f * r1;
        float r_xy = sqrtf_(fmaxn(0.0f, 1.0f - z * z));
        float phi = 2.0f * kPI * r2;
        float sp, cp;
        sincosf_(phi, sp, cp);
        f3 local_dir = mk3(r_xy * cp, r_xy * sp, z);
        s.pos = lp + local_dir * L.v[0];
        s.normal = local_dir;
        s.pdf = 1.0f / L.area;
    }
    return s;
}
FRT_HD float eval_pdf(f3 normal, f3 wi, f3 wo, const MatParams& m, f3 base_color) {   // :249-276
    float n_dot_l = dot(normal, wi);
    float n_dot_v = dot(normal, wo);
    if (m.transmission > 0.01f) return 0.0f;
    if (n_dot_l <= 0.0f || n_dot_v <= 0.0f) return 0.0f;
    f3 F0 = mix3(splat3(0.04f), base_color, m.metallic);
    f3 F = fresnel_schlick(F0, fmaxn(dot(normal, wo), 0.0f));
    float lum_spec = luminance(F);
    float lum_diff = luminance(base_color * (1.0f - m.metallic));
    float prob_spec = clampf(lum_spec / (lum_spec + lum_diff + 0.0001f), 0.001f, 0.999f);
    f3 h = normalize(wi + wo);
    float n_dot_h = fmaxn(dot(normal, h), 0.0f);
    float d = ndf_ggx(n_dot_h, m.roughness);
    float g1 = geometry_schlick_ggx(n_dot_v, m.roughness);
    float pdf_spec = (d * g1) / (4.0f * n_dot_v);
    float pdf_diff = fmaxn(n_dot_l, 0.0f) * kInvPI;
    return prob_spec * pdf_spec + (1.0f - prob_spec) * pdf_diff;
}
FRT_HD f3 eval_bsdf(f3 normal, f3 wi, f3 wo, const MatParams& m, f3 base_color) {   // :278-305
    float n_dot_l = dot(normal, wi);
    float n_dot_v = dot(normal, wo);
    if (m.transmission > 0.01f) return splat3(0.0f);
    if (n_dot_l <= 0.0f || n_dot_v <= 0.0f) return splat3(0.0f);
    f3 h = normalize(wi + wo);
    float n_dot_h = fmaxn(dot(normal, h), 0.0f);
    float h_dot_v = fmaxn(dot(h, wo), 0.0f);
    f3 F0 = mix3(splat3(0.04f), base_color, m.metallic);
    float D = ndf_ggx(n_dot_h, m.roughness);
    float G = geometry_smith(n_dot_l, n_dot_v, m.roughness);
    f3 F = fresnel_schlick(F0, h_dot_v);
    f3 specular = (D * G * F) / fmaxn(4.0f * n_dot_l * n_dot_v, 0.001f);
    f3 kD = (splat3(1.0f) - F) * (1.0f - m.metallic);
    f3 diffuse = kD * base_color / kPI;
    return diffuse + specular;
}
FRT_HD f3 random_unit_vector(PathCtx& c) {   // :143-150
    float z = c.rand() * 2.0f - 1.0f;
    float a = c.rand() * 2.0f * kPI;
    float r = sqrtf_(1.0f - z * z);
    float sa, ca;
    sincosf_(a, sa, ca);
    return mk3(r * ca, r * sa, z);
}
FRT_HD BsdfSmp sample_bsdf(PathCtx& c, f3 wo, f3 ffnormal, bool front_face, const MatParams& m, f3 base_color) {   // :307-371
    BsdfSmp s;
    if (m.transmission > 0.01f) {
        s.pdf = 0.0f;
        float refraction_ratio = front_face ? 1.0f / m.ior : m.ior;
        float cos_theta = fminn(dot(wo, ffnormal), 1.0f);
        float sin_theta = sqrtf_(1.0f - cos_theta * cos_theta);
        bool refl = refraction_ratio * sin_theta > 1.0f;
        if (!refl) refl = reflectance(cos_theta, refraction_ratio) > c.rand();   // short-circuit ||, :318
        s.wi = refl ? reflect(-wo, ffnormal) : refract(-wo, ffnormal, refraction_ratio);
        s.weight = base_color;
        return s;
    }
    f3 F0 = mix3(splat3(0.04f), base_color, m.metallic);
    f3 F_view = fresnel_schlick(F0, fmaxn(dot(ffnormal, wo), 0.0f));
    float lum_spec = luminance(F_view);
    float lum_diff = luminance(base_color * (1.0f - m.metallic));
    float prob_spec = clampf(lum_spec / (lum_spec + lum_diff + 0.0001f), 0.001f, 0.999f);
    float rnd = c.rand();
    if (rnd < prob_spec) {
        f3 tb, bt, n = ffnormal;
        make_orthonormal_basis(n, tb, bt);
        f3 wo_local = mk3(dot(tb, wo), dot(bt, wo), dot(n, wo));
        float ru = c.rand();
        float rv = c.rand();
        f3 wm_local = sample_ggx_vndf(wo_local, m.roughness, ru, rv);
        f3 wm = tb * wm_local.x + bt * wm_local.y + n * wm_local.z;
        s.wi = reflect(-wo, wm);
    } else {
        s.wi = normalize(ffnormal + random_unit_vector(c));
    }
    float n_dot_l = dot(ffnormal, s.wi);
    float n_dot_v = dot(ffnormal, wo);
    if (n_dot_l <= 0.0f || n_dot_v <= 0.0f) { s.weight = splat3(0.0f); s.pdf = 0.0f; return s; }
    f3 bsdf_val = eval_bsdf(ffnormal, s.wi, wo, m, base_color);
    s.pdf = eval_pdf(ffnormal, s.wi, wo, m, base_color);
    if (s.pdf > 0.0f) s.weight = bsdf_val * n_dot_l / s.pdf;
    else s.weight = splat3(0.0f);
    return s;
}

// Attribute fetch + interpolation for a committed hit: gbuffer.wgsl:124-174 and restir.wgsl:383-441
// Attribute fetch + interpolation for a committed hit: gbuffer.wgsl:124-174 and restir.wgsl:383-441.
// The reference walks instance -> mesh_infos -> indices -> attributes (four dependent loads) and decodes three octahedral normals
// per hit. Here the host flattens that chain once per triangle into a 128-byte shading record indexed by the flattened triangle id
// (decoded vertex normals, tangents, uvs, tangent sign, material id): one level of 16-byte loads, issued together with the
// instance's world_to_object. The decoded normals are produced by the very function below compiled for the host, so the
// values are bit-identical to decoding on the fly.
//   q0 (n0.xyz, uv0.x) q1 (n1.xyz, uv0.y) q2 (n2.xyz, uv1.x) q3 (t0.xyz, uv1.y) q4 (t1.xyz, uv2.x) q5 (t2.xyz, uv2.y) q6 (tangent_sign, mat_id, -, -)
struct HitGeom { f3 normal_w; f2 uv; uint32_t mat_id; float u, v; uint32_t tri, inst; };
FRT_HD HitGeom fetch_hit_geometry(const SceneView& sc, const HitRec& h) {
    const float4* rec = sc.shade_tris + (size_t)h.tri * 8u;
    float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q6 = rec[6];
    const InstanceView& in = sc.instances[h.inst];
    f3 n0 = mk3(q0.x, q0.y, q0.z), n1 = mk3(q1.x, q1.y, q1.z), n2 = mk3(q2.x, q2.y, q2.z);
    float4 q3 = rec[3], q4 = rec[4], q5 = rec[5];
    float u = h.u, v = h.v, w = 1.0f - u - v;
    f3 local_normal = normalize(n0 * w + n1 * u + n2 * v);
    HitGeom g;
    g.uv = mk2(q0.w, q1.w) * w + mk2(q2.w, q3.w) * u + mk2(q4.w, q5.w) * v;
    // v * mat3x3(w2o[0], w2o[1], w2o[2]) = (dot(v, col0), dot(v, col1), dot(v, col2))
    f3 c0 = mk3(in.w2o[0], in.w2o[1], in.w2o[2]), c1 = mk3(in.w2o[3], in.w2o[4], in.w2o[5]), c2 = mk3(in.w2o[6], in.w2o[7], in.w2o[8]);
    g.normal_w = normalize(mk3(dot(local_normal, c0), dot(local_normal, c1), dot(local_normal, c2)));
    g.mat_id = f2u(q6.y);
    g.u = u; g.v = v; g.tri = h.tri; g.inst = h.inst;
    return g;
}
// World-space tangent + sign of the hit: only needed under a normal map (gbuffer.wgsl:152-160, restir.wgsl:423-432), so it is
// evaluated lazily; the value is the one the reference computes unconditionally.
FRT_HD f4 hit_tangent(const SceneView& sc, const HitGeom& g) {
    const float4* rec = sc.shade_tris + (size_t)g.tri * 8u;
    float4 q3 = rec[3], q4 = rec[4], q5 = rec[5], q6 = rec[6];
    const InstanceView& in = sc.instances[g.inst];
    float w = 1.0f - g.u - g.v;
    f3 local_tangent = normalize(mk3(q3.x, q3.y, q3.z) * w + mk3(q4.x, q4.y, q4.z) * g.u + mk3(q5.x, q5.y, q5.z) * g.v);
    f3 c0 = mk3(in.w2o[0], in.w2o[1], in.w2o[2]), c1 = mk3(in.w2o[3], in.w2o[4], in.w2o[5]), c2 = mk3(in.w2o[6], in.w2o[7], in.w2o[8]);
    f3 tw = normalize(mk3(dot(local_tangent, c0), dot(local_tangent, c1), dot(local_tangent, c2)));
    return mk4(tw, q6.x);
}
FRT_HD f3 perturb_normal(f3 N_ff, f3 tangent_w, float tangent_sign, f3 nm) {   // gbuffer.wgsl:206-219, restir.wgsl:657-671
    f3 normal_local = normalize(nm * 2.0f - splat3(1.0f));
    f3 T_ff = normalize(tangent_w - N_ff * dot(N_ff, tangent_w));
    f3 B_ff = normalize(cross(N_ff, T_ff)) * tangent_sign;
    return normalize(T_ff * normal_local.x + B_ff * normal_local.y + N_ff * normal_local.z);
}

FRT_HD ReservoirView zero_reservoir() { ReservoirView r; r.y = 0u; r.w_sum = 0.0f; r.M = 0u; r.W = 0.0f; r.sx = r.sy = r.sz = 0.0f; r.p_hat = 0.0f; return r; }

// ================================================================================================ stage 0
// gbuffer.wgsl:91-255
template <class Ctx>
FRT_HD void gbuffer_pixel(Ctx& c, uint32_t px, uint32_t py) {
    const SceneView& sc = c.sc; const FrameView& fv = c.fv;
    uint32_t pix = py * fv.W + px;
    f2 size = mk2((float)fv.W, (float)fv.H);
    f2 uv = (mk2((float)px, (float)py) + mk2(0.5f, 0.5f)) / size;
    f2 ndc = mk2(uv.x * 2.0f - 1.0f, 1.0f - uv.y * 2.0f);
    m4 view_inv = load_m4(fv.cam.view_inverse), proj_inv = load_m4(fv.cam.proj_inverse);
    f3 origin = xyz(view_inv.c[3]);
    f4 target = mul(mul(view_inv, proj_inv), mk4(ndc.x, ndc.y, 1.0f, 1.0f));   // (view_inv * proj_inv) * v, :104
    f3 direction = normalize(xyz(target) / target.w - origin);
    HitRec h;
    c.closest(origin, direction, 0.001f, 1000.0f, h);
    if (h.tri == 0xFFFFFFFFu) {
        fv.gpos[pix] = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
        fv.gnormal[pix] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        fv.galbedo[pix] = pack_rgba8(mk4(0.0f, 0.0f, 0.0f, 1.0f));
        fv.gmotion[pix] = make_float2(0.0f, 0.0f);
        return;
    }
    HitGeom g = fetch_hit_geometry(sc, h);
    f3 ffnormal = h.front ? g.normal_w : -g.normal_w;
    f3 pos = origin + direction * h.t;
    const MaterialView& mat = sc.materials[g.mat_id];
    f4 tex_color = mk4(1.0f, 1.0f, 1.0f, 1.0f);
    uint32_t tex_id = mat.tex_info_0 & 0xFFFFu, normal_tex_id = mat.tex_info_0 >> 16u;
    if (tex_id != 65535u) tex_color = sample_layer<true>(sc, tex_id, g.uv);
    float occlusion = 1.0f;
    uint32_t occlusion_tex_id = mat.tex_info_1 & 0xFFFFu;
    if (occlusion_tex_id != 65535u) occlusion = sample_layer<false>(sc, occlusion_tex_id, g.uv).x;
    f3 final_normal = ffnormal;
    if (normal_tex_id != 65535u) {
        f3 nm = xyz(sample_layer<false>(sc, normal_tex_id, g.uv));
        f4 tg = hit_tangent(sc, g);
        final_normal = perturb_normal(ffnormal, xyz(tg), tg.w, nm);
    }
    f3 base_color = mk3(mat.base_color[0], mat.base_color[1], mat.base_color[2]) * xyz(tex_color) * occlusion;
    m4 view_proj = load_m4(fv.cam.view_proj), prev_view_proj = load_m4(fv.cam.prev_view_proj);
    f4 curr_clip = mul(view_proj, mk4(pos, 1.0f));
    f4 prev_clip = mul(prev_view_proj, mk4(pos, 1.0f));
    f2 curr_ndc = mk2(curr_clip.x / curr_clip.w, curr_clip.y / curr_clip.w);
    f2 prev_ndc = mk2(prev_clip.x / prev_clip.w, prev_clip.y / prev_clip.w);
    f2 curr_uv = curr_ndc * mk2(0.5f, -0.5f) + mk2(0.5f, 0.5f);
    f2 prev_uv = prev_ndc * mk2(0.5f, -0.5f) + mk2(0.5f, 0.5f);
    f2 motion = prev_uv - curr_uv;
    f2 en = encode_octahedral_normal(final_normal);
    fv.gpos[pix] = make_float4(pos.x, pos.y, pos.z, (float)g.mat_id);
    fv.gnormal[pix] = make_float4(en.x, en.y, g.uv.x, g.uv.y);
    fv.galbedo[pix] = pack_rgba8(mk4(base_color, 1.0f));
    fv.gmotion[pix] = make_float2(motion.x, motion.y);
}

// ================================================================================================ stage 3
// post.wgsl:61-282. The centre / neighbour colour and albedo are textureSampleLevel(raw_tex / albedo_tex, smp, uv + unjitter_offset)
// (post.wgsl:72-78, :97-109, :152-158) under the Repeat / Linear sampler of renderer.rs:240-249. Contract: with jitter == (0, 0)
// (the shipped reference, camera.rs:202-203) the sample point is the texel centre and the sample IS the texel (what a sampler's
// fixed-point weights give); otherwise f32 bilinear on texel centres with Repeat addressing, like sample_layer above.
FRT_HD float gauss(float x, float sigma) {   // post.wgsl:21-26 (sigma >= 0.001 at every call site)
    return expf_(-(x * x) * (1.0f / (2.0f * sigma * sigma)));   // x / c evaluated as x * (1 / c) (contract); sigma is a constant
}
FRT_HD f3 rgb_to_ycocg(f3 c) {
    return mk3(c.x * 0.25f + c.y * 0.5f + c.z * 0.25f, c.x * 0.5f + c.y * 0.0f + c.z * -0.5f, c.x * -0.25f + c.y * 0.5f + c.z * -0.25f);
}
FRT_HD f3 ycocg_to_rgb(f3 c) { return mk3(c.x + c.y - c.z, c.x + c.z, c.x - c.y - c.z); }
FRT_HD f3 resolve_tonemap(f3 c) { return c / (1.0f + fmaxn(c.x, fmaxn(c.y, c.z))); }
FRT_HD f3 resolve_inverse_tonemap(f3 c) { return c / (1.0f - fmaxn(c.x, fmaxn(c.y, c.z))); }

// What the 5x5 bilateral and 3x3 variance loops read of a neighbour pixel. GlobalTaps decodes it from the per-pixel buffers in
// HBM on every tap (host check, reference form); the post kernel stages a 20x20 tile of already decoded values in LDS instead
// (frt_kernels.hip: TileTaps), so each pixel is decoded once per workgroup rather than 25 times. Same functions, same values.
struct TapData { f3 color, albedo, normal, pos; };
FRT_HD TapData decode_tap(const FrameView& fv, uint32_t nidx) {
    TapData t;
    t.color = xyz(unpack_rgba16f(fv.raw[nidx]));
    t.albedo = xyz(unpack_rgba8(fv.galbedo[nidx]));
    float4 sn = fv.gnormal[nidx];
    t.normal = decode_octahedral_normal(sn.x, sn.y);
    float4 sp4 = fv.gpos[nidx];
    t.pos = mk3(sp4.x, sp4.y, sp4.z);
    return t;
}
struct GlobalTaps {
    const FrameView& fv;
    FRT_HD TapData get(int nx, int ny) const { return decode_tap(fv, (uint32_t)ny * fv.W + (uint32_t)nx); }
    FRT_HD f3 color(int nx, int ny) const { return xyz(unpack_rgba16f(fv.raw[(uint32_t)ny * fv.W + (uint32_t)nx])); }
};
// jitter != 0: bilinear taps of the radiance and albedo targets at uv + unjitter_offset; normal and position stay texel loads
// (post.wgsl:80-83, :111-113). Straight from HBM: the jittered path is dead in the shipped reference and is not tuned.
struct JitterTaps {
    const FrameView& fv;
    struct Foot { uint32_t i00, i10, i01, i11; float ax, ay; };
    FRT_HD Foot foot(int nx, int ny) const {
        f2 size = mk2((float)fv.W, (float)fv.H);
        f2 uv = (mk2((float)nx, (float)ny) + mk2(0.5f, 0.5f)) / size;
        f2 unjitter_offset = mk2(-fv.jitter_x, fv.jitter_y) * 0.5f;   // post.wgsl:73
        f2 sample_uv = uv + unjitter_offset;
        float x = sample_uv.x * size.x - 0.5f, y = sample_uv.y * size.y - 0.5f;
        float fx = floorf_(x), fy = floorf_(y);
        int W = (int)fv.W, H = (int)fv.H;
        int x0 = (((int)fx % W) + W) % W, y0 = (((int)fy % H) + H) % H;   // AddressMode::Repeat
        int x1 = (x0 + 1) % W, y1 = (y0 + 1) % H;
        Foot f;
        f.i00 = (uint32_t)y0 * fv.W + (uint32_t)x0; f.i10 = (uint32_t)y0 * fv.W + (uint32_t)x1;
        f.i01 = (uint32_t)y1 * fv.W + (uint32_t)x0; f.i11 = (uint32_t)y1 * fv.W + (uint32_t)x1;
        f.ax = x - fx; f.ay = y - fy;
        return f;
    }
    FRT_HD static f3 lerp4(f3 t00, f3 t10, f3 t01, f3 t11, float ax, float ay) {
        f3 top = t00 * (1.0f - ax) + t10 * ax;
        f3 bot = t01 * (1.0f - ax) + t11 * ax;
        return top * (1.0f - ay) + bot * ay;
    }
    FRT_HD f3 raw_at(uint32_t i) const { return xyz(unpack_rgba16f(fv.raw[i])); }
    FRT_HD f3 albedo_at(uint32_t i) const { return xyz(unpack_rgba8(fv.galbedo[i])); }
    FRT_HD f3 color(int nx, int ny) const {
        const Foot f = foot(nx, ny);
        return lerp4(raw_at(f.i00), raw_at(f.i10), raw_at(f.i01), raw_at(f.i11), f.ax, f.ay);
    }
    FRT_HD TapData get(int nx, int ny) const {
        const Foot f = foot(nx, ny);
        TapData t;
        t.color = lerp4(raw_at(f.i00), raw_at(f.i10), raw_at(f.i01), raw_at(f.i11), f.ax, f.ay);
        t.albedo = lerp4(albedo_at(f.i00), albedo_at(f.i10), albedo_at(f.i01), albedo_at(f.i11), f.ax, f.ay);
        const uint32_t nidx = (uint32_t)ny * fv.W + (uint32_t)nx;
        float4 sn = fv.gnormal[nidx];
        t.normal = decode_octahedral_normal(sn.x, sn.y);
        float4 sp4 = fv.gpos[nidx];
        t.pos = mk3(sp4.x, sp4.y, sp4.z);
        return t;
    }
};

template <class Taps>
FRT_HD void post_pixel_t(const FrameView& fv, uint32_t px, uint32_t py, const Taps& taps) {
    int W = (int)fv.W, H = (int)fv.H;
    uint32_t idx = py * fv.W + px;
    const TapData ctr = taps.get((int)px, (int)py);
    f3 center_color = ctr.color, center_albedo = ctr.albedo, center_normal = ctr.normal, center_pos = ctr.pos;
    f3 sum_color = splat3(0.0f);
    float sum_weight = 0.0f;
    // fully unrolled: dx, dy become literals and the 25 spatial weights fold to constants (same IEEE operations, evaluated by the compiler)
#pragma unroll
    for (int dy = -2; dy <= 2; dy++) {
#pragma unroll
        for (int dx = -2; dx <= 2; dx++) {
            int nx = (int)px + dx, ny = (int)py + dy;
            if (nx < 0 || ny < 0 || nx >= W || ny >= H) continue;
            const TapData t = taps.get(nx, ny);
            float w_spatial = gauss(length2(mk2((float)dx, (float)dy)), 1.5f);
            // equal albedo (every tap on a surface of one material: most of them): gauss(length(0)) = exp2(-0) = 1 exactly (frt_math.hpp: the
            // polynomial of exp2f_ at -0 is 1), so the square root and the exponential are skipped — by the whole wave when all its lanes agree
            const f3 d_albedo = t.albedo - center_albedo;
            float w_color = 1.0f;
            if (!(d_albedo.x == 0.0f && d_albedo.y == 0.0f && d_albedo.z == 0.0f)) w_color = gauss(length(d_albedo), 0.2f);
            float dot_normal = clampf(dot(center_normal, t.normal), 0.0f, 1.0f);
            float w_normal = pow20_(dot_normal);
            float w_pos = gauss(length(t.pos - center_pos), 0.1f);
            float weight = w_spatial * w_color * w_normal * w_pos;
            sum_color = sum_color + t.color * weight;
            sum_weight += weight;
        }
    }
    f3 filtered_color = center_color;
    if (sum_weight > 0.001f) filtered_color = sum_color / sum_weight;
    f3 m1 = splat3(0.0f), m2 = splat3(0.0f);
    f3 tm_filtered = resolve_tonemap(filtered_color);
    for (int dy = -1; dy <= 1; dy++) {
        for (int dx = -1; dx <= 1; dx++) {
            int nx = (int)px + dx, ny = (int)py + dy;
            f3 s_col = filtered_color;
            if (nx >= 0 && ny >= 0 && nx < W && ny < H) s_col = taps.color(nx, ny);
            f3 s_ycocg = rgb_to_ycocg(resolve_tonemap(s_col));
            m1 = m1 + s_ycocg;
            m2 = m2 + s_ycocg * s_ycocg;
        }
    }
    m1 = m1 / 9.0f; m2 = m2 / 9.0f;
    f3 var = max3(splat3(0.0f), m2 - m1 * m1);
    f3 sigma = mk3(sqrtf_(var.x), sqrtf_(var.y), sqrtf_(var.z));
    f3 c_min = m1 - sigma * 1.2f;
    f3 c_max = m1 + sigma * 1.2f;
    f3 history_color = tm_filtered;
    bool valid_history = false;
    f2 structure_motion = mk2(0.0f, 0.0f);
    if (fv.frame_count > 0u) {
        float2 mv = fv.gmotion[idx];
        structure_motion = mk2(mv.x, mv.y);
        f2 size = mk2((float)fv.W, (float)fv.H);
        f2 uv = (mk2((float)px, (float)py) + mk2(0.5f, 0.5f)) / size;
        f2 prev_uv = uv + structure_motion;
        f2 prev_pos = prev_uv * size - mk2(0.5f, 0.5f);
        float fpx = floorf_(prev_pos.x), fpy = floorf_(prev_pos.y);
        int p0x = (int)fpx, p0y = (int)fpy;
        float fx = prev_pos.x - fpx, fy = prev_pos.y - fpy;
        if (prev_uv.x >= 0.0f && prev_uv.y >= 0.0f && prev_uv.x <= 1.0f && prev_uv.y <= 1.0f) {
            f3 c4[4];
            for (int k = 0; k < 4; ++k) {
                int x = p0x + (k & 1), y = p0y + (k >> 1);
                c4[k] = splat3(0.0f);
                if (x >= 0 && y >= 0 && x < W && y < H) {
                    if ((uint32_t)y + 1u < fv.prev_y0 || (uint32_t)y > fv.prev_y1) note_halo_overflow(fv);   // history rows: prev rows +- 1
                    float4 hv = fv.history[(uint32_t)y * fv.W + (uint32_t)x];
                    c4[k] = resolve_tonemap(mk3(hv.x, hv.y, hv.z));
                }
            }
            f3 c01 = mix3(c4[0], c4[1], fx);
            f3 c23 = mix3(c4[2], c4[3], fx);
            history_color = mix3(c01, c23, fy);
            valid_history = true;
        }
    }
    f3 final_tm = tm_filtered;
    if (valid_history) {
        f3 clamped_history = ycocg_to_rgb(clamp3(rgb_to_ycocg(history_color), c_min, c_max));
        f2 motion_px = structure_motion * mk2((float)fv.W, (float)fv.H);
        float speed = length2(motion_px);
        if (speed < 0.5f) {
            float accum_blend = 1.0f - (1.0f / (float)(fv.frame_count + 1u));
            final_tm = mix3(tm_filtered, history_color, clampf(accum_blend, 0.0f, 1.0f));
        } else {
            float dynamic_feedback = mixf(0.98f, 0.85f, smoothstepf(0.0f, 2.0f, speed));
            final_tm = mix3(tm_filtered, clamped_history, dynamic_feedback);
        }
    }
    f3 final_color = max3(splat3(0.0f), resolve_inverse_tonemap(final_tm));
    fv.accum[idx] = make_float4(final_color.x, final_color.y, final_color.z, 1.0f);
    const float inv_gamma = (float)(1.0 / 2.2);
    fv.display[idx] = pack_rgba8(mk4(powf_(final_color.x, inv_gamma), powf_(final_color.y, inv_gamma), powf_(final_color.z, inv_gamma), 1.0f));
}
FRT_HD bool post_is_jittered(const FrameView& fv) { return fv.jitter_x != 0.0f || fv.jitter_y != 0.0f; }
FRT_HD void post_pixel(const FrameView& fv, uint32_t px, uint32_t py) {
    if (post_is_jittered(fv)) { JitterTaps taps{fv}; post_pixel_t(fv, px, py, taps); }
    else { GlobalTaps taps{fv}; post_pixel_t(fv, px, py, taps); }
}

} // namespace frt
