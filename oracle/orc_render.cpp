// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path. PARITY UNPINNED (see orc_math.hpp).
// Line-by-line scalar restatement of the four compute shaders of the reference:
//   src/shaders/gbuffer.wgsl, restir.wgsl, restir_spatial.wgsl, post.wgsl
// with the storage formats of src/renderer.rs:52-170 (rgba8unorm albedo, rgba16float radiance, rgba8unorm display)
// and the ping-pong wiring of src/passes/{gbuffer,restir,restir_spatial,post}.rs.
#include "orc_render.hpp"
#include <thread>
#include <algorithm>

namespace orc {

static const float PI = 3.14159265359f;   // restir.wgsl:4

// ------------------------------------------------------------------ texture sampling
// textureSampleLevel(..., level 0) with the sampler of renderer.rs:240-249 (Repeat, Linear). Filter weights of
// the hardware are unpinned; fixed here as plain f32 bilinear on texel centres. sRGB layers decode before filtering.
static vec4 texel(const std::vector<uint8_t>& tex, const float* lut, int x, int y) {
    const uint8_t* p = &tex[((size_t)y * 1024u + (size_t)x) * 4u];
    if (lut) return V4(lut[p[0]], lut[p[1]], lut[p[2]], (float)p[3] / 255.0f);
    return V4((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, (float)p[3] / 255.0f);
}
static vec4 sample_tex(const std::vector<uint8_t>& tex, const float* lut, vec2 uv) {
    float x = uv.x * 1024.0f - 0.5f, y = uv.y * 1024.0f - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float ax = x - fx, ay = y - fy;
    int x0 = (int)fx & 1023, y0 = (int)fy & 1023, x1 = (x0 + 1) & 1023, y1 = (y0 + 1) & 1023;
    vec4 t00 = texel(tex, lut, x0, y0), t10 = texel(tex, lut, x1, y0);
    vec4 t01 = texel(tex, lut, x0, y1), t11 = texel(tex, lut, x1, y1);
    vec4 top = t00 * (1.0f - ax) + t10 * ax;
    vec4 bot = t01 * (1.0f - ax) + t11 * ax;
    return top * (1.0f - ay) + bot * ay;
}

static inline vec4 unpack_rgba8(uint32_t p) {
    return V4(unorm8_to_f32((uint8_t)(p & 0xff)), unorm8_to_f32((uint8_t)((p >> 8) & 0xff)),
              unorm8_to_f32((uint8_t)((p >> 16) & 0xff)), unorm8_to_f32((uint8_t)(p >> 24)));
}
static inline uint32_t pack_rgba8(vec4 c) {
    return (uint32_t)f32_to_unorm8(c.x) | ((uint32_t)f32_to_unorm8(c.y) << 8) | ((uint32_t)f32_to_unorm8(c.z) << 16) |
           ((uint32_t)f32_to_unorm8(c.w) << 24);
}
static inline uint64_t pack_rgba16f(vec4 c) {
    return (uint64_t)f32_to_f16(c.x) | ((uint64_t)f32_to_f16(c.y) << 16) | ((uint64_t)f32_to_f16(c.z) << 32) |
           ((uint64_t)f32_to_f16(c.w) << 48);
}
static inline vec4 unpack_rgba16f(uint64_t p) {
    return V4(f16_to_f32((uint16_t)(p & 0xffff)), f16_to_f32((uint16_t)((p >> 16) & 0xffff)),
              f16_to_f32((uint16_t)((p >> 32) & 0xffff)), f16_to_f32((uint16_t)(p >> 48)));
}

// gbuffer.wgsl:38-44 / restir.wgsl:152-158 / post.wgsl:28-34
static vec3 decode_octahedral_normal(vec2 e) {
    vec3 n = V3(e.x, e.y, 1.0f - fabsf(e.x) - fabsf(e.y));
    float t = fmax_(-n.z, 0.0f);
    n.x += (n.x >= 0.0f) ? -t : t;
    n.y += (n.y >= 0.0f) ? -t : t;
    return normalize(n);
}
// gbuffer.wgsl:46-62
static vec2 encode_octahedral_normal_dev(vec3 n) {
    float l1 = fabsf(n.x) + fabsf(n.y) + fabsf(n.z);
    float s = 1.0f / fmax_(l1, 1e-6f);
    vec2 res_base = V2(n.x * s, n.y * s);
    vec2 res = l1 > 0.0f ? res_base : V2(0, 0);
    if (n.z < 0.0f) {
        float x = res.x, y = res.y;
        float sx = x >= 0.0f ? 1.0f : -1.0f, sy = y >= 0.0f ? 1.0f : -1.0f;
        return V2((1.0f - fabsf(y)) * sx, (1.0f - fabsf(x)) * sy);
    }
    return res;
}
static inline float luminance(vec3 c) { return c.x * 0.2126f + c.y * 0.7152f + c.z * 0.0722f; }   // restir.wgsl:742-744

// restir.wgsl:132-136
static inline uint32_t pcg_hash(uint32_t input) {
    uint32_t state = input * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
// restir.wgsl:781-786
static inline float rand_lcg(uint32_t& state) {
    uint32_t old = state;
    state = old * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (float)((word >> 22u) ^ word) / 4294967296.0f;   // literal 4294967295.0 rounds to 2^32 in f32
}

struct HitInfo {   // restir.wgsl:81-90
    vec3 pos, normal, ffnormal; vec2 uv; uint32_t mat_id; bool front_face; float t; vec4 tangent;
};
struct BsdfSample { vec3 wi; float pdf; vec3 weight; bool is_delta; };
struct LightSample { vec3 pos, normal; float pdf; vec4 emission; };
struct PathResult { vec3 radiance; bool valid_v1; vec3 v1_pos, v1_normal; };

struct Ctx {
    const Renderer& R;
    const Scene& S;
    Tracer tracer;
    const CameraUniform& cam;
    TraceStats st;
    uint32_t rng_seed = 0;      // var<private> rng_seed (restir.wgsl:130)
    int variant = 0;            // 0 = restir.wgsl, 1 = restir_spatial.wgsl
    uint32_t cur;               // frame_count % 2
    Ctx(const Renderer& r, const CameraUniform& c) : R(r), S(*r.scene), tracer(*r.scene, r.use_bvh), cam(c), cur(r.frame_count % 2) {}

    float rand() { rng_seed = pcg_hash(rng_seed); return (float)rng_seed / 4294967296.0f; }   // restir.wgsl:138-141
    vec3 random_unit_vector() {   // restir.wgsl:143-150
        float z = rand() * 2.0f - 1.0f;
        float a = rand() * 2.0f * PI;
        float r = sqrtf(1.0f - z * z);
        float x = r * cos_(a);
        float y = r * sin_(a);
        return V3(x, y, z);
    }
    vec4 sample_color(uint32_t layer, vec2 uv) const { return sample_tex(S.color_textures[layer], S.srgb_lut, uv); }
    vec4 sample_data(uint32_t layer, vec2 uv) const { return sample_tex(S.data_textures[layer], nullptr, uv); }
};

// restir.wgsl:161-168
static void make_orthonormal_basis(vec3 n, vec3& tangent, vec3& bitangent) {
    float sign = n.z >= 0.0f ? 1.0f : -1.0f;
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    tangent = V3(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    bitangent = V3(b, sign + n.y * n.y * a, -n.y);
}
static vec3 fresnel_schlick(vec3 f0, float v_dot_h) {   // :170-172
    return f0 + (1.0f - f0) * pow5_(clamp_(1.0f - v_dot_h, 0.0f, 1.0f));
}
static float reflectance(float cosine, float ref_idx) {   // :175-180
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * pow5_(1.0f - cosine);
}
static float ndf_ggx(float n_dot_h, float roughness) {   // :182-187
    float a = roughness * roughness;
    float a2 = a * a;
    float d = n_dot_h * n_dot_h * (a2 - 1.0f) + 1.0f;
    return a2 / (PI * d * d);
}
static float geometry_schlick_ggx(float n_dot_v, float roughness) {   // :189-196
    float a2 = roughness * roughness;
    return 2.0f * n_dot_v / (n_dot_v + sqrtf(a2 + (1.0f - a2) * n_dot_v * n_dot_v));
}
static float geometry_smith(float n_dot_l, float n_dot_v, float roughness) {   // :198-200
    return geometry_schlick_ggx(n_dot_l, roughness) * geometry_schlick_ggx(n_dot_v, roughness);
}
static vec3 sample_ggx_vndf(vec3 wo, float roughness, vec2 u) {   // :202-216
    float alpha = roughness * roughness;
    vec3 Vh = normalize(V3(alpha * wo.x, alpha * wo.y, wo.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    vec3 T1 = lensq > 0.0f ? V3(-Vh.y, Vh.x, 0.0f) * inversesqrt_(lensq) : V3(1.0f, 0.0f, 0.0f);
    vec3 T2 = cross(Vh, T1);
    float r = sqrtf(u.x);
    float phi = 2.0f * PI * u.y;
    float t1 = r * cos_(phi);
    float t2 = r * sin_(phi);
    float s = 0.5f * (1.0f + Vh.z);
    float t2_lerp = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
    vec3 Nh = t1 * T1 + t2_lerp * T2 + sqrtf(fmax_(0.0f, 1.0f - t1 * t1 - t2_lerp * t2_lerp)) * Vh;
    return normalize(V3(alpha * Nh.x, alpha * Nh.y, fmax_(0.0f, Nh.z)));
}
static LightSample sample_light(Ctx& c, uint32_t light_idx) {   // :219-245
    const LightUniform& light = c.S.lights[light_idx];
    LightSample smp{};
    smp.emission = V4(light.emission[0], light.emission[1], light.emission[2], light.emission[3]);
    float r1 = c.rand();
    float r2 = c.rand();
    vec3 lpos = V3(light.position[0], light.position[1], light.position[2]);
    vec3 lu = V3(light.u[0], light.u[1], light.u[2]), lv = V3(light.v[0], light.v[1], light.v[2]);
    if (light.type_ == 0u) {
        float su = r1 * 2.0f - 1.0f;
        float sv = r2 * 2.0f - 1.0f;
        smp.pos = lpos + lu * su + lv * sv;
        smp.normal = normalize(cross(lu, lv));
        smp.pdf = 1.0f / light.area;
    } else {
        float z = 1.0f - 2.0f * r1;
        float r_xy = sqrtf(fmax_(0.0f, 1.0f - z * z));
        float phi = 2.0f * PI * r2;
        float x = r_xy * cos_(phi);
        float y = r_xy * sin_(phi);
        vec3 local_dir = V3(x, y, z);
        smp.pos = lpos + local_dir * light.v[0];
        smp.normal = local_dir;
        smp.pdf = 1.0f / light.area;
    }
    return smp;
}
static float eval_pdf(vec3 normal, vec3 wi, vec3 wo, const Material& mat, vec3 base_color) {   // :249-276
    float n_dot_l = dot(normal, wi);
    float n_dot_v = dot(normal, wo);
    if (mat.transmission > 0.01f) return 0.0f;
    if (n_dot_l <= 0.0f || n_dot_v <= 0.0f) return 0.0f;
    vec3 F0 = mix3(V3(0.04f), base_color, mat.metallic);
    vec3 F = fresnel_schlick(F0, fmax_(dot(normal, wo), 0.0f));
    float lum_spec = luminance(F);
    float lum_diff = luminance(base_color * (1.0f - mat.metallic));
    float prob_spec = clamp_(lum_spec / (lum_spec + lum_diff + 0.0001f), 0.001f, 0.999f);
    vec3 h = normalize(wi + wo);
    float n_dot_h = fmax_(dot(normal, h), 0.0f);
    float d = ndf_ggx(n_dot_h, mat.roughness);
    float g1 = geometry_schlick_ggx(n_dot_v, mat.roughness);
    float pdf_spec = (d * g1) / (4.0f * n_dot_v);
    float pdf_diff = text_mode() ? fmax_(n_dot_l, 0.0f) / PI : fmax_(n_dot_l, 0.0f) * (1.0f / PI);   // x / PI evaluated as x * (1 / PI) (contract)
    return prob_spec * pdf_spec + (1.0f - prob_spec) * pdf_diff;
}
static vec3 eval_bsdf(vec3 normal, vec3 wi, vec3 wo, const Material& mat, vec3 base_color) {   // :278-305
    float n_dot_l = dot(normal, wi);
    float n_dot_v = dot(normal, wo);
    if (mat.transmission > 0.01f) return V3(0.0f);
    if (n_dot_l <= 0.0f || n_dot_v <= 0.0f) return V3(0.0f);
    vec3 h = normalize(wi + wo);
    float n_dot_h = fmax_(dot(normal, h), 0.0f);
    float h_dot_v = fmax_(dot(h, wo), 0.0f);
    vec3 F0 = mix3(V3(0.04f), base_color, mat.metallic);
    float D = ndf_ggx(n_dot_h, mat.roughness);
    float G = geometry_smith(n_dot_l, n_dot_v, mat.roughness);
    vec3 F = fresnel_schlick(F0, h_dot_v);
    vec3 specular = (D * G * F) / fmax_(4.0f * n_dot_l * n_dot_v, 0.001f);
    vec3 kD = (V3(1.0f) - F) * (1.0f - mat.metallic);
    vec3 diffuse = kD * base_color / PI;
    return diffuse + specular;
}
static BsdfSample sample_bsdf(Ctx& c, vec3 wo, const HitInfo& hit, const Material& mat, vec3 base_color) {   // :307-371
    BsdfSample smp{};
    smp.is_delta = false;
    if (mat.transmission > 0.01f) {
        smp.is_delta = true;
        smp.pdf = 0.0f;
        float refraction_ratio = hit.front_face ? 1.0f / mat.ior : mat.ior;
        float cos_theta = fmin_(dot(wo, hit.ffnormal), 1.0f);
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        // short-circuit ||: rand() is only consumed when the first operand is false (:318)
        if (refraction_ratio * sin_theta > 1.0f || reflectance(cos_theta, refraction_ratio) > c.rand()) {
            smp.wi = reflect(-wo, hit.ffnormal);
        } else {
            smp.wi = refract(-wo, hit.ffnormal, refraction_ratio);
        }
        smp.weight = base_color;
        return smp;
    }
    vec3 F0 = mix3(V3(0.04f), base_color, mat.metallic);
    vec3 F_view = fresnel_schlick(F0, fmax_(dot(hit.ffnormal, wo), 0.0f));
    float lum_spec = luminance(F_view);
    float lum_diff = luminance(base_color * (1.0f - mat.metallic));
    float prob_spec = clamp_(lum_spec / (lum_spec + lum_diff + 0.0001f), 0.001f, 0.999f);
    float rnd = c.rand();
    if (rnd < prob_spec) {
        vec3 tb, bt;
        make_orthonormal_basis(hit.ffnormal, tb, bt);
        vec3 n = hit.ffnormal;
        vec3 wo_local = V3(dot(tb, wo), dot(bt, wo), dot(n, wo));   // transpose(tbn) * wo
        float ru = c.rand();
        float rv = c.rand();
        vec3 wm_local = sample_ggx_vndf(wo_local, mat.roughness, V2(ru, rv));
        vec3 wm = tb * wm_local.x + bt * wm_local.y + n * wm_local.z;   // tbn * wm_local
        smp.wi = reflect(-wo, wm);
    } else {
        smp.wi = normalize(hit.ffnormal + c.random_unit_vector());
    }
    float n_dot_l = dot(hit.ffnormal, smp.wi);
    float n_dot_v = dot(hit.ffnormal, wo);
    if (n_dot_l <= 0.0f || n_dot_v <= 0.0f) {
        smp.weight = V3(0.0f);
        smp.pdf = 0.0f;
        return smp;
    }
    vec3 bsdf_val = eval_bsdf(hit.ffnormal, smp.wi, wo, mat, base_color);
    smp.pdf = eval_pdf(hit.ffnormal, smp.wi, wo, mat, base_color);
    if (smp.pdf > 0.0f) smp.weight = bsdf_val * n_dot_l / smp.pdf;
    else smp.weight = V3(0.0f);
    return smp;
}
// restir.wgsl:375-381 (variant 0) vs restir_spatial.wgsl:380-400 (variant 1)
static bool trace_shadow_ray(Ctx& c, vec3 origin, vec3 dir, float dist) {
    float t_max = fmax_(dist * 0.999f, 0.0f);
    if (c.variant == 0) return !c.tracer.any(origin, dir, 0.001f, t_max, c.st);
    float t_min = 0.0001f;
    if (t_min >= t_max) return true;
    return !c.tracer.any(origin, dir, t_min, t_max, c.st);
}
// restir.wgsl:383-441 — attribute fetch/interpolation for a committed intersection
static HitInfo reconstruct_geometry_hit(const Ctx& c, const Hit& h, vec3 ray_origin, vec3 ray_dir) {
    const Scene& S = c.S;
    HitInfo hit{};
    const Instance& in = S.instances[S.tri_instance[h.tri]];
    uint32_t prim_idx = h.tri - in.first_tri;
    hit.mat_id = in.mat_id;
    const MeshInfo& mi = S.mesh_infos[in.mesh_id];
    uint32_t idx_offset = mi.index_offset + prim_idx * 3u;
    uint32_t i0 = S.indices[idx_offset + 0u] + mi.vertex_offset;
    uint32_t i1 = S.indices[idx_offset + 1u] + mi.vertex_offset;
    uint32_t i2 = S.indices[idx_offset + 2u] + mi.vertex_offset;
    const VertexAttributes &v0 = S.attributes[i0], &v1 = S.attributes[i1], &v2 = S.attributes[i2];
    vec3 n0 = decode_octahedral_normal(V2(v0.normal[0], v0.normal[1]));
    vec3 n1 = decode_octahedral_normal(V2(v1.normal[0], v1.normal[1]));
    vec3 n2 = decode_octahedral_normal(V2(v2.normal[0], v2.normal[1]));
    vec3 t0 = V3(v0.tangent[0], v0.tangent[1], v0.tangent[2]);
    vec3 t1 = V3(v1.tangent[0], v1.tangent[1], v1.tangent[2]);
    vec3 t2 = V3(v2.tangent[0], v2.tangent[1], v2.tangent[2]);
    float u = h.u, v = h.v, w = 1.0f - u - v;
    vec3 local_normal = normalize(n0 * w + n1 * u + n2 * v);
    vec3 local_tangent = normalize(t0 * w + t1 * u + t2 * v);
    vec2 uv_interp = V2(v0.uv[0], v0.uv[1]) * w + V2(v1.uv[0], v1.uv[1]) * u + V2(v2.uv[0], v2.uv[1]) * v;
    // m_inv = mat3x3f(w2o[0], w2o[1], w2o[2]); v * m_inv = (dot(v, col0), dot(v, col1), dot(v, col2))
    vec3 c0 = V3(in.w2o[0], in.w2o[1], in.w2o[2]), c1 = V3(in.w2o[3], in.w2o[4], in.w2o[5]), c2 = V3(in.w2o[6], in.w2o[7], in.w2o[8]);
    hit.normal = normalize(V3(dot(local_normal, c0), dot(local_normal, c1), dot(local_normal, c2)));
    vec3 tangent_w = normalize(V3(dot(local_tangent, c0), dot(local_tangent, c1), dot(local_tangent, c2)));
    hit.tangent = V4(tangent_w, v0.tangent[3]);
    hit.uv = uv_interp;
    hit.front_face = h.front;
    hit.ffnormal = hit.front_face ? hit.normal : -hit.normal;
    hit.t = h.t;
    hit.pos = ray_origin + ray_dir * hit.t;
    return hit;
}
// shared by gbuffer.wgsl:206-219 and restir.wgsl:657-671
static vec3 perturb_normal(vec3 N_ff, vec3 tangent_w, float tangent_sign, vec3 normal_map_rgb) {
    vec3 normal_local = normalize(normal_map_rgb * 2.0f - V3(1.0f));
    vec3 T_ff = normalize(tangent_w - N_ff * dot(N_ff, tangent_w));
    vec3 B_ff = normalize(cross(N_ff, T_ff)) * tangent_sign;
    return normalize(T_ff * normal_local.x + B_ff * normal_local.y + N_ff * normal_local.z);
}
static vec3 eval_direct_lighting(Ctx& c, const HitInfo& hit, vec3 wo, const Material& mat, vec3 base_color,
                                 const LightSample& ls, float weight) {   // restir.wgsl:443-459
    vec3 offset_pos = hit.pos + hit.ffnormal * 0.001f;
    vec3 L = normalize(ls.pos - offset_pos);
    float dist = distance(ls.pos, offset_pos);
    float n_dot_l = fmax_(dot(hit.ffnormal, L), 0.0f);
    float l_dot_n = fmax_(dot(-L, ls.normal), 0.0f);
    if (n_dot_l > 0.0f && l_dot_n > 0.0f) {
        if (trace_shadow_ray(c, offset_pos, L, dist)) {
            vec3 f = eval_bsdf(hit.ffnormal, L, wo, mat, base_color);
            float G = (n_dot_l * l_dot_n) / (dist * dist);
            return xyz(ls.emission) * ls.emission.w * f * G * weight;
        }
    }
    return V3(0.0f);
}
static vec3 nee(Ctx& c, const HitInfo& hit, vec3 wo, const Material& mat, vec3 base_color, vec3 throughput) {
    // restir.wgsl:558-571 and :707-720 (identical blocks)
    vec3 add = V3(0.0f);
    uint32_t nl = c.cam.num_lights;
    if (nl > 0u) {
        uint32_t light_idx = (uint32_t)(c.rand() * (float)nl);
        if (light_idx < nl) {
            LightSample ls = sample_light(c, light_idx);
            float pdf_nee = ls.pdf * (1.0f / (float)nl);
            float p_bsdf = eval_pdf(hit.ffnormal, normalize(ls.pos - hit.pos), wo, mat, base_color);
            float mis_weight_nee = pdf_nee / (pdf_nee + p_bsdf);
            float weight = mis_weight_nee / pdf_nee;
            add = eval_direct_lighting(c, hit, wo, mat, base_color, ls, weight) * throughput;
        }
    }
    return add;
}

// restir.wgsl:460-737 (variant 0) / restir_spatial.wgsl:480-762 (variant 1)
static PathResult trace_path(Ctx& c, int cx, int cy, uint32_t seed) {
    const Renderer& R = c.R; const Scene& S = c.S;
    c.rng_seed = seed;
    PathResult result{}; result.radiance = V3(0.0f); result.valid_v1 = false; result.v1_pos = V3(0.0f); result.v1_normal = V3(0.0f);
    size_t pix = (size_t)cy * R.W + (size_t)cx;
    vec4 pos_w = R.gpos[c.cur][pix];
    if (pos_w.w < 0.0f) return result;
    vec4 normal_w = R.gnormal[c.cur][pix];
    vec4 albedo_raw = unpack_rgba8(R.galbedo[c.cur][pix]);

    HitInfo hit{};
    hit.pos = xyz(pos_w);
    hit.normal = decode_octahedral_normal(V2(normal_w.x, normal_w.y));
    hit.front_face = true;
    hit.ffnormal = hit.normal;
    hit.uv = V2(normal_w.z, normal_w.w);

    Material mat{};
    uint32_t mat_id = (uint32_t)(pos_w.w + 0.1f);
    uint32_t nmat = (uint32_t)S.materials.size();
    if (mat_id < nmat) {
        mat = S.materials[mat_id];
        if (c.variant == 0 || mat.transmission < 0.01f) {   // restir.wgsl:494 vs restir_spatial.wgsl:514-516
            mat.base_color[0] = albedo_raw.x; mat.base_color[1] = albedo_raw.y; mat.base_color[2] = albedo_raw.z; mat.base_color[3] = 1.0f;
        }
    } else {
        mat.base_color[0] = albedo_raw.x; mat.base_color[1] = albedo_raw.y; mat.base_color[2] = albedo_raw.z; mat.base_color[3] = 1.0f;
        mat.roughness = 0.0f; mat.metallic = albedo_raw.w; mat.ior = 1.0f; mat.light_index = -1;
    }
    uint32_t mr_tex_id = mat.tex_info_2 & 0xFFFFu;
    if (mr_tex_id != 65535u) {
        vec4 mr = c.sample_data(mr_tex_id, hit.uv);
        mat.metallic = mr.z * mat.metallic;
        mat.roughness = mr.y * mat.roughness;
    }
    vec3 base_color = V3(mat.base_color[0], mat.base_color[1], mat.base_color[2]);
    vec3 accumulated_color = V3(0.0f);
    vec3 throughput = V3(1.0f);
    vec3 wo = normalize(V3(c.cam.view_pos[0], c.cam.view_pos[1], c.cam.view_pos[2]) - hit.pos);
    uint32_t emissive_tex_id = mat.tex_info_1 >> 16u;
    vec3 emissive_factor = V3(mat.emissive_factor[0], mat.emissive_factor[1], mat.emissive_factor[2]);

    if (mat_id < nmat) {   // :523-533
        if (mat.light_index == -1) {
            vec3 emission = emissive_factor;
            if (emissive_tex_id != 65535u) emission = emission * xyz(c.sample_color(emissive_tex_id, hit.uv));
            accumulated_color += emission;
        }
    }
    vec3 next_dir = V3(0.0f);
    float last_bsdf_pdf = 0.0f;
    bool previous_was_diffuse = false;

    if (mat.light_index >= 0) {   // :543-552
        vec3 emission = emissive_factor;
        if (emissive_tex_id != 65535u) emission = emission * xyz(c.sample_color(emissive_tex_id, hit.uv));
        accumulated_color += emission;
        result.radiance = accumulated_color;
        return result;
    }
    bool is_glass = mat.transmission > 0.01f;
    {
        bool is_specular = is_glass || (mat.roughness < 0.05f);   // :556
        if (!is_specular) {
            accumulated_color += nee(c, hit, wo, mat, base_color, throughput);
            previous_was_diffuse = true;
        } else previous_was_diffuse = false;
    }
    BsdfSample sc = sample_bsdf(c, wo, hit, mat, base_color);
    if (sc.weight.x <= 0.0f && sc.weight.y <= 0.0f && sc.weight.z <= 0.0f) {
        result.radiance = accumulated_color;
        return result;
    }
    last_bsdf_pdf = sc.pdf;
    throughput *= sc.weight;
    next_dir = sc.wi;

    for (uint32_t depth = 1u; depth < R.max_depth; depth++) {   // :590
        if (depth >= 3u) {
            float p = fmax_(throughput.x, fmax_(throughput.y, throughput.z));
            float survival_prob = clamp_(p, 0.05f, 0.95f);
            if (c.rand() > survival_prob) break;
            throughput /= survival_prob;
        }
        vec3 offset_dir = hit.ffnormal * sign_(dot(hit.ffnormal, next_dir));
        vec3 origin = hit.pos + offset_dir * 0.001f;
        Hit h = c.tracer.closest(origin, next_dir, 0.001f, 100.0f, c.st);
        if (!h.hit) break;
        hit = reconstruct_geometry_hit(c, h, origin, next_dir);
        if (depth == 1u) { result.valid_v1 = true; result.v1_pos = hit.pos; result.v1_normal = hit.normal; }
        wo = -next_dir;
        mat = S.materials[hit.mat_id];
        vec4 tex_color = V4(1.0f, 1.0f, 1.0f, 1.0f);
        uint32_t tex_id = mat.tex_info_0 & 0xFFFFu, normal_tex_id = mat.tex_info_0 >> 16u;
        if (tex_id != 65535u) tex_color = c.sample_color(tex_id, hit.uv);
        float occlusion = 1.0f;
        uint32_t occlusion_tex_id = mat.tex_info_1 & 0xFFFFu;
        uint32_t emissive_tex_id_b = mat.tex_info_1 >> 16u;
        if (occlusion_tex_id != 65535u) occlusion = c.sample_data(occlusion_tex_id, hit.uv).x;
        base_color = V3(mat.base_color[0], mat.base_color[1], mat.base_color[2]) * xyz(tex_color) * occlusion;
        if (normal_tex_id != 65535u) {   // :657-671
            vec3 nm = xyz(c.sample_data(normal_tex_id, hit.uv));
            hit.ffnormal = perturb_normal(hit.ffnormal, xyz(hit.tangent), hit.tangent.w, nm);
        }
        if (mat.light_index == -1 && emissive_tex_id_b != 65535u) {   // :675-678
            vec3 emissive_col = xyz(c.sample_color(emissive_tex_id_b, hit.uv));
            accumulated_color += emissive_col * throughput;
        }
        if (mat.light_index >= 0) {   // :683-700
            if (hit.front_face) {
                const LightUniform& light = S.lights[mat.light_index];
                vec3 Le = V3(light.emission[0], light.emission[1], light.emission[2]) * light.emission[3];
                float mis_weight = 1.0f;
                if (previous_was_diffuse) {
                    float dist_sq = hit.t * hit.t;
                    float light_cos = fmax_(dot(hit.ffnormal, -wo), 0.0f);
                    float p_bsdf = last_bsdf_pdf;
                    float p_nee = (1.0f / light.area) * (dist_sq / light_cos) * (1.0f / (float)c.cam.num_lights);
                    if (light_cos > 0.001f) mis_weight = p_bsdf / (p_bsdf + p_nee);
                    else mis_weight = 0.0f;
                }
                accumulated_color += Le * throughput * mis_weight;
            }
            break;
        }
        {
            bool is_specular = is_glass || (mat.roughness < 0.05f);   // :705 — uses the PRIMARY hit's is_glass
            if (!is_specular) {
                accumulated_color += nee(c, hit, wo, mat, base_color, throughput);
                previous_was_diffuse = true;
            } else previous_was_diffuse = false;
        }
        BsdfSample sb = sample_bsdf(c, wo, hit, mat, base_color);
        if (sb.weight.x <= 0.0f && sb.weight.y <= 0.0f && sb.weight.z <= 0.0f) break;
        last_bsdf_pdf = sb.pdf;
        throughput *= sb.weight;
        next_dir = sb.wi;
    }
    result.radiance = accumulated_color;
    return result;
}

// restir.wgsl:746-756
static bool update_reservoir(Reservoir& r, uint32_t seed_cand, float w, float rnd, uint32_t cnt, float p_hat_new, vec3 s_path_new) {
    r.w_sum += w;
    r.M += cnt;
    if (rnd * r.w_sum < w) {
        r.y = seed_cand; r.p_hat = p_hat_new;
        r.s_path[0] = s_path_new.x; r.s_path[1] = s_path_new.y; r.s_path[2] = s_path_new.z;
        return true;
    }
    return false;
}

// ------------------------------------------------------------------ stage 0: gbuffer.wgsl:91-255
static void gbuffer_pixel(Ctx& c, Renderer& R, uint32_t px, uint32_t py) {
    const Scene& S = c.S; const CameraUniform& cam = c.cam;
    size_t pix = (size_t)py * R.W + px;
    vec2 size = V2((float)R.W, (float)R.H);
    vec2 uv = (V2((float)px, (float)py) + V2(0.5f, 0.5f)) / size;
    vec2 ndc = V2(uv.x * 2.0f - 1.0f, 1.0f - uv.y * 2.0f);
    mat4 view_inv, proj_inv, view_proj, prev_view_proj;
    auto load = [](mat4& m, const float* p) { for (int k = 0; k < 4; ++k) m.c[k] = V4(p[4 * k], p[4 * k + 1], p[4 * k + 2], p[4 * k + 3]); };
    load(view_inv, cam.view_inverse); load(proj_inv, cam.proj_inverse); load(view_proj, cam.view_proj); load(prev_view_proj, cam.prev_view_proj);
    vec3 origin = xyz(view_inv.c[3]);
    vec4 target_pos = mul(mul(view_inv, proj_inv), V4(ndc.x, ndc.y, 1.0f, 1.0f));   // (view_inv * proj_inv) * v, :104
    vec3 direction = normalize(xyz(target_pos) / target_pos.w - origin);
    Hit h = c.tracer.closest(origin, direction, 0.001f, 1000.0f, c.st);
    uint32_t wi = c.cur;
    if (!h.hit) {
        R.gpos[wi][pix] = V4(0, 0, 0, -1.0f);
        R.gnormal[wi][pix] = V4(0, 0, 0, 0);
        R.galbedo[wi][pix] = pack_rgba8(V4(0, 0, 0, 1.0f));
        R.gmotion[pix] = V2(0, 0);
        return;
    }
    const Instance& in = S.instances[S.tri_instance[h.tri]];
    uint32_t mesh_id = in.mesh_id, mat_id = in.mat_id;
    uint32_t prim = h.tri - in.first_tri;
    const MeshInfo& mi = S.mesh_infos[mesh_id];
    uint32_t idx_offset = mi.index_offset + prim * 3u;
    uint32_t i0 = S.indices[idx_offset + 0u] + mi.vertex_offset;
    uint32_t i1 = S.indices[idx_offset + 1u] + mi.vertex_offset;
    uint32_t i2 = S.indices[idx_offset + 2u] + mi.vertex_offset;
    const VertexAttributes &v0 = S.attributes[i0], &v1 = S.attributes[i1], &v2 = S.attributes[i2];
    vec3 n0 = decode_octahedral_normal(V2(v0.normal[0], v0.normal[1]));
    vec3 n1 = decode_octahedral_normal(V2(v1.normal[0], v1.normal[1]));
    vec3 n2 = decode_octahedral_normal(V2(v2.normal[0], v2.normal[1]));
    vec3 t0 = V3(v0.tangent[0], v0.tangent[1], v0.tangent[2]);
    vec3 t1 = V3(v1.tangent[0], v1.tangent[1], v1.tangent[2]);
    vec3 t2 = V3(v2.tangent[0], v2.tangent[1], v2.tangent[2]);
    float u_bary = h.u, v_bary = h.v, w_bary = 1.0f - u_bary - v_bary;
    vec3 local_normal = normalize(n0 * w_bary + n1 * u_bary + n2 * v_bary);
    vec3 local_tangent = normalize(t0 * w_bary + t1 * u_bary + t2 * v_bary);
    float tangent_sign = v0.tangent[3];
    vec3 c0 = V3(in.w2o[0], in.w2o[1], in.w2o[2]), c1 = V3(in.w2o[3], in.w2o[4], in.w2o[5]), c2 = V3(in.w2o[6], in.w2o[7], in.w2o[8]);
    vec3 normal_w = normalize(V3(dot(local_normal, c0), dot(local_normal, c1), dot(local_normal, c2)));
    vec3 tangent_w = normalize(V3(dot(local_tangent, c0), dot(local_tangent, c1), dot(local_tangent, c2)));
    vec3 ffnormal = h.front ? normal_w : -normal_w;
    vec3 pos = origin + direction * h.t;
    const Material& mat = S.materials[mat_id];
    vec2 tex_uv = V2(v0.uv[0], v0.uv[1]) * w_bary + V2(v1.uv[0], v1.uv[1]) * u_bary + V2(v2.uv[0], v2.uv[1]) * v_bary;
    vec4 tex_color = V4(1, 1, 1, 1);
    uint32_t tex_id = mat.tex_info_0 & 0xFFFFu, normal_tex_id = mat.tex_info_0 >> 16u;
    if (tex_id != 65535u) tex_color = c.sample_color(tex_id, tex_uv);
    float occlusion = 1.0f;
    uint32_t occlusion_tex_id = mat.tex_info_1 & 0xFFFFu;
    if (occlusion_tex_id != 65535u) occlusion = c.sample_data(occlusion_tex_id, tex_uv).x;
    vec3 final_normal = ffnormal;
    if (normal_tex_id != 65535u) {
        vec3 nm = xyz(c.sample_data(normal_tex_id, tex_uv));
        final_normal = perturb_normal(ffnormal, tangent_w, tangent_sign, nm);
    }
    vec3 base_color = V3(mat.base_color[0], mat.base_color[1], mat.base_color[2]) * xyz(tex_color) * occlusion;
    vec4 curr_clip = mul(view_proj, V4(pos, 1.0f));
    vec4 prev_clip = mul(prev_view_proj, V4(pos, 1.0f));
    vec2 curr_ndc = V2(curr_clip.x / curr_clip.w, curr_clip.y / curr_clip.w);
    vec2 prev_ndc = V2(prev_clip.x / prev_clip.w, prev_clip.y / prev_clip.w);
    vec2 curr_uv = curr_ndc * V2(0.5f, -0.5f) + V2(0.5f, 0.5f);
    vec2 prev_uv = prev_ndc * V2(0.5f, -0.5f) + V2(0.5f, 0.5f);
    vec2 motion = prev_uv - curr_uv;
    R.gpos[wi][pix] = V4(pos, (float)mat_id);
    vec2 en = encode_octahedral_normal_dev(final_normal);
    R.gnormal[wi][pix] = V4(en.x, en.y, tex_uv.x, tex_uv.y);
    R.galbedo[wi][pix] = pack_rgba8(V4(base_color, 1.0f));
    R.gmotion[pix] = motion;
}

// ------------------------------------------------------------------ stage 1: restir.wgsl:788-918
static bool is_valid_neighbor_temporal(vec3 curr_pos, vec3 curr_normal, uint32_t curr_mat, vec3 prev_pos, vec3 prev_normal,
                                       uint32_t prev_mat, vec3 camera_pos) {   // restir.wgsl:758-778
    if (curr_mat != prev_mat) return false;
    if (dot(curr_normal, prev_normal) < 0.99f) return false;
    float dist_diff_sq = dot(curr_pos - prev_pos, curr_pos - prev_pos);
    float dist_to_camera_sq = dot(curr_pos - camera_pos, curr_pos - camera_pos);
    float threshold = fmax_(0.00001f, dist_to_camera_sq * 0.001f);
    if (dist_diff_sq > threshold) return false;
    return true;
}
static void temporal_pixel(Ctx& c, Renderer& R, uint32_t px, uint32_t py) {
    const Scene& S = c.S; const CameraUniform& cam = c.cam;
    c.variant = 0;
    uint32_t pixel_idx = px + py * R.W;
    uint32_t seed_base = pixel_idx + cam.frame_count * 927163u;
    uint32_t seed_candidate = pcg_hash(seed_base);
    uint32_t local_seed = seed_base;
    uint32_t cur = c.cur, prv = cur ^ 1u;
    vec4 pos_w = R.gpos[cur][pixel_idx];
    std::vector<Reservoir>& curr_res = R.reservoirs[0];
    const std::vector<Reservoir>& prev_res = R.reservoirs[1];
    if (pos_w.w < 0.0f) { curr_res[pixel_idx] = Reservoir{}; return; }
    Reservoir r{};
    PathResult path_result = trace_path(c, (int)px, (int)py, seed_candidate);
    float p_hat = luminance(path_result.radiance);
    update_reservoir(r, seed_candidate, p_hat, 0.5f, 1u, p_hat, path_result.v1_pos);
    r.W = p_hat > 0.0f ? 1.0f : 0.0f;

    vec2 motion = R.gmotion[pixel_idx];
    vec2 size = V2((float)R.W, (float)R.H);
    vec2 uv = (V2((float)px, (float)py) + V2(0.5f, 0.5f)) / size;
    vec2 prev_uv = uv + motion;
    const uint32_t MAX_RESERVOIR_M_TEMPORAL = 16u;
    if (prev_uv.x >= 0.0f && prev_uv.x <= 1.0f && prev_uv.y >= 0.0f && prev_uv.y <= 1.0f) {
        vec2 pf = prev_uv * size;
        uint32_t pxx = (uint32_t)pf.x, pyy = (uint32_t)pf.y;   // vec2u(): truncation; non-negative here
        // prev_uv == 1.0 would index one past the edge; WGSL robust access returns zeros / clamps. Treated as zeros.
        bool inb = pxx < R.W && pyy < R.H;
        uint32_t prev_pixel_idx = pyy * R.W + pxx;
        vec4 prev_pos_data = inb ? R.gpos[prv][prev_pixel_idx] : V4(0, 0, 0, 0);
        vec4 prev_normal_data = inb ? R.gnormal[prv][prev_pixel_idx] : V4(0, 0, 0, 0);
        vec3 prev_normal = decode_octahedral_normal(V2(prev_normal_data.x, prev_normal_data.y));
        uint32_t prev_mat_id = (uint32_t)(prev_pos_data.w + 0.1f);
        vec4 curr_normal_data = R.gnormal[cur][pixel_idx];
        vec3 curr_normal = decode_octahedral_normal(V2(curr_normal_data.x, curr_normal_data.y));
        uint32_t curr_mat_id = (uint32_t)(pos_w.w + 0.1f);
        const Material& mat = S.materials[curr_mat_id];
        bool is_specular = mat.roughness < 0.2f || mat.metallic > 0.8f || (mat.transmission > 0.01f);
        vec3 campos = V3(cam.view_pos[0], cam.view_pos[1], cam.view_pos[2]);
        if (is_valid_neighbor_temporal(xyz(pos_w), curr_normal, curr_mat_id, xyz(prev_pos_data), prev_normal, prev_mat_id, campos) &&
            !is_specular) {
            Reservoir prev_r = inb ? prev_res[prev_pixel_idx] : Reservoir{};
            vec3 curr_albedo = xyz(unpack_rgba8(R.galbedo[cur][pixel_idx]));
            vec3 prev_albedo = inb ? xyz(unpack_rgba8(R.galbedo[prv][prev_pixel_idx])) : V3(0.0f);
            float l_curr = luminance(curr_albedo) + 0.001f;
            float l_prev = luminance(prev_albedo) + 0.001f;
            float albedo_ratio = l_curr / l_prev;
            if (albedo_ratio < 3.0f && albedo_ratio > 0.33f) {
                float p_hat_new = prev_r.p_hat * albedo_ratio;
                if (p_hat_new > 0.0f) {
                    uint32_t clamped_M = std::min(prev_r.M, MAX_RESERVOIR_M_TEMPORAL);
                    float w_prev = p_hat_new * prev_r.W * (float)clamped_M;
                    update_reservoir(r, prev_r.y, w_prev, rand_lcg(local_seed), clamped_M, p_hat_new,
                                     V3(prev_r.s_path[0], prev_r.s_path[1], prev_r.s_path[2]));
                }
            }
        }
    }
    float p_hat_final = r.p_hat;
    if (p_hat_final > 0.0f) r.W = (1.0f / p_hat_final) * (r.w_sum / (float)r.M);
    else { r.W = 0.0f; r.p_hat = 0.0f; }
    curr_res[pixel_idx] = r;
}

// ------------------------------------------------------------------ stage 2: restir_spatial.wgsl:857-1016
static bool is_valid_neighbor_spatial(const Scene& S, vec3 curr_pos, vec3 curr_normal, uint32_t curr_mat_id, vec3 prev_pos,
                                      vec3 prev_normal, uint32_t prev_mat_id, vec3 camera_pos) {   // restir_spatial.wgsl:783-814
    if (curr_mat_id != prev_mat_id) return false;
    const Material& mat = S.materials[curr_mat_id];
    bool is_specular = mat.roughness < 0.2f || mat.metallic > 0.8f || (mat.transmission > 0.01f);
    if (is_specular) {
        if (dot(curr_normal, prev_normal) < 0.998f) return false;
        float dist_diff = distance(curr_pos, prev_pos);
        if (dist_diff > 0.01f) return false;
    } else {
        if (dot(curr_normal, prev_normal) < 0.995f) return false;
        float dist_to_camera_sq = dot(curr_pos - camera_pos, curr_pos - camera_pos);
        float threshold = fmax_(0.00001f, dist_to_camera_sq * 0.001f);
        float dist_diff_sq = dot(curr_pos - prev_pos, curr_pos - prev_pos);
        if (dist_diff_sq > threshold) return false;
    }
    return true;
}
static float calculate_jacobian(vec3 curr_pos, vec3 curr_normal, vec3 curr_albedo, vec3 neighbor_v1_pos, vec3 neighbor_pos,
                                vec3 neighbor_normal, vec3 neighbor_albedo) {   // restir_spatial.wgsl:822-854
    vec3 dir_curr = neighbor_v1_pos - curr_pos;
    float cos_curr = fmax_(dot(curr_normal, normalize(dir_curr)), 0.0f);
    vec3 dir_neigh = neighbor_v1_pos - neighbor_pos;
    float cos_neigh = fmax_(dot(neighbor_normal, normalize(dir_neigh)), 0.0f);
    if (cos_neigh <= 0.001f) return 0.0f;
    float jacobian = cos_curr / cos_neigh;
    float lum_curr = luminance(curr_albedo) + 0.001f;
    float lum_neigh = luminance(neighbor_albedo) + 0.001f;
    jacobian *= (lum_curr / lum_neigh);
    jacobian = clamp_(jacobian, 0.1f, 10.0f);
    return jacobian;
}
static void spatial_pixel(Ctx& c, Renderer& R, uint32_t px, uint32_t py) {
    const Scene& S = c.S; const CameraUniform& cam = c.cam;
    c.variant = 1;
    uint32_t pixel_idx = py * R.W + px;
    uint32_t seed_init = py * R.W + px + R.frame_count * 0x12345678u;   // scene_info.y == frame_count (restir_spatial.rs execute)
    uint32_t local_seed = seed_init;
    uint32_t cur = c.cur;
    const std::vector<Reservoir>& in_res = R.reservoirs[0];
    std::vector<Reservoir>& out_res = R.reservoirs[1];
    vec4 pos_w = R.gpos[cur][pixel_idx];
    if (pos_w.w < 0.0f) {
        out_res[pixel_idx] = Reservoir{};
        R.raw[pixel_idx] = pack_rgba16f(V4(0, 0, 0, 0));
        return;
    }
    vec4 normal_w = R.gnormal[cur][pixel_idx];
    vec3 normal = decode_octahedral_normal(V2(normal_w.x, normal_w.y));
    uint32_t mat_id = (uint32_t)(pos_w.w + 0.1f);
    vec3 albedo = xyz(unpack_rgba8(R.galbedo[cur][pixel_idx]));
    Reservoir r = in_res[pixel_idx];
    if (r.M > 20u) { r.w_sum *= 20.0f / (float)r.M; r.M = 20u; }
    vec3 camera_pos = V3(cam.view_pos[0], cam.view_pos[1], cam.view_pos[2]);
    uint32_t num_neighbors = 5u;
    float radius = 10.0f;
    const Material& mat = S.materials[mat_id];
    if (mat.roughness < 0.1f || mat.metallic > 0.9f || mat.transmission > 0.1f) { num_neighbors = 3u; radius = 4.0f; }
    for (uint32_t i = 0u; i < num_neighbors; i++) {
        float r1 = rand_lcg(local_seed);
        float r2 = rand_lcg(local_seed);
        float angle = 2.0f * PI * r1;
        float rad = sqrtf(r2) * radius;
        vec2 offset = V2(cos_(angle), sin_(angle)) * rad;
        int nx = (int)px + (int)offset.x, ny = (int)py + (int)offset.y;   // vec2<i32>(offset): truncation toward zero
        if (nx < 0 || nx >= (int)R.W || ny < 0 || ny >= (int)R.H) continue;
        uint32_t neighbor_idx = (uint32_t)ny * R.W + (uint32_t)nx;
        vec4 n_pos_w = R.gpos[cur][neighbor_idx];
        if (n_pos_w.w < 0.0f) continue;
        vec4 n_normal_w = R.gnormal[cur][neighbor_idx];
        vec3 n_normal = decode_octahedral_normal(V2(n_normal_w.x, n_normal_w.y));
        uint32_t n_mat_id = (uint32_t)(n_pos_w.w + 0.1f);
        vec3 n_albedo = xyz(unpack_rgba8(R.galbedo[cur][neighbor_idx]));
        if (!is_valid_neighbor_spatial(S, xyz(pos_w), normal, mat_id, xyz(n_pos_w), n_normal, n_mat_id, camera_pos)) continue;
        Reservoir neighbor_r = in_res[neighbor_idx];
        if (neighbor_r.p_hat <= 0.0f) continue;
        vec3 n_s_path = V3(neighbor_r.s_path[0], neighbor_r.s_path[1], neighbor_r.s_path[2]);
        float jacobian = calculate_jacobian(xyz(pos_w), normal, albedo, n_s_path, xyz(n_pos_w), n_normal, n_albedo);
        bool is_specular = mat.roughness < 0.1f || mat.metallic > 0.9f || mat.transmission > 0.1f;
        if (is_specular) { if (jacobian < 0.5f || jacobian > 2.0f) continue; }
        vec3 dir_to_v1 = n_s_path - xyz(pos_w);
        float dist_to_v1 = length(dir_to_v1);
        bool visible = false;
        if (dot(normal, dir_to_v1) > 0.0f) {
            if (dist_to_v1 > 0.001f) {
                vec3 origin = xyz(pos_w);
                vec3 ray_dir = normalize(dir_to_v1);
                float t_max = fmax_(dist_to_v1, 0.0f);
                if (trace_shadow_ray(c, origin, ray_dir, t_max)) visible = true;
            } else visible = false;
        }
        if (!visible) continue;
        float p_hat_corrected = neighbor_r.p_hat * jacobian;
        uint32_t M_new = std::min(neighbor_r.M, 20u);
        float weight = p_hat_corrected * neighbor_r.W * (float)M_new;
        update_reservoir(r, neighbor_r.y, weight, rand_lcg(local_seed), M_new, p_hat_corrected, n_s_path);
    }
    PathResult final_res = trace_path(c, (int)px, (int)py, r.y);
    vec3 final_color = V3(0.0f);
    float p_hat_final = luminance(final_res.radiance);
    r.s_path[0] = final_res.v1_pos.x; r.s_path[1] = final_res.v1_pos.y; r.s_path[2] = final_res.v1_pos.z;
    if (p_hat_final > 0.0f) {
        float w_unclamped = (1.0f / p_hat_final) * (r.w_sum / (float)r.M);
        r.W = clamp_(w_unclamped, 0.0f, 20.0f);
        final_color = final_res.radiance * r.W;
        r.p_hat = p_hat_final;
    } else { r.W = 0.0f; r.p_hat = 0.0f; }
    out_res[pixel_idx] = r;
    R.raw[pixel_idx] = pack_rgba16f(V4(final_color, 1.0f));
}

// ------------------------------------------------------------------ stage 3: post.wgsl:61-282
static float gauss(float x, float sigma) {   // post.wgsl:21-26
    if (sigma < 0.001f) return fabsf(x) < 0.001f ? 1.0f : 0.0f;
    if (text_mode()) return exp_(-(x * x) / (2.0f * sigma * sigma));
    return exp_(-(x * x) * (1.0f / (2.0f * sigma * sigma)));   // x / c evaluated as x * (1 / c) (contract)
}
static vec3 rgb_to_ycocg(vec3 rgb) {
    return V3(rgb.x * 0.25f + rgb.y * 0.5f + rgb.z * 0.25f, rgb.x * 0.5f + rgb.y * 0.0f + rgb.z * -0.5f,
              rgb.x * -0.25f + rgb.y * 0.5f + rgb.z * -0.25f);
}
static vec3 ycocg_to_rgb(vec3 c) { float y = c.x, co = c.y, cg = c.z; return V3(y + co - cg, y + cg, y - co - cg); }
static vec3 resolve_tonemap(vec3 c) { return c / (1.0f + fmax_(c.x, fmax_(c.y, c.z))); }
static vec3 resolve_inverse_tonemap(vec3 c) { return c / (1.0f - fmax_(c.x, fmax_(c.y, c.z))); }

// textureSampleLevel(raw_tex | albedo_tex, smp, uv + unjitter_offset, 0).rgb of post.wgsl:72-78, :97-109, :152-158 at the sample
// point of pixel (nx, ny). Sampler of renderer.rs:240-249: Linear, Repeat. jitter == (0, 0) (the shipped reference, camera.rs:202-203):
// the sample point is the texel centre, the sample is that texel (contract: DESIGN.md §3). Otherwise an f32 bilinear blend of the four
// texels around it, in the order written here, wrapping at the image border.
struct PostSampler {
    const Renderer& R; uint32_t cur; bool jittered; vec2 unjitter_offset;
    PostSampler(const Renderer& r, uint32_t cur_) : R(r), cur(cur_) {
        jittered = R.jitter[0] != 0.0f || R.jitter[1] != 0.0f;
        unjitter_offset = V2(-R.jitter[0], R.jitter[1]) * 0.5f;   // post.wgsl:73
    }
    vec3 raw_texel(int x, int y) const { return xyz(unpack_rgba16f(R.raw[(uint32_t)y * R.W + (uint32_t)x])); }
    vec3 albedo_texel(int x, int y) const { return xyz(unpack_rgba8(R.galbedo[cur][(uint32_t)y * R.W + (uint32_t)x])); }
    template <class Texel> vec3 sample(int nx, int ny, Texel texel) const {
        if (!jittered) return texel(nx, ny);
        int W = (int)R.W, H = (int)R.H;
        vec2 size = V2((float)R.W, (float)R.H);
        vec2 uv = (V2((float)nx, (float)ny) + V2(0.5f, 0.5f)) / size;
        vec2 sample_uv = uv + unjitter_offset;
        float x = sample_uv.x * size.x - 0.5f, y = sample_uv.y * size.y - 0.5f;
        float fx = floorf(x), fy = floorf(y);
        float ax = x - fx, ay = y - fy;
        int x0 = ((int)fx % W + W) % W, y0 = ((int)fy % H + H) % H;
        int x1 = (x0 + 1) % W, y1 = (y0 + 1) % H;
        vec3 top = texel(x0, y0) * (1.0f - ax) + texel(x1, y0) * ax;
        vec3 bot = texel(x0, y1) * (1.0f - ax) + texel(x1, y1) * ax;
        return top * (1.0f - ay) + bot * ay;
    }
    vec3 color(int nx, int ny) const { return sample(nx, ny, [this](int x, int y) { return raw_texel(x, y); }); }
    vec3 albedo(int nx, int ny) const { return sample(nx, ny, [this](int x, int y) { return albedo_texel(x, y); }); }
};

static void post_pixel(Ctx& c, Renderer& R, uint32_t px, uint32_t py) {
    uint32_t cur = c.cur;
    const std::vector<vec4>& history = R.accum[cur ^ 1u];   // post.rs:209-224
    std::vector<vec4>& accumulation = R.accum[cur];
    int W = (int)R.W, H = (int)R.H;
    uint32_t idx = py * R.W + px;
    const PostSampler smp(R, cur);
    vec3 center_color = smp.color((int)px, (int)py);
    vec3 center_albedo = smp.albedo((int)px, (int)py);
    vec4 cn = R.gnormal[cur][idx];
    vec3 center_normal = decode_octahedral_normal(V2(cn.x, cn.y));
    vec3 center_pos = xyz(R.gpos[cur][idx]);
    vec3 sum_color = V3(0.0f);
    float sum_weight = 0.0f;
    const float sigma_spatial = 1.5f, sigma_color = 0.2f, sigma_pos = 0.1f;
    const int kernel_radius = 2;
    for (int dy = -kernel_radius; dy <= kernel_radius; dy++) {
        for (int dx = -kernel_radius; dx <= kernel_radius; dx++) {
            int nx = (int)px + dx, ny = (int)py + dy;
            if (nx < 0 || ny < 0 || nx >= W || ny >= H) continue;
            uint32_t nidx = (uint32_t)ny * R.W + (uint32_t)nx;
            vec3 sample_color = smp.color(nx, ny);
            vec3 sample_albedo = smp.albedo(nx, ny);
            vec4 sn = R.gnormal[cur][nidx];
            vec3 sample_normal = decode_octahedral_normal(V2(sn.x, sn.y));
            vec3 sample_pos = xyz(R.gpos[cur][nidx]);
            float dist_spatial = length2(V2((float)dx, (float)dy));
            float w_spatial = gauss(dist_spatial, sigma_spatial);
            float dist_color = length(sample_albedo - center_albedo);
            float w_color = gauss(dist_color, sigma_color);
            float dot_normal = clamp_(dot(center_normal, sample_normal), 0.0f, 1.0f);
            float w_normal = pow20_(dot_normal);
            float dist_pos = length(sample_pos - center_pos);
            float w_pos = gauss(dist_pos, sigma_pos);
            float weight = w_spatial * w_color * w_normal * w_pos;
            sum_color += sample_color * weight;
            sum_weight += weight;
        }
    }
    vec3 filtered_color = center_color;
    if (sum_weight > 0.001f) filtered_color = sum_color / sum_weight;

    vec3 m1 = V3(0.0f), m2 = V3(0.0f);
    vec3 tm_filtered = resolve_tonemap(filtered_color);
    for (int dy = -1; dy <= 1; dy++) {
        for (int dx = -1; dx <= 1; dx++) {
            int nx = (int)px + dx, ny = (int)py + dy;
            vec3 s_col;
            if (nx >= 0 && ny >= 0 && nx < W && ny < H) s_col = smp.color(nx, ny);
            else s_col = filtered_color;
            vec3 s_ycocg = rgb_to_ycocg(resolve_tonemap(s_col));
            m1 += s_ycocg;
            m2 += s_ycocg * s_ycocg;
        }
    }
    m1 /= 9.0f; m2 /= 9.0f;
    vec3 var = max3(V3(0.0f), m2 - m1 * m1);
    vec3 sigma = V3(sqrtf(var.x), sqrtf(var.y), sqrtf(var.z));
    const float gamma = 1.2f;
    vec3 c_min = m1 - sigma * gamma;
    vec3 c_max = m1 + sigma * gamma;

    vec3 history_color = tm_filtered;
    bool valid_history = false;
    vec2 structure_motion = V2(0, 0);
    if (R.frame_count > 0u) {
        structure_motion = R.gmotion[idx];
        vec2 size = V2((float)R.W, (float)R.H);
        vec2 uv = (V2((float)px, (float)py) + V2(0.5f, 0.5f)) / size;
        vec2 prev_uv = uv + structure_motion;
        vec2 prev_pos = prev_uv * size - V2(0.5f, 0.5f);
        float fpx = floorf(prev_pos.x), fpy = floorf(prev_pos.y);
        int p0x = (int)fpx, p0y = (int)fpy;
        vec2 f = V2(prev_pos.x - fpx, prev_pos.y - fpy);   // fract
        if (prev_uv.x >= 0.0f && prev_uv.y >= 0.0f && prev_uv.x <= 1.0f && prev_uv.y <= 1.0f) {
            auto tap = [&](int x, int y) -> vec3 {
                if (x >= 0 && y >= 0 && x < W && y < H) return resolve_tonemap(xyz(history[(uint32_t)y * R.W + (uint32_t)x]));
                return V3(0.0f);
            };
            vec3 c0 = tap(p0x, p0y), c1 = tap(p0x + 1, p0y), c2 = tap(p0x, p0y + 1), c3 = tap(p0x + 1, p0y + 1);
            vec3 c01 = mix3(c0, c1, f.x);
            vec3 c23 = mix3(c2, c3, f.x);
            history_color = mix3(c01, c23, f.y);
            valid_history = true;
        }
    }
    vec3 final_tm = tm_filtered;
    if (valid_history) {
        vec3 hist_ycocg = rgb_to_ycocg(history_color);
        vec3 clipped_ycocg = clamp3(hist_ycocg, c_min, c_max);
        vec3 clamped_history = ycocg_to_rgb(clipped_ycocg);
        vec2 motion_px = structure_motion * V2((float)R.W, (float)R.H);
        float speed = length2(motion_px);
        if (speed < 0.5f) {
            float accum_blend = 1.0f - (1.0f / (float)(R.frame_count + 1u));
            final_tm = mix3(tm_filtered, history_color, clamp_(accum_blend, 0.0f, 1.0f));
        } else {
            float dynamic_feedback = mix_(0.98f, 0.85f, smoothstep_(0.0f, 2.0f, speed));
            final_tm = mix3(tm_filtered, clamped_history, dynamic_feedback);
        }
    }
    vec3 final_color = resolve_inverse_tonemap(final_tm);
    final_color = max3(V3(0.0f), final_color);
    accumulation[idx] = V4(final_color, 1.0f);
    vec3 display_color = pow3(final_color, (float)(1.0 / 2.2));
    R.display[idx] = pack_rgba8(V4(display_color, 1.0f));
}

// ------------------------------------------------------------------ driver
Renderer::Renderer(const Scene* s, uint32_t w, uint32_t h, uint32_t md, bool bvh, int nt)
    : scene(s), W(w), H(h), max_depth(md), use_bvh(bvh), nthreads(nt < 1 ? 1 : nt) { reset(); }

void Renderer::reset() {
    size_t n = (size_t)W * H;
    for (int i = 0; i < 2; ++i) {
        gpos[i].assign(n, V4(0, 0, 0, 0)); gnormal[i].assign(n, V4(0, 0, 0, 0)); galbedo[i].assign(n, 0u);
        reservoirs[i].assign(n, Reservoir{}); accum[i].assign(n, V4(0, 0, 0, 0));
    }
    gmotion.assign(n, V2(0, 0)); raw.assign(n, 0ull); display.assign(n, 0u);
    frame_count = 0;
    stats_total = TraceStats{};
    for (auto& s : stats_stage) s = TraceStats{};
}

template <class F>
static void run_rows(Renderer& R, const CameraUniform& cam, uint32_t y0, uint32_t y1, int stage, F f) {
    int nt = R.nthreads;
    std::vector<TraceStats> st((size_t)nt);
    auto work = [&](int tid) {
        Ctx c(R, cam);
        for (uint32_t y = y0 + (uint32_t)tid; y < y1; y += (uint32_t)nt)
            for (uint32_t x = 0; x < R.W; ++x) f(c, R, x, y);
        st[(size_t)tid] = c.st;
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    for (auto& s : st) { R.stats_total.add(s); R.stats_stage[stage].add(s); }
}

void Renderer::render_phases(const CameraUniform& cam, int phases, uint32_t y0, uint32_t y1) {
    if (y1 > H) y1 = H;
    if (phases & PH_GBUFFER) run_rows(*this, cam, y0, y1, 0, gbuffer_pixel);
    if (phases & PH_TEMPORAL) run_rows(*this, cam, y0, y1, 1, temporal_pixel);
    if (phases & PH_SPATIAL) run_rows(*this, cam, y0, y1, 2, spatial_pixel);
    if (phases & PH_POST) run_rows(*this, cam, y0, y1, 3, post_pixel);
}

void Renderer::render(const CameraUniform& cam) {
    render_phases(cam, PH_ALL, 0, H);
    end_frame();
}

} // namespace orc
