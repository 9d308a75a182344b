// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path. PARITY UNPINNED (see orc_math.hpp).
// Restates: src/geometry.rs, src/scene/{material,builder,scenes}.rs, src/camera.rs and the glam 0.30.9
// (Cargo.lock:876) matrix helpers those files call (glam is not vendored in /root/reference; its
// published scalar algorithms are restated here).
#include "orc_scene.hpp"
#include <cstring>
#include <map>
#include <array>
#include <utility>
#include <cmath>

namespace orc {

// ---------------------------------------------------------------- glam restatements
mat4 mat4_identity() {
    mat4 m; m.c[0] = V4(1, 0, 0, 0); m.c[1] = V4(0, 1, 0, 0); m.c[2] = V4(0, 0, 1, 0); m.c[3] = V4(0, 0, 0, 1);
    return m;
}
mat4 mat4_from_translation(vec3 t) { mat4 m = mat4_identity(); m.c[3] = V4(t.x, t.y, t.z, 1); return m; }
mat4 mat4_from_scale(vec3 s) {
    mat4 m = mat4_identity(); m.c[0].x = s.x; m.c[1].y = s.y; m.c[2].z = s.z; return m;
}
mat4 mat4_from_rotation_x(float a) {
    float sa = sinf(a), ca = cosf(a);
    mat4 m = mat4_identity(); m.c[1] = V4(0, ca, sa, 0); m.c[2] = V4(0, -sa, ca, 0); return m;
}
mat4 mat4_from_rotation_y(float a) {
    float sa = sinf(a), ca = cosf(a);
    mat4 m = mat4_identity(); m.c[0] = V4(ca, 0, -sa, 0); m.c[2] = V4(sa, 0, ca, 0); return m;
}
mat4 mat4_from_rotation_z(float a) {
    float sa = sinf(a), ca = cosf(a);
    mat4 m = mat4_identity(); m.c[0] = V4(ca, sa, 0, 0); m.c[1] = V4(-sa, ca, 0, 0); return m;
}
// glam scalar Mat4::inverse (cofactor expansion, GLM ordering)
mat4 mat4_inverse(const mat4& s) {
    float m00 = s.c[0].x, m01 = s.c[0].y, m02 = s.c[0].z, m03 = s.c[0].w;
    float m10 = s.c[1].x, m11 = s.c[1].y, m12 = s.c[1].z, m13 = s.c[1].w;
    float m20 = s.c[2].x, m21 = s.c[2].y, m22 = s.c[2].z, m23 = s.c[2].w;
    float m30 = s.c[3].x, m31 = s.c[3].y, m32 = s.c[3].z, m33 = s.c[3].w;
    float coef00 = m22 * m33 - m32 * m23, coef02 = m12 * m33 - m32 * m13, coef03 = m12 * m23 - m22 * m13;
    float coef04 = m21 * m33 - m31 * m23, coef06 = m11 * m33 - m31 * m13, coef07 = m11 * m23 - m21 * m13;
    float coef08 = m21 * m32 - m31 * m22, coef10 = m11 * m32 - m31 * m12, coef11 = m11 * m22 - m21 * m12;
    float coef12 = m20 * m33 - m30 * m23, coef14 = m10 * m33 - m30 * m13, coef15 = m10 * m23 - m20 * m13;
    float coef16 = m20 * m32 - m30 * m22, coef18 = m10 * m32 - m30 * m12, coef19 = m10 * m22 - m20 * m12;
    float coef20 = m20 * m31 - m30 * m21, coef22 = m10 * m31 - m30 * m11, coef23 = m10 * m21 - m20 * m11;
    float fac0[4] = {coef00, coef00, coef02, coef03}, fac1[4] = {coef04, coef04, coef06, coef07};
    float fac2[4] = {coef08, coef08, coef10, coef11}, fac3[4] = {coef12, coef12, coef14, coef15};
    float fac4[4] = {coef16, coef16, coef18, coef19}, fac5[4] = {coef20, coef20, coef22, coef23};
    float v0[4] = {m10, m00, m00, m00}, v1[4] = {m11, m01, m01, m01};
    float v2[4] = {m12, m02, m02, m02}, v3[4] = {m13, m03, m03, m03};
    const float sa[4] = {1, -1, 1, -1}, sb[4] = {-1, 1, -1, 1};
    float inv[4][4];
    for (int i = 0; i < 4; ++i) {
        inv[0][i] = ((v1[i] * fac0[i] - v2[i] * fac1[i]) + v3[i] * fac2[i]) * sa[i];
        inv[1][i] = ((v0[i] * fac0[i] - v2[i] * fac3[i]) + v3[i] * fac4[i]) * sb[i];
        inv[2][i] = ((v0[i] * fac1[i] - v1[i] * fac3[i]) + v3[i] * fac5[i]) * sa[i];
        inv[3][i] = ((v0[i] * fac2[i] - v1[i] * fac4[i]) + v2[i] * fac5[i]) * sb[i];
    }
    float d0 = m00 * inv[0][0], d1 = m01 * inv[1][0], d2 = m02 * inv[2][0], d3 = m03 * inv[3][0];
    float det = ((d0 + d1) + d2) + d3;
    float rcp = 1.0f / det;
    mat4 r;
    for (int c = 0; c < 4; ++c) r.c[c] = V4(inv[c][0] * rcp, inv[c][1] * rcp, inv[c][2] * rcp, inv[c][3] * rcp);
    return r;
}
static vec3 glam_normalize(vec3 v) { float r = 1.0f / sqrtf(dot(v, v)); return v * r; }
static vec3 transform_vector3(const mat4& m, vec3 v) {
    vec4 r = (m.c[0] * v.x + m.c[1] * v.y) + m.c[2] * v.z;
    return xyz(r);
}

// ---------------------------------------------------------------- src/geometry.rs
// geometry.rs:56-76
vec2 encode_octahedral_normal(vec3 n) {
    float l1 = fabsf(n.x) + fabsf(n.y) + fabsf(n.z);
    vec2 res = l1 > 0.0f ? V2(n.x / l1, n.y / l1) : V2(0, 0);
    if (n.z < 0.0f) {
        float x = res.x, y = res.y;
        float sx = x >= 0.0f ? 1.0f : -1.0f, sy = y >= 0.0f ? 1.0f : -1.0f;
        return V2((1.0f - fabsf(y)) * sx, (1.0f - fabsf(x)) * sy);
    }
    return res;
}
static VertexAttributes attr(vec2 n, float u, float v, float tx, float ty, float tz, float tw) {
    VertexAttributes a; a.normal[0] = n.x; a.normal[1] = n.y; a.uv[0] = u; a.uv[1] = v;
    a.tangent[0] = tx; a.tangent[1] = ty; a.tangent[2] = tz; a.tangent[3] = tw; return a;
}
// geometry.rs:79-117
Geometry create_plane() {
    Geometry g;
    g.positions = {V4(-0.5f, 0, 0.5f, 1), V4(0.5f, 0, 0.5f, 1), V4(-0.5f, 0, -0.5f, 1), V4(0.5f, 0, -0.5f, 1)};
    vec2 en = encode_octahedral_normal(V3(0, 1, 0));
    g.attributes = {attr(en, 0, 1, 1, 0, 0, 1), attr(en, 1, 1, 1, 0, 0, 1), attr(en, 0, 0, 1, 0, 0, 1), attr(en, 1, 0, 1, 0, 0, 1)};
    g.indices = {0, 1, 2, 2, 1, 3};
    return g;
}
// geometry.rs:120-219
Geometry create_cube() {
    struct Side { float n[3], t[4], v[4][3]; };
    static const Side sides[6] = {
        {{0, 0, 1}, {1, 0, 0, 1}, {{-0.5f, -0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.5f, 0.5f, 0.5f}, {-0.5f, 0.5f, 0.5f}}},
        {{0, 0, -1}, {-1, 0, 0, 1}, {{0.5f, -0.5f, -0.5f}, {-0.5f, -0.5f, -0.5f}, {-0.5f, 0.5f, -0.5f}, {0.5f, 0.5f, -0.5f}}},
        {{0, 1, 0}, {1, 0, 0, 1}, {{-0.5f, 0.5f, 0.5f}, {0.5f, 0.5f, 0.5f}, {0.5f, 0.5f, -0.5f}, {-0.5f, 0.5f, -0.5f}}},
        {{0, -1, 0}, {1, 0, 0, 1}, {{-0.5f, -0.5f, -0.5f}, {0.5f, -0.5f, -0.5f}, {0.5f, -0.5f, 0.5f}, {-0.5f, -0.5f, 0.5f}}},
        {{1, 0, 0}, {0, 0, -1, 1}, {{0.5f, -0.5f, 0.5f}, {0.5f, -0.5f, -0.5f}, {0.5f, 0.5f, -0.5f}, {0.5f, 0.5f, 0.5f}}},
        {{-1, 0, 0}, {0, 0, 1, 1}, {{-0.5f, -0.5f, -0.5f}, {-0.5f, -0.5f, 0.5f}, {-0.5f, 0.5f, 0.5f}, {-0.5f, 0.5f, -0.5f}}},
    };
    static const float uvs[4][2] = {{0, 1}, {1, 1}, {1, 0}, {0, 0}};
    Geometry g;
    uint32_t v_idx = 0;
    for (const Side& s : sides) {
        vec2 en = encode_octahedral_normal(V3(s.n[0], s.n[1], s.n[2]));
        for (int k = 0; k < 4; ++k) {
            g.positions.push_back(V4(s.v[k][0], s.v[k][1], s.v[k][2], 1));
            g.attributes.push_back(attr(en, uvs[k][0], uvs[k][1], s.t[0], s.t[1], s.t[2], s.t[3]));
        }
        uint32_t idx[6] = {v_idx, v_idx + 1, v_idx + 2, v_idx, v_idx + 2, v_idx + 3};
        g.indices.insert(g.indices.end(), idx, idx + 6);
        v_idx += 4;
    }
    return g;
}
// geometry.rs:222-346
Geometry create_sphere(uint32_t subdivisions) {
    Geometry g;
    float t = (1.0f + sqrtf(5.0f)) / 2.0f;
    auto add_vertex = [&](float px, float py, float pz) -> uint32_t {
        float len = sqrtf(px * px + py * py + pz * pz);
        vec3 n = V3(px / len, py / len, pz / len);
        g.positions.push_back(V4(n.x * 0.5f, n.y * 0.5f, n.z * 0.5f, 1));
        g.attributes.push_back(attr(encode_octahedral_normal(n), 0, 0, 1, 0, 0, 1));
        return (uint32_t)g.positions.size() - 1;
    };
    add_vertex(-1, t, 0); add_vertex(1, t, 0); add_vertex(-1, -t, 0); add_vertex(1, -t, 0);
    add_vertex(0, -1, t); add_vertex(0, 1, t); add_vertex(0, -1, -t); add_vertex(0, 1, -t);
    add_vertex(t, 0, -1); add_vertex(t, 0, 1); add_vertex(-t, 0, -1); add_vertex(-t, 0, 1);
    std::vector<std::array<uint32_t, 3>> faces = {
        {0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11}, {1, 5, 9}, {5, 11, 4}, {11, 10, 2}, {10, 7, 6}, {7, 1, 8},
        {3, 9, 4}, {3, 4, 2}, {3, 2, 6}, {3, 6, 8}, {3, 8, 9}, {4, 9, 5}, {2, 4, 11}, {6, 2, 10}, {8, 6, 7}, {9, 8, 1}};
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> cache;   // geometry.rs:282 (HashMap; only point lookups)
    auto midpoint = [&](uint32_t p1, uint32_t p2) -> uint32_t {   // geometry.rs:309-346
        auto key = p1 < p2 ? std::make_pair(p1, p2) : std::make_pair(p2, p1);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
        vec4 a = g.positions[p1], b = g.positions[p2];
        float mx = (a.x + b.x) * 0.5f, my = (a.y + b.y) * 0.5f, mz = (a.z + b.z) * 0.5f;
        float len = sqrtf(mx * mx + my * my + mz * mz);
        vec3 n = V3(mx / len, my / len, mz / len);
        g.positions.push_back(V4(n.x * 0.5f, n.y * 0.5f, n.z * 0.5f, 1));
        g.attributes.push_back(attr(encode_octahedral_normal(n), 0, 0, 1, 0, 0, 1));
        uint32_t idx = (uint32_t)g.positions.size() - 1;
        cache[key] = idx;
        return idx;
    };
    for (uint32_t s = 0; s < subdivisions; ++s) {
        std::vector<std::array<uint32_t, 3>> nf;
        nf.reserve(faces.size() * 4);
        for (auto& tri : faces) {
            uint32_t v1 = tri[0], v2 = tri[1], v3 = tri[2];
            uint32_t a = midpoint(v1, v2), b = midpoint(v2, v3), c = midpoint(v3, v1);
            nf.push_back({v1, a, c}); nf.push_back({v2, b, a}); nf.push_back({v3, c, b}); nf.push_back({a, b, c});
        }
        faces.swap(nf);
    }
    for (auto& tri : faces) { g.indices.push_back(tri[0]); g.indices.push_back(tri[1]); g.indices.push_back(tri[2]); }
    return g;
}
// geometry.rs:350-434
Geometry create_crystal() {
    Geometry g;
    vec3 top_tip = V3(0, 1, 0), bottom_tip = V3(0, -1, 0);
    vec3 top_ring[4] = {V3(0.3f, 0.5f, 0.3f), V3(-0.3f, 0.5f, 0.3f), V3(-0.3f, 0.5f, -0.3f), V3(0.3f, 0.5f, -0.3f)};
    vec3 bot_ring[4] = {V3(0.3f, -0.5f, 0.3f), V3(-0.3f, -0.5f, 0.3f), V3(-0.3f, -0.5f, -0.3f), V3(0.3f, -0.5f, -0.3f)};
    auto add_face = [&](vec3 p0, vec3 p1, vec3 p2) {
        vec3 n = glam_normalize(cross(p1 - p0, p2 - p0));
        vec2 en = encode_octahedral_normal(n);
        uint32_t base = (uint32_t)g.positions.size();
        vec3 ps[3] = {p0, p1, p2};
        for (vec3 p : ps) {
            g.positions.push_back(V4(p.x, p.y, p.z, 1));
            g.attributes.push_back(attr(en, 0, 0, 1, 0, 0, 1));
        }
        g.indices.push_back(base); g.indices.push_back(base + 1); g.indices.push_back(base + 2);
    };
    for (int i = 0; i < 4; ++i) add_face(top_tip, top_ring[(i + 1) % 4], top_ring[i]);
    for (int i = 0; i < 4; ++i) {
        int in = (i + 1) % 4;
        add_face(top_ring[i], top_ring[in], bot_ring[in]);
        add_face(top_ring[i], bot_ring[in], bot_ring[i]);
    }
    for (int i = 0; i < 4; ++i) add_face(bottom_tip, bot_ring[i], bot_ring[(i + 1) % 4]);
    return g;
}

// ---------------------------------------------------------------- src/scene/material.rs
Material material_new(float r, float g, float b, float a) {   // material.rs:31-47
    Material m{};
    m.base_color[0] = r; m.base_color[1] = g; m.base_color[2] = b; m.base_color[3] = a;
    m.roughness = 0.5f; m.metallic = 0.0f; m.transmission = 0.0f; m.ior = 1.0f; m.light_index = -1;
    m.tex_info_0 = m.tex_info_1 = m.tex_info_2 = 0xFFFFFFFFu; m.pad_final = 0;
    return m;
}
static uint32_t pack16(uint32_t cur, uint32_t val, bool high) {   // material.rs:79-86
    uint32_t v = val & 0xFFFFu;
    return high ? ((cur & 0x0000FFFFu) | (v << 16)) : ((cur & 0xFFFF0000u) | v);
}

// ---------------------------------------------------------------- src/scene/builder.rs
static std::vector<uint8_t> solid_tex(uint8_t r, uint8_t g, uint8_t b, uint8_t a) {
    std::vector<uint8_t> t(1024u * 1024u * 4u);
    for (size_t i = 0; i < t.size(); i += 4) { t[i] = r; t[i + 1] = g; t[i + 2] = b; t[i + 3] = a; }
    return t;
}
Scene::Scene() {   // builder.rs:24-91
    color_textures.push_back(solid_tex(255, 255, 255, 255));
    std::vector<uint8_t> checker(1024u * 1024u * 4u);
    for (uint32_t y = 0; y < 1024; ++y)
        for (uint32_t x = 0; x < 1024; ++x) {
            bool check = ((x / 64) + (y / 64)) % 2 == 0;
            uint8_t c = check ? 255 : 0;
            uint8_t* p = &checker[(y * 1024u + x) * 4u];
            p[0] = c; p[1] = c; p[2] = c; p[3] = 255;
        }
    color_textures.push_back(checker);
    color_textures.push_back(solid_tex(0, 0, 0, 255));
    data_textures.push_back(solid_tex(255, 255, 255, 255));
    data_textures.push_back(solid_tex(128, 128, 255, 255));
    data_textures.push_back(solid_tex(0, 0, 0, 255));
    // sRGB8 -> linear decode table of an Rgba8UnormSrgb view (builder.rs:489); IEC 61966-2-1, double then f32
    for (int i = 0; i < 256; ++i) {
        double c = i / 255.0;
        double l = c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4);
        srgb_lut[i] = (float)l;
    }
}
uint32_t Scene::add_color_texture(const uint8_t* rgba8) {   // builder.rs:93-103
    color_textures.emplace_back(rgba8, rgba8 + 1024u * 1024u * 4u);
    return (uint32_t)color_textures.size() - 1;
}
uint32_t Scene::add_data_texture(const uint8_t* rgba8) {    // builder.rs:105-115
    data_textures.emplace_back(rgba8, rgba8 + 1024u * 1024u * 4u);
    return (uint32_t)data_textures.size() - 1;
}
uint32_t Scene::add_material(const Material& m) { materials.push_back(m); return (uint32_t)materials.size() - 1; }   // :117-121
uint32_t Scene::add_mesh(const Geometry& g) {   // builder.rs:123-141
    uint32_t id = (uint32_t)mesh_infos.size();
    MeshInfo mi{(uint32_t)attributes.size(), (uint32_t)indices.size(), {0, 0}};
    attributes.insert(attributes.end(), g.attributes.begin(), g.attributes.end());
    indices.insert(indices.end(), g.indices.begin(), g.indices.end());
    mesh_infos.push_back(mi);
    mesh_positions.push_back(g.positions);
    mesh_index_count.push_back((uint32_t)g.indices.size());
    return id;
}
void Scene::add_instance(uint32_t mesh_id, uint32_t mat_id, const mat4& t) {   // builder.rs:181-189 (mask ignored)
    Instance in{};
    in.mesh_id = mesh_id; in.mat_id = mat_id;
    for (int c = 0; c < 4; ++c) { in.m[4 * c] = t.c[c].x; in.m[4 * c + 1] = t.c[c].y; in.m[4 * c + 2] = t.c[c].z; in.m[4 * c + 3] = t.c[c].w; }
    instances.push_back(in);
}
void Scene::add_quad_light(const float position[3], const float u[3], const float v[3], const float emission[4]) {   // :392-415
    vec3 uv = V3(u[0], u[1], u[2]), vv = V3(v[0], v[1], v[2]);
    float area = length(cross(uv, vv)) * 4.0f;
    LightUniform l{};
    for (int i = 0; i < 3; ++i) { l.position[i] = position[i]; l.u[i] = u[i]; l.v[i] = v[i]; }
    l.type_ = 0; l.area = area; l.pad = 0;
    for (int i = 0; i < 4; ++i) l.emission[i] = emission[i];
    lights.push_back(l);
}
void Scene::add_sphere_light(const float center[3], float radius, const float emission[4]) {   // :418-429
    float area = 4.0f * 3.14159265358979323846f * radius * radius;
    LightUniform l{};
    for (int i = 0; i < 3; ++i) { l.position[i] = center[i]; l.u[i] = 0.0f; }
    l.type_ = 1; l.area = area; l.v[0] = radius; l.v[1] = 0; l.v[2] = 0; l.pad = 0;
    for (int i = 0; i < 4; ++i) l.emission[i] = emission[i];
    lights.push_back(l);
}
static Material light_material(int32_t light_index, const float color[3], float intensity) {   // :324-338, :360-371
    Material m = material_new(1, 1, 1, 1);
    m.light_index = light_index;
    for (int i = 0; i < 3; ++i) m.emissive_factor[i] = color[i] * intensity;
    m.tex_info_0 = pack16(m.tex_info_0, 0, false);
    return m;
}
void Scene::register_quad_light(uint32_t mesh_id, const mat4& t, const float color[3], float intensity) {   // :316-351
    uint32_t mat_id = add_material(light_material((int32_t)lights.size(), color, intensity));
    add_instance(mesh_id, mat_id, t);
    float pos[3] = {t.c[3].x, t.c[3].y, t.c[3].z};
    vec3 u = transform_vector3(t, V3(1, 0, 0)) * 0.5f;
    vec3 v = transform_vector3(t, V3(0, 0, -1)) * 0.5f;
    float ua[3] = {u.x, u.y, u.z}, va[3] = {v.x, v.y, v.z};
    float em[4] = {color[0], color[1], color[2], intensity};
    add_quad_light(pos, ua, va, em);
}
void Scene::register_sphere_light(uint32_t mesh_id, const mat4& t, const float color[3], float intensity) {   // :353-385
    uint32_t mat_id = add_material(light_material((int32_t)lights.size(), color, intensity));
    add_instance(mesh_id, mat_id, t);
    float pos[3] = {t.c[3].x, t.c[3].y, t.c[3].z};
    float scale = length(transform_vector3(t, V3(1, 0, 0)));
    float em[4] = {color[0], color[1], color[2], intensity};
    add_sphere_light(pos, scale * 0.5f, em);
}

// Replaces the driver BLAS/TLAS build (builder.rs:143-179, :454-468): flatten every instance to
// world-space triangles. Contract: world = ((c0*x + c1*y) + c2*z) + c3 in f32;
// e1 = v1w - v0w, e2 = v2w - v0w; world_to_object = inverse(m3x3) via cofactors in double, rounded to f32.
void Scene::build() {
    tris.clear(); tri_instance.clear();
    for (size_t ii = 0; ii < instances.size(); ++ii) {
        Instance& in = instances[ii];
        const float* m = in.m;
        double a = m[0], b = m[4], c = m[8], d = m[1], e = m[5], f = m[9], g = m[2], h = m[6], i = m[10];
        double co00 = e * i - f * h, co01 = f * g - d * i, co02 = d * h - e * g;
        double det = a * co00 + b * co01 + c * co02;
        double inv[3][3];   // inv[r][c]
        inv[0][0] = co00 / det; inv[0][1] = (c * h - b * i) / det; inv[0][2] = (b * f - c * e) / det;
        inv[1][0] = co01 / det; inv[1][1] = (a * i - c * g) / det; inv[1][2] = (c * d - a * f) / det;
        inv[2][0] = co02 / det; inv[2][1] = (b * g - a * h) / det; inv[2][2] = (a * e - b * d) / det;
        for (int cc = 0; cc < 3; ++cc) for (int rr = 0; rr < 3; ++rr) in.w2o[3 * cc + rr] = (float)inv[rr][cc];
        in.flip = det < 0.0 ? 1u : 0u;
        in.first_tri = (uint32_t)tris.size();
        const std::vector<vec4>& P = mesh_positions[in.mesh_id];
        const MeshInfo& mi = mesh_infos[in.mesh_id];
        uint32_t nidx = mesh_index_count[in.mesh_id];
        auto xf = [&](vec4 p) -> vec3 {
            return V3(((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12],
                      ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13],
                      ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14]);
        };
        for (uint32_t k = 0; k + 2 < nidx; k += 3) {
            vec3 w0 = xf(P[indices[mi.index_offset + k]]);
            vec3 w1 = xf(P[indices[mi.index_offset + k + 1]]);
            vec3 w2 = xf(P[indices[mi.index_offset + k + 2]]);
            tris.push_back({w0, w1 - w0, w2 - w0});
            tri_instance.push_back((uint32_t)ii);
        }
        in.tri_count = (uint32_t)tris.size() - in.first_tri;
    }
    built = true;
}

// ---------------------------------------------------------------- src/scene/scenes.rs
static const float PI_F = 3.14159265358979323846f;      // std::f32::consts::PI
static const float FRAC_PI_2_F = 1.57079632679489661923f;

void create_cornell_box(Scene& b) {   // scenes.rs:9-130
    uint32_t plane_id = b.add_mesh(create_plane());
    uint32_t cube_id = b.add_mesh(create_cube());
    uint32_t sphere_id = b.add_mesh(create_sphere(3));
    uint32_t crystal_id = b.add_mesh(create_crystal());

    uint32_t mat_red = b.add_material(material_new(0.65f, 0.05f, 0.05f, 1.0f));
    uint32_t mat_green = b.add_material(material_new(0.12f, 0.45f, 0.15f, 1.0f));
    uint32_t mat_white = b.add_material(material_new(0.73f, 0.73f, 0.73f, 1.0f));
    Material checker = material_new(0.73f, 0.73f, 0.73f, 1.0f);
    checker.roughness = 0.99f; checker.tex_info_0 = pack16(checker.tex_info_0, 1, false);
    uint32_t mat_checker = b.add_material(checker);
    Material metal = material_new(0.8f, 0.8f, 0.8f, 1.0f);
    metal.metallic = 1.0f; metal.roughness = 0.01f;     // Material::metallic(r) quirk, material.rs:54-58
    uint32_t mat_rough_metal = b.add_material(metal);
    Material crystal = material_new(0.5f, 0.8f, 1.0f, 1.0f);
    crystal.metallic = 0.0f; crystal.roughness = 0.0f; crystal.ior = 1.5f; crystal.transmission = 1.0f;   // glass(1.5)
    uint32_t mat_crystal = b.add_material(crystal);

    auto T = [](float x, float y, float z) { return mat4_from_translation(V3(x, y, z)); };
    auto S = [](float s) { return mat4_from_scale(V3(s)); };
    b.add_instance(plane_id, mat_checker, mul(T(0, -1, 0), S(2)));
    b.add_instance(plane_id, mat_white, mul(mul(T(0, 1, 0), mat4_from_rotation_x(PI_F)), S(2)));
    b.add_instance(plane_id, mat_white, mul(mul(T(0, 0, -1), mat4_from_rotation_x(FRAC_PI_2_F)), S(2)));
    b.add_instance(plane_id, mat_red, mul(mul(T(-1, 0, 0), mat4_from_rotation_z(-FRAC_PI_2_F)), S(2)));
    b.add_instance(plane_id, mat_green, mul(mul(T(1, 0, 0), mat4_from_rotation_z(FRAC_PI_2_F)), S(2)));
    const float white[3] = {1, 1, 1};
    b.register_quad_light(plane_id, mul(mul(T(0, 0.99f, 0), mat4_from_rotation_x(PI_F)), S(0.5f)), white, 10.0f);
    b.add_instance(crystal_id, mat_crystal, mul(T(0.4f, -0.5f, 0.3f), S(0.5f)));
    const float blue[3] = {0.02f, 0.02f, 0.9f};
    b.register_sphere_light(sphere_id, mul(T(0.4f, -0.5f, 0.3f), S(0.1f)), blue, 10.0f);
    b.add_instance(cube_id, mat_rough_metal,
                   mul(mul(T(-0.35f, -0.4f + 0.002f, -0.3f), mat4_from_rotation_y(0.4f)), mat4_from_scale(V3(0.6f, 1.2f, 0.6f))));
    b.build();
}

static void hsv_to_rgb(float h, float s, float v, float out[3]) {   // scenes.rs:226-246
    float c = v * s;
    float x = c * (1.0f - fabsf(fmodf(h * 6.0f, 2.0f) - 1.0f));
    float m = v - c;
    float r, g, b;
    if (h < 1.0f / 6.0f) { r = c; g = x; b = 0; }
    else if (h < 2.0f / 6.0f) { r = x; g = c; b = 0; }
    else if (h < 3.0f / 6.0f) { r = 0; g = c; b = x; }
    else if (h < 4.0f / 6.0f) { r = 0; g = x; b = c; }
    else if (h < 5.0f / 6.0f) { r = x; g = 0; b = c; }
    else { r = c; g = 0; b = x; }
    out[0] = r + m; out[1] = g + m; out[2] = b + m;
}
void create_restir_scene(Scene& b) {   // scenes.rs:133-223
    uint32_t plane_id = b.add_mesh(create_plane());
    uint32_t sphere_id = b.add_mesh(create_sphere(2));
    uint32_t cube_id = b.add_mesh(create_cube());
    Material fl = material_new(0.73f, 0.73f, 0.73f, 1.0f); fl.roughness = 0.99f;
    uint32_t mat_floor = b.add_material(fl);
    uint32_t mat_wall = b.add_material(fl);
    Material me = material_new(1, 1, 1, 1); me.metallic = 1.0f; me.roughness = 0.2f;
    uint32_t mat_metal = b.add_material(me);
    b.add_instance(plane_id, mat_floor, mul(mat4_from_translation(V3(0, -1, 0)), mat4_from_scale(V3(10.0f))));
    b.add_instance(plane_id, mat_wall,
                   mul(mul(mat4_from_translation(V3(0, 5, -5)), mat4_from_rotation_x(FRAC_PI_2_F)), mat4_from_scale(V3(10.0f))));
    const int rows = 10, cols = 10;
    const float spacing = 1.0f, light_radius = 0.05f, emission_strength = 20.0f;
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            float x = ((float)c - (float)cols / 2.0f) * spacing;
            float z = ((float)r - (float)rows / 2.0f) * spacing;
            float y = -0.9f;
            float hue = (float)(r * cols + c) / (float)(rows * cols);
            float color[3]; hsv_to_rgb(hue, 0.8f, 1.0f, color);
            Material m = material_new(color[0], color[1], color[2], 1.0f);
            m.light_index = r * cols + c;
            for (int k = 0; k < 3; ++k) m.emissive_factor[k] = color[k] * emission_strength;
            uint32_t mat_id = b.add_material(m);
            b.add_instance(sphere_id, mat_id, mul(mat4_from_translation(V3(x, y, z)), mat4_from_scale(V3(light_radius))));
            float pos[3] = {x, y, z}, em[4] = {color[0], color[1], color[2], emission_strength};
            b.add_sphere_light(pos, light_radius, em);
        }
    b.add_instance(cube_id, mat_metal, mul(mat4_from_translation(V3(0, -0.5f, 0)), mat4_from_scale(V3(0.5f))));
    b.build();
}

// ---------------------------------------------------------------- src/camera.rs
// camera.rs:207-256 build_uniform for any controller state. prev_view_proj == nullptr: the initial IDENTITY (first frame, :233-238).
CameraUniform camera_build(vec3 position, float yaw, float pitch, const float* prev_view_proj, float aspect, uint32_t frame_count,
                           uint32_t num_lights, float jitter_x, float jitter_y, float* unjittered_out) {
    float sy = sinf(yaw), cy = cosf(yaw), sp = sinf(pitch), cp = cosf(pitch);
    vec3 forward = glam_normalize(V3(cp * cy, sp, cp * sy));
    // look_at_rh(eye, eye + forward, Y) = look_to_rh(eye, (eye + forward) - eye, Y)
    vec3 dir = (position + forward) - position;
    vec3 f = glam_normalize(dir);
    vec3 s = glam_normalize(cross(f, V3(0, 1, 0)));
    vec3 u = cross(s, f);
    mat4 view;
    view.c[0] = V4(s.x, u.x, -f.x, 0); view.c[1] = V4(s.y, u.y, -f.y, 0); view.c[2] = V4(s.z, u.z, -f.z, 0);
    view.c[3] = V4(-dot(position, s), -dot(position, u), dot(position, f), 1);
    // perspective_rh(45deg, aspect, 0.1, 100)
    float fov = 45.0f * (3.14159265358979323846f / 180.0f);
    float sf = sinf(0.5f * fov), cf = cosf(0.5f * fov);
    float h = cf / sf, w = h / aspect, r = 100.0f / (0.1f - 100.0f);
    mat4 proj_base;
    proj_base.c[0] = V4(w, 0, 0, 0); proj_base.c[1] = V4(0, h, 0, 0); proj_base.c[2] = V4(0, 0, r, -1); proj_base.c[3] = V4(0, 0, r * 0.1f, 0);
    mat4 vp_unjittered = mul(proj_base, view);
    mat4 proj = proj_base;
    proj.c[2].x += jitter_x;   // camera.rs:226
    proj.c[2].y += jitter_y;   // camera.rs:227
    mat4 vp = mul(proj, view);
    mat4 vi = mat4_inverse(view), pi = mat4_inverse(proj);
    CameraUniform cu{};
    auto store = [](float* dst, const mat4& m) {
        for (int c = 0; c < 4; ++c) { dst[4 * c] = m.c[c].x; dst[4 * c + 1] = m.c[c].y; dst[4 * c + 2] = m.c[c].z; dst[4 * c + 3] = m.c[c].w; }
    };
    store(cu.view_proj, vp); store(cu.view_inverse, vi); store(cu.proj_inverse, pi);
    if (prev_view_proj) memcpy(cu.prev_view_proj, prev_view_proj, 64); else store(cu.prev_view_proj, vp_unjittered);
    cu.view_pos[0] = position.x; cu.view_pos[1] = position.y; cu.view_pos[2] = position.z; cu.view_pos[3] = 1.0f;
    cu.frame_count = frame_count; cu.num_lights = num_lights;
    if (unjittered_out) store(unjittered_out, vp_unjittered);
    return cu;
}
// camera.rs:182-205; `scale` = the literal 0 of :202-203
void camera_halton_jitter(uint32_t index, uint32_t width, uint32_t height, float scale, float out[2]) {
    auto halton = [](uint32_t i, uint32_t base) {
        float f = 1.0f, r = 0.0f;
        while (i > 0) { f = f / (float)base; r = r + f * (float)(i % base); i = i / base; }
        return r;
    };
    float halton_x = halton(index + 1u, 2u) - 0.5f;
    float halton_y = halton(index + 1u, 3u) - 0.5f;
    out[0] = (halton_x * scale) / (float)width;
    out[1] = (halton_y * scale) / (float)height;
}
CameraUniform camera_default(float aspect, uint32_t frame_count, uint32_t num_lights) {
    // camera.rs:40-42 pose; :207-256 build_uniform with jitter == (0,0) (camera.rs:202-203) and
    // prev_view_proj == IDENTITY -> unjittered view_proj (static camera: same every frame, state.rs:172)
    return camera_build(V3(0.0f, 0.0f, 3.0f), -90.0f * (3.14159265358979323846f / 180.0f), 0.0f, nullptr, aspect, frame_count, num_lights, 0.0f, 0.0f, nullptr);
}

} // namespace orc
