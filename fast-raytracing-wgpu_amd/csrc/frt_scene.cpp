// frt_scene.cpp — host scene model, geometry generators, scene factories, camera.
// Mirrors src/geometry.rs, src/scene/{builder,material,scenes}.rs, src/camera.rs; matrix helpers follow glam 0.30.9
// (Cargo.lock:876), f32 throughout.
#include "frt_scene.hpp"
#include <cmath>
#include <cstring>
#include <map>
#include <algorithm>

namespace frt {

// ---------------------------------------------------------------------------------------------- Mat4 (glam)
Mat4 mat4_identity() { Mat4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }
Mat4 mat4_mul(const Mat4& a, const Mat4& b) {
    // glam Mat4 * Mat4: column j = ((a.x*b.x + a.y*b.y) + a.z*b.z) + a.w*b.w
    Mat4 r;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            r.m[4 * j + i] = ((a.m[i] * b.m[4 * j] + a.m[4 + i] * b.m[4 * j + 1]) + a.m[8 + i] * b.m[4 * j + 2]) + a.m[12 + i] * b.m[4 * j + 3];
    return r;
}
Mat4 mat4_translation(float x, float y, float z) { Mat4 r = mat4_identity(); r.m[12] = x; r.m[13] = y; r.m[14] = z; return r; }
Mat4 mat4_scale(float x, float y, float z) { Mat4 r = mat4_identity(); r.m[0] = x; r.m[5] = y; r.m[10] = z; return r; }
Mat4 mat4_rotation_x(float a) { float s = sinf(a), c = cosf(a); Mat4 r = mat4_identity(); r.m[5] = c; r.m[6] = s; r.m[9] = -s; r.m[10] = c; return r; }
Mat4 mat4_rotation_y(float a) { float s = sinf(a), c = cosf(a); Mat4 r = mat4_identity(); r.m[0] = c; r.m[2] = -s; r.m[8] = s; r.m[10] = c; return r; }
Mat4 mat4_rotation_z(float a) { float s = sinf(a), c = cosf(a); Mat4 r = mat4_identity(); r.m[0] = c; r.m[1] = s; r.m[4] = -s; r.m[5] = c; return r; }
Mat4 mat4_inverse(const Mat4& a) {
    // glam scalar Mat4::inverse: 2x2 sub-determinants ("coef"), four cofactor rows, sign masks, 1/det
    const float* m = a.m;
    auto M = [&](int c, int r) { return m[4 * c + r]; };
    float c00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3), c02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3), c03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
    float c04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3), c06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3), c07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
    float c08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2), c10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2), c11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
    float c12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3), c14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3), c15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
    float c16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2), c18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2), c19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
    float c20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1), c22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1), c23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
    float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
    float f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    float v0[4] = {M(1, 0), M(0, 0), M(0, 0), M(0, 0)}, v1[4] = {M(1, 1), M(0, 1), M(0, 1), M(0, 1)};
    float v2[4] = {M(1, 2), M(0, 2), M(0, 2), M(0, 2)}, v3[4] = {M(1, 3), M(0, 3), M(0, 3), M(0, 3)};
    Mat4 inv;
    for (int i = 0; i < 4; ++i) {
        float sa = (i & 1) ? -1.0f : 1.0f, sb = -sa;
        inv.m[0 + i] = ((v1[i] * f0[i] - v2[i] * f1[i]) + v3[i] * f2[i]) * sa;
        inv.m[4 + i] = ((v0[i] * f0[i] - v2[i] * f3[i]) + v3[i] * f4[i]) * sb;
        inv.m[8 + i] = ((v0[i] * f1[i] - v1[i] * f3[i]) + v3[i] * f5[i]) * sa;
        inv.m[12 + i] = ((v0[i] * f2[i] - v1[i] * f4[i]) + v2[i] * f5[i]) * sb;
    }
    float det = ((M(0, 0) * inv.m[0] + M(0, 1) * inv.m[4]) + M(0, 2) * inv.m[8]) + M(0, 3) * inv.m[12];
    float rcp = 1.0f / det;
    for (float& x : inv.m) x *= rcp;
    return inv;
}
static void xform_vec3(const Mat4& t, float x, float y, float z, float out[3]) {   // glam transform_vector3
    for (int i = 0; i < 3; ++i) out[i] = (t.m[i] * x + t.m[4 + i] * y) + t.m[8 + i] * z;
}
static void v3_normalize_glam(float v[3]) {   // Vec3::normalize = v * (1 / length)
    float r = 1.0f / sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] *= r; v[1] *= r; v[2] *= r;
}

// ---------------------------------------------------------------------------------------------- geometry.rs
namespace geometry {

void encode_octahedral_normal(const float n[3], float out[2]) {   // geometry.rs:56-76
    float l1 = fabsf(n[0]) + fabsf(n[1]) + fabsf(n[2]);
    float rx = 0.0f, ry = 0.0f;
    if (l1 > 0.0f) { rx = n[0] / l1; ry = n[1] / l1; }
    if (n[2] < 0.0f) {
        float fx = (1.0f - fabsf(ry)) * (rx >= 0.0f ? 1.0f : -1.0f);
        float fy = (1.0f - fabsf(rx)) * (ry >= 0.0f ? 1.0f : -1.0f);
        rx = fx; ry = fy;
    }
    out[0] = rx; out[1] = ry;
}

static void push_vertex(Geometry& g, float x, float y, float z, const float nrm[3], float u, float v, const float tan[4]) {
    g.positions.insert(g.positions.end(), {x, y, z, 1.0f});
    frt_vertex_attr a;
    encode_octahedral_normal(nrm, a.normal);
    a.uv[0] = u; a.uv[1] = v;
    memcpy(a.tangent, tan, 16);
    g.attributes.push_back(a);
}

Geometry create_plane() {   // geometry.rs:79-117: unit XZ quad, normal +Y
    Geometry g;
    const float up[3] = {0, 1, 0}, tan[4] = {1, 0, 0, 1};
    const float P[4][5] = {{-0.5f, 0.5f, 0, 1}, {0.5f, 0.5f, 1, 1}, {-0.5f, -0.5f, 0, 0}, {0.5f, -0.5f, 1, 0}};   // x, z, u, v
    for (auto& p : P) push_vertex(g, p[0], 0.0f, p[1], up, p[2], p[3], tan);
    g.indices = {0, 1, 2, 2, 1, 3};
    return g;
}

Geometry create_cube() {   // geometry.rs:120-219: 6 faces x 4 vertices, per-face normal + tangent
    struct Face { float n[3]; float t[4]; float c[4][3]; };
    const float h = 0.5f;
    const Face faces[6] = {
        {{0, 0, 1}, {1, 0, 0, 1}, {{-h, -h, h}, {h, -h, h}, {h, h, h}, {-h, h, h}}},        // front
        {{0, 0, -1}, {-1, 0, 0, 1}, {{h, -h, -h}, {-h, -h, -h}, {-h, h, -h}, {h, h, -h}}},  // back
        {{0, 1, 0}, {1, 0, 0, 1}, {{-h, h, h}, {h, h, h}, {h, h, -h}, {-h, h, -h}}},        // top
        {{0, -1, 0}, {1, 0, 0, 1}, {{-h, -h, -h}, {h, -h, -h}, {h, -h, h}, {-h, -h, h}}},   // bottom
        {{1, 0, 0}, {0, 0, -1, 1}, {{h, -h, h}, {h, -h, -h}, {h, h, -h}, {h, h, h}}},       // right
        {{-1, 0, 0}, {0, 0, 1, 1}, {{-h, -h, -h}, {-h, -h, h}, {-h, h, h}, {-h, h, -h}}},   // left
    };
    const float uv[4][2] = {{0, 1}, {1, 1}, {1, 0}, {0, 0}};
    Geometry g;
    for (uint32_t f = 0; f < 6; ++f) {
        for (int k = 0; k < 4; ++k) push_vertex(g, faces[f].c[k][0], faces[f].c[k][1], faces[f].c[k][2], faces[f].n, uv[k][0], uv[k][1], faces[f].t);
        uint32_t b = 4 * f;
        g.indices.insert(g.indices.end(), {b, b + 1, b + 2, b, b + 2, b + 3});
    }
    return g;
}

Geometry create_sphere(uint32_t subdivisions) {   // geometry.rs:222-346: icosphere, radius 0.5, midpoint cache
    Geometry g;
    const float tan[4] = {1, 0, 0, 1};
    auto add_unit = [&](float x, float y, float z) -> uint32_t {
        float len = sqrtf(x * x + y * y + z * z);
        float n[3] = {x / len, y / len, z / len};
        push_vertex(g, n[0] * 0.5f, n[1] * 0.5f, n[2] * 0.5f, n, 0.0f, 0.0f, tan);
        return (uint32_t)g.attributes.size() - 1u;
    };
    const float t = (1.0f + sqrtf(5.0f)) / 2.0f;
    const float seed[12][3] = {{-1, t, 0}, {1, t, 0}, {-1, -t, 0}, {1, -t, 0}, {0, -1, t}, {0, 1, t},
                               {0, -1, -t}, {0, 1, -t}, {t, 0, -1}, {t, 0, 1}, {-t, 0, -1}, {-t, 0, 1}};
    for (auto& s : seed) add_unit(s[0], s[1], s[2]);
    std::vector<uint32_t> faces = {0, 11, 5, 0, 5, 1, 0, 1, 7, 0, 7, 10, 0, 10, 11, 1, 5, 9, 5, 11, 4, 11, 10, 2, 10, 7, 6, 7, 1, 8,
                                   3, 9, 4, 3, 4, 2, 3, 2, 6, 3, 6, 8, 3, 8, 9, 4, 9, 5, 2, 4, 11, 6, 2, 10, 8, 6, 7, 9, 8, 1};
    std::map<uint64_t, uint32_t> cache;   // geometry.rs:282 — point lookups only, so ordering of the map is irrelevant
    auto midpoint = [&](uint32_t a, uint32_t b) -> uint32_t {
        uint64_t key = a < b ? ((uint64_t)a << 32 | b) : ((uint64_t)b << 32 | a);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
        const float* pa = &g.positions[4 * a];
        const float* pb = &g.positions[4 * b];
        float mx = (pa[0] + pb[0]) * 0.5f, my = (pa[1] + pb[1]) * 0.5f, mz = (pa[2] + pb[2]) * 0.5f;
        uint32_t id = add_unit(mx, my, mz);
        cache.emplace(key, id);
        return id;
    };
    for (uint32_t level = 0; level < subdivisions; ++level) {
        std::vector<uint32_t> next;
        next.reserve(faces.size() * 4);
        for (size_t f = 0; f < faces.size(); f += 3) {
            uint32_t v1 = faces[f], v2 = faces[f + 1], v3 = faces[f + 2];
            uint32_t a = midpoint(v1, v2), b = midpoint(v2, v3), c = midpoint(v3, v1);
            next.insert(next.end(), {v1, a, c, v2, b, a, v3, c, b, a, b, c});
        }
        faces.swap(next);
    }
    g.indices = faces;
    return g;
}

Geometry create_crystal() {   // geometry.rs:350-434: 16 flat faces, unshared vertices
    Geometry g;
    const float tan[4] = {1, 0, 0, 1};
    const float top[3] = {0, 1, 0}, bottom[3] = {0, -1, 0};
    const float ring[4][2] = {{0.3f, 0.3f}, {-0.3f, 0.3f}, {-0.3f, -0.3f}, {0.3f, -0.3f}};   // x, z
    auto ringp = [&](int i, float y, float out[3]) { out[0] = ring[i & 3][0]; out[1] = y; out[2] = ring[i & 3][1]; };
    auto face = [&](const float p0[3], const float p1[3], const float p2[3]) {
        float e1[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, e2[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
        float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        v3_normalize_glam(n);
        uint32_t base = (uint32_t)g.attributes.size();
        push_vertex(g, p0[0], p0[1], p0[2], n, 0, 0, tan);
        push_vertex(g, p1[0], p1[1], p1[2], n, 0, 0, tan);
        push_vertex(g, p2[0], p2[1], p2[2], n, 0, 0, tan);
        g.indices.insert(g.indices.end(), {base, base + 1, base + 2});
    };
    float a[3], b[3], c[3], d[3];
    for (int i = 0; i < 4; ++i) { ringp(i + 1, 0.5f, a); ringp(i, 0.5f, b); face(top, a, b); }
    for (int i = 0; i < 4; ++i) {
        ringp(i, 0.5f, a); ringp(i + 1, 0.5f, b); ringp(i + 1, -0.5f, c); ringp(i, -0.5f, d);
        face(a, b, c);
        face(a, c, d);
    }
    for (int i = 0; i < 4; ++i) { ringp(i, -0.5f, a); ringp(i + 1, -0.5f, b); face(bottom, a, b); }
    return g;
}

} // namespace geometry

// ---------------------------------------------------------------------------------------------- material.rs
MaterialBuilder::MaterialBuilder(float r, float g, float b, float a) {
    memset(&m, 0, sizeof(m));
    m.base_color[0] = r; m.base_color[1] = g; m.base_color[2] = b; m.base_color[3] = a;
    m.roughness = 0.5f; m.ior = 1.0f; m.light_index = -1;
    m.tex_info_0 = m.tex_info_1 = m.tex_info_2 = 0xFFFFFFFFu;
}

// ---------------------------------------------------------------------------------------------- builder.rs
static const size_t kTexBytes = 1024u * 1024u * 4u;   // src/scene/mod.rs:12-13

static std::vector<uint8_t> make_texture(uint8_t (*fn)(uint32_t, uint32_t, int)) {
    std::vector<uint8_t> t(kTexBytes);
    for (uint32_t y = 0; y < 1024; ++y)
        for (uint32_t x = 0; x < 1024; ++x)
            for (int ch = 0; ch < 4; ++ch) t[(y * 1024u + x) * 4u + ch] = fn(x, y, ch);
    return t;
}

SceneBuilder::SceneBuilder() {
    // builder.rs:41-91 — colour {white, 64-px checker, black}; data {white, flat normal, black}
    color_textures.push_back(make_texture([](uint32_t, uint32_t, int) -> uint8_t { return 255; }));
    color_textures.push_back(make_texture([](uint32_t x, uint32_t y, int ch) -> uint8_t {
        if (ch == 3) return 255;
        return (((x / 64) + (y / 64)) % 2 == 0) ? 255 : 0;
    }));
    color_textures.push_back(make_texture([](uint32_t, uint32_t, int ch) -> uint8_t { return ch == 3 ? 255 : 0; }));
    data_textures.push_back(color_textures[0]);
    data_textures.push_back(make_texture([](uint32_t, uint32_t, int ch) -> uint8_t { return ch < 2 ? 128 : 255; }));
    data_textures.push_back(color_textures[2]);
    for (int i = 0; i < 256; ++i) {   // Rgba8UnormSrgb decode (builder.rs:489), IEC 61966-2-1
        double c = i / 255.0;
        srgb_lut[i] = (float)(c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4));
    }
}

uint32_t SceneBuilder::add_color_texture(const uint8_t* p) { color_textures.emplace_back(p, p + kTexBytes); return (uint32_t)color_textures.size() - 1; }
uint32_t SceneBuilder::add_data_texture(const uint8_t* p) { data_textures.emplace_back(p, p + kTexBytes); return (uint32_t)data_textures.size() - 1; }
uint32_t SceneBuilder::add_material(const frt_material& m) { materials.push_back(m); return (uint32_t)materials.size() - 1; }
uint32_t SceneBuilder::add_light(const frt_light& l) { lights.push_back(l); return (uint32_t)lights.size() - 1; }

uint32_t SceneBuilder::add_mesh(const Geometry& g) {
    MeshInfo mi = {(uint32_t)attributes.size(), (uint32_t)indices.size(), {0, 0}};
    attributes.insert(attributes.end(), g.attributes.begin(), g.attributes.end());
    indices.insert(indices.end(), g.indices.begin(), g.indices.end());
    mesh_infos.push_back(mi);
    mesh_positions.push_back(g.positions);
    mesh_index_counts.push_back((uint32_t)g.indices.size());
    return (uint32_t)mesh_infos.size() - 1;
}

void SceneBuilder::add_instance(uint32_t mesh_id, uint32_t mat_id, const Mat4& t) {
    InstanceRec r{};
    r.mesh_id = mesh_id; r.mat_id = mat_id;
    memcpy(r.m, t.m, sizeof(r.m));
    instances.push_back(r);
    built = false;
}

void SceneBuilder::add_quad_light(const float pos[3], const float u[3], const float v[3], const float emission[4]) {
    float cx = u[1] * v[2] - u[2] * v[1], cy = u[2] * v[0] - u[0] * v[2], cz = u[0] * v[1] - u[1] * v[0];
    frt_light l{};
    memcpy(l.position, pos, 12); memcpy(l.u, u, 12); memcpy(l.v, v, 12); memcpy(l.emission, emission, 16);
    l.type_ = 0;
    l.area = sqrtf(cx * cx + cy * cy + cz * cz) * 4.0f;   // |(2u) x (2v)|
    lights.push_back(l);
}
void SceneBuilder::add_sphere_light(const float center[3], float radius, const float emission[4]) {
    frt_light l{};
    memcpy(l.position, center, 12); memcpy(l.emission, emission, 16);
    l.type_ = 1;
    l.area = 4.0f * 3.14159265358979323846f * radius * radius;
    l.v[0] = radius;
    lights.push_back(l);
}
static frt_material emissive_material(size_t light_index, const float color[3], float intensity) {
    return MaterialBuilder(1, 1, 1, 1).light_index((int32_t)light_index)
        .emissive_factor(color[0] * intensity, color[1] * intensity, color[2] * intensity).texture(0);
}
void SceneBuilder::register_quad_light(uint32_t mesh_id, const Mat4& t, const float color[3], float intensity) {
    uint32_t mat = add_material(emissive_material(lights.size(), color, intensity));
    add_instance(mesh_id, mat, t);
    float u[3], v[3];
    xform_vec3(t, 1, 0, 0, u); xform_vec3(t, 0, 0, -1, v);
    for (int i = 0; i < 3; ++i) { u[i] *= 0.5f; v[i] *= 0.5f; }
    const float em[4] = {color[0], color[1], color[2], intensity};
    add_quad_light(&t.m[12], u, v, em);
}
void SceneBuilder::register_sphere_light(uint32_t mesh_id, const Mat4& t, const float color[3], float intensity) {
    uint32_t mat = add_material(emissive_material(lights.size(), color, intensity));
    add_instance(mesh_id, mat, t);
    float x[3];
    xform_vec3(t, 1, 0, 0, x);
    float scale = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    const float em[4] = {color[0], color[1], color[2], intensity};
    add_sphere_light(&t.m[12], scale * 0.5f, em);
}

// Instances -> world-space triangles (contract): p_world = ((c0*x + c1*y) + c2*z) + c3 in f32,
// e1 = v1 - v0, e2 = v2 - v0; world_to_object 3x3 by cofactors in double, rounded once to f32.
void SceneBuilder::flatten() {
    tris.clear(); tri_instance.clear();
    for (size_t ii = 0; ii < instances.size(); ++ii) {
        InstanceRec& in = instances[ii];
        const float* m = in.m;
        double a = m[0], b = m[4], c = m[8], d = m[1], e = m[5], f = m[9], g = m[2], h = m[6], i = m[10];
        double k00 = e * i - f * h, k01 = f * g - d * i, k02 = d * h - e * g;
        double det = a * k00 + b * k01 + c * k02;
        double inv[3][3] = {{k00 / det, (c * h - b * i) / det, (b * f - c * e) / det},
                            {k01 / det, (a * i - c * g) / det, (c * d - a * f) / det},
                            {k02 / det, (b * g - a * h) / det, (a * e - b * d) / det}};
        for (int col = 0; col < 3; ++col) for (int row = 0; row < 3; ++row) in.w2o[3 * col + row] = (float)inv[row][col];
        in.flip = det < 0.0 ? 1u : 0u;
        in.first_tri = (uint32_t)tris.size();
        const std::vector<float>& P = mesh_positions[in.mesh_id];
        const uint32_t* idx = &indices[mesh_infos[in.mesh_id].index_offset];
        uint32_t nidx = mesh_index_counts[in.mesh_id];
        auto world = [&](uint32_t vi, float out[3]) {
            float x = P[4 * vi], y = P[4 * vi + 1], z = P[4 * vi + 2];
            for (int r = 0; r < 3; ++r) out[r] = ((m[r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r];
        };
        for (uint32_t k = 0; k + 2 < nidx; k += 3) {
            float w0[3], w1[3], w2[3];
            world(idx[k], w0); world(idx[k + 1], w1); world(idx[k + 2], w2);
            TriRec t;
            for (int r = 0; r < 3; ++r) { t.v0[r] = w0[r]; t.e1[r] = w1[r] - w0[r]; t.e2[r] = w2[r] - w0[r]; }
            tris.push_back(t);
            tri_instance.push_back((uint32_t)ii);
        }
        in.tri_count = (uint32_t)tris.size() - in.first_tri;
    }
}

void SceneBuilder::build() {
    error.clear();
    // Texture layers and light indices reach the kernels unchecked (sample_layer: base + layer * 4 MiB): validate them here, once.
    // wgpu would reject an out-of-range layer at bind time / clamp the fetch; here it would be an out-of-bounds read on the GPU.
    for (size_t i = 0; i < materials.size() && error.empty(); ++i) {
        const frt_material& m = materials[i];
        const struct { uint32_t id; size_t layers; const char* what; } slots[5] = {
            {m.tex_info_0 & 0xFFFFu, color_textures.size(), "base colour"}, {m.tex_info_0 >> 16, data_textures.size(), "normal"},
            {m.tex_info_1 & 0xFFFFu, data_textures.size(), "occlusion"}, {m.tex_info_1 >> 16, color_textures.size(), "emissive"},
            {m.tex_info_2 & 0xFFFFu, data_textures.size(), "metallic-roughness"}};
        for (const auto& t : slots)
            if (t.id != 0xFFFFu && t.id >= t.layers)
                error = "material " + std::to_string(i) + ": " + t.what + " texture layer " + std::to_string(t.id) + " does not exist (" + std::to_string(t.layers) + " layers)";
        if (m.light_index >= 0 && (size_t)m.light_index >= lights.size())
            error = "material " + std::to_string(i) + ": light_index " + std::to_string(m.light_index) + " does not exist (" + std::to_string(lights.size()) + " lights)";
    }
    if (!error.empty()) { built = false; return; }
    flatten();
    build_bvh2();
    build_gpu_layout();
    built = error.empty();
}

// ---------------------------------------------------------------------------------------------- scenes.rs
namespace scenes {
static const float kPi = 3.14159265358979323846f, kHalfPi = 1.57079632679489661923f;

void create_cornell_box(SceneBuilder& b) {
    uint32_t plane = b.add_mesh(geometry::create_plane());
    uint32_t cube = b.add_mesh(geometry::create_cube());
    uint32_t sphere = b.add_mesh(geometry::create_sphere(3));
    uint32_t crystal = b.add_mesh(geometry::create_crystal());

    uint32_t red = b.add_material(MaterialBuilder(0.65f, 0.05f, 0.05f, 1.0f));
    uint32_t green = b.add_material(MaterialBuilder(0.12f, 0.45f, 0.15f, 1.0f));
    uint32_t white = b.add_material(MaterialBuilder(0.73f, 0.73f, 0.73f, 1.0f));
    uint32_t checker = b.add_material(MaterialBuilder(0.73f, 0.73f, 0.73f, 1.0f).roughness(0.99f).texture(1));
    uint32_t metal = b.add_material(MaterialBuilder(0.8f, 0.8f, 0.8f, 1.0f).metallic(0.01f));
    uint32_t glass = b.add_material(MaterialBuilder(0.5f, 0.8f, 1.0f, 1.0f).glass(1.5f));

    auto TRS = [](const Mat4& t, const Mat4& r, float s) { return mat4_mul(mat4_mul(t, r), mat4_scale(s, s, s)); };
    b.add_instance(plane, checker, mat4_mul(mat4_translation(0, -1, 0), mat4_scale(2, 2, 2)));                 // floor
    b.add_instance(plane, white, TRS(mat4_translation(0, 1, 0), mat4_rotation_x(kPi), 2.0f));                    // ceiling
    b.add_instance(plane, white, TRS(mat4_translation(0, 0, -1), mat4_rotation_x(kHalfPi), 2.0f));               // back
    b.add_instance(plane, red, TRS(mat4_translation(-1, 0, 0), mat4_rotation_z(-kHalfPi), 2.0f));                // left
    b.add_instance(plane, green, TRS(mat4_translation(1, 0, 0), mat4_rotation_z(kHalfPi), 2.0f));                // right
    const float white_light[3] = {1.0f, 1.0f, 1.0f};
    b.register_quad_light(plane, TRS(mat4_translation(0, 0.99f, 0), mat4_rotation_x(kPi), 0.5f), white_light, 10.0f);
    b.add_instance(crystal, glass, mat4_mul(mat4_translation(0.4f, -0.5f, 0.3f), mat4_scale(0.5f, 0.5f, 0.5f)));
    const float blue_light[3] = {0.02f, 0.02f, 0.9f};
    b.register_sphere_light(sphere, mat4_mul(mat4_translation(0.4f, -0.5f, 0.3f), mat4_scale(0.1f, 0.1f, 0.1f)), blue_light, 10.0f);
    b.add_instance(cube, metal, mat4_mul(mat4_mul(mat4_translation(-0.35f, -0.4f + 0.002f, -0.3f), mat4_rotation_y(0.4f)), mat4_scale(0.6f, 1.2f, 0.6f)));
    b.build();
}

static void hsv_to_rgb(float h, float s, float v, float rgb[3]) {   // scenes.rs:226-246
    float c = v * s, x = c * (1.0f - fabsf(fmodf(h * 6.0f, 2.0f) - 1.0f)), m = v - c;
    int sector = h < 1.0f / 6.0f ? 0 : h < 2.0f / 6.0f ? 1 : h < 3.0f / 6.0f ? 2 : h < 4.0f / 6.0f ? 3 : h < 5.0f / 6.0f ? 4 : 5;
    const float table[6][3] = {{c, x, 0}, {x, c, 0}, {0, c, x}, {0, x, c}, {x, 0, c}, {c, 0, x}};
    for (int i = 0; i < 3; ++i) rgb[i] = table[sector][i] + m;
}

void create_restir_scene(SceneBuilder& b) {
    uint32_t plane = b.add_mesh(geometry::create_plane());
    uint32_t sphere = b.add_mesh(geometry::create_sphere(2));
    uint32_t cube = b.add_mesh(geometry::create_cube());
    uint32_t mat_floor = b.add_material(MaterialBuilder(0.73f, 0.73f, 0.73f, 1.0f).roughness(0.99f));
    uint32_t mat_wall = b.add_material(MaterialBuilder(0.73f, 0.73f, 0.73f, 1.0f).roughness(0.99f));
    uint32_t mat_metal = b.add_material(MaterialBuilder(1, 1, 1, 1).metallic(0.2f));
    b.add_instance(plane, mat_floor, mat4_mul(mat4_translation(0, -1, 0), mat4_scale(10, 10, 10)));
    b.add_instance(plane, mat_wall, mat4_mul(mat4_mul(mat4_translation(0, 5, -5), mat4_rotation_x(kHalfPi)), mat4_scale(10, 10, 10)));
    const int rows = 10, cols = 10;
    const float spacing = 1.0f, radius = 0.05f, strength = 20.0f;
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            float x = ((float)c - (float)cols / 2.0f) * spacing, z = ((float)r - (float)rows / 2.0f) * spacing, y = -0.9f;
            float col[3];
            hsv_to_rgb((float)(r * cols + c) / (float)(rows * cols), 0.8f, 1.0f, col);
            uint32_t mat = b.add_material(MaterialBuilder(col[0], col[1], col[2], 1.0f).light_index(r * cols + c)
                                              .emissive_factor(col[0] * strength, col[1] * strength, col[2] * strength));
            b.add_instance(sphere, mat, mat4_mul(mat4_translation(x, y, z), mat4_scale(radius, radius, radius)));
            const float pos[3] = {x, y, z}, em[4] = {col[0], col[1], col[2], strength};
            b.add_sphere_light(pos, radius, em);
        }
    b.add_instance(cube, mat_metal, mat4_mul(mat4_translation(0, -0.5f, 0), mat4_scale(0.5f, 0.5f, 0.5f)));
    b.build();
}
} // namespace scenes

// ---------------------------------------------------------------------------------------------- camera.rs
// CameraController::build_uniform (camera.rs:207-256) for any pose, jitter and previous view-projection.
// prev_view_proj == null stands for the controller's initial Mat4::IDENTITY ("first frame": the unjittered view_proj is sent, :233-238).
// unjittered_out (may be null) receives the second tuple element, which State stores as the next frame's prev_view_proj (state.rs:172).
void camera_build_uniform(const float position[3], float yaw, float pitch, const float* prev_view_proj, float aspect, uint32_t frame_count,
                          uint32_t num_lights, float jitter_x, float jitter_y, frt_camera_uniform* out, float* unjittered_out) {
    const float eye[3] = {position[0], position[1], position[2]};
    const float rad_per_deg = 3.14159265358979323846f / 180.0f;
    float fwd[3] = {cosf(pitch) * cosf(yaw), sinf(pitch), cosf(pitch) * sinf(yaw)};
    v3_normalize_glam(fwd);
    // Mat4::look_at_rh(eye, eye + fwd, Y) -> look_to_rh(eye, (eye + fwd) - eye, Y)
    float f[3] = {(eye[0] + fwd[0]) - eye[0], (eye[1] + fwd[1]) - eye[1], (eye[2] + fwd[2]) - eye[2]};
    v3_normalize_glam(f);
    float s[3] = {f[1] * 0.0f - f[2] * 1.0f, f[2] * 0.0f - f[0] * 0.0f, f[0] * 1.0f - f[1] * 0.0f};   // f x (0,1,0)
    v3_normalize_glam(s);
    float u[3] = {s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]};   // s x f
    auto dot3 = [](const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    Mat4 view = mat4_identity();
    for (int c = 0; c < 3; ++c) { view.m[4 * c] = s[c]; view.m[4 * c + 1] = u[c]; view.m[4 * c + 2] = -f[c]; }
    view.m[12] = -dot3(eye, s); view.m[13] = -dot3(eye, u); view.m[14] = dot3(eye, f);
    // Mat4::perspective_rh(45 deg, aspect, 0.1, 100): depth 0..1
    float half = 0.5f * (45.0f * rad_per_deg);
    float hh = cosf(half) / sinf(half), ww = hh / aspect, rr = 100.0f / (0.1f - 100.0f);
    Mat4 proj_base{};
    proj_base.m[0] = ww; proj_base.m[5] = hh; proj_base.m[10] = rr; proj_base.m[11] = -1.0f; proj_base.m[14] = rr * 0.1f;
    Mat4 vp_unjittered = mat4_mul(proj_base, view);
    Mat4 proj = proj_base;
    proj.m[8] += jitter_x;      // proj_cols[2][0] += jitter.0 (:226): shear of the projection
    proj.m[9] += jitter_y;      // proj_cols[2][1] += jitter.1 (:227)
    Mat4 vp = mat4_mul(proj, view), vi = mat4_inverse(view), pi = mat4_inverse(proj);
    memset(out, 0, sizeof(*out));
    memcpy(out->view_proj, vp.m, 64); memcpy(out->view_inverse, vi.m, 64); memcpy(out->proj_inverse, pi.m, 64);
    memcpy(out->prev_view_proj, prev_view_proj ? prev_view_proj : vp_unjittered.m, 64);
    out->view_pos[0] = eye[0]; out->view_pos[1] = eye[1]; out->view_pos[2] = eye[2]; out->view_pos[3] = 1.0f;
    out->frame_count = frame_count; out->num_lights = num_lights;
    if (unjittered_out) memcpy(unjittered_out, vp_unjittered.m, 64);
}
// CameraController::get_halton_jitter (camera.rs:182-205). The reference multiplies the Halton offsets by 0 (:202-203) — `scale`
// stands for that literal: 0 reproduces the shipped reference, 1 is the sequence the comment above it describes.
void camera_halton_jitter(uint32_t index, uint32_t width, uint32_t height, float scale, float out[2]) {
    auto halton = [](uint32_t i, uint32_t base) {
        float f = 1.0f, r = 0.0f;
        while (i > 0) { f /= (float)base; r += f * (float)(i % base); i /= base; }
        return r;
    };
    float hx = halton(index + 1u, 2u) - 0.5f, hy = halton(index + 1u, 3u) - 0.5f;
    out[0] = (hx * scale) / (float)width;
    out[1] = (hy * scale) / (float)height;
}
void camera_default(float aspect, uint32_t frame_count, uint32_t num_lights, frt_camera_uniform* out) {
    // CameraController::new (camera.rs:40-42) + build_uniform (:207-256), jitter = 0 (:202-203), prev_view_proj = IDENTITY
    // on the first call -> the unjittered view_proj; with a static camera it stays that value (state.rs:172).
    const float eye[3] = {0.0f, 0.0f, 3.0f};
    const float rad_per_deg = 3.14159265358979323846f / 180.0f;
    camera_build_uniform(eye, -90.0f * rad_per_deg, 0.0f, nullptr, aspect, frame_count, num_lights, 0.0f, 0.0f, out, nullptr);
}

} // namespace frt
