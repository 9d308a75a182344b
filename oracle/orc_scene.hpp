// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
// Scene model restated from the reference host code (PARITY UNPINNED, see orc_math.hpp):
//   src/geometry.rs, src/scene/{material,light,resources,builder,scenes}.rs, src/camera.rs
#pragma once
#include "orc_math.hpp"
#include <vector>
#include <cstdint>

namespace orc {

// src/geometry.rs:4-10 — 32 bytes
struct VertexAttributes { float normal[2]; float uv[2]; float tangent[4]; };
static_assert(sizeof(VertexAttributes) == 32, "VertexAttributes");

// src/geometry.rs:12-18 (without the wgpu BLAS handle)
struct Geometry {
    std::vector<vec4> positions;            // [f32;4], w = 1
    std::vector<VertexAttributes> attributes;
    std::vector<uint32_t> indices;
};

// src/scene/material.rs:2-28 — 64 bytes
struct Material {
    float base_color[4];
    float emissive_factor[3];
    float roughness;
    float metallic, transmission, ior;
    int32_t light_index;
    uint32_t tex_info_0, tex_info_1, tex_info_2, pad_final;
};
static_assert(sizeof(Material) == 64, "Material");

// src/scene/light.rs:1-16 — 64 bytes
struct LightUniform {
    float position[3]; uint32_t type_;
    float u[3]; float area;
    float v[3]; uint32_t pad;
    float emission[4];
};
static_assert(sizeof(LightUniform) == 64, "LightUniform");

// src/scene/resources.rs:2-8
struct MeshInfo { uint32_t vertex_offset, index_offset, pad[2]; };

// src/camera.rs:4-15 — 288 bytes
struct CameraUniform {
    float view_proj[16];
    float view_inverse[16];
    float proj_inverse[16];
    float view_pos[4];
    float prev_view_proj[16];
    uint32_t frame_count, num_lights, padding[2];
};
static_assert(sizeof(CameraUniform) == 288, "CameraUniform");

// One TLAS instance (src/scene/builder.rs:181-189) + what the ray query reports for it.
struct Instance {
    uint32_t mesh_id, mat_id;       // custom index = (mesh_id << 16) | mat_id
    float m[16];                    // object->world, column-major
    float w2o[9];                   // world_to_object 3x3, w2o[3*c + r] = column c, row r
    uint32_t first_tri, tri_count;  // range in the flattened world triangle list
    uint32_t flip;                  // 1 if det(m3x3) < 0 (front-face test is done in object space)
};

// Flattened world-space triangle: v0, e1 = v1 - v0, e2 = v2 - v0 (48 bytes of payload)
struct Tri { vec3 v0, e1, e2; };

// Canonical BVH2 node (32 bytes). count > 0: leaf over tri_index[first .. first+count);
// count == 0: inner, children at nodes[left] and nodes[left + 1].
struct BvhNode { float bmin[3]; uint32_t left_first; float bmax[3]; uint32_t count; };
static_assert(sizeof(BvhNode) == 32, "BvhNode");

struct Scene {
    // SceneBuilder state (src/scene/builder.rs:11-21)
    std::vector<Material> materials;
    std::vector<VertexAttributes> attributes;
    std::vector<uint32_t> indices;
    std::vector<MeshInfo> mesh_infos;
    std::vector<LightUniform> lights;
    std::vector<std::vector<uint8_t>> color_textures;   // 1024x1024 RGBA8 (sRGB)
    std::vector<std::vector<uint8_t>> data_textures;    // 1024x1024 RGBA8 (linear)
    // positions are kept per mesh only for the acceleration structure (builder.rs:149)
    std::vector<std::vector<vec4>> mesh_positions;
    std::vector<uint32_t> mesh_index_count;
    std::vector<Instance> instances;
    // built
    std::vector<Tri> tris;
    std::vector<uint32_t> tri_instance;
    std::vector<BvhNode> bvh_nodes;
    std::vector<uint32_t> bvh_tri_index;
    float srgb_lut[256];
    bool built = false;

    Scene();
    uint32_t add_mesh(const Geometry& g);
    uint32_t add_material(const Material& m);
    void add_instance(uint32_t mesh_id, uint32_t mat_id, const mat4& transform);
    void register_quad_light(uint32_t mesh_id, const mat4& transform, const float color[3], float intensity);
    void register_sphere_light(uint32_t mesh_id, const mat4& transform, const float color[3], float intensity);
    void add_quad_light(const float position[3], const float u[3], const float v[3], const float emission[4]);
    void add_sphere_light(const float center[3], float radius, const float emission[4]);
    uint32_t add_color_texture(const uint8_t* rgba8);
    uint32_t add_data_texture(const uint8_t* rgba8);
    void build();   // flatten instances to world triangles (replaces BLAS/TLAS build)
};

// src/geometry.rs
vec2 encode_octahedral_normal(vec3 n);
Geometry create_plane();
Geometry create_cube();
Geometry create_sphere(uint32_t subdivisions);
Geometry create_crystal();

// src/scene/material.rs:31-47 and builder-style helpers
Material material_new(float r, float g, float b, float a);

// glam 0.30.9 restatements used by scenes.rs / camera.rs
mat4 mat4_identity();
mat4 mat4_from_translation(vec3 t);
mat4 mat4_from_scale(vec3 s);
mat4 mat4_from_rotation_x(float a);
mat4 mat4_from_rotation_y(float a);
mat4 mat4_from_rotation_z(float a);
mat4 mat4_inverse(const mat4& m);

// src/scene/scenes.rs:9-130
void create_cornell_box(Scene& s);
// src/scene/scenes.rs:133-223
void create_restir_scene(Scene& s);

// src/camera.rs:38-56, :207-256 with the fixed initial pose and zero jitter (camera.rs:202-203)
CameraUniform camera_default(float aspect, uint32_t frame_count, uint32_t num_lights);
// src/camera.rs:207-256 for any controller state / jitter / previous view-projection, and :182-205 (scale = the literal 0 of :202-203)
CameraUniform camera_build(vec3 position, float yaw, float pitch, const float* prev_view_proj, float aspect, uint32_t frame_count,
                           uint32_t num_lights, float jitter_x, float jitter_y, float* unjittered_out);
void camera_halton_jitter(uint32_t index, uint32_t width, uint32_t height, float scale, float out[2]);

} // namespace orc
