// frt_scene.hpp — host-side scene model: C++ mirror of the reference's SceneBuilder / geometry / scenes API
// (src/scene/builder.rs, src/geometry.rs, src/scene/scenes.rs) with the driver-built BLAS/TLAS replaced by an
// explicit host SAH-BVH over the flattened world-space triangle list.
#pragma once
#include "../../include/frt.h"
#include "frt_bvh8.hpp"
#include <vector>
#include <string>
#include <stdint.h>

namespace frt {

struct Mat4 { float m[16]; };   // column-major, m[4*c + r]

Mat4 mat4_identity();
Mat4 mat4_mul(const Mat4& a, const Mat4& b);
Mat4 mat4_translation(float x, float y, float z);
Mat4 mat4_scale(float x, float y, float z);
Mat4 mat4_rotation_x(float a);
Mat4 mat4_rotation_y(float a);
Mat4 mat4_rotation_z(float a);
Mat4 mat4_inverse(const Mat4& a);

// src/geometry.rs:12-18 (no BLAS handle: the acceleration structure is built in SceneBuilder::build)
struct Geometry {
    std::vector<float> positions;            // xyzw per vertex
    std::vector<frt_vertex_attr> attributes;
    std::vector<uint32_t> indices;
};
namespace geometry {
void encode_octahedral_normal(const float n[3], float out[2]);   // geometry.rs:56
Geometry create_plane();                                         // geometry.rs:79  create_plane_blas
Geometry create_cube();                                          // geometry.rs:120 create_cube_blas
Geometry create_sphere(uint32_t subdivisions);                   // geometry.rs:222 create_sphere_blas
Geometry create_crystal();                                       // geometry.rs:350 create_crystal_blas
}

// src/scene/material.rs builder-style helpers
struct MaterialBuilder {
    frt_material m;
    explicit MaterialBuilder(float r, float g, float b, float a);   // Material::new, :31
    MaterialBuilder& light_index(int32_t i) { m.light_index = i; return *this; }
    MaterialBuilder& metallic(float roughness) { m.metallic = 1.0f; m.roughness = roughness; return *this; }   // :54-58 (sic)
    MaterialBuilder& roughness(float r) { m.roughness = r; return *this; }
    MaterialBuilder& glass(float ior) { m.metallic = 0.0f; m.roughness = 0.0f; m.ior = ior; m.transmission = 1.0f; return *this; }
    MaterialBuilder& texture(uint32_t id) { m.tex_info_0 = (m.tex_info_0 & 0xFFFF0000u) | (id & 0xFFFFu); return *this; }
    MaterialBuilder& emissive_factor(float r, float g, float b) { m.emissive_factor[0] = r; m.emissive_factor[1] = g; m.emissive_factor[2] = b; return *this; }
    operator frt_material() const { return m; }
};

struct MeshInfo { uint32_t vertex_offset, index_offset, pad[2]; };   // src/scene/resources.rs:2-8

struct InstanceRec {     // TLAS instance (builder.rs:181-189) + derived data
    uint32_t mesh_id, mat_id, first_tri, tri_count, flip;
    float m[16];
    float w2o[9];        // world_to_object 3x3: w2o[3*c + r]
};

struct TriRec { float v0[3], e1[3], e2[3]; };

// GPU layouts -------------------------------------------------------------------------------------------
// Pair node, 64 B: the boxes of both children of one BVH2 inner node + two child references.
//   one float4 per axis: qa = (c0.min.a, c1.min.a, c0.max.a, c1.max.a) for a = x, y, z (pairs for v_pk_fma_f32, frt_trace.hpp: slab2);
//   q3 = (ref0, ref1, 0, 0) as bits
// ref: bit 31 clear -> pair-node index; bit 31 set -> leaf: bits 0..23 first triangle slot, bits 24..30 triangle count.
// An absent child has ref = 0xFFFFFFFF and an inverted box.
struct PairNode { float q[16]; };
struct QuadNode { float q[32]; };   // frt_trace.hpp: trace4 (four child boxes per node, 128 B)
// Triangle slot, 48 B, in leaf order: (v0.xyz, flattened id bits) (e1.xyz, instance index bits) (e2.xyz, 0)
struct TriSlot { float q[12]; };
// Shading record, 128 B per flattened triangle (frt_shade.hpp: fetch_hit_geometry)
struct ShadeTri { float q[32]; };
struct InstanceDev { uint32_t mesh_id, mat_id, first_tri, flip; float w2o[9]; float pad[3]; };   // 64 B

static const uint32_t kLeafFlag = 0x80000000u;
static const uint32_t kNoChild = 0xFFFFFFFFu;
static const int kMaxBvhDepth = 30;     // traversal stack (frt_trace.hpp kStackDepth = 32) must cover it

class SceneBuilder {
public:
    SceneBuilder();                                    // builder.rs:24 (+ default textures :41-91)
    uint32_t add_mesh(const Geometry& g);              // :123
    uint32_t add_material(const frt_material& m);      // :117
    void add_instance(uint32_t mesh_id, uint32_t mat_id, const Mat4& transform);   // :181 (mask ignored, as there)
    uint32_t add_light(const frt_light& l);
    void register_quad_light(uint32_t mesh_id, const Mat4& t, const float color[3], float intensity);     // :316
    void register_sphere_light(uint32_t mesh_id, const Mat4& t, const float color[3], float intensity);   // :353
    void add_quad_light(const float pos[3], const float u[3], const float v[3], const float emission[4]);   // :392
    void add_sphere_light(const float center[3], float radius, const float emission[4]);                   // :418
    uint32_t add_color_texture(const uint8_t* rgba8);  // :93
    uint32_t add_data_texture(const uint8_t* rgba8);   // :105
    void build();                                      // :431 — flatten + SAH BVH (host only)

    // SceneResources-equivalent host data (src/scene/resources.rs:10-22)
    std::vector<frt_material> materials;
    std::vector<frt_vertex_attr> attributes;
    std::vector<uint32_t> indices;
    std::vector<MeshInfo> mesh_infos;
    std::vector<frt_light> lights;
    std::vector<std::vector<uint8_t>> color_textures, data_textures;
    std::vector<std::vector<float>> mesh_positions;
    std::vector<uint32_t> mesh_index_counts;
    std::vector<InstanceRec> instances;
    // built
    bool built = false;
    std::vector<TriRec> tris;
    std::vector<uint32_t> tri_instance;
    std::vector<frt_bvh2_node> bvh2;
    std::vector<uint32_t> bvh2_tri_index;
    uint32_t bvh_depth = 0, bvh_leaves = 0, bvh_max_leaf = 0;
    std::vector<PairNode> pair_nodes;
    std::vector<QuadNode> quad_nodes;       // the same tree with every other level folded away (build_quad_nodes)
    uint32_t quad_stack_need = 0;           // deepest traversal stack a ray can need in the quad tree
    uint32_t quad_fold = 0;                 // how the quad tree was folded (frt_bvh.cpp: build_quad_nodes): 2 surface-area programme, 1 programme + greedy where the stack bound asks, 0 greedy
    std::vector<uint32_t> qnode_a, qnode_b;     // quantized pair nodes, 4 words per node each (frt_trace.hpp: QBvh)
    float qmin[3] = {0, 0, 0}, qstep[3] = {1, 1, 1};
    std::vector<TriSlot> tri_slots;
    // The same tree as 8-wide nodes with grid boxes (frt_bvh8.hpp; frt_trace.hpp: trace8), built on first use (ensure_wide8): the product's kernels walk
    // the quad tree; the 8-wide walk is an experiment (lib/libfrt_exp.so) and its tree is otherwise read by tests and tools/bvh_quality.cpp only.
    mutable Wide8 wide8;                    // wide8.ok = false: not walkable that way (more than 65,536 nodes)
    mutable std::vector<TriSlot> tri_slots8;   // the triangle slots in the wide tree's order (a node's leaf triangles contiguous)
    mutable bool wide8_built = false;
    void ensure_wide8() const;
    std::vector<ShadeTri> shade_tris;
    std::vector<InstanceDev> instances_dev;
    float srgb_lut[256];
    std::string error;

private:
    void flatten();
    void build_bvh2();
    void build_gpu_layout();
};

namespace scenes {
void create_cornell_box(SceneBuilder& b);    // scenes.rs:9-130
void create_restir_scene(SceneBuilder& b);   // scenes.rs:133-223
}

// src/camera.rs:207-256 at the initial pose
void camera_default(float aspect, uint32_t frame_count, uint32_t num_lights, frt_camera_uniform* out);
void camera_build_uniform(const float position[3], float yaw, float pitch, const float* prev_view_proj, float aspect, uint32_t frame_count,
                          uint32_t num_lights, float jitter_x, float jitter_y, frt_camera_uniform* out, float* unjittered_out);
void camera_halton_jitter(uint32_t index, uint32_t width, uint32_t height, float scale, float out[2]);

} // namespace frt

struct frt_scene { frt::SceneBuilder b; };
