/* frt.h — C ABI of the MI355X-native path tracer (drop-in boundary for the hot path of
 * kokutoupan/fast-raytracing-wgpu: G-buffer -> ReSTIR-PT temporal -> ReSTIR-PT spatial + shade -> post/accumulate).
 *
 * The reference has no FFI; the seam is the Rust API between `State` and `SceneBuilder` / `Renderer`
 * (src/state.rs:57-80, :192-204). Every entry point below names the reference interface it replaces.
 * All structs are byte-identical to the reference's #[repr(C)] types. Plain pointers and sizes only.
 *
 * Conventions: functions returning int return 0 (FRT_OK) or a negative frt_status; the message for the last
 * failure on the calling thread is frt_last_error(). Handles are not thread-safe (like the reference, which
 * drives everything from the winit main thread). Inputs are copied; outputs go to caller-owned buffers.
 * There is NO CPU rendering path: every render entry point fails with FRT_ERR_NO_DEVICE without a HIP device.
 */
#ifndef FRT_H
#define FRT_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum frt_status {
    FRT_OK = 0,
    FRT_ERR_INVALID_ARG = -1,
    FRT_ERR_NO_DEVICE = -2,    /* no HIP device / HIP runtime error: the product never falls back to the CPU */
    FRT_ERR_HIP = -3,
    FRT_ERR_STATE = -4,        /* call order (e.g. render before build) */
    FRT_ERR_LIMIT = -5         /* scene exceeds a compiled limit (BVH depth vs traversal stack, u16 ids) */
} frt_status;

/* ---- data types ------------------------------------------------------------------------- */

/* src/geometry.rs:4-10 — VertexAttributes, 32 B */
typedef struct frt_vertex_attr { float normal[2]; float uv[2]; float tangent[4]; } frt_vertex_attr;

/* src/scene/material.rs:2-28 — Material, 64 B. Texture ids are u16 pairs, 0xFFFF = none. */
typedef struct frt_material {
    float base_color[4];
    float emissive_factor[3];
    float roughness;
    float metallic, transmission, ior;
    int32_t light_index;
    uint32_t tex_info_0;   /* [base colour tex (low), normal tex (high)] */
    uint32_t tex_info_1;   /* [occlusion tex (low), emissive tex (high)] */
    uint32_t tex_info_2;   /* [metallic-roughness tex (low), pad] */
    uint32_t pad_final;
} frt_material;

/* src/scene/light.rs:1-16 — LightUniform, 64 B. type_: 0 quad, 1 sphere (v[0] = radius). */
typedef struct frt_light {
    float position[3]; uint32_t type_;
    float u[3]; float area;
    float v[3]; uint32_t pad;
    float emission[4];
} frt_light;

/* src/camera.rs:4-15 — CameraUniform, 288 B, matrices column-major. */
typedef struct frt_camera_uniform {
    float view_proj[16];
    float view_inverse[16];
    float proj_inverse[16];
    float view_pos[4];
    float prev_view_proj[16];
    uint32_t frame_count, num_lights, padding[2];
} frt_camera_uniform;

/* src/passes/restir.rs:5-14 — Reservoir, 32 B */
typedef struct frt_reservoir { uint32_t y; float w_sum; uint32_t M; float W; float s_path[3]; float p_hat; } frt_reservoir;

/* Canonical BVH2 node, 32 B (replaces the opaque driver BLAS/TLAS of src/scene/builder.rs:143-179, :454-468).
 * count > 0: leaf over tri_index[left_first .. left_first + count); count == 0: children left_first, left_first + 1. */
typedef struct frt_bvh2_node { float bmin[3]; uint32_t left_first; float bmax[3]; uint32_t count; } frt_bvh2_node;

typedef struct frt_scene frt_scene;
typedef struct frt_renderer frt_renderer;

/* ---- errors --------------------------------------------------------------------------------------------- */
const char* frt_last_error(void);
/* Number of visible HIP devices (0 if none / runtime unavailable). Does not initialise a device context. */
int frt_device_count(void);

/* ---- geometry generators: src/geometry.rs:79-434 (create_*_blas without the wgpu BLAS handle) --------------
 * which: 0 plane (:79), 1 cube (:120), 2 icosphere(subdiv) (:222), 3 crystal (:350).
 * Call with null buffers to get counts; then with buffers of nverts*16, nverts*32, nidx*4 bytes. */
int frt_geometry_create(int which, uint32_t subdiv, uint32_t* nverts, uint32_t* nidx,
                        float* pos4, frt_vertex_attr* attrs, uint32_t* idx);
/* src/geometry.rs:56-76 */
void frt_encode_octahedral_normal(const float n[3], float out[2]);
/* src/scene/material.rs:31-47 — Material::new defaults */
void frt_material_default(const float base_color[4], frt_material* out);

/* ---- scene: src/scene/builder.rs SceneBuilder --------------------------------------------------------------- */
frt_scene* frt_scene_create(void);                                   /* SceneBuilder::new, :24 (default textures :41-91) */
void frt_scene_destroy(frt_scene* s);
int frt_scene_add_mesh(frt_scene* s, const float* pos4, uint32_t nverts, const frt_vertex_attr* attrs,
                       const uint32_t* idx, uint32_t nidx);          /* add_mesh, :123 -> mesh id */
int frt_scene_add_material(frt_scene* s, const frt_material* m);     /* add_material, :117 -> material id */
int frt_scene_add_instance(frt_scene* s, uint32_t mesh_id, uint32_t mat_id, const float m_colmajor[16]);   /* add_instance, :181 (mask ignored there too) */
int frt_scene_add_light(frt_scene* s, const frt_light* l);           /* lights.push, :406 / :420 -> light index */
int frt_scene_register_quad_light(frt_scene* s, uint32_t mesh_id, const float m_colmajor[16], const float color[3], float intensity);   /* :316 */
int frt_scene_register_sphere_light(frt_scene* s, uint32_t mesh_id, const float m_colmajor[16], const float color[3], float intensity); /* :353 */
int frt_scene_add_texture(frt_scene* s, int kind /*0 colour (sRGB), 1 data*/, const uint8_t* rgba8_1024x1024);   /* :93-115 -> layer id */
/* SceneBuilder::build, :431 — flattens instances, builds the SAH BVH on the host. No device work. */
int frt_scene_build(frt_scene* s);
/* src/scene/scenes.rs:9-130 and :133-223 — whole-scene factories (built). */
frt_scene* frt_scene_create_cornell_box(void);
frt_scene* frt_scene_create_restir_scene(void);

/* ---- model import: src/scene/loader.rs:9-181 load_gltf (+ a Wavefront OBJ subset, an extension) -------------------------
 * frt_model = the (geometries, materials, images, material_indices) tuple load_gltf returns: one geometry per mesh primitive,
 * images (PNG, JPEG) decoded and Lanczos3-resized to 1024 x 1024 RGBA8 (grey / 16-bit / arithmetic-coded files become the white fallback texture of
 * loader.rs:35-44; see frt_model_warning), materials built as loader.rs:58-99 does (metallic is always 1: material.rs:54-58).
 * Material texture slots hold IMAGE indices until frt_scene_add_gltf_materials remaps them to texture-array layers. */
typedef struct frt_model frt_model;
frt_model* frt_model_load(const char* path);                 /* .gltf / .glb / .vrm, or .obj; NULL + frt_last_error on failure */
void frt_model_destroy(frt_model* m);
int frt_model_counts(const frt_model* m, uint32_t counts[4]);      /* geometries, materials, images, warnings */
int frt_model_geometry_counts(const frt_model* m, uint32_t geo, uint32_t* nverts, uint32_t* nidx, uint32_t* material_index);
int frt_model_geometry_get(const frt_model* m, uint32_t geo, float* pos4, frt_vertex_attr* attrs, uint32_t* idx);   /* any pointer may be NULL */
int frt_model_material_get(const frt_model* m, uint32_t i, frt_material* out);
int frt_model_material_set(frt_model* m, uint32_t i, const frt_material* in);   /* scenes.rs:392-410 rewrites loaded materials before adding them */
int frt_model_image_get(const frt_model* m, uint32_t i, uint8_t* rgba8_1024x1024);
const char* frt_model_warning(const frt_model* m, uint32_t i);    /* what the reference prints to stdout; NULL past the end */
/* src/scene/builder.rs:191-292, :294-300, :302-314. ids arrays are caller-owned ([materials] / [geometries]); return = count written */
int frt_scene_add_gltf_materials(frt_scene* s, const frt_model* m, uint32_t* mat_ids);
int frt_scene_add_gltf_meshes(frt_scene* s, const frt_model* m, uint32_t* mesh_ids);
int frt_scene_add_gltf_instances(frt_scene* s, const frt_model* m, const uint32_t* mesh_ids, uint32_t n_mesh, const uint32_t* mat_ids, uint32_t n_mat,
                                 const float transform_colmajor[16]);
/* src/scene/scenes.rs:246-322 create_gltf_scene (floor plane, 15-intensity quad light, the model; built). Differs from the
 * reference in one error case: there a model that fails to load is logged and an EMPTY scene is built; here the call returns
 * NULL with the loader's message in frt_last_error (scenes without triangles cannot be built). */
frt_scene* frt_scene_create_gltf_scene(const char* path, const float model_transform_colmajor[16], const float light_transform_colmajor[16]);

/* Introspection (tests): counts[8] = tris, instances, materials, lights, meshes, attributes, indices, bvh2 nodes */
int frt_scene_counts(const frt_scene* s, uint32_t counts[8]);
/* which: 0 tris (9 f32: v0,e1,e2), 1 tri_instance (u32), 2 materials, 3 lights, 4 attributes, 5 indices, 6 mesh infos (16 B),
 * 7 instances (120 B: mesh,mat,first_tri,tri_count,flip u32; m[16]; w2o[9] f32), 8 bvh2 nodes (32 B), 9 bvh2 tri_index (u32);
 * the device forms of the tree (frt_scene_tree_stats gives the counts): 10 quad nodes (128 B), 11 8-wide compressed nodes (80 B, csrc/frt_bvh8.hpp),
 * 12 triangle slots in the 8-wide tree's order (48 B: v0, id; e1, instance; e2, 0), 13 triangle slots in BVH2 leaf order (48 B),
 * 14 the float boxes behind the 8-wide nodes' grid boxes (192 B per node: 8 x lo.xyz, hi.xyz; host data for tools/bvh_quality.cpp) */
int frt_scene_get(const frt_scene* s, int which, void* out);
/* stats[8]: quad nodes, deepest traversal stack of the quad tree, 8-wide nodes (0: the scene has no 8-wide tree: more than 65,536 nodes), deepest stack of
 * the 8-wide tree, its levels, sum of its nodes' child counts, its triangle slots, how the quad tree was folded (2 surface-area programme, 1 programme where
 * the traversal-stack bound allows and the greedy fold elsewhere, 0 greedy fold) */
int frt_scene_tree_stats(const frt_scene* s, uint32_t stats[8]);
/* bvh stats[4]: max depth, leaves, max leaf size, wide-node count */
int frt_scene_bvh_stats(const frt_scene* s, uint32_t stats[4]);

/* ---- camera: src/camera.rs:207-256 build_uniform at the initial pose (:40-42), jitter 0 (:202-203) ------------ */
void frt_camera_default(float aspect, uint32_t frame_count, uint32_t num_lights, frt_camera_uniform* out);
/* CameraController::build_uniform, src/camera.rs:207-256, for any pose (position, yaw, pitch: the controller's state, :38-56), jitter
 * (the projection shear of :224-228; NULL = (0, 0)) and previous view-projection (NULL = the controller's initial IDENTITY, i.e. the
 * first frame, :233-238). unjittered_view_proj (16 floats, may be NULL) is the second element of the returned tuple, which the caller
 * keeps as the next frame's prev_view_proj (state.rs:172). */
int frt_camera_build_uniform(const float position[3], float yaw, float pitch, const float* prev_view_proj_colmajor, float aspect,
                             uint32_t frame_count, uint32_t num_lights, const float jitter[2], frt_camera_uniform* out, float* unjittered_view_proj);
/* CameraController::get_halton_jitter, src/camera.rs:182-205. `scale` stands for the literal 0 the reference multiplies the Halton
 * offsets by (:202-203): 0 reproduces the shipped reference (no jitter), 1 gives the sequence its comments describe. */
void frt_camera_halton_jitter(uint32_t index, uint32_t width, uint32_t height, float scale, float out[2]);

/* ---- renderer: src/renderer.rs ---------------------------------------------------------------------------- */
typedef struct frt_render_opts {
    uint32_t max_depth;       /* MAX_DEPTH, restir.wgsl:5; 0 -> 8 */
    int32_t device;           /* HIP device ordinal */
    void* stream;             /* hipStream_t to enqueue on; NULL -> a stream owned by the renderer, unless FRT_FLAG_USE_STREAM */
    uint32_t row_begin;       /* rows [row_begin, row_end) owned by this renderer (image strip); 0,0 -> whole image */
    uint32_t row_end;
    void* device_arena;       /* optional caller-owned device memory for all per-pixel buffers (frt_renderer_arena_bytes) */
    uint64_t arena_bytes;
    uint32_t flags;           /* FRT_FLAG_* */
    uint32_t motion_halo_rows;/* strips only: rows beyond the strip for which the caller keeps PREVIOUS-frame state valid (spatial reservoirs,
                                 accumulation: frt/dist.py exchanges them before the temporal stage) so that temporal reprojection and the
                                 history fetch of a MOVING camera may land there; the G-buffer halo grows to cover them. 0 = static camera.
                                 Reads that fall outside are counted in frt_stats.halo_overflow (the frame then differs from a 1-GPU frame). */
    uint32_t queue_capacity;  /* slots of each continuation queue (paths parked between two launches of a traced stage); 0 -> sized from the
                                 share of paths that reach the first cut and grown by frt_renderer_stats after an overflow. Any value is
                                 safe: a path that finds its queue full is finished in place (frt_stats.queue_overflow counts them). */
    uint32_t cut_depths[4];   /* ascending bounce depths at which the traced stages park their surviving paths in the continuation queues and resume
                                 them, dense again, in a further launch. All zero -> the library's choice (3 and 4; renderers of
                                 fewer than 0.8 M pixels: 3). cut_depths[0] = 0xFFFFFFFF -> never cut. Entries that are not ascending are skipped.
                                 Pixels do not depend on it. */
} frt_render_opts;
#define FRT_FLAG_TIMING 1u          /* record per-stage HIP events every frame (frt_stats.ms_*) */
#define FRT_FLAG_PIPELINE 8u        /* two-stream schedule: the G-buffer and the T-trace half of the temporal stage of the NEXT
                                       frame run on a second stream beside this frame's spatial continuation launches and post (the latency-bound
                                       part of the frame). The next frame's camera is speculated (this camera, frame_count + 1, prev_view_proj =
                                       view_proj: a camera that did not move) and checked against the real uniform at the next render call; a wrong
                                       guess is dropped and redone in order. Same pixels; reads, frt_renderer_sync and frt_renderer_stats see
                                       completed frames as without the flag. Callers that read the G-buffer / motion targets through
                                       frt_renderer_buffer_info on their own stream call frt_renderer_fence first. */
#define FRT_FLAG_OVERLAP_POST FRT_FLAG_PIPELINE   /* round-1 name */
#define FRT_FLAG_USE_STREAM 4u      /* opts->stream is authoritative even when NULL (= the legacy default stream, e.g. torch's current stream) */
#define FRT_FLAG_COMPACTION 2u      /* EXPERIMENTS BUILD ONLY (lib/libfrt_exp.so, `make experiments`): temporal / spatial stages through the
                                       workgroup-compacting kernels (measured slower, profiles/r1_v3_*). The product library rejects the flag. */
#define FRT_FLAG_THIRD_GSET 16u     /* with FRT_FLAG_PIPELINE: a whole-frame renderer also owns the third G-buffer / motion / candidate set (60 B per
                                       pixel outside the arena) that strip renderers own, so that the next frame's G-buffer + T-trace need not wait
                                       for this frame's T-merge. No gain for a whole frame; lets tests drive that schedule. */

/* EXPERIMENTS BUILD ONLY (lib/libfrt_exp.so, `make experiments`; the product library rejects them): round 4's two measured-and-not-kept walks. Same pixels. */
#define FRT_FLAG_WALK_WIDE 32u      /* the traced kernels walk the scene's 8-WIDE tree with 16-bit grid boxes (csrc/frt_bvh8.hpp, frt_trace.hpp: trace8; a tree of at
                                       most 28 KiB is copied into every traced workgroup's LDS) instead of the 4-wide one: Cornell Box 1.57 vs 1.54 ms per frame,
                                       larger scenes 17 - 33 % slower (profiles/r4_experiments/wide8.md) */
#define FRT_FLAG_WALK_WIDE_HBM 64u  /* the same walk with the tree read from HBM / L1 even when it would fit a workgroup's LDS */
#define FRT_FLAG_WG_TRACE 128u      /* the traced kernels walk their rays COLLECTIVELY: before every walk a 16x16 workgroup re-deals its rays to dense waves sorted by
                                       direction octant through LDS (csrc/experiments/frt_round4_walks.hpp: wg_trace): 1.65 vs 1.43 ms per frame
                                       (profiles/r4_experiments/collective_walks.md) */

/* FRT_PHASE_SPATIAL = the whole spatial stage. A strip renderer may issue it in two parts so that the halo exchange overlaps with
 * work: FRT_PHASE_SPATIAL_INNER (rows whose 10-row reuse neighbourhood lies inside the strip: needs nothing from a neighbour),
 * then, once the neighbours' temporal reservoirs have arrived, FRT_PHASE_SPATIAL_EDGE (the remaining rows + the continuations). */
enum { FRT_PHASE_GBUFFER = 1, FRT_PHASE_TEMPORAL = 2, FRT_PHASE_SPATIAL = 4, FRT_PHASE_POST = 8, FRT_PHASE_ALL = 15,
       FRT_PHASE_SPATIAL_INNER = 16, FRT_PHASE_SPATIAL_EDGE = 32 };

/* Per-pixel buffers (RenderTargets, src/renderer.rs:26-170; reservoirs src/passes/restir.rs:329-348) */
enum {
    FRT_BUF_GPOS = 0,        /* rgba32f  16 B/px, x2 ping-pong */
    FRT_BUF_GNORMAL = 1,     /* rgba32f  16 B/px, x2 */
    FRT_BUF_GALBEDO = 2,     /* rgba8    4 B/px, x2 */
    FRT_BUF_GMOTION = 3,     /* rg32f    8 B/px; index 0 = the last rendered frame (the reference has one motion texture); under
                                FRT_FLAG_OVERLAP_POST there are two slots internally and index 1 is the other one */
    FRT_BUF_RESERVOIR = 4,   /* 32 B/px, [0] temporal result, [1] spatial result */
    FRT_BUF_RAW = 5,         /* rgba16f  8 B/px */
    FRT_BUF_DISPLAY = 6,     /* rgba8    4 B/px */
    FRT_BUF_ACCUM = 7,       /* vec4f    16 B/px, x2 */
    FRT_BUF_CANDIDATE = 8    /* vec4f    16 B/px: (v1_pos, p_hat) of the temporal stage's fresh candidate path, T-trace -> T-merge (no reference
                                counterpart: restir.wgsl keeps it in registers between :825 and :826) */
};

typedef struct frt_stats {
    uint64_t rays_closest;    /* closest-hit rays issued (device-counted), since create/reset */
    uint64_t rays_any;        /* any-hit (shadow / visibility) rays issued */
    uint64_t frames;          /* frames rendered since create/reset */
    double ms_stage[4];       /* summed kernel time per stage (gbuffer, temporal, spatial, post); FRT_FLAG_TIMING only */
    uint64_t launches[4];     /* launches per stage */
    uint64_t rays_stage[4][2];/* per stage {closest, any}; post issues none */
    uint64_t halo_overflow;   /* strips: previous-frame reads (reprojection, history) outside own rows +- motion_halo_rows; 0 for a whole frame */
    double ms_merge;          /* summed T-merge kernel time (ms_stage[1] is T-trace); FRT_FLAG_TIMING only */
    uint64_t queue_overflow;  /* paths that found their continuation queue full and were finished in place */
    uint64_t queue_capacity;  /* current slots of the largest continuation queue (the spatial stage's first); the others are sized in proportion */
    uint64_t speculated_frames;       /* FRT_FLAG_PIPELINE: frames whose G-buffer + T-trace ran ahead and were adopted */
    uint64_t discarded_speculations;  /* ... and speculated work that did not match the next camera and was dropped */
    uint64_t queue_bytes;     /* device bytes of all continuation queues at the current capacity */
} frt_stats;

uint64_t frt_renderer_arena_bytes(uint32_t width, uint32_t height);
/* Renderer::new, src/renderer.rs:206. Uploads the scene replica to opts->device. */
frt_renderer* frt_renderer_create(const frt_scene* s, uint32_t width, uint32_t height, const frt_render_opts* opts);
void frt_renderer_destroy(frt_renderer* r);
/* Renderer::render, src/renderer.rs:349 — enqueue the four stages for one frame, then frame_count += 1 (:515). Asynchronous. */
int frt_renderer_render(frt_renderer* r, const frt_camera_uniform* cam);
/* Strip form: enqueue only `phases` (multi-GPU: a halo exchange sits between TEMPORAL and SPATIAL); frt_renderer_end_frame advances frame_count. */
int frt_renderer_render_phases(frt_renderer* r, const frt_camera_uniform* cam, int phases);
int frt_renderer_end_frame(frt_renderer* r);
/* PostParams.jitter (renderer.rs:14, :361-379): render(..., jitter) writes it before the post pass. set_jitter applies to the post
 * stages enqueued after it; render_jittered = set_jitter + render. (0, 0) — the shipped reference, camera.rs:202-203 — is the default.
 * Non-zero jitter makes post take bilinear radiance / albedo taps (post.wgsl:72-78, :97-109, :152-158); whole-frame renderers only. */
int frt_renderer_set_jitter(frt_renderer* r, float jitter_x, float jitter_y);
int frt_renderer_render_jittered(frt_renderer* r, const frt_camera_uniform* cam, float jitter_x, float jitter_y);
int frt_renderer_sync(frt_renderer* r);                       /* block until enqueued work is done */
/* Stream-level fence, no host wait: the renderer's stream (opts->stream) is ordered behind everything the renderer has enqueued on
 * its internal second stream (FRT_FLAG_PIPELINE). Call before touching buffers from frt_renderer_buffer_info on the caller's stream
 * (halo exchange, zero-copy views). */
int frt_renderer_fence(frt_renderer* r);
/* Orders the edge stream (frt_renderer_stream(r, 2)) behind the open frame's T-merge, now, with the event the renderer recorded behind T-merge anyway
 * (FRT_PHASE_SPATIAL_EDGE does the same later): for a caller that places a transfer of the T-merge's output IN that stream, in front of the edge rows'
 * launches (frt/rccl.py). Call after frt_renderer_render_phases(... TEMPORAL). A renderer without an edge stream: no-op. */
int frt_renderer_order_edge_stream(frt_renderer* r);
/* The streams the renderer enqueues on: which = 0 the main stream (opts->stream: T-merge, spatial, post — everything a halo exchange
 * reads), 1 the second stream of FRT_FLAG_PIPELINE (G-buffer + T-trace of the next frame), 2 the stream on which a strip renderer
 * under FRT_FLAG_PIPELINE launches FRT_PHASE_SPATIAL_EDGE's pixel kernels: a caller that receives halo rows orders THAT stream behind
 * the transfer before it issues the phase (frt/dist.py). Both equal stream 0 without the flag. */
void* frt_renderer_stream(const frt_renderer* r, int which);
uint32_t frt_renderer_frame_count(const frt_renderer* r);     /* renderer.frame_count, :198 */
int frt_renderer_reset(frt_renderer* r);                      /* frame_count = 0 only, as state.rs:152 / renderer.rs:346 (buffers keep their contents) */
int frt_renderer_clear(frt_renderer* r);                      /* back to the state right after create: zeroed targets, frame_count = 0, stats = 0 */
/* Read-back of post_processed_texture (state.rs:226-278) and of any other target; syncs first. index = ping-pong slot. */
int frt_renderer_read_display(frt_renderer* r, uint8_t* rgba8);
int frt_renderer_read_accum(frt_renderer* r, float* rgba32f);  /* the slot written by the last frame */
int frt_renderer_read_buffer(frt_renderer* r, int buf, int index, void* out);
/* Row-range copies to / from the host: rows [y0, y1) of a target, tightly packed (gathers of image strips, halo exchange through the host). */
int frt_renderer_read_rows(frt_renderer* r, int buf, int index, uint32_t y0, uint32_t y1, void* out);
int frt_renderer_write_rows(frt_renderer* r, int buf, int index, uint32_t y0, uint32_t y1, const void* in);
/* Device address / geometry of a target, for halo exchange and gathers by the caller (rows are contiguous, full-frame pitch). */
int frt_renderer_buffer_info(const frt_renderer* r, int buf, int index, void** device_ptr, uint32_t* bytes_per_pixel);
/* Rows this renderer computes per phase given its strip: out[0..1] gbuffer, [2..3] temporal, [4..5] spatial, [6..7] post */
int frt_renderer_phase_rows(const frt_renderer* r, uint32_t out[8]);
int frt_renderer_stats(frt_renderer* r, frt_stats* out);      /* syncs first */
/* Switch FRT_FLAG_TIMING on or off after creation (the per-stage HIP events cost ~25 us per frame: too much for a thin strip) */
int frt_renderer_set_timing(frt_renderer* r, int on);

/* ---- N GPUs behind one call (`ngpus`; section 8e) -----------------------------------------------------------------
 * In the reference one call renders one frame: Renderer::render, src/renderer.rs:349-518, called from State::render, src/state.rs:192-204.
 * frt_multi_renderer is that call for a node with several GPUs: ONE process, `ndev` strip renderers (two-stream schedule), the scene
 * replicated on every device, the frame cut into horizontal strips of equal work, the per-frame halo rows moved by peer copies
 * (hipMemcpyPeerAsync over xGMI) on the right streams, the image gathered when it is read. A Rust `Renderer` over this handle gets N GPUs
 * without knowing about strips. Images are bit-identical to a single frt_renderer's.
 * devices: `ndev` HIP ordinals (NULL = 0 .. ndev-1); an ordinal may repeat (several strips on one GPU: how the path is tested on a 1-GPU box).
 * opts: max_depth, motion_halo_rows (moving camera: rows of previous-frame state exchanged around every strip), queue_capacity and
 * FRT_FLAG_TIMING are honoured; device, stream, rows and arena are set per strip by the library. */
typedef struct frt_multi_renderer frt_multi_renderer;
frt_multi_renderer* frt_multi_renderer_create(const frt_scene* s, uint32_t width, uint32_t height, uint32_t ndev, const int32_t* devices,
                                              const frt_render_opts* opts);                       /* Renderer::new, src/renderer.rs:206 */
void frt_multi_renderer_destroy(frt_multi_renderer* m);
int frt_multi_renderer_render(frt_multi_renderer* m, const frt_camera_uniform* cam);              /* Renderer::render, :349 — asynchronous */
int frt_multi_renderer_sync(frt_multi_renderer* m);
uint32_t frt_multi_renderer_frame_count(const frt_multi_renderer* m);                             /* renderer.frame_count, :198 */
int frt_multi_renderer_reset(frt_multi_renderer* m);                                              /* frame_count = 0, state.rs:152 */
/* A strip whose step fails (a HIP error in the middle of a frame) leaves the other strips with a half-enqueued frame: the handle is then FAILED and
 * every render call returns FRT_ERR_STATE until frt_multi_renderer_clear, which waits for the devices, closes every strip's open frame and puts every
 * strip back into the state right after create (frt_renderer_clear: zeroed targets, frame_count = 0, stats = 0). */
int frt_multi_renderer_clear(frt_multi_renderer* m);
/* PostParams.jitter (renderer.rs:14, :361-379). Only (0, 0) — the shipped reference, camera.rs:202-203 — is accepted when the frame is cut into
 * strips: post's bilinear taps use Repeat addressing and read the opposite image edge (post.wgsl:72-78), which lives on another device. */
int frt_multi_renderer_set_jitter(frt_multi_renderer* m, float jitter_x, float jitter_y);
/* Device-side gather (state.rs:226-278 reads ONE texture per frame): every strip's own rows of `buf`[index] are copied into the full-frame buffer
 * `dst` in the memory of HIP device `device` — peer copies over xGMI on the strips' copy streams, ordered behind the frames enqueued so far, no host
 * staging. `stream` (a hipStream_t of `device`) is ordered behind the copies; stream = NULL: the call returns when the rows have arrived.
 * The next frame's writers of those rows are ordered behind the copies by the library. dst: width * height * bytes-per-pixel of `buf`. */
int frt_multi_renderer_gather(frt_multi_renderer* m, int buf, int index, int32_t device, void* dst, void* stream);
/* out[0] = neighbouring strip pairs on different devices, out[1] = of those, pairs with direct peer access enabled in both directions */
int frt_multi_renderer_peer_access(const frt_multi_renderer* m, uint32_t out[2]);
/* TESTING: the next render call fails on strip `strip` in step `step` (0: T-merge half of the frame, 1: spatial + post half) with FRT_ERR_HIP,
 * as if a HIP call had failed there — how tests reach the failed state above without breaking a device. */
int frt_multi_renderer_inject_failure(frt_multi_renderer* m, uint32_t strip, int step);
/* post_processed_texture, state.rs:226-278: frt_multi_renderer_gather on the first strip's device + ONE device-to-host copy */
int frt_multi_renderer_read_display(frt_multi_renderer* m, uint8_t* rgba8);
int frt_multi_renderer_read_accum(frt_multi_renderer* m, float* rgba32f);
int frt_multi_renderer_read_buffer(frt_multi_renderer* m, int buf, int index, void* out);         /* any target, every strip's own rows */
int frt_multi_renderer_stats(frt_multi_renderer* m, frt_stats* out);                              /* summed over the strips */
int frt_multi_renderer_boundaries(const frt_multi_renderer* m, uint32_t* rows_out);               /* ndev + 1 row indices; returns ndev */

#ifdef __cplusplus
}
#endif
#endif /* FRT_H */
