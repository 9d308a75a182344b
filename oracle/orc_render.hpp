// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path. PARITY UNPINNED (see orc_math.hpp).
// Scalar restatement of the reference frame: src/renderer.rs:349-518 (pass order, ping-pong, frame_count)
// driving gbuffer.wgsl, restir.wgsl, restir_spatial.wgsl, post.wgsl.
#pragma once
#include "orc_trace.hpp"
#include <vector>

namespace orc {

// src/passes/restir.rs:5-14 — 32 bytes
struct Reservoir { uint32_t y; float w_sum; uint32_t M; float W; float s_path[3]; float p_hat; };
static_assert(sizeof(Reservoir) == 32, "Reservoir");

enum Phase { PH_GBUFFER = 1, PH_TEMPORAL = 2, PH_SPATIAL = 4, PH_POST = 8, PH_ALL = 15 };

struct Renderer {
    const Scene* scene;
    uint32_t W, H, max_depth;
    bool use_bvh;
    int nthreads;
    uint32_t frame_count = 0;
    float jitter[2] = {0.0f, 0.0f};          // PostParams.jitter (renderer.rs:14, :376): what render(..., jitter) writes before the post pass

    // RenderTargets (src/renderer.rs:26-170); textures/buffers start zeroed like wgpu resources
    std::vector<vec4> gpos[2], gnormal[2];
    std::vector<uint32_t> galbedo[2];        // rgba8unorm, r in the low byte
    std::vector<vec2> gmotion;               // rg32float
    std::vector<Reservoir> reservoirs[2];    // [0] temporal result, [1] spatial result (restir.rs:362-378, renderer.rs:292-293)
    std::vector<uint64_t> raw;               // rgba16float, r in the low 16 bits
    std::vector<uint32_t> display;           // rgba8unorm
    std::vector<vec4> accum[2];

    TraceStats stats_total;                  // accumulated over frames
    TraceStats stats_stage[4];               // per stage (gbuffer, temporal, spatial, post=unused), accumulated

    Renderer(const Scene* s, uint32_t w, uint32_t h, uint32_t max_depth, bool use_bvh, int nthreads);
    void reset();
    // One reference frame = phases G,T,S,P over all rows, then frame_count += 1 (renderer.rs:515).
    void render(const CameraUniform& cam);
    // Strip form used by the multi-rank tests: run `phases` over rows [y0, y1) only; frame_count is advanced by end_frame().
    void render_phases(const CameraUniform& cam, int phases, uint32_t y0, uint32_t y1);
    void end_frame() { frame_count += 1; }
};

} // namespace orc
