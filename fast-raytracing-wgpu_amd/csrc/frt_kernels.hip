// frt_kernels.hip — gfx950 kernels of the hot path (replaces src/passes/*.rs dispatches + src/shaders/*.wgsl).
//
// Launch shape: one wave64 owns an 8x8 pixel tile (the reference's @workgroup_size(8,8), gbuffer.rs:302), four waves
// per 256-thread workgroup = a 16x16 block. Each lane's BVH traversal stack is a column of an LDS array
// (kStackDepth x 256 words = 32 KiB per workgroup; lane-consecutive addresses -> conflict-free ds_read/ds_write_b32).
// Scene data (pair nodes 64 B, triangle slots 48 B) is read with 16-byte loads; per-pixel buffers are pixel-linear
// float4 / 8-byte / 4-byte streams, so every wave-level access is a set of full 128-byte row segments.
#include "frt_path.hpp"
#include "frt_kernels.hpp"
#include <algorithm>

namespace frt {

static constexpr int kBlock = 256;

__device__ __forceinline__ bool tile_pixel(const FrameView& fv, uint32_t& px, uint32_t& py) {
    uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    px = blockIdx.x * 16u + (wave & 1u) * 8u + (lane & 7u);
    py = fv.y0 + blockIdx.y * 16u + (wave >> 1) * 8u + (lane >> 3);
    return px < fv.W && py < fv.y1;
}

__device__ __forceinline__ void flush_ray_counters(const FrameView& fv, uint32_t n_closest, uint32_t n_any, uint32_t* s_cnt) {
    // wave-level sum via cross-lane adds, then one LDS atomic per wave and one global atomic pair per workgroup
    for (int off = 32; off > 0; off >>= 1) {
        n_closest += __shfl_down(n_closest, off, 64);
        n_any += __shfl_down(n_any, off, 64);
    }
    if ((threadIdx.x & 63u) == 0u) { atomicAdd(&s_cnt[0], n_closest); atomicAdd(&s_cnt[1], n_any); }
    __syncthreads();
    if (threadIdx.x == 0u) {
        if (s_cnt[0]) atomicAdd(&fv.ray_counters[0], (unsigned long long)s_cnt[0]);
        if (s_cnt[1]) atomicAdd(&fv.ray_counters[1], (unsigned long long)s_cnt[1]);
    }
}

// G-buffer: one primary ray per pixel, coherent within the 8x8 tile (lane utilisation 96 %): plain thread-per-pixel launch.
__global__ void __launch_bounds__(kBlock) gbuffer_kernel(SceneView sc, FrameView fv) {
    __shared__ uint32_t s_stack[kStackDepth * kBlock];
    __shared__ uint32_t s_cnt[2];
    if (threadIdx.x < 2u) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t px, py;
    bool active = tile_pixel(fv, px, py);
    PathCtx c(sc, fv, &s_stack[threadIdx.x], (uint32_t)kBlock);
    if (active) gbuffer_pixel(c, px, py);
    bool counted = active && py >= fv.own_y0 && py < fv.own_y1;
    flush_ray_counters(fv, counted ? c.n_closest : 0u, counted ? c.n_any : 0u, s_cnt);
}

// Temporal (STAGE 1, trace_path variant 0) and spatial + shade (STAGE 2, variant 1) as a persistent wave-level state machine.
//
// Every lane carries one pixel's resumable state (frt_path.hpp). A lane whose pixel has finished is refilled at once from a
// tile-ordered global queue: the idle lanes are found with a wave ballot, the leader takes popcount(ballot) queue slots
// with ONE atomic, and each idle lane picks slot base + (its rank in the ballot). So the wave is kept compact — lanes at
// different bounce depths, fresh lanes at depth 0 and (spatial) lanes still walking their neighbour list all share the two
// traversal phases of an iteration:
//     A  closest-hit rays of lanes at depth >= 1        C  any-hit rays: NEE shadow rays + spatial visibility rays
//     B  shading up to the shadow ray / neighbour prep  D  NEE add + BSDF sample / reservoir merge; finalise finished pixels
// Results do not depend on the order pixels are taken in: seeds are functions of (pixel, frame) only (restir.wgsl:797-798).
enum : uint32_t { LANE_EMPTY = 0, LANE_NEIGH = 1, LANE_PATH = 2 };

template <int STAGE>
__global__ void __launch_bounds__(kBlock) wavefront_kernel(SceneView sc, FrameView fv, uint32_t* queue) {
    __shared__ uint32_t s_stack[kStackDepth * kBlock];
    __shared__ uint32_t s_cnt[2];
    if (threadIdx.x < 2u) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    constexpr int VARIANT = STAGE == 1 ? 0 : 1;
    const uint32_t tiles_x = (fv.W + 7u) >> 3, tiles_y = (fv.y1 - fv.y0 + 7u) >> 3;
    const uint32_t total = tiles_x * tiles_y * 64u;
    PathCtx c(sc, fv, &s_stack[threadIdx.x], (uint32_t)kBlock);
    PathState st;
    SpatialState ss;
    st.done = true; st.depth = 0u; ss.i = 0u; ss.n = 0u;
    uint32_t mode = LANE_EMPTY, tot_closest = 0u, tot_any = 0u;
    bool exhausted = false, owned = false;

    for (;;) {
        // ---- regeneration (ballot compaction): idle lanes pull pixels until each has work or the queue is dry
        while (mode == LANE_EMPTY && !exhausted) {
            unsigned long long idle = __ballot(1);
            uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            uint32_t base = 0u;
            if (rank == 0u) base = atomicAdd(queue, (uint32_t)__popcll(idle));
            base = __shfl(base, __ffsll((long long)idle) - 1, 64);
            uint32_t idx = base + rank;
            if (idx >= total) { exhausted = true; break; }
            uint32_t tile = idx >> 6, l = idx & 63u;
            uint32_t px = (tile % tiles_x) * 8u + (l & 7u);
            uint32_t py = fv.y0 + (tile / tiles_x) * 8u + (l >> 3);
            if (px >= fv.W || py >= fv.y1) continue;
            uint32_t pix = py * fv.W + px;
            owned = py >= fv.own_y0 && py < fv.own_y1;
            if (STAGE == 1) { if (temporal_begin(c, st, pix)) mode = LANE_PATH; }
            else { if (spatial_begin(c, ss, pix)) mode = ss.n > 0u ? LANE_NEIGH : LANE_PATH; }
        }
        if (!__any(mode != LANE_EMPTY)) break;   // every lane idle => every lane saw the queue dry

        // ---- A: closest-hit rays
        HitRec h;
        h.tri = 0xFFFFFFFFu; h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.inst = 0u; h.front = false;
        f3 origin = splat3(0.0f);
        if (mode == LANE_PATH && st.depth >= 1u) {
            if (path_pre_closest(c, st, origin)) {
                c.n_closest++;
                trace<false>(sc, origin, st.next_dir, 0.001f, 100.0f, c.stk, c.stride, h);
            }
        }
        // ---- B: shade up to the shadow ray / prepare the next spatial neighbour
        AnyReq req;
        req.want = false; req.o = splat3(0.0f); req.d = splat3(0.0f); req.tmin = 0.0f; req.tmax = 0.0f;
        if (mode == LANE_PATH) { if (!st.done) path_shade<VARIANT>(c, st, h, origin, req); }
        else if (STAGE == 2 && mode == LANE_NEIGH) spatial_neighbor_prepare(c, ss, req);
        // ---- C: any-hit rays (NEE shadow rays and reconnection visibility rays together)
        bool visible = true;
        if (req.want) {
            HitRec s;
            c.n_any++;
            trace<true>(sc, req.o, req.d, req.tmin, req.tmax, c.stk, c.stride, s);
            visible = s.tri == 0xFFFFFFFFu;
        }
        // ---- D: finish the step; retire finished pixels
        if (mode == LANE_PATH) {
            if (!st.done) path_post_any(c, st, visible);
            if (st.done) {
                if (STAGE == 1) temporal_finalize(c, st); else spatial_finalize(c, ss, st);
                if (owned) { tot_closest += c.n_closest; tot_any += c.n_any; }
                c.n_closest = 0u; c.n_any = 0u;
                mode = LANE_EMPTY;
            }
        } else if (STAGE == 2 && mode == LANE_NEIGH) {
            spatial_neighbor_finish(ss, visible);
            if (ss.i >= ss.n) { path_begin(c, st, ss.pix, ss.r.y); mode = LANE_PATH; }
        }
    }
    flush_ray_counters(fv, tot_closest, tot_any, s_cnt);
}

__global__ void __launch_bounds__(kBlock) post_kernel(FrameView fv) {
    uint32_t px, py;
    if (tile_pixel(fv, px, py)) post_pixel(fv, px, py);
}

static dim3 grid_for(const FrameView& fv) { return dim3((fv.W + 15u) / 16u, (fv.y1 - fv.y0 + 15u) / 16u, 1u); }

// Resident workgroups per CU of the persistent kernels (VGPR / LDS limited); the grid is CUs x this, so every workgroup is
// resident from the start and the queue drains evenly. (No workgroup waits on another: a smaller residency only adds a tail.)
uint32_t persistent_blocks_per_cu(int stage) {
    int n = 0;
    hipError_t e = stage == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wavefront_kernel<1>, kBlock, 0)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wavefront_kernel<2>, kBlock, 0);
    if (e != hipSuccess || n < 1) n = 2;
    return (uint32_t)n;
}

hipError_t launch_stage(int stage, const SceneView& sc, const FrameView& fv, hipStream_t stream, uint32_t* queue, uint32_t persistent_blocks) {
    if (fv.y1 <= fv.y0 || fv.W == 0u) return hipSuccess;
    dim3 grid = grid_for(fv), block(kBlock);
    uint32_t tiles = ((fv.W + 7u) / 8u) * ((fv.y1 - fv.y0 + 7u) / 8u);
    dim3 pgrid(std::max(1u, std::min(persistent_blocks, (tiles + 3u) / 4u)));
    switch (stage) {
    case 0: hipLaunchKernelGGL(gbuffer_kernel, grid, block, 0, stream, sc, fv); break;
    case 1: hipLaunchKernelGGL(wavefront_kernel<1>, pgrid, block, 0, stream, sc, fv, queue); break;
    case 2: hipLaunchKernelGGL(wavefront_kernel<2>, pgrid, block, 0, stream, sc, fv, queue); break;
    case 3: hipLaunchKernelGGL(post_kernel, grid, block, 0, stream, fv); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace frt
