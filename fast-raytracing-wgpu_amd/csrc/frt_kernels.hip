// frt_kernels.hip — gfx950 kernels of the hot path (replaces src/passes/*.rs dispatches + src/shaders/*.wgsl).
//
// Launch shape: one wave64 owns an 8x8 pixel tile (the reference's @workgroup_size(8,8), gbuffer.rs:302), four waves
// per 256-thread workgroup = a 16x16 block. Each lane's BVH traversal stack is a column of an LDS array
// (kStackDepth x 256 words = 32 KiB per workgroup; lane-consecutive addresses -> conflict-free ds_read/ds_write_b32).
// Scene data (pair nodes 64 B, triangle slots 48 B) is read with 16-byte loads; per-pixel buffers are pixel-linear
// float4 / 8-byte / 4-byte streams, so every wave-level access is a set of full 128-byte row segments.
#include "frt_shade.hpp"
#include "frt_kernels.hpp"

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

// STAGE 0 = G-buffer, 1 = ReSTIR temporal (trace_path variant 0), 2 = ReSTIR spatial + final shade (variant 1)
template <int STAGE>
__global__ void __launch_bounds__(kBlock) trace_stage_kernel(SceneView sc, FrameView fv) {
    __shared__ uint32_t s_stack[kStackDepth * kBlock];
    __shared__ uint32_t s_cnt[2];
    if (threadIdx.x < 2u) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t px, py;
    bool active = tile_pixel(fv, px, py);
    PathCtx c(sc, fv, &s_stack[threadIdx.x], (uint32_t)kBlock);
    if (active) {
        if (STAGE == 0) gbuffer_pixel(c, px, py);
        else if (STAGE == 1) temporal_pixel(c, px, py);
        else spatial_pixel(c, px, py);
    }
    bool counted = active && py >= fv.own_y0 && py < fv.own_y1;
    flush_ray_counters(fv, counted ? c.n_closest : 0u, counted ? c.n_any : 0u, s_cnt);
}

__global__ void __launch_bounds__(kBlock) post_kernel(FrameView fv) {
    uint32_t px, py;
    if (tile_pixel(fv, px, py)) post_pixel(fv, px, py);
}

static dim3 grid_for(const FrameView& fv) { return dim3((fv.W + 15u) / 16u, (fv.y1 - fv.y0 + 15u) / 16u, 1u); }

hipError_t launch_stage(int stage, const SceneView& sc, const FrameView& fv, hipStream_t stream) {
    if (fv.y1 <= fv.y0 || fv.W == 0u) return hipSuccess;
    dim3 grid = grid_for(fv), block(kBlock);
    switch (stage) {
    case 0: hipLaunchKernelGGL(trace_stage_kernel<0>, grid, block, 0, stream, sc, fv); break;
    case 1: hipLaunchKernelGGL(trace_stage_kernel<1>, grid, block, 0, stream, sc, fv); break;
    case 2: hipLaunchKernelGGL(trace_stage_kernel<2>, grid, block, 0, stream, sc, fv); break;
    case 3: hipLaunchKernelGGL(post_kernel, grid, block, 0, stream, fv); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace frt
