// frt_kernels.hip — gfx950 kernels of the hot path (replaces src/passes/*.rs dispatches + src/shaders/*.wgsl).
//
// Launch shape: one wave64 owns an 8x8 pixel tile (the reference's @workgroup_size(8,8), gbuffer.rs:302), four waves
// per 256-thread workgroup = a 16x16 block. Each lane's BVH traversal stack is a column of an LDS array
// (kStackDepth x 256 words = 32 KiB per workgroup; lane-consecutive addresses -> conflict-free ds_read/ds_write_b32).
// Scene data (quad nodes 128 B, triangle slots 48 B: frt_trace.hpp) is read with 16-byte loads; per-pixel buffers are pixel-linear
// float4 / 8-byte / 4-byte streams, so every wave-level access is a set of full 128-byte row segments.
#include "frt_mono.hpp"
#include "frt_kernels.hpp"
#include <algorithm>

namespace frt {

static constexpr int kBlock = 256;
// LDS of a traced workgroup = the stack array and nothing else: exactly 32 KiB, five workgroups per CU by LDS. The quad tree of any scene the
// builder accepts needs at most kStackDepth - 1 entries per lane (frt_bvh.cpp: build_quad_nodes folds only within that budget; a plain binary
// subtree is at most kMaxBvhDepth = 30 deep), so the LAST row of the array is never a stack entry: the workgroup's few shared words (ray-count
// partial sums, the queue reservation scratch) live there.
static constexpr int kMiscRow = kStackDepth - 1;
// Kernels that walk the 8-wide tree (WALK >= kWalkWide; frt_trace.hpp: trace8) need one stack word per level: kStack8 rows + one row for the shared
// words = 9 KiB, and — kWalkWideLds — the whole tree behind them in dynamic LDS (128 bytes per node: 27 KiB for the Cornell Box's 216 nodes).
template <int WALK> struct WalkLds { static constexpr int kRows = kStackDepth, kMisc = kMiscRow; };
template <> struct WalkLds<kWalkWide> { static constexpr int kRows = kStack8 + 1, kMisc = kStack8; };
template <> struct WalkLds<kWalkWideLds> { static constexpr int kRows = kStack8 + 1, kMisc = kStack8; };
extern __shared__ uint4 s_wide_nodes[];      // dynamic LDS of the kWalkWideLds kernels
#ifndef FRT_WAVES
#define FRT_WAVES 4      // waves per SIMD the traced kernels are built for (A/B builds: 5 needs <= 96 VGPRs and <= 32 KiB of LDS per workgroup)
#endif

__device__ __forceinline__ bool tile_pixel_at(const FrameView& fv, uint32_t tx, uint32_t ty, uint32_t& px, uint32_t& py) {
    uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    px = tx * 16u + (wave & 1u) * 8u + (lane & 7u);
    py = fv.y0 + ty * 16u + (wave >> 1) * 8u + (lane >> 3);
    return px < fv.W && py < fv.y1;
}
__device__ __forceinline__ bool tile_pixel(const FrameView& fv, uint32_t& px, uint32_t& py) { return tile_pixel_at(fv, blockIdx.x, blockIdx.y, px, py); }

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

// The top of the quad tree — nodes 0 .. kLdsTopNodes - 1: the root and, numbered breadth-first, its children — copied into the workgroup's LDS, in the
// row of the stack array that no stack entry reaches (behind the shared words): the first two node steps of every walk read it there (frt_trace.hpp:
// trace4). Call before the workgroup's first barrier; null for a tree too small to have those nodes.
__device__ __forceinline__ const uint32_t* stage_top_nodes(const SceneView& sc, uint32_t* s_cnt) {
    uint32_t* const s_top = s_cnt + 32;      // (128-byte aligned)
    if (sc.num_nodes4 < (uint32_t)kLdsTopNodes) return nullptr;
    if (threadIdx.x < (uint32_t)kLdsTopNodes * 32u) s_top[threadIdx.x] = reinterpret_cast<const uint32_t*>(sc.nodes4)[threadIdx.x];
    return s_top;
}

// The whole 8-wide tree copied into the workgroup's dynamic LDS (kWalkWideLds): every node step of every walk then reads its node at the LDS's
// latency and leaves the L1 to the triangles. Call before the workgroup's first barrier.
__device__ __forceinline__ void stage_wide_nodes(const SceneView& sc) {
    const uint32_t n = sc.num_nodes8 * 8u;      // uint4s
    for (uint32_t i = threadIdx.x; i < n; i += (uint32_t)kBlock) s_wide_nodes[i] = sc.nodes8[i];
}
// VOTE: the traced kernels of scenes with a deep tree walk it with the voting loop (frt_trace.hpp: trace4<ANY, VOTE>; frt_renderer.hip: kVoteMinQuadNodes).
// WALK: 0 the quad walk, 1 the quad walk with the voting loop, kWalkWide / kWalkWideLds the 8-wide walk.
template <int WALK> struct CtxOf { typedef PathCtx type; };
template <> struct CtxOf<1> { typedef VotePathCtx type; };
template <> struct CtxOf<kWalkWide> { typedef Wide8PathCtx type; };
template <> struct CtxOf<kWalkWideLds> { typedef Wide8PathCtx type; };
// Workgroup prologue shared by the traced kernels: clears the shared words, stages the tree's top (quad walk) or the whole tree (kWalkWideLds) in LDS,
// returns the context's tree pointer set up. One barrier inside.
template <int WALK, class Ctx>
__device__ __forceinline__ void walk_prologue(const SceneView& sc, uint32_t* s_cnt, Ctx& c) {
    if (threadIdx.x < 2u) s_cnt[threadIdx.x] = 0u;
    if constexpr (WALK == kWalkWideLds) { stage_wide_nodes(sc); c.nb = reinterpret_cast<const char*>(s_wide_nodes); }
    else if constexpr (WALK < kWalkWide) c.lds_top = stage_top_nodes(sc, s_cnt);
    __syncthreads();
}

// G-buffer: one primary ray per pixel, coherent within the 8x8 tile (lane utilisation 96 %): plain thread-per-pixel launch.
template <int WALK>
__global__ void __launch_bounds__(kBlock) gbuffer_kernel(SceneView sc, FrameView fv) {
    __shared__ uint32_t s_stack[WalkLds<WALK>::kRows * kBlock];
    uint32_t* const s_cnt = &s_stack[WalkLds<WALK>::kMisc * kBlock];
    typename CtxOf<WALK>::type c(sc, fv, &s_stack[threadIdx.x], (uint32_t)kBlock);
    walk_prologue<WALK>(sc, s_cnt, c);
    uint32_t px, py;
    bool active = tile_pixel(fv, px, py);
    if (active) gbuffer_pixel(c, px, py);
    bool counted = active && py >= fv.own_y0 && py < fv.own_y1;
    flush_ray_counters(fv, counted ? c.n_closest : 0u, counted ? c.n_any : 0u, s_cnt);
}

// Default T-trace (STAGE 1) / spatial + shade (STAGE 2) kernels: one thread per pixel runs the primary hit and the bounces
// below `cut` (frt_mono.hpp). 83 % of the rays of a Cornell frame are fired at depth <= 2, but almost every 8x8 tile has a lane
// that survives to depth 5-7, so an uncut wave spends most of its bounce iterations with a handful of live lanes (24 % lane
// utilisation, profiles/r1_v4_pmc.txt). Paths still alive at `cut` are therefore PARKED: a wave ballot finds them, the leader
// reserves popcount(ballot) slots of a continuation queue in HBM with one atomic, every survivor writes its 22-30 words of
// loop state at slot base + rank (SoA, so the wave writes full rows), and the lane retires. continue_kernel then resumes the
// parked paths one per lane — dense waves again — and may park its own survivors for a further launch. Pixels are unchanged:
// a path's arithmetic and rand() sequence do not depend on the lane or launch that runs it.
// The queue is NOT sized for the worst case (every pixel parking): a lane whose slot lies beyond the capacity keeps its path and
// finishes it in place (second trip of the segment loop below), counted in q.overflow so that the host can grow the queue.
__device__ __forceinline__ uint32_t wave_reserve(uint32_t* count, bool want) {
    const unsigned long long m = __ballot(want);
    if (m == 0ull) return 0u;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0u;
    if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = __shfl(base, leader, 64);
    return base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ void note_queue_overflow(const ContQueue& q, bool mine) {
    const unsigned long long m = __ballot(mine);
    if (m != 0ull && q.overflow && (int)(threadIdx.x & 63u) == __ffsll((long long)m) - 1) atomicAdd(q.overflow, (uint32_t)__popcll(m));
}

template <int STAGE>
__device__ __forceinline__ void finish_path(PathCtx& c, uint32_t pix, const ReservoirView& r, const LoopState& s) {
    if (STAGE == 1) c.fv.cand[pix] = temporal_candidate(s.accumulated, s.v1_pos);   // T-trace ends here; merge_kernel does the rest
    else spatial_tail(c, pix, r, s.accumulated, s.v1_pos);
}

// Slots for the lanes of a workgroup that want one, with ONE atomic for the whole workgroup. Every thread of the workgroup must call it
// (three barriers inside). s_tmp: 8 words of LDS.
__device__ __forceinline__ uint32_t workgroup_reserve(uint32_t* counter, bool want, uint32_t* s_tmp) {
    const unsigned long long m = __ballot(want);
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0u) s_tmp[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0u) {
        const uint32_t total = s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
        s_tmp[4] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t base = s_tmp[4];
    for (uint32_t w = 0; w < wave; ++w) base += s_tmp[w];
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    __syncthreads();
    return base + rank;
}

// Runs bounces [d0, d1) of the lane's path and parks a survivor in `q`; when the queue is full the lane goes on to MAX_DEPTH itself.
// On return the path is either parked (true: state stored) or finished (false: s holds the final radiance).
template <int VARIANT, class Ctx>
__device__ __forceinline__ bool run_segment_and_park(Ctx& c, LoopState& s, uint32_t d0, uint32_t d1, const ContQueue& q, uint32_t pix, bool owned,
                                                     const ReservoirView* r) {
    bool run = s.alive, parked = false;
#pragma nounroll
    for (int trip = 0; trip < 2; ++trip) {
        if (run) path_loop<VARIANT>(c, s, d0, d1);
        if (trip == 1) break;
        uint32_t region = 0u, rcap = q.capacity;
        if (q.nsub > 1u) { region = (blockIdx.x + blockIdx.y * gridDim.x) % q.nsub; rcap = q.capacity / q.nsub; }
        const uint32_t slot = wave_reserve(q.count + region, s.alive);
        parked = s.alive && slot < rcap;
        run = s.alive && !parked;
        if (parked) cont_store(q, region * rcap + slot, pix, c.rng, owned, s, r);
        if (__ballot(run) == 0ull) break;      // wave-uniform
        note_queue_overflow(q, run);
        d0 = d1; d1 = c.fv.max_depth;          // no further cut for a path that could not be parked
    }
    return parked;
}

template <int STAGE, int WALK>
__global__ void __launch_bounds__(kBlock, FRT_WAVES) pixel_kernel(SceneView sc, FrameView fv, ContQueue q, uint32_t cut, uint32_t* zero_counts, bool wg_park) {
    constexpr int THREADS = kBlock;
    __shared__ uint32_t s_stack[WalkLds<WALK>::kRows * THREADS];
    uint32_t* const s_cnt = &s_stack[WalkLds<WALK>::kMisc * THREADS];
    uint32_t* const s_tmp = s_cnt + 8;
    typename CtxOf<WALK>::type c(sc, fv, &s_stack[threadIdx.x], (uint32_t)THREADS);
    walk_prologue<WALK>(sc, s_cnt, c);
    constexpr int VARIANT = STAGE == 1 ? 0 : 1;
    // the queue counters this stage's NEXT launch will use (the other set of the pair) are cleared here instead of by a memset
    if (zero_counts && blockIdx.x == 0u && blockIdx.y == 0u && threadIdx.x <= (uint32_t)kMaxCuts) zero_counts[threadIdx.x] = 0u;
    uint32_t px, py;
    const bool active = tile_pixel(fv, px, py);      // tile rows top to bottom (a sweep from the expensive end of the image was measured: 1-2 % slower, profiles/r2_experiments)
    const uint32_t pix = py * fv.W + px;
    const bool counted = active && py >= fv.own_y0 && py < fv.own_y1;
    LoopState s;
    s.alive = false;
    ReservoirView r = zero_reservoir();
    bool traced = false;
    if (active) {
        uint32_t seed = 0u;
        if (STAGE == 1) {
            if (!(fv.gpos[pix].w < 0.0f)) { seed = temporal_seed(fv, pix); traced = true; }   // background: merge_kernel writes the zero reservoir (restir.wgsl:805-811)
        } else if (spatial_neighbors(c, pix, r)) { seed = r.y; traced = true; }
        if (traced) path_head<VARIANT>(c, pix, seed, s);
    }
    // bounces [1, cut), then park the survivors: ONE atomic for the workgroup's four waves (tens of thousands of atomics per launch on one
    // address are served at ~13 ns each: 32,640 wave-level reservations = 0.42 ms of a 0.8 ms kernel's life); a lane that finds the queue
    // full keeps its path and finishes it in place
    uint32_t d0 = 1u, d1 = cut < fv.max_depth ? cut : fv.max_depth;
    bool run = s.alive, parked = false;
#pragma nounroll
    for (int trip = 0; trip < 2; ++trip) {      // (a loop so that path_loop is instantiated once)
        if (run) path_loop<VARIANT>(c, s, d0, d1);
        if (trip == 1) break;
        uint32_t region = 0u, rcap = q.capacity;
        if (q.nsub > 1u) { region = (blockIdx.x + blockIdx.y * gridDim.x) % q.nsub; rcap = q.capacity / q.nsub; }
        const uint32_t slot = wg_park ? workgroup_reserve(q.count + region, s.alive, s_tmp) : wave_reserve(q.count + region, s.alive);
        parked = s.alive && slot < rcap;
        run = s.alive && !parked;
        if (parked) cont_store(q, region * rcap + slot, pix, c.rng, counted, s, STAGE == 2 ? &r : nullptr);
        if (__ballot(run) == 0ull) break;      // wave-uniform; no barrier follows
        note_queue_overflow(q, run);
        d0 = d1; d1 = fv.max_depth;
    }
    if (traced && !parked) finish_path<STAGE>(c, pix, r, s);
    flush_ray_counters(fv, counted ? c.n_closest : 0u, counted ? c.n_any : 0u, s_cnt);   // (contains a barrier: every wave is done)
}

// Resumes parked paths for bounces [d0, d1); survivors are parked again in `qout` (d1 < MAX_DEPTH) or finished here.
template <int STAGE, int WALK>
__global__ void __launch_bounds__(kBlock, FRT_WAVES) continue_kernel(SceneView sc, FrameView fv, ContQueue qin, ContQueue qout, uint32_t d0, uint32_t d1) {
    constexpr int THREADS = kBlock;
    __shared__ uint32_t s_stack[WalkLds<WALK>::kRows * THREADS];
    uint32_t* const s_cnt = &s_stack[WalkLds<WALK>::kMisc * THREADS];
    const uint32_t filled = *qin.count;
    const uint32_t n = filled < qin.capacity ? filled : qin.capacity;
    // The launch that parked these paths ran out of slots (its counter ran past the capacity; the surplus paths were finished in place): tell the
    // host through the mapped flag whose address sits behind the overflow counter (frt_mono.hpp: ContQueue). One thread per launch, here where
    // few registers are live — in the parking code the 64-bit address cost the traced kernels two VGPRs (112 -> 114, +1.5 % per frame).
    if (blockIdx.x == 0u && threadIdx.x == 0u && filled > qin.capacity && qin.overflow) {
        uint32_t* const seen = *reinterpret_cast<uint32_t* const*>(qin.overflow + 2);
        if (seen) *reinterpret_cast<volatile uint32_t*>(seen) = 1u;
    }
    if (blockIdx.x * (uint32_t)THREADS >= n) return;   // uniform per workgroup
    constexpr int VARIANT = STAGE == 1 ? 0 : 1;
    typename CtxOf<WALK>::type c(sc, fv, &s_stack[threadIdx.x], (uint32_t)THREADS);
    walk_prologue<WALK>(sc, s_cnt, c);
    uint32_t cnt_closest = 0u, cnt_any = 0u;
    // stride loop: one trip with the grid of launch_trace_continuations; uniform per workgroup for any grid
    for (uint32_t base = blockIdx.x * (uint32_t)THREADS; base < n; base += gridDim.x * (uint32_t)THREADS) {
        const uint32_t slot_in = base + threadIdx.x;
        LoopState s;
        s.alive = false;
        ReservoirView r = zero_reservoir();
        uint32_t pix = 0u;
        bool owned = false;
        c.n_closest = 0u; c.n_any = 0u;
        const bool have = slot_in < n;
        if (have) cont_load(qin, slot_in, pix, c.rng, owned, s, STAGE == 2 ? &r : nullptr);
        const bool parked = run_segment_and_park<VARIANT>(c, s, d0, d1, qout, pix, owned, STAGE == 2 ? &r : nullptr);
        if (have && !parked) finish_path<STAGE>(c, pix, r, s);
        if (owned) { cnt_closest += c.n_closest; cnt_any += c.n_any; }
    }
    flush_ray_counters(fv, cnt_closest, cnt_any, s_cnt);
}

// T-merge (frt_path.hpp: temporal_merge): RIS of the fresh candidate with the reprojected previous spatial reservoir, one thread
// per pixel over rows [y0, y1), pixel-linear (every access is a full row segment). No rays; the only kernel of the temporal stage
// that sits on the frame-to-frame dependency chain. Its first lanes also COMMIT the ray counts of a G-buffer + T-trace pair that
// ran ahead of its frame (frt_renderer.hip: speculation): `pending` -> `committed`, four words.
__global__ void __launch_bounds__(kBlock) merge_kernel(SceneView sc, FrameView fv, unsigned long long* pending, unsigned long long* committed) {
    if (pending && blockIdx.x == 0u && threadIdx.x < 4u) {
        const unsigned long long v = atomicExch(&pending[threadIdx.x], 0ull);
        if (v) atomicAdd(&committed[threadIdx.x], v);
    }
    const size_t first = (size_t)fv.y0 * fv.W, last = (size_t)fv.y1 * fv.W;
    const size_t pix = first + (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (pix < last) temporal_merge_pixel(sc, fv, (uint32_t)pix);
}

// Post / accumulate: a 5x5 bilateral + 3x3 variance stencil (post.wgsl:95-170). The 16x16 pixel workgroup first stages its
// 20x20 neighbourhood in LDS as DECODED values (f16 radiance -> f32, unorm8 albedo -> f32, octahedral normal -> unit vector,
// position): 12 floats per entry, SoA with a row pitch of 24 words so that the 8x8 tile a wave reads maps to 32 distinct banks.
// Each entry is decoded once per workgroup instead of once per tap (25x fewer normalisations and unorm divisions), and the
// taps become ds_read_b32 instead of four global loads.
static constexpr int kTileW = 20, kTilePitch = 24, kTileH = 20, kTileN = kTilePitch * kTileH;
struct TileTaps {
    const float* t;   // [12][kTileN] in LDS
    int ox, oy;       // image coordinates of tile entry (0, 0)
    __device__ __forceinline__ int at(int nx, int ny) const { return (ny - oy) * kTilePitch + (nx - ox); }
    __device__ __forceinline__ TapData get(int nx, int ny) const {
        const int i = at(nx, ny);
        TapData d;
        d.color = mk3(t[0 * kTileN + i], t[1 * kTileN + i], t[2 * kTileN + i]);
        d.albedo = mk3(t[3 * kTileN + i], t[4 * kTileN + i], t[5 * kTileN + i]);
        d.normal = mk3(t[6 * kTileN + i], t[7 * kTileN + i], t[8 * kTileN + i]);
        d.pos = mk3(t[9 * kTileN + i], t[10 * kTileN + i], t[11 * kTileN + i]);
        return d;
    }
    __device__ __forceinline__ f3 color(int nx, int ny) const {
        const int i = at(nx, ny);
        return mk3(t[0 * kTileN + i], t[1 * kTileN + i], t[2 * kTileN + i]);
    }
};

__global__ void __launch_bounds__(kBlock) post_kernel(FrameView fv) {
    __shared__ float s_tile[12 * kTileN];
    const int ox = (int)(blockIdx.x * 16u) - 2, oy = (int)(fv.y0 + blockIdx.y * 16u) - 2;
    for (int e = (int)threadIdx.x; e < kTileW * kTileH; e += kBlock) {
        const int tx = e % kTileW, ty = e / kTileW;
        const int nx = ox + tx, ny = oy + ty;
        if (nx >= 0 && ny >= 0 && nx < (int)fv.W && ny < (int)fv.H) {
            const TapData d = decode_tap(fv, (uint32_t)ny * fv.W + (uint32_t)nx);
            const int i = ty * kTilePitch + tx;
            s_tile[0 * kTileN + i] = d.color.x; s_tile[1 * kTileN + i] = d.color.y; s_tile[2 * kTileN + i] = d.color.z;
            s_tile[3 * kTileN + i] = d.albedo.x; s_tile[4 * kTileN + i] = d.albedo.y; s_tile[5 * kTileN + i] = d.albedo.z;
            s_tile[6 * kTileN + i] = d.normal.x; s_tile[7 * kTileN + i] = d.normal.y; s_tile[8 * kTileN + i] = d.normal.z;
            s_tile[9 * kTileN + i] = d.pos.x; s_tile[10 * kTileN + i] = d.pos.y; s_tile[11 * kTileN + i] = d.pos.z;
        }
    }
    __syncthreads();
    uint32_t px, py;
    if (tile_pixel(fv, px, py)) {
        TileTaps taps{s_tile, ox, oy};
        post_pixel_t(fv, px, py, taps);
    }
}

// jitter != 0 (PostParams.jitter): bilinear radiance / albedo taps straight from HBM (frt_shade.hpp: JitterTaps). Untuned on purpose.
__global__ void __launch_bounds__(kBlock) post_jitter_kernel(FrameView fv) {
    uint32_t px, py;
    if (tile_pixel(fv, px, py)) { JitterTaps taps{fv}; post_pixel_t(fv, px, py, taps); }
}

#ifndef FRT_EXPERIMENTS
#define FRT_EXPERIMENTS 0
#endif
#if FRT_EXPERIMENTS
#include "experiments/frt_experiment_kernels.hpp"      // lib/libfrt_exp.so only (`make experiments`)
#include "experiments/frt_round4_walks.hpp"            // ... round 4: collective walks (wg_trace)
#endif

static dim3 grid_for(const FrameView& fv) { return dim3((fv.W + 15u) / 16u, (fv.y1 - fv.y0 + 15u) / 16u, 1u); }
static bool empty_rows(const FrameView& fv) { return fv.y1 <= fv.y0 || fv.W == 0u; }
static ContQueue queue_of(const TraceLaunch& L, uint32_t k) {
    ContQueue q; q.words = L.qwords[k & 1u]; q.count = L.counts + k; q.capacity = (k & 1u) ? L.capacity_odd : L.capacity; q.overflow = L.overflow;
#if FRT_EXPERIMENTS
    q.nsub = L.wavefront ? (uint32_t)kWfSub : 1u;
#else
    q.nsub = 1u;
#endif
    return q;
}
static uint32_t first_cut(const TraceLaunch& L, const FrameView& fv) { return L.ncuts ? L.cuts[0] : fv.max_depth; }

hipError_t launch_gbuffer(const SceneView& sc, const FrameView& fv, hipStream_t stream, uint32_t walk) {
    if (empty_rows(fv)) return hipSuccess;
#if FRT_EXPERIMENTS
    if (walk == (uint32_t)kWalkWide || walk == (uint32_t)kWalkWideLds) { hipLaunchKernelGGL(gbuffer_kernel<kWalkWide>, grid_for(fv), dim3(kBlock), 0, stream, sc, fv); return hipGetLastError(); }
#endif
    (void)walk;
    hipLaunchKernelGGL(gbuffer_kernel<kWalkQuad>, grid_for(fv), dim3(kBlock), 0, stream, sc, fv);
    return hipGetLastError();
}
hipError_t launch_post(const FrameView& fv, hipStream_t stream) {
    if (empty_rows(fv)) return hipSuccess;
    if (post_is_jittered(fv)) hipLaunchKernelGGL(post_jitter_kernel, grid_for(fv), dim3(kBlock), 0, stream, fv);
    else hipLaunchKernelGGL(post_kernel, grid_for(fv), dim3(kBlock), 0, stream, fv);
    return hipGetLastError();
}
hipError_t launch_merge(const SceneView& sc, const FrameView& fv, hipStream_t stream, unsigned long long* pending, unsigned long long* committed) {
    if (empty_rows(fv)) return hipSuccess;
    const size_t n = (size_t)(fv.y1 - fv.y0) * fv.W;
    hipLaunchKernelGGL(merge_kernel, dim3((uint32_t)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, sc, fv, pending, committed);
    return hipGetLastError();
}
#if !FRT_EXPERIMENTS
hipError_t launch_compact(int, const SceneView&, const FrameView&, hipStream_t) { return hipErrorNotSupported; }      // experiments build only
void resident_plan(const SceneView&, uint32_t& nodes, bool& tris) { nodes = 0u; tris = false; }
#endif
// Pixel kernel of a traced stage over rows [fv.y0, fv.y1) (fv.y0 on the stage's 16-row tile grid). A stage may be launched in
// several row ranges (a strip's interior before its halo-dependent edge rows): they share the queue; only one of them should
// carry the tile-order state (L.tile_state) and the counter clearing (L.zero_counts).
hipError_t launch_trace_pixels(int stage, const SceneView& sc, const FrameView& fv, hipStream_t stream, const TraceLaunch& L) {
    if (stage != 1 && stage != 2) return hipErrorInvalidValue;
    if (empty_rows(fv)) return hipSuccess;
    const dim3 grid = grid_for(fv);
#if FRT_EXPERIMENTS
    if (L.resident) return exp_launch_resident_pixels(stage, sc, fv, stream, L);
#endif
    auto go = [&](auto kernel, uint32_t lds = 0u) { hipLaunchKernelGGL(kernel, grid, dim3(kBlock), lds, stream, sc, fv, queue_of(L, 0), first_cut(L, fv), L.zero_counts, L.wg_park); };
#if FRT_EXPERIMENTS
    if (L.walk == (uint32_t)kWalkQuadWg) {      // collective walks: dynamic LDS = stack rows + exchange rows
        const uint32_t lds = (L.wg_rows + (uint32_t)kXRows) * (uint32_t)kBlock * 4u;
        auto gw = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, dim3(kBlock), lds, stream, sc, fv, queue_of(L, 0), first_cut(L, fv), L.zero_counts, L.wg_rows); };
        if (stage == 1) { if (L.vote) gw(pixel_kernel_wg<1, true>); else gw(pixel_kernel_wg<1, false>); }
        else { if (L.vote) gw(pixel_kernel_wg<2, true>); else gw(pixel_kernel_wg<2, false>); }
        return hipGetLastError();
    }
    if (L.walk == (uint32_t)kWalkWideLds) { if (stage == 1) go(pixel_kernel<1, kWalkWideLds>, L.wide_lds_bytes); else go(pixel_kernel<2, kWalkWideLds>, L.wide_lds_bytes); return hipGetLastError(); }
    if (L.walk == (uint32_t)kWalkWide) { if (stage == 1) go(pixel_kernel<1, kWalkWide>); else go(pixel_kernel<2, kWalkWide>); return hipGetLastError(); }
#endif
    if (stage == 1) { if (L.vote) go(pixel_kernel<1, 1>); else go(pixel_kernel<1, 0>); }
    else { if (L.vote) go(pixel_kernel<2, 1>); else go(pixel_kernel<2, 0>); }
    return hipGetLastError();
}
bool trace_has_continuations(const TraceLaunch& L, uint32_t max_depth) { return L.ncuts > 0 && L.cuts[0] < max_depth; }
// One continuation launch per further path segment, each over min(queue length, capacity) parked paths (the grid covers the capacity;
// workgroups beyond the queue's length leave at once). Resident form: persistent workgroups take 64-path chunks from a counter.
hipError_t launch_trace_continuations(int stage, const SceneView& sc, const FrameView& fv, hipStream_t stream, const TraceLaunch& L) {
    if (stage != 1 && stage != 2) return hipErrorInvalidValue;
#if FRT_EXPERIMENTS
    { hipError_t e_; if (exp_launch_continuations(stage, sc, fv, stream, L, e_)) return e_; }      // wavefront / stream / refill / resident forms
#endif
    for (uint32_t k = 0; k < L.ncuts && L.cuts[k] < fv.max_depth; ++k) {
        const uint32_t gslots = std::max(queue_of(L, k).capacity, L.grid_min_slots);   // (workgroups beyond the queue's fill retire at once)
        const dim3 cgrid((gslots + (uint32_t)kBlock - 1u) / (uint32_t)kBlock);
        const uint32_t d0 = L.cuts[k], d1 = (k + 1 < L.ncuts && L.cuts[k + 1] < fv.max_depth) ? L.cuts[k + 1] : fv.max_depth;
        auto go = [&](auto kernel, uint32_t lds = 0u) { hipLaunchKernelGGL(kernel, cgrid, dim3(kBlock), lds, stream, sc, fv, queue_of(L, k), queue_of(L, k + 1), d0, d1); };
#if FRT_EXPERIMENTS
        if (L.walk == (uint32_t)kWalkQuadWg) {
            const uint32_t lds = (L.wg_rows + (uint32_t)kXRows) * (uint32_t)kBlock * 4u;
            auto gw = [&](auto kernel) { hipLaunchKernelGGL(kernel, cgrid, dim3(kBlock), lds, stream, sc, fv, queue_of(L, k), queue_of(L, k + 1), d0, d1, L.wg_rows); };
            if (stage == 1) { if (L.vote) gw(continue_kernel_wg<1, true>); else gw(continue_kernel_wg<1, false>); }
            else { if (L.vote) gw(continue_kernel_wg<2, true>); else gw(continue_kernel_wg<2, false>); }
            continue;
        }
        if (L.walk == (uint32_t)kWalkWideLds) { if (stage == 1) go(continue_kernel<1, kWalkWideLds>, L.wide_lds_bytes); else go(continue_kernel<2, kWalkWideLds>, L.wide_lds_bytes); continue; }
        if (L.walk == (uint32_t)kWalkWide) { if (stage == 1) go(continue_kernel<1, kWalkWide>); else go(continue_kernel<2, kWalkWide>); continue; }
#endif
        if (stage == 1) { if (L.vote) go(continue_kernel<1, 1>); else go(continue_kernel<1, 0>); }
        else { if (L.vote) go(continue_kernel<2, 1>); else go(continue_kernel<2, 0>); }
    }
    return hipGetLastError();
}
} // namespace frt
