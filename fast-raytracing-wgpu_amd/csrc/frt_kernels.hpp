// frt_kernels.hpp — host-callable launch entry points of frt_kernels.hip.
#pragma once
#include "frt_mono.hpp"

namespace frt {
// How to launch a traced stage (1 = T-trace, 2 = spatial + shade): pixel kernel cut at cuts[0] + one continuation launch per further
// segment. The two word buffers are used alternately by the segments (the second one only exists when there are two cuts or more);
// every segment has its OWN counter (counts[0 .. ncuts], zero before the stage runs), so a buffer that is written again two launches
// later starts from slot 0. `zero_counts`: the counter set of this stage's NEXT launch, cleared in passing by the pixel kernel.
static constexpr int kMaxCuts = 4;
enum { kWalkQuad = 0, kWalkWide = 2, kWalkWideLds = 3, kWalkQuadWg = 4 };      // (1 = the quad walk with the voting loop: TraceLaunch::vote; kWalkQuadWg: the quad walk, a workgroup's rays re-dealt to dense direction-sorted waves: frt_kernels.hip: wg_trace)
#ifndef FRT_EXPERIMENTS
#define FRT_EXPERIMENTS 0
#endif
struct TraceLaunch {
    uint32_t ncuts; uint32_t cuts[kMaxCuts]; uint32_t* qwords[2]; uint32_t* counts;
    uint32_t capacity, capacity_odd, grid_min_slots;   // capacity_odd: slots of qwords[1] (the odd segments: far fewer paths get that far); grid_min_slots: the continuation grids cover at least this many slots
    uint32_t* overflow;
    uint32_t* zero_counts;
    bool wg_park;   // pixel kernel: one queue reservation per workgroup instead of one per wave (the product: always)
    bool vote;      // the kernels whose BVH walk votes for its next step (frt_trace.hpp: trace4<ANY, VOTE>): scenes with a deep tree
    uint32_t walk;  // which tree the traced kernels walk: kWalkQuad (trace4; `vote` picks its loop), kWalkWide (trace8, nodes read from HBM), kWalkWideLds (trace8, the whole 8-wide tree copied into every workgroup's LDS: wide_lds_bytes of dynamic LDS)
    uint32_t wide_lds_bytes;
    uint32_t wg_rows;   // kWalkQuadWg: stack rows of a workgroup's dynamic LDS (the quad tree's stack need + the shared row)
#if FRT_EXPERIMENTS
    // lib/libfrt_exp.so only (csrc/experiments/frt_experiment_kernels.hpp): the measured-and-not-kept kernel designs
    uint32_t* tile_state;   // the stage's sweep-direction state (TileOrder) or null = tile rows top to bottom
    bool wavefront; uint32_t* wf_words[2]; uint32_t* wf_items[2]; uint32_t* wf_hits;   // ray-level wavefront (wf_*_kernel); counts = its 96-word counter block
    bool stream; uint32_t shade_min, slice;   // single cut: stream_kernel (resumable traversal + lane refill); shade when >= shade_min lanes wait
    bool refill; uint32_t refill_min;   // single cut: bounce_kernel (lane refill) instead of the continuation launches; refill when >= refill_min lanes are free
    bool resident; uint32_t res_nodes; bool res_tris; uint32_t num_cus; uint32_t res_batch;   // resident_*_kernel: BVH cached in LDS, persistent workgroups; res_batch: 0 = chosen from the tile count
    uint32_t* work;   // 2 x (1 + kMaxCuts) words: {next, ticket} of the pixel launch, then of each continuation launch; zero between launches
#endif
};
// All launches are asynchronous on `stream` and cover rows [fv.y0, fv.y1).
hipError_t launch_gbuffer(const SceneView& sc, const FrameView& fv, hipStream_t stream, uint32_t walk = kWalkQuad);      // walk: kWalkQuad or kWalkWide (primary rays are coherent: their nodes stay in HBM / L1)
hipError_t launch_trace_pixels(int stage, const SceneView& sc, const FrameView& fv, hipStream_t stream, const TraceLaunch& L);
bool trace_has_continuations(const TraceLaunch& L, uint32_t max_depth);
hipError_t launch_trace_continuations(int stage, const SceneView& sc, const FrameView& fv, hipStream_t stream, const TraceLaunch& L);
// LDS plan of the resident kernels for this scene: pair nodes cached (0 = the scene does not qualify), all triangle slots cached?
void resident_plan(const SceneView& sc, uint32_t& nodes, bool& tris);
// T-merge; `pending` (may be null): four ray counters {G closest, G any, T-trace closest, T-trace any} of a G-buffer + T-trace pair that
// ran ahead of its frame, added to `committed` and cleared.
hipError_t launch_merge(const SceneView& sc, const FrameView& fv, hipStream_t stream, unsigned long long* pending, unsigned long long* committed);
hipError_t launch_post(const FrameView& fv, hipStream_t stream);
// Opt-in workgroup-compacting kernels (FRT_FLAG_COMPACTION): stage 1 = the FUSED temporal stage (trace + merge), stage 2 = spatial.
hipError_t launch_compact(int stage, const SceneView& sc, const FrameView& fv, hipStream_t stream);
}
