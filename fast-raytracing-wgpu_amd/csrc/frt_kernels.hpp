// frt_kernels.hpp — host-callable launch entry points of frt_kernels.hip.
#pragma once
#include "frt_mono.hpp"

namespace frt {
// How to launch the traced stages: default = pixel kernel cut at cuts[0] + one continuation launch per further segment. The two
// word buffers are used alternately; every segment has its OWN counter (counts[0 .. ncuts], zero before the stage runs), so a
// buffer that is written again two launches later starts from slot 0. compaction = the opt-in workgroup-compacting kernels.
static constexpr int kMaxCuts = 4;
struct StageLaunch { bool compaction; bool pair_tail; uint32_t ncuts; uint32_t cuts[kMaxCuts]; uint32_t* qwords[2]; uint32_t* counts; uint32_t capacity;
                     uint32_t* row_order[2]; uint32_t* row_cost[2]; uint32_t nrows[2];     // per traced stage; null = tile rows top to bottom
                     uint32_t* zero_in_pixel; uint32_t* zero_in_sort; };   // queue counters of the other stage to clear in passing (or null)
// stage: 0 G-buffer, 1 temporal, 2 spatial + shade, 3 post. Rows [fv.y0, fv.y1). Asynchronous on `stream`.
// Traced stages with a cut: `ev` (optional) is recorded on `stream` right after the pixel kernel; with `tail` set the continuation
// launches go to that stream, ordered behind `ev`. *has_cont tells whether the stage has continuation launches at all.
hipError_t launch_stage(int stage, const SceneView& sc, const FrameView& fv, hipStream_t stream, const StageLaunch& L,
                        hipStream_t tail = nullptr, hipEvent_t ev = nullptr, bool* has_cont = nullptr);
}
