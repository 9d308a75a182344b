// frt_kernels.hpp — host-callable launch entry points of frt_kernels.hip.
#pragma once
#include "frt_mono.hpp"

namespace frt {
// How to launch the traced stages: default = pixel kernel cut at cuts[0] + continuation launches over the queues (two ContQueue
// per stage, zero counts before the stage runs); compaction = the opt-in workgroup-compacting kernels.
struct StageLaunch { bool compaction; uint32_t ncuts; uint32_t cuts[4]; ContQueue queues[2]; };
// stage: 0 G-buffer, 1 temporal, 2 spatial + shade, 3 post. Rows [fv.y0, fv.y1). Asynchronous on `stream`.
hipError_t launch_stage(int stage, const SceneView& sc, const FrameView& fv, hipStream_t stream, const StageLaunch& L);
}
