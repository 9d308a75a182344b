// frt_kernels.hpp — host-callable launch entry points of frt_kernels.hip.
#pragma once
#include "frt_mono.hpp"

namespace frt {
// stage: 0 G-buffer, 1 temporal, 2 spatial + shade, 3 post. Rows [fv.y0, fv.y1). Asynchronous on `stream`.
hipError_t launch_stage(int stage, const SceneView& sc, const FrameView& fv, hipStream_t stream, bool compaction);
}
