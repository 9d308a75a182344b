// frt_kernels.hpp — host-callable launch entry points of frt_kernels.hip.
#pragma once
#include "frt_path.hpp"

namespace frt {
// stage: 0 G-buffer, 1 temporal, 2 spatial + shade, 3 post. Rows [fv.y0, fv.y1). Asynchronous on `stream`.
// `queue`: a zeroed device word for the persistent stages (1, 2); `persistent_blocks`: resident workgroups to launch for them.
uint32_t persistent_blocks_per_cu(int stage);
hipError_t launch_stage(int stage, const SceneView& sc, const FrameView& fv, hipStream_t stream, uint32_t* queue, uint32_t persistent_blocks);
}
