#!/bin/bash
# (the knobs below are read by the EXPERIMENTS build only: make -C fast-raytracing-wgpu_amd experiments)
export FRT_LIB=${FRT_LIB:-$(pwd)/fast-raytracing-wgpu_amd/lib/libfrt_exp.so}
# experiment: ray-level wavefront (FRT_WAVEFRONT=1): parity, then timing on one and two streams
export FRT_RESIDENT=0
FRT_WAVEFRONT=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kernels_match or golden or moving_camera_on_gpu or full_size" 2>&1 | tail -4
for fl in 1 9; do
  echo "== flags $fl"
  FRT_FLAGS=$fl python3 tools/frame_time.py 2>&1 | tail -1
  for rf in 8 16 32; do for sl in 8 32; do
    echo -n "refill $rf slice $sl: "; FRT_FLAGS=$fl FRT_WAVEFRONT=1 FRT_REFILL=$rf FRT_STREAM_SLICE=$sl python3 tools/frame_time.py 2>&1 | tail -1
  done; done
done
