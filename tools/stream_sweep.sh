#!/bin/bash
# (the knobs below are read by the EXPERIMENTS build only: make -C fast-raytracing-wgpu_amd experiments)
export FRT_LIB=${FRT_LIB:-$(pwd)/fast-raytracing-wgpu_amd/lib/libfrt_exp.so}
# experiment: stream kernel (resumable traversal + lane refill, FRT_STREAM = shade_min) at several cut depths
export FRT_RESIDENT=0
FRT_STREAM=1 FRT_CUTS=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kernels_match or golden or moving_camera_on_gpu" 2>&1 | tail -3
FRT_STREAM=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_cut" 2>&1 | tail -3
for fl in 1 9; do
  echo "== flags $fl"
  FRT_FLAGS=$fl python3 tools/frame_time.py 2>&1 | tail -1
  for c in 1 2 3; do
    for sm in 16 32 48; do
      echo -n "cut $c shade_min $sm: "; FRT_FLAGS=$fl FRT_CUTS=$c FRT_STREAM=$sm python3 tools/frame_time.py 2>&1 | tail -1
    done
  done
done
