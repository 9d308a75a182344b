#!/bin/bash
# (the knobs below are read by the EXPERIMENTS build only: make -C fast-raytracing-wgpu_amd experiments)
export FRT_LIB=${FRT_LIB:-$(pwd)/fast-raytracing-wgpu_amd/lib/libfrt_exp.so}
# experiment: resident kernels (BVH in LDS) vs plain, one stream (FRT_FLAGS=1) and two streams (9)
for fl in 1 9; do
  echo "== flags $fl"
  FRT_FLAGS=$fl FRT_RESIDENT=0 python3 tools/frame_time.py 2>&1 | tail -1
  for b in 1 2 4; do FRT_FLAGS=$fl FRT_RES_BATCH=$b python3 tools/frame_time.py 2>&1 | tail -1; done
  FRT_FLAGS=$fl FRT_RES_BATCH=1 FRT_RES_TRIS=0 python3 tools/frame_time.py 2>&1 | tail -1
done
