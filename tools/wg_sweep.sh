#!/bin/bash
# (the knobs below are read by the EXPERIMENTS build only: make -C fast-raytracing-wgpu_amd experiments)
export FRT_LIB=${FRT_LIB:-$(pwd)/fast-raytracing-wgpu_amd/lib/libfrt_exp.so}
for fl in 1 9; do
  echo "== flags $fl"
  FRT_FLAGS=$fl FRT_RESIDENT=0 python3 tools/frame_time.py 2>&1 | tail -1
  FRT_FLAGS=$fl FRT_RESIDENT=0 FRT_WG64=1 python3 tools/frame_time.py 2>&1 | tail -1
  FRT_FLAGS=$fl FRT_RES_BATCH=1 python3 tools/frame_time.py 2>&1 | tail -1
done
FRT_RESIDENT=0 FRT_WG64=1 python3 tools/strip_time.py 2>&1 | grep "cuts=default" | grep slowest
