#!/bin/bash
# (the knobs below are read by the EXPERIMENTS build only: make -C fast-raytracing-wgpu_amd experiments)
export FRT_LIB=${FRT_LIB:-$(pwd)/fast-raytracing-wgpu_amd/lib/libfrt_exp.so}
# experiment: cut depths of the traced stages (FRT_CUTS) under the current schedule, 1080p frame
for c in 0 2 3 4 "2,4" "3,5" "2,3" "2,3,4" "3,4,5" "1,2,3"; do
  FRT_CUTS=$c python3 tools/frame_time.py 2>&1 | tail -1
done
