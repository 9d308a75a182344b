#!/bin/bash
# (the knobs below are read by the EXPERIMENTS build only: make -C fast-raytracing-wgpu_amd experiments)
export FRT_LIB=${FRT_LIB:-$(pwd)/fast-raytracing-wgpu_amd/lib/libfrt_exp.so}
# experiment: stream kernel policy (shade_min x slice x refill_min), one stream, cut 1; T and S stage ms are columns 2 and 3 of "stages"
export FRT_RESIDENT=0 FRT_FLAGS=1 FRT_CUTS=1
for sm in 2 16 32 48 64; do for sl in 2 8 32 1000; do for rf in 8 32; do
  echo -n "shade_min $sm slice $sl refill $rf: "; FRT_STREAM=$sm FRT_STREAM_SLICE=$sl FRT_REFILL=$rf python3 tools/frame_time.py 2>&1 | tail -1 | sed 's/default flags=1 FRT_CUTS=1//'
done; done; done
