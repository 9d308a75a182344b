#!/bin/bash
# A/B of the traced kernels built for 4 / 5 / 6 waves per SIMD (_ab/r4w/libfrt_w5.so, _w6.so: make FLAGS+=-DFRT_WAVES=n) under the three walks
# (FRT_FLAGS: 41 = the 8-wide tree in LDS, 73 = the 8-wide tree from HBM, 9 = the quad tree; two streams), same box, interleaved.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FRT_LIB=${FRT_LIB:-$R/fast-raytracing-wgpu_amd/lib/libfrt_exp.so}      # the walks compared here live in the experiments build
cd $R
for rnd in 1 2; do
  for fl in 41 73 9; do
    for lib in "" _ab/r4w/libfrt_w5.so _ab/r4w/libfrt_w6.so; do FRT_FLAGS=$fl python3 tools/frame_time.py $lib 2>&1 | tail -1; done
  done
done
