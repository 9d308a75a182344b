#!/bin/bash
# A/B of the three walks on the GPU box (same box, interleaved; FRT_FLAGS = timing 1 | pipeline 8 | FLAG_WALK_WIDE 32 | FLAG_WALK_WIDE_HBM 64): the 8-wide
# tree in LDS, the 8-wide tree from HBM, the quad tree (the default); last block: one stream.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FRT_LIB=${FRT_LIB:-$R/fast-raytracing-wgpu_amd/lib/libfrt_exp.so}      # the walks compared here live in the experiments build
cd $R
for rnd in 1 2; do
  for fl in 41 73 9; do FRT_FLAGS=$fl python3 tools/frame_time.py 2>&1 | tail -1; done
done
for fl in 33 65 1; do FRT_FLAGS=$fl python3 tools/frame_time.py 2>&1 | tail -1; done
