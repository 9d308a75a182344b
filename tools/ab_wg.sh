#!/bin/bash
# A/B: plain walks (FRT_FLAGS 9 = timing | pipeline) vs collective walks (137 = + FLAG_WG_TRACE 128), two streams; then one stream (1 vs 129). Same box, interleaved.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FRT_LIB=${FRT_LIB:-$R/fast-raytracing-wgpu_amd/lib/libfrt_exp.so}      # the walks compared here live in the experiments build
cd $R
for rnd in 1 2; do for fl in 9 137; do FRT_FLAGS=$fl python3 tools/frame_time.py 2>&1 | tail -1; done; done
for fl in 1 129; do FRT_FLAGS=$fl python3 tools/frame_time.py 2>&1 | tail -1; done
