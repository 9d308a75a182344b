#!/bin/bash
# (the knobs below are read by the EXPERIMENTS build only: make -C fast-raytracing-wgpu_amd experiments)
export FRT_LIB=${FRT_LIB:-$(pwd)/fast-raytracing-wgpu_amd/lib/libfrt_exp.so}
# experiment: bounce kernel with lane refill (FRT_REFILL) at several cut depths, vs the continuation launches
export FRT_RESIDENT=0
FRT_REFILL=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_cut or kernels_match" 2>&1 | tail -3
for fl in 9 1; do
  echo "== flags $fl"
  FRT_FLAGS=$fl python3 tools/frame_time.py 2>&1 | tail -1
  for c in 1 2 3; do
    for rf in 1 16 32 48; do
      echo -n "cut $c refill_min $rf: "; FRT_FLAGS=$fl FRT_CUTS=$c FRT_REFILL=$rf python3 tools/frame_time.py 2>&1 | tail -1
    done
  done
done
