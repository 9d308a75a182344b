#!/bin/bash
# A/B frame times of alternative libfrt.so builds on the GPU box: tools/abrun.sh <lib> [<lib> ...]  (two streams, then one stream; two rounds each, interleaved)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rnd in 1 2; do
  for lib in "$@"; do python3 tools/frame_time.py $lib 2>&1 | tail -1; done
done
for lib in "$@"; do FRT_FLAGS=1 python3 tools/frame_time.py $lib 2>&1 | tail -1; done
