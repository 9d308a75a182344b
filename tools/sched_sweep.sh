#!/bin/bash
# experiment: when the next frame's G-buffer + T-trace start (FRT_AHEAD_AFTER = tm | spix) vs frame time and strip time
for p in tm spix; do
  echo "== FRT_AHEAD_AFTER=$p"
  FRT_AHEAD_AFTER=$p python3 tools/frame_time.py 2>&1 | tail -1
  FRT_AHEAD_AFTER=$p python3 tools/strip_time.py 2>&1 | grep "cuts=default"
done
