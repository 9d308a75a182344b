#!/bin/bash
# experiment: priority of the second stream (FRT_AHEAD_PRIO) vs frame time and strip time
for p in normal low high; do
  echo "== FRT_AHEAD_PRIO=$p"
  FRT_AHEAD_PRIO=$p python3 tools/frame_time.py 2>&1 | tail -1
  FRT_AHEAD_PRIO=$p python3 tools/strip_time.py 2>&1 | grep "cuts=default" | grep slowest
done
