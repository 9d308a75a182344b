#!/bin/bash
# experiment: cut depths of the traced stages (FRT_CUTS) under the current schedule, 1080p frame
for c in 0 2 3 4 "2,4" "3,5" "2,3" "2,3,4" "3,4,5" "1,2,3"; do
  FRT_CUTS=$c python3 tools/frame_time.py 2>&1 | tail -1
done
