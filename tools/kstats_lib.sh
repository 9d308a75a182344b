#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of tools/frame_time.py with an alternative library, one stream: tools/kstats_lib.sh <tag> <lib>
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; LIB=$2
export FRT_FLAGS=1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kst_$TAG -- python3 $R/tools/frame_time.py $R/$LIB > $R/gpurun_out/kst_$TAG.log 2>&1 || { tail -5 $R/gpurun_out/kst_$TAG.log; exit 1; }
find $R/gpurun_out/kst_$TAG -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/kernel_stats_$TAG.csv \;
cut -d, -f1-4 $R/gpurun_out/kernel_stats_$TAG.csv | sed 's/frt:://g' | cut -c1-110 | head -9
