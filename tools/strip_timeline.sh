#!/bin/bash
# Kernel timeline of one thin strip (what bounds strong scaling): tools/strip_timeline.sh <tag> [rank world [4k]]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$TAG -- python3 $R/tools/strip_one.py "$@" > $R/gpurun_out/tl_$TAG.log 2>&1 || { tail -5 $R/gpurun_out/tl_$TAG.log; exit 1; }
cd $R && tail -1 gpurun_out/tl_$TAG.log && python3 tools/timeline.py gpurun_out/tl_$TAG 2
