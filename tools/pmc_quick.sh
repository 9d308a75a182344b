#!/bin/bash
# quick PMC pass over tools/frame_time.py under the environment given on the command line (VAR=value ...), tag = $1
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/pmcq_$TAG -- python3 $R/tools/frame_time.py > $R/gpurun_out/pmcq_$TAG.log 2>&1 || exit 1
cd $R && python3 tools/pmc_summary.py gpurun_out/pmcq_$TAG | cut -c1-420
