#!/bin/bash
# SQ counters of tools/frame_time.py with an alternative library, one stream: tools/pmc_lib.sh <tag> <lib>   (A/B of kernel variants: instruction and
# load counts per launch tell whether a change reached the kernels' dynamic work, whatever the frame time says)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; LIB=$2
export FRT_FLAGS=1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/pmcl_$TAG -- python3 $R/tools/frame_time.py $R/$LIB > $R/gpurun_out/pmcl_$TAG.log 2>&1 || { tail -5 $R/gpurun_out/pmcl_$TAG.log; exit 1; }
cd $R && python3 tools/pmc_summary.py gpurun_out/pmcl_$TAG | cut -c1-420
