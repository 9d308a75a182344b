#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_icache -- python3 $R/tools/frame_time.py > $R/gpurun_out/pmc_icache.log 2>&1 || { tail -5 $R/gpurun_out/pmc_icache.log; exit 1; }
cd $R && python3 tools/pmc_summary.py gpurun_out/pmc_icache | cut -c1-300
