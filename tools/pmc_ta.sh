#!/bin/bash
# Texture-addresser / vector-L1 counters of the one-stream frame (is the per-CU load path the bound of the traced kernels?).
# LIB=<libfrt.so> selects an A/B build. Two --pmc passes over tools/frame_time.py with FRT_FLAGS=1 (one stream: a kernel's counters are its own). Extra VAR=value args are exported.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-ta}; shift
export FRT_FLAGS=1
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
run() { timeout -k 10 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$1 -- python3 $R/tools/frame_time.py $LIB > $R/gpurun_out/pmc_${TAG}_$1.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${TAG}_$1.log; return 1; }; }
run a "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum" &&
run b "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" || exit 1
# (the TA_*_STALLED_* and TCP_*_STALL_CYCLES groups are refused by the profiler on this box: "exceeds the capabilities of the hardware")
cd $R && python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_a gpurun_out/pmc_${TAG}_b > gpurun_out/pmc_${TAG}.txt && cut -c1-900 gpurun_out/pmc_${TAG}.txt
