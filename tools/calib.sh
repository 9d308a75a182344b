#!/bin/bash
# VALU issue calibration on the GPU box: plain run, then the same binary under the SQ counters (one --pmc pass, kernel trace only).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
$R/tools/_build/valu_calib > $R/gpurun_out/valu_calib.txt 2>&1 || exit 1
cat $R/gpurun_out/valu_calib.txt
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --kernel-trace --output-format csv -d $R/gpurun_out/pmc_calib -- $R/tools/_build/valu_calib > $R/gpurun_out/pmc_calib.log 2>&1 || exit 1
cd $R && python3 tools/valu_calib_summary.py gpurun_out/pmc_calib | tee gpurun_out/valu_calib_pmc.txt
