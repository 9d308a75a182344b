#!/bin/bash
# Collects the evidence bench.py's roofline block cites, on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench command      -> gpurun_out/prof_<tag>/…kernel_stats.csv
#   2. separate --pmc passes (never combined with other trace domains)     -> gpurun_out/pmc_<tag>_*/…counter_collection.csv
#   3. the bench line itself, with the CPU baseline                        -> gpurun_out/bench_<tag>.json
# then prints tools/pmc_summary.py over the PMC passes. Afterwards, in the repository: python tools/pmc_to_json.py <tag> > profiles/<tag>_pmc.json
# (bench.py's roofline block reads it) and copy what should be judged into profiles/.
set -o pipefail
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --cpu-frames 0 --no-4k"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- $B > $R/gpurun_out/prof_$TAG.log 2>&1 || exit 1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- $B --steps 16 --warmup 4 > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1 || exit 1
    i=$((i+1))
done
cd $R
bash tools/calib.sh > gpurun_out/calib_$TAG.log 2>&1 || exit 1
bash tools/pmc_ta.sh ${TAG}ta > gpurun_out/pmc_ta_$TAG.log 2>&1 || { tail -3 gpurun_out/pmc_ta_$TAG.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_0 gpurun_out/pmc_${TAG}_1 gpurun_out/pmc_${TAG}_2 gpurun_out/pmc_${TAG}_3 > gpurun_out/pmc_$TAG.txt
cat gpurun_out/pmc_$TAG.txt
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats_$TAG.csv \;
timeout -k 10 400 python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
cut -c1-400 gpurun_out/bench_$TAG.json
