#!/bin/bash
# rocprofv3 kernel stats of tools/frame_time.py under the environment given (VAR=value ...), tag = $1
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kst_$TAG -- python3 $R/tools/frame_time.py > $R/gpurun_out/kst_$TAG.log 2>&1 || exit 1
find $R/gpurun_out/kst_$TAG -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/kernel_stats_$TAG.csv \;
python3 - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/kernel_stats_$TAG.csv")):
    n=r["Name"].replace("frt::","").split("(")[0].replace("void ","")
    print(f"{n:28s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f} total {float(r['TotalDurationNs'])/1e6:9.1f} ms")
PY
tail -1 $R/gpurun_out/kst_$TAG.log
