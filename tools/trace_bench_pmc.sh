#!/bin/bash
# SQ counters of the traversal microbenchmark (tools/trace_bench.hip), one --pmc pass, kernel trace only. Usage: tools/trace_bench_pmc.sh [scene] [variant]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_tb
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD \
    --kernel-trace --output-format csv -d $R/gpurun_out/pmc_tb -- $R/tools/_build/trace_bench "$@" > $R/gpurun_out/pmc_tb.log 2>&1 || { tail -5 $R/gpurun_out/pmc_tb.log; exit 1; }
cd $R && python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_tb/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
import sys
rays = 2 * (1 << 23)
print(f"{'kernel':28s} {'VALU/ray-lane':>13s} {'wave-instr':>11s} {'lane_util':>9s} {'SALU/VALU':>9s} {'VMEM_RD':>10s} {'wave_cyc/VALU':>13s} {'busy_cyc':>10s}")
for k, v in sorted(agg.items()):
    m = {c: sum(x) / len(x) for c, x in v.items()}
    name = k.split("(")[0].replace("void ", "")
    lu = m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_ACTIVE_INST_VALU"]) if m.get("SQ_ACTIVE_INST_VALU") else 0
    print(f"{name:28s} {m['SQ_INSTS_VALU'] * 64 / rays:13.1f} {m['SQ_INSTS_VALU']:11.0f} {lu:9.3f} {m['SQ_INSTS_SALU'] / m['SQ_INSTS_VALU']:9.3f} {m['SQ_INSTS_VMEM_RD']:10.0f} "
          f"{4 * m['SQ_WAVE_CYCLES'] / m['SQ_INSTS_VALU']:13.1f} {m['SQ_BUSY_CYCLES']:10.0f}")
PY
