"""Per-kernel SQ counters of tools/valu_calib (rocprofv3 --pmc): counter value per VALU wave-instruction, so that the counters of
the renderer's kernels can be read in units of instructions and SIMD cycles."""
import csv, collections, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{sys.argv[1]}/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    if "fma_kernel" not in k:
        continue
    m = {c: x[-1] for c, x in v.items()}          # the timed launch (second of two)
    n = m.get("SQ_INSTS_VALU", 0.0)
    name = k.split("(")[0].replace("void ", "")
    print(f"{name:28s} waves={m.get('SQ_WAVES', 0):.0f} INSTS_VALU={n:.4g} ACTIVE_INST_VALU/inst={m.get('SQ_ACTIVE_INST_VALU', 0) / n:.3f} "
          f"WAVE_CYCLES/inst={m.get('SQ_WAVE_CYCLES', 0) / n:.3f} BUSY_CYCLES={m.get('SQ_BUSY_CYCLES', 0):.4g} "
          f"THREAD_CYCLES_VALU/(64*ACTIVE)={m.get('SQ_THREAD_CYCLES_VALU', 0) / max(64 * m.get('SQ_ACTIVE_INST_VALU', 1), 1):.3f} "
          f"WAIT_ANY/WAVE_CYCLES={m.get('SQ_WAIT_ANY', 0) / max(m.get('SQ_WAVE_CYCLES', 1), 1):.3f} "
          f"WAIT_INST_ANY/WAVE_CYCLES={m.get('SQ_WAIT_INST_ANY', 0) / max(m.get('SQ_WAVE_CYCLES', 1), 1):.3f}")
