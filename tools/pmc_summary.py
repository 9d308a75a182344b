"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel means per launch."""
import csv, collections, glob, sys
tab = collections.defaultdict(dict)
for d in sys.argv[1:]:
    for f in glob.glob(f'{d}/**/*_counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in agg.items():
            if 'frt' in k:
                for c, x in v.items(): tab[k][c] = sum(x) / len(x)
for k, v in tab.items():
    name = k.replace('frt::', '').split('(')[0].replace('void ', '')
    line = f"{name:28s} " + " ".join(f"{c}={x:.4g}" for c, x in sorted(v.items()))
    if 'SQ_THREAD_CYCLES_VALU' in v and 'SQ_ACTIVE_INST_VALU' in v:
        line += f" lane_util={v['SQ_THREAD_CYCLES_VALU'] / (64 * v['SQ_ACTIVE_INST_VALU']):.3f}"
    if 'SQ_WAIT_ANY' in v and 'SQ_WAVE_CYCLES' in v:
        line += f" wait_frac={v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.3f} valu_frac={v.get('SQ_ACTIVE_INST_VALU', 0) / v['SQ_WAVE_CYCLES']:.3f}"
    print(line)
