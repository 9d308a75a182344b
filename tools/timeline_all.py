"""Like tools/timeline.py, for every kernel of the process (RCCL's included): the last `n` kernels of a rocprofv3 --kernel-trace run.  python tools/timeline_all.py <dir> [n]"""
import csv, glob, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void frt::", "").replace("frt::", "")[:44], r["Queue_Id"]) for r in csv.DictReader(open(f)))
win = ks[-n:]
t0 = win[0][0]; last = {}
for s, e, name, q in win:
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  queue {q}  {name:44s} {'idle before: %.1f us' % gap if gap > 0.05 else ''}")
    last[q] = e
