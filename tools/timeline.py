"""Kernel timeline of the two-stream frame from a rocprofv3 --kernel-trace run (tools/profile_all.sh writes one under gpurun_out/prof_<tag>/):
per hardware queue the kernels of a few frames in the middle of the run with their start, end and duration, the idle time between consecutive
kernels of a queue, and per frame the busy and idle time of each queue.   python tools/timeline.py gpurun_out/prof_<tag> [frames]"""
import csv, glob, sys, collections

d = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
ks = []
for r in csv.DictReader(open(f)):
    if "frt::" in r["Kernel_Name"]:
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void frt::", "").replace("frt::", ""), r["Queue_Id"]))
ks.sort()
merges = [i for i, k in enumerate(ks) if k[2] == "merge_kernel"]
a, b = merges[len(merges) // 2], merges[len(merges) // 2 + frames]
win = ks[a:b]
t0 = win[0][0]
last = {}
busy, idle = collections.Counter(), collections.Counter()
for s, e, n, q in win:
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  queue {q}  {n:28s} {'idle before: %.1f us' % gap if gap > 0.05 else ''}")
    busy[q] += (e - s) / 1e3
    if q in last: idle[q] += max(gap, 0.0)
    last[q] = e
period = (ks[b][0] - ks[a][0]) / 1e3 / frames
print(f"frame period {period:.1f} us; per frame and queue: " + "; ".join(f"queue {q}: busy {busy[q] / frames:.1f} us, idle {idle[q] / frames:.1f} us" for q in sorted(busy)))
