"""One step's device timeline out of a rocprofv3 --kernel-trace (+ --memory-copy-trace) CSV directory: kernels and copies of the
last-but-one step (a step starts at the kernel whose name holds <first>), start relative to the step, duration, gap to
the end of whatever ran before.  usage: timeline_step.py <dir> <first-kernel-substring>"""
import csv
import glob
import sys

d, first = sys.argv[1], sys.argv[2]
ev = []
for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]))
for p in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
ev.sort()
starts = [i for i, e in enumerate(ev) if first in e[2]]
if len(starts) < 3:
    print("no steps found; kernels:", sorted({e[2] for e in ev})[:40])
    sys.exit(0)
a, b = starts[-2], starts[-1]
t0 = ev[a][0]
prev = None
busy = 0
for s, e, name in ev[a:b]:
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:7.1f}  {name}")
    busy += e - s
    prev = max(prev or 0, e)
print(f"step: {(ev[b][0] - t0) / 1e3:.1f} us start to start, {busy / 1e3:.1f} us of kernels and copies")
