"""The last <ms> milliseconds of a rocprofv3 --kernel-trace (+ --memory-copy-trace) CSV directory as one device timeline: start
(us, relative), duration, queue, name — kernels of under <min_us> are folded into a count per gap.
usage: timeline_tail.py <dir> [ms=6] [min_us=0] [skip_ms=0]"""
import csv
import glob
import sys

d = sys.argv[1]
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
skip_ms = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
ev = []
for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-40:]))
for p in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
end = ev[-1][1] - int(skip_ms * 1e6)
t0 = end - int(ms * 1e6)
small = 0
for s, e, q, name in ev:
    if s < t0 or s > end:
        continue
    if (e - s) / 1e3 < min_us:
        small += 1
        continue
    if small:
        print(f"            ... {small} shorter")
        small = 0
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  {q:>5}  {name}")
