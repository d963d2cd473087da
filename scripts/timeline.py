"""Print the device timeline of the last steps of a rocprofv3 --kernel-trace CSV: start relative to
the previous k_bkt_part, duration, gap to the previous kernel's end, queue.  usage: timeline.py <csv> [n_rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-n - 60:-60] if len(rows) > n + 60 else rows[-n:]
t0 = None
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-42:]
    if "k_bkt_part" in name or t0 is None:
        t0 = s
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:7.1f}  q{r.get('Queue_Id', '?'):>3}  {name}")
    prev_end = max(prev_end or 0, e)
