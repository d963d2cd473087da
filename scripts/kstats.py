"""Per-kernel launch count / mean / min / max duration (us) of a rocprofv3 --kernel-trace CSV, engine kernels only.
usage: kstats.py <kernel_trace.csv> [skip_first_n_launches_per_kernel]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
d = defaultdict(list)
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    name = r["Kernel_Name"].split("(")[0]
    if "rl::" not in name:
        continue
    d[name.replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    w = v[skip:] if len(v) > skip else v
    print(f"{k[:60]:60s} n={len(v):5d} mean={sum(w) / len(w):8.1f} min={min(w):8.1f} max={max(w):8.1f} us")
