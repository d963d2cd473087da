#!/usr/bin/env python
"""Copy the rocprofv3 --kernel-trace --stats summary of a bench run into profiles/ (tracked).

usage: python scripts/summarize_prof.py gpurun_out/prof/r01 profiles/r01_n1
Writes <out>_kernel_stats.csv (all kernels) and <out>_engine_kernels.md (the engine's own kernels,
average duration per launch — the numbers bench.py's HIP-event timings must agree with).
"""
import csv
import sys

src, out = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src + "_kernel_stats.csv")))
with open(out + "_kernel_stats.csv", "w") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    w.writerows(rows)
eng = [r for r in rows if r["Name"].startswith(("rl::", "void rl::"))]
with open(out + "_engine_kernels.md", "w") as f:
    f.write("| kernel | calls | avg us | min us | max us | total ms |\n|---|---|---|---|---|---|\n")
    for r in sorted(eng, key=lambda r: -float(r["TotalDurationNs"])):
        name = r["Name"].split("(")[0].replace("void ", "")
        f.write(f"| `{name}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
                f"{float(r['MaxNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.3f} |\n")
print(open(out + "_engine_kernels.md").read())
