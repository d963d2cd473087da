#!/usr/bin/env python
"""SQ counters + the random-line model's inputs of one scripts/gpu_final_r06.sh visit ->
  profiles/<tag>_sq.md        the two hot-path kernels, in the pipeline and alone: waves, instructions per wave, where a wave's
                              cycles go (issuing / waiting to issue / parked), VALU busy, LDS bank conflicts
  profiles/random_line.json   {transactions_per_launch (TCP -> TCC read + write requests of k_bkt_step per launch),
                               lines_per_s (random 32-byte reads per second, chip-wide: scripts/microbench/random_slope.hip)} —
                              what bench.py's roofline.random_line_frac divides; both measured in THIS visit
usage: python scripts/summarize_sq.py gpurun_out/prof_r06z r06z"""
import csv
import glob
import json
import os
import re
import subprocess
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("k_bkt_step", "k_bkt_part_c")


def short(name):
    return name.split("(")[0].split("<")[0].replace("void ", "").replace("rl::", "").strip()


def per_launch(dirglob):
    """{kernel: {counter: mean over the last 20 launches}}"""
    acc = defaultdict(lambda: defaultdict(list))
    for d in glob.glob(os.path.join(src, dirglob)):
        p = os.path.join(d, "p_counter_collection.csv")
        if not os.path.exists(p):
            continue
        rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v[-20:]) / len(v[-20:]) for c, v in d.items()} for k, d in acc.items()}


commit = open(os.path.join(src, "commit.txt")).read().strip() if os.path.exists(os.path.join(src, "commit.txt")) else None
L = [f"# SQ counters of the two hot-path kernels ({tag}, tree {commit})\n",
     "`scripts/gpu_final_r06.sh`: `rocprofv3 --pmc` in runs of their own with `--kernel-trace` only, `bench.py --steps 20 --warmup 5`; mean over "
     "the last 20 launches; sums over the chip per launch.  SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles.  "
     "\"in the pipeline\" = three batches in flight (the partition of the next batches beside the replay); \"alone\" = `--depth 1`.\n",
     "| kernel | form | waves | VALU / wave | SALU / wave | LDS / wave | wave quad-cycles / wave | issuing | waiting to issue | parked (memory / barrier) | "
     "VALU busy (of wave-cycles) | LDS bank-conflict cycles / LDS active cycles |",
     "|---|---|---|---|---|---|---|---|---|---|---|---|"]
for form, pat in (("in the pipeline", "sq_pipe_*"), ("alone", "sq_alone_*")):
    C = per_launch(pat)
    for k in KERNELS:
        c = C.get(k)
        if not c:
            continue
        w = c.get("SQ_WAVES", float("nan"))
        wc = c.get("SQ_WAVE_CYCLES", float("nan"))
        pct = lambda x: f"{100 * c.get(x, float('nan')) / wc:.0f} %" if wc == wc and wc else "n/a"  # noqa: E731
        lds_act = c.get("SQ_LDS_IDX_ACTIVE", float("nan"))
        L.append(f"| `{k}` | {form} | {w:.0f} | {c.get('SQ_INSTS_VALU', float('nan')) / w:.0f} | {c.get('SQ_INSTS_SALU', float('nan')) / w:.0f} | "
                 f"{c.get('SQ_INSTS_LDS', float('nan')) / w:.0f} | {wc / w:.0f} | {pct('SQ_ACTIVE_INST_ANY')} | {pct('SQ_WAIT_INST_ANY')} | "
                 f"{pct('SQ_WAIT_ANY')} | {pct('SQ_ACTIVE_INST_VALU')} | "
                 f"{100 * c.get('SQ_LDS_BANK_CONFLICT', float('nan')) / lds_act if lds_act == lds_act and lds_act else float('nan'):.0f} % |")
L.append("")
# ---- the random-line model -----------------------------------------------------------------------------------------
T = per_launch("tcp_req")
step = T.get("k_bkt_step", {})
tx = step.get("TCP_TCC_READ_REQ_sum", 0.0) + step.get("TCP_TCC_WRITE_REQ_sum", 0.0)
rate = None
p = os.path.join(src, "random_slope.txt")
if os.path.exists(p):
    big = False
    for ln in open(p):
        if ln.startswith("== table"):
            big = "2^25" in ln or "2^26" in ln
        m = re.search(r"^\s+read32\s+1M\s.*=>\s+([0-9.]+) G accesses/s", ln)  # (the plain row, not "read32 (2nd half late)")
        if m and big and rate is None:
            rate = float(m.group(1)) * 1e9
L.append("## The random-line model's inputs, measured in this visit\n")
L.append(f"`k_bkt_step`, per launch (1 M-hit Zipf-0.99 batch): {step.get('TCP_TCC_READ_REQ_sum', float('nan')):.0f} read requests + "
         f"{step.get('TCP_TCC_WRITE_REQ_sum', float('nan')):.0f} write requests of the vector L1s to the L2 = **{tx:.0f} transactions**; "
         f"the chip's random 32-byte read rate over a table that does not fit the caches (`scripts/microbench/random_slope.hip`, slope "
         f"4 M .. 64 M accesses): **{(rate or float('nan')) / 1e9:.1f} G lines/s**.  -> `profiles/random_line.json`, read by bench.py "
         "(`roofline.random_line_frac` = transactions / rate / the kernel's launch time).\n")
open(os.path.join(ROOT, "profiles", tag + "_sq.md"), "w").write("\n".join(L) + "\n")
if tx and rate:
    json.dump({"source": tag, "commit": commit, "transactions_per_launch": tx, "read_requests": step.get("TCP_TCC_READ_REQ_sum"),
               "write_requests": step.get("TCP_TCC_WRITE_REQ_sum"), "lines_per_s": rate,
               "files": [f"profiles/{tag}_sq.md", f"profiles/{tag}_random_slope.txt"]},
              open(os.path.join(ROOT, "profiles", "random_line.json"), "w"), indent=1)
if os.path.exists(p):
    open(os.path.join(ROOT, "profiles", tag + "_random_slope.txt"), "w").write(open(p).read())
for name in ("pytest_gpu.txt", "smoke.log", "gpu.txt", "host.txt"):
    q = os.path.join(src, name)
    if os.path.exists(q):
        open(os.path.join(ROOT, "profiles", f"{tag}_{name}"), "w").write(open(q).read())
print("\n".join(L))
