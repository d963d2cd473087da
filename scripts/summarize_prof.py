#!/usr/bin/env python
"""Turn one scripts/gpu_profile.sh visit (gpurun_out/prof_<tag>/) into the tracked summaries under
profiles/:

  profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats, all kernels
  profiles/<tag>_engine_kernels.md    the engine's kernels: calls, avg/min/max duration
  profiles/<tag>_pmc.md               per engine kernel: FETCH_SIZE, WRITE_SIZE, TCC hit rate, EA
                                      requests per launch, the calibration kernels beside them, and
                                      the HBM bytes per launch derived from them
  profiles/traffic.json               {kernel: {"hbm_bytes_per_launch": ...}} — read by bench.py

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are
collected in separate passes, are in KiB (x1024), and FETCH_SIZE under-reports a wide coalesced
read by exactly 2x on gfx950; other access patterns are uncalibrated there, so the factors are
measured here on kernels with known byte counts (scripts/microbench/pmc_calib.hip) and stated in
the table.

usage: python scripts/summarize_prof.py gpurun_out/prof_r01b r01b
"""
import csv
import json
import os
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles", tag)
NAN = float("nan")


def short(name):
    # `void rl::k_bkt_part<4>(rl::Cell*, ...)` -> `rl::k_bkt_part` (template arguments dropped: one
    # instantiation of each kernel runs in the bench)
    return name.split("(")[0].split("<")[0].replace("void ", "").strip()


# ---- timing ----------------------------------------------------------------------------------
rows = list(csv.DictReader(open(os.path.join(src, "trace", "t_kernel_stats.csv"))))
with open(out + "_kernel_stats.csv", "w") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    w.writerows(rows)
eng = [r for r in rows if short(r["Name"]).startswith("rl::")]
avg_us = {}
with open(out + "_engine_kernels.md", "w") as f:
    f.write(f"rocprofv3 --kernel-trace --stats of `python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --secondary 0` ({tag})\n\n")
    f.write("| kernel | calls | avg us | min us | max us | total ms |\n|---|---|---|---|---|---|\n")
    for r in sorted(eng, key=lambda r: -float(r["TotalDurationNs"])):
        name = short(r["Name"])
        avg_us[name] = float(r["AverageNs"]) / 1e3
        f.write(f"| `{name}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
                f"{float(r['MaxNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.3f} |\n")
# steady state: the 20 timed launches of each hot-path kernel (bench.py runs 5 warm-up batches first;
# the first ones also run with an empty hot-key set)
per = defaultdict(list)
for r in csv.DictReader(open(os.path.join(src, "trace", "t_kernel_trace.csv"))):
    per[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(out + "_engine_kernels.md", "a") as f:
    f.write("\nTimed launches only (the last 20):\n\n| kernel | avg us | min us | max us |\n|---|---|---|---|\n")
    for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1][-20:])):
        if (name.startswith("rl::k_bkt") or name.startswith("rl::k_hot")) and len(v) >= 25:
            t = v[-20:]
            avg_us[name] = sum(t) / len(t)
            f.write(f"| `{name}` | {avg_us[name]:.1f} | {min(t):.1f} | {max(t):.1f} |\n")
print(open(out + "_engine_kernels.md").read())


# ---- counters -----------------------------------------------------------------------------------
def counters(dirname, prefix):
    """{kernel: {counter: [values per dispatch]}}"""
    path = os.path.join(src, dirname, prefix + "_counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(list))
    if not os.path.exists(path):
        return acc
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def mean_tail(v, skip=0):
    v = v[skip:] if len(v) > skip else v
    return sum(v) / len(v) if v else NAN


PASSES = ["FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum+TCC_MISS_sum", "TCC_EA0_RDREQ_sum+TCC_EA0_WRREQ_sum"]
eng_c, cal_c = defaultdict(dict), defaultdict(dict)
for p in PASSES:
    for k, d in counters("pmc_" + p, "p").items():
        for c, v in d.items():
            # bench.py: 5 warm-up + 20 timed batches; skip the warm-up launches of the hot-path kernels
            eng_c[k][c] = mean_tail(v, 5 if len(v) >= 25 else 0)
    for k, d in counters("calib_" + p, "c").items():
        for c, v in d.items():
            cal_c[k][c] = mean_tail(v, 1)

KNOWN = {  # bytes per launch of the calibration kernels
    "k_calib_stream_read": (1 << 30, 0),
    "k_calib_stream_write": (0, 1 << 30),
    "k_calib_random_read32": (64 << 20, 0),
    "k_calib_random_store8": (0, 8 << 20),
}
fac = {}
lines = []
lines.append(f"PMC passes of `python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --secondary 0` ({tag}); one counter group per run, "
             "mean over the 20 timed launches.\n")
lines.append("## Calibration (scripts/microbench/pmc_calib.hip, known bytes per launch)\n")
lines.append("| kernel | known read MiB | known write MiB | FETCH_SIZE (KiB) | WRITE_SIZE (KiB) | EA RDREQ | EA WRREQ | "
             "bytes / FETCH_SIZE KiB | bytes / WRITE_SIZE KiB |")
lines.append("|---|---|---|---|---|---|---|---|---|")
for k, (rd, wr) in KNOWN.items():
    c = cal_c.get(k, {})
    fs, ws = c.get("FETCH_SIZE", NAN), c.get("WRITE_SIZE", NAN)
    fr = rd / fs if rd and fs == fs and fs > 0 else NAN
    fw = wr / ws if wr and ws == ws and ws > 0 else NAN
    fac[k] = (fr, fw)
    lines.append(f"| `{k}` | {rd / 2**20:.0f} | {wr / 2**20:.0f} | {fs:.0f} | {ws:.0f} | "
                 f"{c.get('TCC_EA0_RDREQ_sum', NAN):.0f} | {c.get('TCC_EA0_WRREQ_sum', NAN):.0f} | {fr:.0f} | {fw:.0f} |")
f_stream, f_random = fac["k_calib_stream_read"][0], fac["k_calib_random_read32"][0]
w_stream, w_random = fac["k_calib_stream_write"][1], fac["k_calib_random_store8"][1]
lines.append("")
lines.append(f"Factors used below: streaming read {f_stream:.0f} B per FETCH_SIZE KiB, random 64-B-cell read {f_random:.0f}; "
             f"streaming write {w_stream:.0f} B per WRITE_SIZE KiB, random 8-B store {w_random:.0f} "
             "(1024 = the counter is exact; 2048 = it under-reports 2x).\n")
lines.append("## Engine kernels (per launch)\n")
lines.append("| kernel | avg us | FETCH_SIZE KiB | WRITE_SIZE KiB | TCC hit rate | EA RDREQ | EA WRREQ | HBM read MB | "
             "HBM write MB | pattern |")
lines.append("|---|---|---|---|---|---|---|---|---|---|")
PATTERN = {"rl::k_bkt_step": "random"}
traffic = {}
for k in sorted(eng_c, key=lambda k: -avg_us.get(k, 0)):
    if not k.startswith("rl::"):
        continue
    c = eng_c[k]
    pat = PATTERN.get(k, "stream")
    fr = f_random if pat == "random" else f_stream
    fw = w_random if pat == "random" else w_stream
    fs, ws = c.get("FETCH_SIZE", NAN), c.get("WRITE_SIZE", NAN)
    hit, miss = c.get("TCC_HIT_sum", NAN), c.get("TCC_MISS_sum", NAN)
    # reads: calibrated per pattern; writes: WRITE_SIZE x 1024 IS the memory-side traffic (32-byte
    # granules: the calibration's 8-byte stores cost 32 B each), so it is used as is
    rd_b, wr_b = fs * fr, ws * 1024.0
    hr = hit / (hit + miss) if hit + miss > 0 else NAN
    traffic[k.replace("rl::", "")] = {
        "hbm_bytes_per_launch": (rd_b if rd_b == rd_b else 0) + (wr_b if wr_b == wr_b else 0),
        "hbm_read_bytes": rd_b, "hbm_write_bytes": wr_b, "fetch_size_kib": fs, "write_size_kib": ws,
        "tcc_hit_rate": hr if hr == hr else None, "avg_us": avg_us.get(k)}
    lines.append(f"| `{k}` | {avg_us.get(k, NAN):.1f} | {fs:.0f} | {ws:.0f} | {hr:.3f} | {c.get('TCC_EA0_RDREQ_sum', NAN):.0f} | "
                 f"{c.get('TCC_EA0_WRREQ_sum', NAN):.0f} | {rd_b / 1e6:.1f} | {wr_b / 1e6:.1f} | {pat} |")
open(out + "_pmc.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
import subprocess
try:
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    commit = None
json.dump({"source": tag, "commit": commit, "kernels": traffic}, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
for name in ("bench.json", "bench_under_trace.json"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        open(out + "_" + name, "w").write(open(p).read())


# ---- the general resolver's timeline (scripts/bench_match.py under rocprofv3 --kernel-trace) --------------------------
gen_trace = os.path.join(src, "gen", "g_kernel_trace.csv")
if os.path.exists(gen_trace):
    rows = sorted(csv.DictReader(open(gen_trace)), key=lambda r: int(r["Start_Timestamp"]))
    first = [i for i, r in enumerate(rows) if "k_match_count2" in r["Kernel_Name"] or "k_match_fast<false>" in r["Kernel_Name"]]
    with open(out + "_general_timeline.md", "w") as f:
        f.write(f"rocprofv3 --kernel-trace of `python scripts/bench_match.py --steps 6` ({tag}): one steady-state call of "
                "rl_match_and_check_batch_device (1 M requests -> 3.1 M counters), kernel by kernel.\n\n")
        try:
            line = [l for l in open(os.path.join(src, "gen_bench.json")).read().splitlines() if l.startswith("{")][-1]
            f.write(f"Without the profiler: `{line}`\n\n")
        except Exception:
            pass
        if len(first) >= 2:
            a, b = first[-2], first[-1]
            t0 = int(rows[a]["Start_Timestamp"])
            prev = t0
            f.write("| start us | duration us | idle before us | kernel |\n|---|---|---|---|\n")
            for r in rows[a:b]:
                st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                f.write(f"| {(st - t0) / 1e3:.1f} | {(en - st) / 1e3:.1f} | {(st - prev) / 1e3:.1f} | `{short(r['Kernel_Name'])}` |\n")
                prev = en
            f.write(f"\nDevice span of the call: {(prev - t0) / 1e3:.1f} us.\n")
    print("wrote", out + "_general_timeline.md")
