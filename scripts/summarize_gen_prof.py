#!/usr/bin/env python
"""Turn one scripts/gpu_profile_gen.sh visit (gpurun_out/prof_<tag>/) into profiles/<tag>_general_kernels.md: the general
resolver's kernels (scripts/bench_match.py: 1 M requests -> 3.1 M counters per call), per CALL — launches, time, HBM bytes
from the PMC passes (counter handling as scripts/summarize_prof.py / MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE in
their own passes, KiB units, factors measured on kernels with known byte counts in the same visit), L2 hit rate, EA
requests — and the traffic / algorithmic ratio of the call (49 B per counter).

usage: python scripts/summarize_gen_prof.py gpurun_out/prof_r06a r06a
"""
import csv
import json
import os
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles", tag + "_general_kernels.md")
NAN = float("nan")


def short(name):
    s = name.split("(")[0].replace("void ", "").strip()
    return s


def is_start(name):  # the first kernel of one rl_match_and_check_batch_device call
    return "k_match_count2" in name or "k_match_fast<false>" in name


def calls_of(path, value_of):
    """-> list of calls, each {kernel: [values]} in launch order; only complete steady-state calls (the last 4)."""
    rows = sorted(csv.DictReader(open(path)), key=lambda r: (int(r.get("Start_Timestamp", 0) or 0), int(r.get("Dispatch_Id", 0) or 0)))
    calls, cur = [], None
    for r in rows:
        name = short(r["Kernel_Name"])
        if not name.startswith("rl::"):
            continue
        if is_start(r["Kernel_Name"]):
            cur = defaultdict(list)
            calls.append(cur)
        if cur is not None:
            cur[name].append(value_of(r))
    return calls[-5:-1] if len(calls) >= 6 else calls[-3:-1]


def per_call(calls):
    """mean over calls of {kernel: (launches, total)}"""
    acc = defaultdict(lambda: [0.0, 0.0])
    for c in calls:
        for k, v in c.items():
            acc[k][0] += len(v)
            acc[k][1] += sum(v)
    n = max(1, len(calls))
    return {k: (a[0] / n, a[1] / n) for k, a in acc.items()}


trace = os.path.join(src, "gen", "g_kernel_trace.csv")
t_calls = calls_of(trace, lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
t = per_call(t_calls)
# device span of a call: first start -> last end
rows = sorted((r for r in csv.DictReader(open(trace)) if short(r["Kernel_Name"]).startswith("rl::")), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if is_start(r["Kernel_Name"])]
spans, busy = [], []
for a, b in zip(starts[-5:-1], starts[-4:]):
    spans.append((int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
    busy.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b]) / 1e3)


def counter_calls(dirname, prefix="p"):
    path = os.path.join(src, dirname, prefix + "_counter_collection.csv")
    if not os.path.exists(path):
        return {}
    by = defaultdict(dict)  # dispatch -> {name, counters}
    for r in csv.DictReader(open(path)):
        d = by[int(r["Dispatch_Id"])]
        d["Kernel_Name"] = r["Kernel_Name"]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    seq = [by[k] for k in sorted(by)]
    res = {}
    names = {c for d in seq for c in d if c != "Kernel_Name"}
    for cname in names:
        calls, cur = [], None
        for d in seq:
            name = short(d["Kernel_Name"])
            if not name.startswith("rl::"):
                continue
            if is_start(d["Kernel_Name"]):
                cur = defaultdict(list)
                calls.append(cur)
            if cur is not None:
                cur[name].append(d.get(cname, 0.0))
        calls = calls[-5:-1] if len(calls) >= 6 else calls[-3:-1]
        res[cname] = per_call(calls)
    return res


def calib(dirname):
    path = os.path.join(src, dirname, "c_counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(list))
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            acc[short(r["Kernel_Name"]).split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v[1:]) / max(1, len(v[1:])) for c, v in d.items()} for k, d in acc.items()}


PASSES = ["FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum+TCC_MISS_sum", "TCC_EA0_RDREQ_sum+TCC_EA0_WRREQ_sum"]
C = {}
cal = defaultdict(dict)
for p in PASSES:
    C.update(counter_calls("pmc_" + p))
    for k, d in calib("calib_" + p).items():
        cal[k].update(d)
KNOWN = {"k_calib_stream_read": (1 << 30, 0), "k_calib_stream_write": (0, 1 << 30), "k_calib_random_read32": (64 << 20, 0),
         "k_calib_random_store8": (0, 8 << 20)}


def factor(kernel, counter, nbytes):
    v = cal.get(kernel, {}).get(counter)
    return nbytes / v if v else NAN


f_stream = factor("k_calib_stream_read", "FETCH_SIZE", 1 << 30)
f_random = factor("k_calib_random_read32", "FETCH_SIZE", 64 << 20)
w_stream = factor("k_calib_stream_write", "WRITE_SIZE", 1 << 30)
w_random = factor("k_calib_random_store8", "WRITE_SIZE", 8 << 20)
# which kernels read mostly at random (cells, flags gathered by request / by index); the others stream
RANDOM_READS = ("k_gen_sort", "k_gen_round", "k_gen_commit", "k_gen_piece_sum", "k_gen_load", "k_gen_reach", "k_gen_check_keys")

line = {}
try:
    line = json.loads([l for l in open(os.path.join(src, "gen_bench.json")).read().splitlines() if l.startswith("{")][-1])
except Exception:
    pass
n_ctr = line.get("counters_per_batch", 0)
algo = 49.0 * n_ctr
L = []
L.append(f"# General resolver, kernel by kernel ({tag})\n")
L.append("`scripts/gpu_profile_gen.sh`: `python scripts/bench_match.py` = rl_match_and_check_batch_device, 1 M requests -> "
         f"{n_ctr} counters per call (4 namespaces x 8 limits, Zipf(1.2) users, no load_counters).  Figures are PER CALL, the mean of "
         f"{len(t_calls)} steady-state calls of the profiled run (`--steps 6`, the last call left out).\n")
if line:
    L.append(f"Without the profiler (`--steps 20`): **{line['ms_per_step']:.4f} ms per call** = {line['requests_per_s'] / 1e9:.2f} G requests/s, "
             f"{n_ctr * 49 / (line['ms_per_step'] * 1e-3) / 1e9:.0f} GB/s algorithmic (49 B per counter) = "
             f"{n_ctr * 49 / (line['ms_per_step'] * 1e-3) / 8e12:.4f} of 8 TB/s.\n")
if spans:
    L.append(f"Under `rocprofv3 --kernel-trace`: device span of a call {sum(spans) / len(spans):.1f} us, of which kernels "
             f"{sum(busy) / len(busy):.1f} us and gaps between launches {(sum(spans) - sum(busy)) / len(spans):.1f} us.\n")
L.append("## Calibration of the counters in this visit (scripts/microbench/pmc_calib.hip)\n")
L.append(f"bytes per FETCH_SIZE unit: streaming read {f_stream:.0f}, random 32-B read {f_random:.0f}; bytes per WRITE_SIZE unit: "
         f"streaming write {w_stream:.0f}, random 8-B store {w_random:.0f} (1024 = the counter's KiB is exact; 2048 = it under-reports 2x; "
         "a random 8-B store costs a 32-B granule: WRITE_SIZE x 1024 IS the memory-side write traffic).\n")
L.append("## Per kernel, per call\n")
L.append("| kernel | launches | us | % of kernel time | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM read MB | HBM write MB | L2 hit rate | EA RDREQ | EA WRREQ | reads |")
L.append("|---|---|---|---|---|---|---|---|---|---|---|---|")
tot_us = sum(v[1] for v in t.values())
tot_rd = tot_wr = 0.0
summary = {}
for k, (n, us) in sorted(t.items(), key=lambda kv: -kv[1][1]):
    g = lambda c: C.get(c, {}).get(k, (0, NAN))[1]  # noqa: E731
    fs, ws, hit, miss = g("FETCH_SIZE"), g("WRITE_SIZE"), g("TCC_HIT_sum"), g("TCC_MISS_sum")
    rnd = any(x in k for x in RANDOM_READS)
    rd = fs * (f_random if rnd else f_stream)
    wr = ws * 1024.0
    if rd == rd:
        tot_rd += rd
    if wr == wr:
        tot_wr += wr
    hr = hit / (hit + miss) if hit + miss > 0 else NAN
    summary[k.replace("rl::", "")] = {"launches": n, "us": us, "hbm_read_bytes": rd, "hbm_write_bytes": wr, "l2_hit_rate": hr}
    L.append(f"| `{k.replace('rl::', '')}` | {n:.1f} | {us:.1f} | {100 * us / tot_us:.1f} | {fs:.0f} | {ws:.0f} | {rd / 1e6:.1f} | {wr / 1e6:.1f} | "
             f"{hr:.3f} | {g('TCC_EA0_RDREQ_sum'):.0f} | {g('TCC_EA0_WRREQ_sum'):.0f} | {'random' if rnd else 'stream'} |")
L.append(f"| **all** | {sum(v[0] for v in t.values()):.0f} | {tot_us:.1f} | 100 | | | {tot_rd / 1e6:.1f} | {tot_wr / 1e6:.1f} | | | | |")
L.append("")
if algo:
    L.append(f"**Traffic / algorithmic**: {(tot_rd + tot_wr) / 1e6:.0f} MB of HBM traffic per call against {algo / 1e6:.0f} MB algorithmic "
             f"({n_ctr} counters x 49 B) = **{(tot_rd + tot_wr) / algo:.2f} x**.  Average rate over the kernels' own time: "
             f"{(tot_rd + tot_wr) / (tot_us * 1e-6) / 1e12:.2f} TB/s — the call is not bound by HBM bytes.\n")
groups = [("matching (k_match*)", lambda k: "k_match" in k or "k_xscan" in k or "k_m_" in k),
          ("ordering the hits: partition (k_bkt_hist / scan / scatter)", lambda k: "k_bkt_" in k),
          ("ordering the hits: k_gen_sort", lambda k: "k_gen_sort" in k),
          ("fixpoint rounds (k_gen_admit*, piece_sum, round)", lambda k: any(x in k for x in ("k_gen_admit", "k_gen_piece_sum", "k_gen_round", "k_gen_load"))),
          ("final / reach / count / commit / post", lambda k: any(x in k for x in ("k_gen_final", "k_gen_reach", "k_gen_count", "k_gen_commit", "k_gen_post", "k_gen_heads")))]
L.append("## By stage\n")
L.append("| stage | us | % |\n|---|---|---|")
seen = set()
for name, pred in groups:
    us = sum(v[1] for k, v in t.items() if pred(k) and k not in seen)
    seen |= {k for k in t if pred(k)}
    L.append(f"| {name} | {us:.1f} | {100 * us / tot_us:.1f} |")
rest = sum(v[1] for k, v in t.items() if k not in seen)
L.append(f"| other | {rest:.1f} | {100 * rest / tot_us:.1f} |")
open(out, "w").write("\n".join(L) + "\n")
json.dump({"source": tag, "line": line, "kernels": summary, "hbm_read_bytes": tot_rd, "hbm_write_bytes": tot_wr,
           "algorithmic_bytes": algo}, open(os.path.join(ROOT, "profiles", tag + "_general_traffic.json"), "w"), indent=1)
print("\n".join(L))
