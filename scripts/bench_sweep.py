#!/usr/bin/env python
"""Streaming operations over the configs[2] table (10 M keys, 2^25 32-byte cells = 1.07 GB): rl_sweep_expired
(the kernel expected near the HBM roofline), rl_get_counters (count only), rl_dump-free scans.
Algorithmic bytes: 32 B per slot scanned (tag, value, expiry, limit) + 8 B per cell removed.  Prints one JSON line;
run under `rocprofv3 --kernel-trace --stats` for the kernel-only durations (k_scan<3> = sweep, k_scan<0> = get)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from limitador_amd import workloads as W  # noqa: E402
from limitador_amd.engine import Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--keys", type=int, default=10_000_000)
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda", 0)
cap = 1 << (int(args.keys * 2.2 - 1).bit_length())
eng = Engine(capacity_cells=cap, max_batch_hits=1 << 20)
eng.set_limits([(W.MAX_VALUE, W.WINDOW_S)])
rows = W.torch_universe_rows(args.keys, dev)
# a tenth of the cells expire at every sweep step: expiry = NOW0 + (1 + i % 10) seconds
rows[:, 3] = W.NOW0_US + (1 + torch.arange(args.keys, device=dev) % 10) * 1_000_000
torch.cuda.synchronize()  # (the engine loads on its own stream)
for lo in range(0, args.keys, 1 << 20):
    part = rows[lo:lo + (1 << 20)].contiguous()
    eng.load_cells_device(part.data_ptr(), part.shape[0])
del rows
torch.cuda.synchronize()
table_bytes = cap * 32
live0 = eng.stats()["live_cells"]
out = {"what": "streaming scans over the counter table", "keys": args.keys, "live_cells": live0, "capacity_cells": cap, "table_bytes": table_bytes}
# get_counters, count only (no rows copied out): every live, unexpired cell of limit 0
t0 = time.perf_counter()
for _ in range(args.reps):
    n = eng.count_counters(0, W.NOW0_US)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.reps
out["get_counters_count_only"] = {"rows_counted": int(n), "ms": dt * 1e3, "GBps_32B_per_slot": table_bytes / dt / 1e9, "frac_of_8TBps": table_bytes / dt / 8e12}
# sweep: one second later each time (10 % of the original cells expire per call; no compaction before 1/8 tombstones)
times, removed = [], []
for i in range(3):
    t0 = time.perf_counter()
    removed.append(eng.sweep_expired(W.NOW0_US + (1 + i) * 1_000_000))
    times.append(time.perf_counter() - t0)
st = eng.stats()
dt = sum(times[:1]) / 1  # the first call is a pure sweep (tombstones below cap/8 afterwards)
out["sweep_expired"] = {"calls_ms": [t * 1e3 for t in times], "removed": removed, "rebuilds_after": st["rebuilds"],
                        "first_call": {"ms": dt * 1e3, "GBps": (table_bytes + 8 * removed[0]) / dt / 1e9,
                                       "frac_of_8TBps": (table_bytes + 8 * removed[0]) / dt / 8e12}}
# compaction in place (k_compact_mark + k_compact_shift) of the table the three sweeps left: 30 % of the cells are tombstones
# (unless the sweeps' own threshold — tombstones > capacity / 8 — has compacted already: rebuilds_after says)
t0 = time.perf_counter()
eng.compact()
dt = time.perf_counter() - t0
st = eng.stats()
out["compact_after_sweeps"] = {"ms": dt * 1e3, "live_after": st["live_cells"], "tombstones_after": st["tombstones"]}
# and with 4 more tenths gone in one sweep (the threshold compaction runs inside the call: sweep + compaction)
t0 = time.perf_counter()
rem = eng.sweep_expired(W.NOW0_US + 7_000_000)
dt = time.perf_counter() - t0
st = eng.stats()
out["sweep_40pct_incl_threshold_compaction"] = {"ms": dt * 1e3, "removed": int(rem), "rebuilds_after": st["rebuilds"], "live_after": st["live_cells"]}
print(json.dumps(out))
eng.close()
