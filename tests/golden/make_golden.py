#!/usr/bin/env python
"""Generate the committed golden traces under tests/golden/ (run from the repo root:
`python tests/golden/make_golden.py`).

The Rust reference cannot be executed in this image, so the expected outputs are produced by the CPU
oracle (oracle/limitador_oracle.c) — itself pinned by the reference's own vectors
(tests/test_oracle_golden.py, tests/scenarios.py).  The traces freeze that pinned behaviour:
`-m "not gpu"` replays them through the oracle (a regression pin: the oracle may not drift), `-m gpu`
replays them through the HIP engine WITHOUT the oracle in the loop.

Each trace is one .npz: limit rows, the simple counters to pre-create, and a list of events
  ("check", hits, req_off | None, load_counters, now_us) -> verdict, first_limited, remaining, expires_in
  ("update", hits, now_us)          update_counter
  ("within", hits, now_us)          -> within
  ("sweep", now_us)                 -> n_removed          (explicit eviction event)
  ("delete", limit) / ("clear",)
followed by the final table (sorted rows of key, limit, value, expiry)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from limitador_amd import workloads as W  # noqa: E402
from limitador_amd.wire import HIT_DTYPE, RL_SIMPLE  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SEC = 1_000_000
NOW = W.NOW0_US


def hits_of(keys, limits, deltas):
    h = np.empty(len(keys), dtype=HIT_DTYPE)
    h["key"], h["limit"], h["delta"] = keys, limits, deltas
    return h


def final_rows(orc, keys, simple):
    rows = []
    for k in sorted(set(int(x) for x in keys)):
        got = orc.peek(k)
        if got is not None:
            rows.append((k, got[2], got[0], got[1]))
    for limit, key in simple:
        got = orc.peek_simple(limit | RL_SIMPLE)
        if got is not None:
            rows.append((key, limit | RL_SIMPLE, got[0], got[1]))
    return np.array(rows, dtype=np.uint64).reshape(-1, 4)


def record(name, rows, simple, events):
    """Run the events through the oracle and save inputs + outputs."""
    orc = oracle.OracleStorage()
    orc.set_limits(rows)
    for limit, key in simple:
        orc.add_counter(limit | RL_SIMPLE, key)
    data = {"limit_rows": np.array(rows, dtype=np.uint64), "simple": np.array(simple, dtype=np.uint64).reshape(-1, 2),
            "n_events": np.array(len(events))}
    all_keys = []
    for i, ev in enumerate(events):
        kind = ev[0]
        data[f"e{i}_kind"] = np.array(kind)
        if kind == "check":
            _, hits, req_off, load, now = ev
            v, f, r, e = orc.check_and_update(hits, now, req_off=req_off, load_counters=load)
            data[f"e{i}_hits"], data[f"e{i}_now"], data[f"e{i}_load"] = hits, np.array(now), np.array(int(load))
            if req_off is not None:
                data[f"e{i}_req_off"] = np.asarray(req_off, dtype=np.uint32)
            data[f"e{i}_verdict"], data[f"e{i}_first"] = v, f
            if load:
                data[f"e{i}_remaining"], data[f"e{i}_expires"] = r, e
            all_keys.append(hits["key"][(hits["limit"] & RL_SIMPLE) == 0])
        elif kind == "update":
            _, hits, now = ev
            orc.update_counters(hits, now)
            data[f"e{i}_hits"], data[f"e{i}_now"] = hits, np.array(now)
            all_keys.append(hits["key"][(hits["limit"] & RL_SIMPLE) == 0])
        elif kind == "within":
            _, hits, now = ev
            data[f"e{i}_hits"], data[f"e{i}_now"] = hits, np.array(now)
            data[f"e{i}_within"] = orc.is_within_limits(hits, now)
        elif kind == "sweep":
            data[f"e{i}_now"] = np.array(ev[1])
            data[f"e{i}_removed"] = np.array(orc.sweep_expired(ev[1]))
        elif kind == "delete":
            data[f"e{i}_limit"] = np.array(ev[1])
            orc.delete_counters(ev[1])
        elif kind == "clear":
            orc.clear()
    data["final"] = final_rows(orc, np.concatenate(all_keys) if all_keys else [], simple)
    orc.close()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in list(data.items())[:4]}, len(events), "events")


def trace_single_counter():
    """k = 1 requests: Zipf keys over five limits incl. a 0-second window and u64::MAX, mixed deltas,
    two simple counters, windows rolling over, a sweep in the middle."""
    rng = np.random.default_rng(101)
    rows = [(5, 1), (50, 10), (1000, 60), (0, 60), (2**64 - 1, 3600), (7, 0)]
    simple = [(4, 9_000_001), (0, 9_000_002)]
    n_keys = 400
    key_limit = rng.integers(0, 4, size=n_keys)
    key_limit[:8] = 5
    events, now = [], NOW
    for step in range(10):
        n = int(rng.integers(200, 2500))
        idx = (rng.zipf(1.3, size=n) - 1) % n_keys if step % 2 else rng.integers(0, n_keys, size=n)
        keys = W.splitmix64(idx.astype(np.uint64))
        limits = key_limit[idx].astype(np.uint32)
        deltas = rng.integers(0, 4, size=n) if step % 3 else np.ones(n, dtype=np.uint32)
        sm = rng.random(n) < 0.05
        which = rng.integers(0, 2, size=n)
        keys[sm] = np.where(which[sm] == 0, 9_000_001, 9_000_002)
        limits[sm] = np.where(which[sm] == 0, 4, 0) | RL_SIMPLE
        events.append(("check", hits_of(keys, limits, deltas), None, False, now))
        if step == 5:
            events.append(("sweep", now + 3 * SEC))
        now += int(rng.integers(0, 3 * SEC))
    record("single_counter", rows, simple, events)


def trace_multi_counter_load():
    """k in [0, 6] counters per request (simple first, then qualified), load_counters on and off."""
    rng = np.random.default_rng(202)
    rows = [(20, 60), (5, 10), (3, 1), (100, 60), (2, 60)]
    simple = [(0, 7_100_000), (1, 7_100_001)]
    events, now = [], NOW
    for step in range(8):
        n_req = int(rng.integers(50, 400))
        hits, off = [], [0]
        for _ in range(n_req):
            k = int(rng.integers(0, 7))
            cs = []
            for s in range(2):
                if len(cs) < k and rng.random() < 0.5:
                    cs.append((7_100_000 + s, s | RL_SIMPLE))
            while len(cs) < k:
                u = int(rng.integers(0, 40))
                lim = 2 + u % 3
                cs.append((int(W.splitmix64(np.array([u * 8 + lim], dtype=np.uint64))[0]), lim))
            delta = int(rng.integers(1, 4))
            hits += [(key, lim, delta) for key, lim in cs]
            off.append(len(hits))
        h = np.array(hits, dtype=HIT_DTYPE) if hits else np.zeros(0, dtype=HIT_DTYPE)
        events.append(("check", h, np.array(off, dtype=np.uint32), bool(step % 2), now))
        now += int(rng.integers(0, 2 * SEC))
    record("multi_counter_load", rows, simple, events)


def trace_storage_surface():
    """update_counter / is_within_limits / delete / clear / sweep between checks."""
    rng = np.random.default_rng(303)
    rows = [(30, 1), (30, 10), (30, 60)]
    simple = [(2, 8_000_001)]
    events, now = [], NOW
    for step in range(6):
        n = int(rng.integers(100, 1500))
        idx = rng.integers(0, 150, size=n)
        h = hits_of(W.splitmix64(idx.astype(np.uint64)), (idx % 2).astype(np.uint32), rng.integers(0, 5, size=n))
        sm = rng.random(n) < 0.05
        h["key"][sm], h["limit"][sm] = 8_000_001, 2 | RL_SIMPLE
        events.append(("within", h, now))
        events.append(("update" if step % 2 else "check", h, now) if step % 2 else ("check", h, None, False, now))
        events.append(("within", h, now))
        now += int(rng.integers(0, 2 * SEC))
        if step == 2:
            events.append(("sweep", now))
        if step == 3:
            events.append(("delete", 1))
        if step == 4:
            events.append(("clear",))
    record("storage_surface", rows, simple, events)


if __name__ == "__main__":
    oracle.build()
    trace_single_counter()
    trace_multi_counter_load()
    trace_storage_surface()
