#!/usr/bin/env python
"""Blocking rl_check_and_update_batch calls with micro-batches of multi-counter requests from host arrays (what the
mirror's MicroBatcher hands over): n_req requests x 3 counters, microseconds per call.
usage: python scripts/bench_micro_batches.py [n_req ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from limitador_amd import workloads as W  # noqa: E402
from limitador_amd.engine import Engine, _ptr  # noqa: E402
from limitador_amd.wire import HIT_DTYPE  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [1, 8, 21, 32, 64, 85, 128, 256, 1024]
eng = Engine(capacity_cells=1 << 20, max_batch_hits=1 << 16)
eng.set_limits([(10**9, 60), (10**9, 60), (10**9, 10)])
rng = np.random.default_rng(W.SEED)
out = {}
now = W.NOW0_US
f, h = eng._lib.rl_check_and_update_batch, eng._h
for n_req in sizes:
    k = 3
    n = n_req * k
    batches = []
    for _ in range(8):
        hits = np.zeros(n, dtype=HIT_DTYPE)
        users = rng.integers(0, 50_000, size=n_req)
        for j in range(k):
            hits["key"][j::k] = W.splitmix64((users * 8 + j).astype(np.uint64))
            hits["limit"][j::k] = j
        hits["delta"] = 1
        batches.append(hits)
    off = (np.arange(n_req + 1) * k).astype(np.uint32)
    verdict = np.zeros(n_req, dtype=np.uint8)
    first = np.zeros(n_req, dtype=np.int32)
    for i in range(20):
        assert f(h, _ptr(batches[i & 7]), n, _ptr(off), n_req, now, 0, _ptr(verdict), _ptr(first), None, None) == 0
        now += 1000
    t = []
    for i in range(300):
        t0 = time.perf_counter()
        rc = f(h, _ptr(batches[i & 7]), n, _ptr(off), n_req, now, 0, _ptr(verdict), _ptr(first), None, None)
        t.append(time.perf_counter() - t0)
        assert rc == 0
        now += 1000
    t = np.sort(np.array(t)) * 1e6
    out[n_req] = {"hits": n, "p50_us": round(float(t[len(t) // 2]), 1), "p99_us": round(float(t[int(len(t) * 0.99)]), 1)}
print(json.dumps({"what": "blocking rl_check_and_update_batch, n_req requests x 3 counters from host arrays",
                  "env": {k: v for k, v in os.environ.items() if k.startswith("RL_")}, "us": out}))
eng.close()
