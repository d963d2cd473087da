#!/usr/bin/env python
"""Per-request calls in a tight loop through the C ABI (ctypes, preallocated arrays): one request of three simple
counters per call (BASELINE.json configs[0]'s shape), microseconds per call.  RL_SERVE=1 (default): answered by a
lingering k_gen_serve through the host-mapped mailbox; RL_SERVE=0: one kernel launch per call.  RL_APPLY_TRACE=1 prints
how many calls were served by how many launches when the engine is destroyed.
usage: python scripts/bench_per_request.py [calls]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from limitador_amd.engine import Engine, _ptr  # noqa: E402
from limitador_amd.wire import HIT_DTYPE, RL_SIMPLE  # noqa: E402

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
eng = Engine(capacity_cells=1 << 12, max_batch_hits=1 << 10)
eng.set_limits([(10, 60), (5, 60), (50000, 10)])  # limitador-server/sandbox/limits.yaml
for i in range(3):
    eng.add_counter(i | RL_SIMPLE, 1000 + i)
hits = np.zeros(3, dtype=HIT_DTYPE)
hits["key"] = [1000, 1001, 1002]
hits["limit"] = np.arange(3, dtype=np.uint32) | RL_SIMPLE
hits["delta"] = 1
off = np.array([0, 3], dtype=np.uint32)
verdict = np.zeros(1, dtype=np.uint8)
first = np.zeros(1, dtype=np.int32)
f, h = eng._lib.rl_check_and_update_batch, eng._h
now = 1_700_000_000_000_000
for _ in range(200):
    f(h, _ptr(hits), 3, _ptr(off), 1, now, 0, _ptr(verdict), _ptr(first), None, None)
t0 = time.perf_counter()
limited = 0
for i in range(n_calls):
    rc = f(h, _ptr(hits), 3, _ptr(off), 1, now + i, 0, _ptr(verdict), _ptr(first), None, None)
    if rc:
        raise SystemExit(f"rl_check_and_update_batch -> {rc}: {eng._lib.rl_last_error(h).decode()}")
    limited += int(verdict[0])
dt = time.perf_counter() - t0
print(json.dumps({"what": "rl_check_and_update_batch, one request x 3 counters per call (ctypes loop)", "calls": n_calls,
                  "us_per_call": dt / n_calls * 1e6, "limited": limited, "serve": os.environ.get("RL_SERVE", "1")}))
eng.close()
