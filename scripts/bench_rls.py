#!/usr/bin/env python
"""Added latency of the wire path vs batch size: rli_serve_batch = decode the serialized
RateLimitRequests + limit matching + check_and_update on the device + the serialized RateLimitResponses, for a
batch of N requests (4 namespaces x 8 limits, Zipf users).  A request that waits for a batch of N pays at most
max_delay (the batcher's budget) + this.  Prints one JSON line: per N the p50 / p99 of the call and requests/s,
without and with the draft-03 headers (load_counters).
usage: python scripts/bench_rls.py [exact|hashed] [sizes, comma-separated]   exact: host dictionaries + packed ids; hashed: the messages decoded on
the device, counters keyed by a hash of their canonical key bytes (rli_set_key_mode, rl_wire.hpp)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from limitador_amd.engine import Engine  # noqa: E402
from limitador_amd.ingest import Ingest  # noqa: E402
from test_ingest_cpu import rls_request  # noqa: E402  (the hand-written wire encoder)

rng = np.random.default_rng(5)
eng = Engine(capacity_cells=1 << 22, max_batch_hits=1 << 21, max_limits=64)
KEYS = sys.argv[1] if len(sys.argv) > 1 else "exact"
g = Ingest(keys=KEYS)
methods = ["GET", "POST", "PUT"]
for n in range(4):
    for j in range(8):
        nv = 0 if j < 2 else (1 if j < 5 else 2)
        conds = [f"descriptors[0]['method'] {'==' if j % 2 == 0 else '!='} '{methods[j % 3]}'"]
        variables = [] if nv == 0 else (["descriptors[0]['user']"] if nv == 1 else ["descriptors[0]['app']", "descriptors[0]['user']"])
        lid = g.add_limit(f"ns{n}", 10**9 if j == 0 else 1000, [1, 10, 60, 3600][(n + j) % 4], conds, variables)
        g.set_limit_name(lid, f"ns{n}-limit{j}")
g.install(eng)


def messages(n):
    out = []
    for _ in range(n):
        out.append(rls_request(f"ns{int(rng.integers(0, 4))}",
                               [[("method", methods[int(rng.integers(0, 3))]), ("path", "/x"),
                                 ("user", f"user{int(rng.zipf(1.2)) % 200000}"), ("app", f"app{int(rng.integers(0, 5))}")]]))
    return out


now = 1_700_000_000_000_000
out = {"what": "rli_serve_batch: wire bytes -> verdicts + RateLimitResponse bytes", "keys": KEYS, "sizes": {}}
SIZES = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 16, 256, 4096, 32768, 262144)
for n in SIZES:
    STRIDE = int(os.environ.get("BENCH_RLS_STRIDE", "1024"))  # the caller's slot per response (out + i * stride)
    prep = g.prepare_batch(messages(n), stride=STRIDE)  # (the ctypes marshalling of the Python harness is not what is measured)
    row = {}
    for hdr in (False, True):
        g.serve_prepared(eng, prep, now, with_headers=hdr)
        ts = []
        for _ in range(30 if n <= 4096 else 10):
            now += 1000
            t0 = time.perf_counter()
            g.serve_prepared(eng, prep, now, with_headers=hdr)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts)
        row["with_headers" if hdr else "codes_only"] = {"p50_ms": float(np.percentile(ts, 50) * 1e3), "p99_ms": float(np.percentile(ts, 99) * 1e3),
                                                       "requests_per_s": n / float(np.percentile(ts, 50))}
    if n >= 32768:  # the Kuadrant service's two methods on the same messages (rli_serve_batch_op: no headers)
        for name, op in (("kuadrant_check", 1), ("kuadrant_report", 2)):
            g.serve_prepared_op(eng, prep, op, now)
            ts = []
            for _ in range(10):
                now += 1000
                t0 = time.perf_counter()
                g.serve_prepared_op(eng, prep, op, now)
                ts.append(time.perf_counter() - t0)
            row[name] = {"p50_ms": float(np.percentile(ts, 50) * 1e3), "requests_per_s": n / float(np.percentile(ts, 50))}
    if n >= 32768 and KEYS == "hashed":
        # SEVERAL calls in flight (round 6): T threads, each with its own prepared batch, call rli_serve_batch back to back; a
        # call takes one of the engine's RL_SERVE_SETS serving sets, so one thread packs / copies in / decides while the
        # others' responses cross PCIe and are handed on.  Sustained time per batch = elapsed / calls; the call's own latency
        # beside it.
        import threading

        preps = [prep] + [g.prepare_batch(messages(n), stride=STRIDE) for _ in range(3)]
        K = 12
        clock = [now]
        clock_mu = threading.Lock()
        for T, name in ((2, "with_headers_two_in_flight"), (3, "with_headers_three_in_flight"), (4, "with_headers_four_in_flight")):
            lat = [[] for _ in range(T)]

            def pump(t, k):
                for _ in range(k):
                    with clock_mu:
                        clock[0] += 1000
                        t_now = clock[0]
                    t0 = time.perf_counter()
                    g.serve_prepared(eng, preps[t], t_now, with_headers=True)
                    lat[t].append(time.perf_counter() - t0)

            # (warm the serving sets — a set's pinned staging and device buffers are allocated by its first calls — with a
            # short concurrent phase that is not timed)
            ths = [threading.Thread(target=pump, args=(t, 3)) for t in range(T)]
            [t.start() for t in ths]
            [t.join() for t in ths]
            lat = [[] for _ in range(T)]
            ths = [threading.Thread(target=pump, args=(t, K)) for t in range(T)]
            t0 = time.perf_counter()
            [t.start() for t in ths]
            [t.join() for t in ths]
            el = time.perf_counter() - t0
            allat = np.array(sum(lat, []))
            row[name] = {"ms_per_batch_sustained": el / (T * K) * 1e3, "requests_per_s": n * T * K / el,
                         "call_p50_ms": float(np.percentile(allat, 50) * 1e3)}
        now = clock[0]
    out["sizes"][str(n)] = row
out["host_threads"] = os.environ.get("RLI_THREADS", "auto: one per 1024 messages, at most 32")
print(json.dumps(out))
