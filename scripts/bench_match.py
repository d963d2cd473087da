#!/usr/bin/env python
"""Time the on-device limit matcher (k_match, rl_match.hpp) and the whole match + check_and_update
call on a BASELINE configs[4]-shaped workload: 4 namespaces x 8 limits (2 simple + 6 qualified on 1-2
variables, ==/!= conditions on method / path), 4 descriptor entries per request, Zipf users.

Prints one JSON line; run it under `rocprofv3 --kernel-trace --stats` for the per-kernel durations
(k_match<false> = count pass, k_match<true> = fill pass).  Algorithmic bytes of the matcher per request:
namespace 4 + delta 4 + entry offsets 4 + entries 4 x 8 (read by both passes) + 16 per derived counter."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from limitador_amd.engine import Engine  # noqa: E402
from limitador_amd.wire import MATCH_COND_DTYPE, MATCH_LIMIT_DTYPE, RL_SIMPLE  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--requests", type=int, default=1_000_000)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--users", type=int, default=200_000)
ap.add_argument("--beside", choices=["none", "d2h", "h2d", "d2h_nocu"], default="none",
                help="a second thread copies 40 MB device -> pinned host (the runtime's blit kernel) or host -> device (SDMA) in a loop meanwhile")
args = ap.parse_args()

rng = np.random.default_rng(42)
K_METHOD, K_PATH, K_USER, K_APP = 0, 1, 2, 3
rows = np.zeros(32, dtype=MATCH_LIMIT_DTYPE)
conds, limit_rows = [], []
for ns in range(4):
    for j in range(8):
        i = ns * 8 + j
        nv = 0 if j < 2 else (1 if j < 5 else 2)
        rows[i]["limit"] = i | (RL_SIMPLE if nv == 0 else 0)
        rows[i]["ns"] = ns
        rows[i]["cond_off"] = len(conds)
        c = [(K_METHOD, j % 2, j % 3)] if j % 4 != 3 else [(K_METHOD, 0, j % 3), (K_PATH, 1, (j + 1) % 3)]
        rows[i]["n_cond"] = len(c)
        conds += c
        rows[i]["n_vars"] = nv
        rows[i]["var_key"] = (K_APP if nv == 2 else K_USER, K_USER if nv == 2 else 0)
        limit_rows.append((1000 if j else 10**9, [1, 10, 60, 3600][(ns + j) % 4]))
eng = Engine(capacity_cells=1 << 23, max_batch_hits=8 * args.requests, max_limits=64)
eng.set_limits(limit_rows)
eng.set_match_table(rows, np.array(conds, dtype=MATCH_COND_DTYPE), 4)
for i in range(32):
    if rows[i]["n_vars"] == 0:
        eng.add_counter(i | RL_SIMPLE, eng.match_key(i))

dev = torch.device("cuda", 0)
n = args.requests
batches = []
for _ in range(3):
    ns = rng.integers(0, 4, size=n).astype(np.uint32)
    users = ((rng.zipf(1.2, size=n) - 1) % args.users).astype(np.uint32) + 16
    ent_key = np.tile(np.array([K_METHOD, K_PATH, K_USER, K_APP], dtype=np.uint32), n)
    ent_val = np.stack([rng.integers(0, 3, size=n), rng.integers(0, 3, size=n), users, rng.integers(3, 8, size=n)],
                       axis=1).astype(np.uint32).reshape(-1)
    ent_off = (np.arange(n + 1, dtype=np.uint32) * 4)
    t = [torch.from_numpy(a.view(np.int32).copy()).to(dev) for a in (ns, ent_off, ent_key, ent_val, np.ones(n, dtype=np.uint32))]
    batches.append(t)
verdict = torch.empty(n, dtype=torch.uint8, device=dev)
limited = torch.empty(n, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
import ctypes as C

n_hits = C.c_uint32(0)


def step(i, now):
    b = batches[i % 3]
    eng._check(eng._lib.rl_match_and_check_batch_device(eng._h, b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(),
                                                         b[3].data_ptr(), b[4].data_ptr(), n, now, 0,
                                                         verdict.data_ptr(), limited.data_ptr(), C.byref(n_hits)))


now = 1_700_000_000_000_000
for i in range(2):
    step(i, now + i * 1000)
torch.cuda.synchronize()
stop, copies = [False], [0]
if args.beside != "none":
    import threading

    d_buf = torch.zeros(10 << 20, dtype=torch.float32, device=dev)
    h_buf = torch.empty(10 << 20, dtype=torch.float32).pin_memory()
    s_c = torch.cuda.Stream()

    hip = None
    if args.beside == "d2h_nocu":  # the same copy with kind hipMemcpyDeviceToDeviceNoCU, through the HIP runtime this process has mapped
        path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)
        hip = C.CDLL(path)
        hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]

    def pump():  # (twenty copies per synchronise: the thread sleeps in the runtime, without the GIL, nearly all the time)
        with torch.cuda.stream(s_c):
            while not stop[0]:
                for _ in range(20):
                    if hip is not None:
                        rc = hip.hipMemcpyAsync(h_buf.data_ptr(), d_buf.data_ptr(), h_buf.numel() * 4, 1024, s_c.cuda_stream)
                        assert rc == 0, rc
                    elif args.beside == "d2h":
                        h_buf.copy_(d_buf, non_blocking=True)
                    else:
                        d_buf.copy_(h_buf, non_blocking=True)
                s_c.synchronize()
                copies[0] += 20

    th = threading.Thread(target=pump)
    th.start()
    time.sleep(0.05)
c0 = copies[0]
t0 = time.perf_counter()
for i in range(args.steps):
    step(i, now + (i + 2) * 1000)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n_copies = copies[0] - c0
stop[0] = True
if args.beside != "none":
    th.join()
print(json.dumps({"what": "rl_match_and_check_batch_device (match + general check_and_update)", "requests_per_batch": n,
                  "counters_per_batch": n_hits.value, "steps": args.steps, "ms_per_step": dt / args.steps * 1e3,
                  "requests_per_s": n * args.steps / dt, "beside": args.beside, "copies_of_40MB_meanwhile": n_copies, "limited_in_last_batch": int(verdict.sum().item())}))
eng.close()
