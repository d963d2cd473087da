"""Latency of ONE blocking rl_check_and_update_batch_device call as a function of the batch size (device
buffers, table of 1 M keys): what a micro-batching transport pays per batch.  Needs a MI355X."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from limitador_amd import workloads as W  # noqa: E402
from limitador_amd.engine import Engine  # noqa: E402

dev = torch.device("cuda", 0)
n_keys = 1_000_000
eng = Engine(capacity_cells=1 << 24, max_batch_hits=1 << 20)
eng.set_limits([(W.MAX_VALUE, W.WINDOW_S)])
eng.load_cells(W.universe_rows(n_keys))
rng = np.random.default_rng(W.SEED)
out = {}
now = W.NOW0_US
for n in (1, 256, 4096, 16384, 65536, 262144, 1 << 20):
    reps = 200 if n <= 65536 else 50
    batches = [torch.from_numpy(W.uniform_batch(n_keys, n, rng).view(np.int64).reshape(-1, 2).copy()).to(dev)
               for _ in range(8)]
    verdict = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for i in range(10):
        eng.check_and_update_device(batches[i & 7].data_ptr(), n, now, verdict.data_ptr())
        now += 1000
    t = []
    for i in range(reps):
        t0 = time.perf_counter()
        eng.check_and_update_device(batches[i & 7].data_ptr(), n, now, verdict.data_ptr())
        t.append(time.perf_counter() - t0)
        now += 1000
    t = np.sort(np.array(t)) * 1e6
    out[n] = {"p50_us": round(float(t[len(t) // 2]), 1), "p99_us": round(float(t[int(len(t) * 0.99)]), 1),
              "decisions_per_s_at_p50": round(n / (t[len(t) // 2] * 1e-6))}
print(json.dumps(out))
eng.close()


def per_request(tiny):
    """One request with 3 counters (1 simple + 2 qualified), the trait's per-request call."""
    os.environ["RL_TINY_MAX"] = "1024" if tiny else "0"
    os.environ["RL_GEN_TINY_MAX"] = "64" if tiny else "0"
    from limitador_amd.wire import HIT_DTYPE, RL_SIMPLE

    eng = Engine(capacity_cells=1 << 16, max_batch_hits=1 << 12)
    eng.set_limits([(10**9, 60), (10**9, 60), (10**9, 60)])
    eng.add_counter(0 | RL_SIMPLE, 7_000_001)
    res = {}
    for load in (False, True):
        t = []
        now_ = W.NOW0_US
        for i in range(300):
            h = np.zeros(3, dtype=HIT_DTYPE)
            h[0] = (7_000_001, 0 | RL_SIMPLE, 1)
            h[1] = (W.splitmix64(np.uint64(i % 50 + 1)), 1, 1)
            h[2] = (W.splitmix64(np.uint64(i % 7 + 100)), 2, 1)
            t0 = time.perf_counter()
            eng.check_and_update(h, now_, req_off=np.array([0, 3], dtype=np.uint32), load_counters=load)
            t.append(time.perf_counter() - t0)
            now_ += 1000
        t = np.sort(np.array(t[50:])) * 1e6
        res["load_counters" if load else "plain"] = {"p50_us": round(float(t[len(t) // 2]), 1),
                                                     "p99_us": round(float(t[int(len(t) * 0.99)]), 1)}
    eng.close()
    return res


print(json.dumps({"one request x 3 counters, host buffers": {"one_launch": per_request(True),
                                                              "general_pipeline": per_request(False)}}))
