#!/usr/bin/env python
"""rl_sharded_check_requests_device at world 1 (the library's own RCCL communicator): requests of three counters each, the
counters sharded by key (all of them here, on the one GPU), ms per step and rounds.  What it shows is the protocol's own
cost — exchanges, the phased resolver's blocking calls, one word per rank per round through the host — not scaling.
usage: python scripts/bench_sharded_requests.py [n_req] [steps] [rccl|local]
(local: the in-process transport at world 1 — the same kernels, the exchanges as the transport's own device copies, no RCCL
bring-up: what bench.py's `secondary` runs)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from limitador_amd import sharded_abi  # noqa: E402
from limitador_amd import workloads as W  # noqa: E402
from limitador_amd.engine import Engine  # noqa: E402

n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
transport = sys.argv[3] if len(sys.argv) > 3 else "rccl"
k = 3
dev = torch.device("cuda", 0)
n = n_req * k
eng = Engine(capacity_cells=1 << 23, max_batch_hits=n)
eng.set_limits([(1000, 60), (200, 60), (50, 10)])
group = sharded_abi.LocalGroup(1) if transport == "local" else None
sh = (sharded_abi.Sharded(eng, 1, 0, n, transport=group.transport(0)) if group else
      sharded_abi.Sharded(eng, 1, 0, n, unique_id=sharded_abi.unique_id()))
rng = np.random.default_rng(W.SEED)
batches = []
for _ in range(4):
    users = (rng.zipf(1.2, size=n_req) - 1) % 500_000
    hits = np.zeros((n_req, k, 2), dtype=np.int64)
    for j in range(k):
        hits[:, j, 0] = W.splitmix64((users * 8 + j).astype(np.uint64)).view(np.int64) & 0x3FFFFFFFFFFFFFFF
        hits[:, j, 1] = j | (1 << 32)  # limit j, delta 1
    batches.append(torch.from_numpy(hits.reshape(n, 2)).to(dev))
off = torch.arange(0, n + 1, k, dtype=torch.int32, device=dev)
v = torch.empty(n_req, dtype=torch.uint8, device=dev)
f = torch.empty(n_req, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
now = W.NOW0_US
rounds = []
for i in range(3):
    sh.check_requests(batches[i % 4].data_ptr(), n, off.data_ptr(), n_req, now, v.data_ptr(), False, f.data_ptr())
    now += 1000
t0 = time.perf_counter()
for i in range(steps):
    rounds.append(sh.check_requests(batches[i % 4].data_ptr(), n, off.data_ptr(), n_req, now, v.data_ptr(), False, f.data_ptr()))
    now += 1000
dt = time.perf_counter() - t0
print(json.dumps({"what": "rl_sharded_check_requests_device, world 1, " + ("in-process transport" if group else "RCCL"), "requests_per_step": n_req, "counters_per_step": n,
                  "ms_per_step": dt / steps * 1e3, "requests_per_s": n_req * steps / dt, "rounds": rounds,
                  "limited_in_last_step": int(v.sum().item())}))
sh.close()
eng.close()
if group:
    group.close()
