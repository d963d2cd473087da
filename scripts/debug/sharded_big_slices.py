"""rl_sharded_check_requests_device at world 1 on growing slices from a cold engine: which sizes are refused, and why.
usage: python scripts/debug/sharded_big_slices.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from limitador_amd import sharded_abi  # noqa: E402
from limitador_amd import workloads as W  # noqa: E402
from limitador_amd.engine import Engine  # noqa: E402

dev = torch.device("cuda", 0)
k = 3
uid = sharded_abi.unique_id()
first = True
for n_req in (131072, 393216, 524288, 1048576):
    n = n_req * k
    eng = Engine(capacity_cells=1 << 23, max_batch_hits=n)
    eng.set_limits([(1000, 60), (200, 60), (50, 10)])
    sh = sharded_abi.Sharded(eng, 1, 0, n, unique_id=uid) if first else sharded_abi.Sharded(eng, 1, 0, n, unique_id=sharded_abi.unique_id())
    first = False
    rng = np.random.default_rng(W.SEED)
    users = (rng.zipf(1.2, size=n_req) - 1) % 500_000
    top = np.bincount(users).max()
    hits = np.zeros((n_req, k, 2), dtype=np.int64)
    for j in range(k):
        hits[:, j, 0] = W.splitmix64((users * 8 + j).astype(np.uint64)).view(np.int64) & 0x3FFFFFFFFFFFFFFF
        hits[:, j, 1] = j | (1 << 32)
    t = torch.from_numpy(hits.reshape(n, 2)).to(dev)
    off = torch.arange(0, n + 1, k, dtype=torch.int32, device=dev)
    v = torch.empty(n_req, dtype=torch.uint8, device=dev)
    f = torch.empty(n_req, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for step in range(3):
        try:
            r = sh.check_requests(t.data_ptr(), n, off.data_ptr(), n_req, W.NOW0_US + step, v.data_ptr(), False, f.data_ptr())
            print(n_req, "top key", top, "step", step, "rounds", r, "limited", int(v.sum().item()), flush=True)
        except sharded_abi.ShardedError as ex:
            print(n_req, "top key", top, "step", step, "REFUSED", ex, flush=True)
    sh.close()
    eng.close()
