"""A key-sharded step refused on both ranks (rank 1's table is far too small for the cells the step creates), then the
process ends WITHOUT closing anything (or, with "close", closing everything): does anything hang?
usage: timeout -s ABRT 60 python -X faulthandler <this> [close]"""
import os
import sys
import threading

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from limitador_amd import sharded_abi  # noqa: E402
from limitador_amd import workloads as W  # noqa: E402
from limitador_amd.engine import Engine  # noqa: E402
from limitador_amd.wire import HIT_DTYPE  # noqa: E402

dev = torch.device("cuda", 0)
world, n_req = 2, 70_000
engines = []
for r in range(world):
    e = Engine(capacity_cells=1 << (18 if r == 0 else 9), max_batch_hits=1 << 18)
    e.set_limits([(100_000, 60), (3, 60)])
    engines.append(e)
group = sharded_abi.LocalGroup(world)
ranks = [sharded_abi.Sharded(engines[r], world, r, 1 << 18, transport=group.transport(r)) for r in range(world)]
h = np.zeros(2 * n_req, dtype=HIT_DTYPE)
h["key"][0::2] = 0x123456789ABC
h["key"][1::2] = W.splitmix64(np.arange(n_req, dtype=np.uint64) % np.uint64(40_000)) & np.uint64(0x3FFFFFFFFFFFFFFF)
h["limit"][1::2] = 1
h["delta"] = 1
off = torch.arange(0, 2 * n_req + 1, 2, dtype=torch.int32, device=dev)
t = torch.from_numpy(h.view(np.int64).reshape(-1, 2).copy()).to(dev)
out = {}


def run(r):
    v = torch.zeros(n_req, dtype=torch.uint8, device=dev)
    f = torch.zeros(n_req, dtype=torch.int32, device=dev)
    try:
        ranks[r].check_requests(t.data_ptr(), 2 * n_req, off.data_ptr(), n_req, W.NOW0_US, v.data_ptr(), False, f.data_ptr())
        out[r] = "applied"
    except sharded_abi.ShardedError as ex:
        out[r] = repr(ex)


th = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
for x in th:
    x.start()
for x in th:
    x.join(timeout=60)
print(out, flush=True)
if "close" in sys.argv[1:]:
    for s in ranks:
        s.close()
    print("ranks closed", flush=True)
    for e in engines:
        e.close()
    print("engines closed", flush=True)
    group.close()
print("leaving", flush=True)
