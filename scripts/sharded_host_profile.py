"""Host-side cost of one sharded step (world 1, RCCL): where the microseconds between kernels go.
Prints per-segment host time averaged over the steps.  Needs a MI355X."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from limitador_amd import workloads as W  # noqa: E402
from limitador_amd.engine import Engine  # noqa: E402
from limitador_amd.sharded import ShardedEngine  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = 1 << 20
eng = Engine(capacity_cells=1 << 22, max_batch_hits=n, max_limits=16)
from limitador_amd.wire import LIMIT_ROW_DTYPE  # noqa: E402
lim = np.zeros(1, dtype=LIMIT_ROW_DTYPE)
lim[0] = (100, 60)
eng.set_limits(lim)
sh = ShardedEngine(eng, dist.group.WORLD, dev, n)
rng = np.random.default_rng(1)
hs = []
for i in range(4):
    h = W.uniform_batch(500_000, n, rng)
    hs.append(torch.from_numpy(h.view(np.int64).reshape(-1, 2).copy()).to(dev))
out = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]

T = {}


def seg(name, t0):
    t1 = time.perf_counter()
    T[name] = T.get(name, 0.0) + (t1 - t0)
    return t1


def step(i, now):
    slot = i & 1
    hits = hs[i & 3]
    cnt = sh._counts[slot]
    t = time.perf_counter()
    sorted_hits, perm = sh.local.partition(hits, 1, slot, cnt[0])
    t = seg("partition", t)
    dist.all_to_all_single(cnt[1], cnt[0])
    t = seg("a2a_counts", t)
    send, recv = cnt.tolist()
    t = seg("tolist", t)
    rh = sh._recv_hits[slot][:sum(recv)]
    dist.all_to_all_single(rh, sorted_hits, output_split_sizes=recv, input_split_sizes=send)
    t = seg("a2a_hits", t)
    rv = sh._recv_verdict[slot][:n]
    eng.submit_device(rh.data_ptr(), n, now, rv.data_ptr())
    t = seg("submit", t)
    sv = sh._sorted_verdict[slot][:n]
    dist.all_to_all_single(sv, rv, output_split_sizes=send, input_split_sizes=recv)
    t = seg("a2a_verdict", t)
    eng.unpermute_u8_device(sv.data_ptr(), perm.data_ptr(), n, out[slot].data_ptr())
    t = seg("unpermute", t)
    eng.collect()
    t = seg("collect(wait)", t)


for i in range(5):
    step(i, 1_000_000 + i)
T.clear()
K = 40
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    step(i, 2_000_000 + i)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / K * 1e6
for k, v in T.items():
    print(f"{k:16s} {v / K * 1e6:8.1f} us")
print(f"{'step':16s} {tot:8.1f} us")
dist.destroy_process_group()
