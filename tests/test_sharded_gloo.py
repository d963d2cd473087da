"""The N>1 data path (limitador_amd/sharded.py) with world_size 2 over gloo on CPU.

The routing choreography (stable partition by owner, counts / descriptor / verdict all-to-alls,
un-permute) is the product code; the three device-side operations of a shard are replaced by a
CPU stand-in backed by the oracle — the GPU versions of those three are covered by
tests/test_gpu_sharded.py against the same stand-in.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from limitador_amd import workloads as W
from limitador_amd.sharded import ShardedEngine, owner_of_tensor
from limitador_amd.wire import HIT_DTYPE

SEED = 0x9E3779B97F4A7C15
ROWS = [(30, 60), (5, 1)]


class OracleLocal:
    """CPU stand-in for HipLocal: same contract, numpy + oracle."""

    def __init__(self, world):
        self.orc = oracle.OracleStorage()
        self.orc.set_limits(ROWS)
        self.world = world

    def partition(self, hits, world, slot, counts_out):
        owners = owner_of_tensor(hits[:, 0], SEED, world).numpy()
        perm = np.argsort(owners, kind="stable")
        counts_out.copy_(torch.from_numpy(np.bincount(owners, minlength=world).astype(np.int32)))
        return hits[torch.from_numpy(perm)].contiguous(), torch.from_numpy(perm.astype(np.int32))

    def check(self, hits, n, now_us, verdict):
        if n == 0:
            return
        h = hits.numpy().view(HIT_DTYPE).reshape(-1)
        v, _, _, _ = self.orc.check_and_update(h, now_us)
        verdict.copy_(torch.from_numpy(v))

    def unpermute(self, src, perm, n, dst):
        dst[perm.long()] = src


def _slices(world, steps, n):
    rng = np.random.default_rng(5)
    out = []
    for _ in range(steps):
        step = []
        for _r in range(world):
            idx = (rng.zipf(1.3, size=n) - 1) % 97
            h = np.empty(n, dtype=HIT_DTYPE)
            h["key"] = W.splitmix64(idx.astype(np.uint64))
            h["limit"] = idx % 2
            h["delta"] = rng.integers(0, 3, size=n)
            step.append(h)
        out.append(step)
    return out


def _worker(rank, world, port, steps, n, q, depth=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        local = OracleLocal(world)
        sh = ShardedEngine(None, dist.group.WORLD, torch.device("cpu"), max_local_hits=n, local=local)
        data = _slices(world, steps, n)
        verdicts = []
        now = W.NOW0_US
        outs = []
        for s in range(steps):
            hits = torch.from_numpy(data[s][rank].view(np.int64).reshape(-1, 2).copy())
            out = torch.empty(n, dtype=torch.uint8)
            outs.append(out)
            if depth == 1:
                sh.check_and_update(hits, now, out)
            else:  # `depth` slices in flight
                sh.submit(hits, now, out)
                if sh.in_flight == depth:
                    sh.collect()
            now += 400_000
        while sh.in_flight:
            sh.collect()
        verdicts = [o.numpy().copy() for o in outs]
        # every key must live on exactly its owner
        for key in W.splitmix64(np.arange(97, dtype=np.uint64)):
            own = int(owner_of_tensor(torch.tensor([int(key)], dtype=torch.uint64).view(torch.int64), SEED, world)[0])
            assert (local.orc.peek(int(key)) is None) or own == rank
        q.put((rank, verdicts))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("depth", [1, 2, 3], ids=["blocking", "two_in_flight", "three_in_flight"])
def test_two_rank_sharded_path_matches_the_sequential_reference(depth):
    world, steps, n = 2, 6, 700
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, steps, n, q, depth)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # sequential reference: one storage, slices concatenated in rank order (the global trace order)
    orc = oracle.OracleStorage()
    orc.set_limits(ROWS)
    data = _slices(world, steps, n)
    now = W.NOW0_US
    denied = 0
    for s in range(steps):
        cat = np.concatenate([data[s][r] for r in range(world)])
        v, _, _, _ = orc.check_and_update(cat, now)
        for r in range(world):
            assert np.array_equal(got[r][s], v[r * n:(r + 1) * n]), f"step {s} rank {r}"
        denied += int(v.sum())
        now += 400_000
    assert 0 < denied < steps * world * n


def test_owner_of_tensor_matches_the_c_abi(engine_lib):
    keys = W.splitmix64(np.arange(5000, dtype=np.uint64))
    t = torch.from_numpy(keys.view(np.int64))
    for world in (1, 2, 3, 8):
        got = owner_of_tensor(t, SEED, world).numpy()
        want = np.array([engine_lib.rl_owner_of(int(k), SEED, world) for k in keys])
        assert np.array_equal(got, want)
