"""Device side of the sharded path on one MI355X: the stable owner partition / un-permute kernels
against numpy, and the whole ShardedEngine over RCCL with world_size 1."""
import os
import socket

import numpy as np
import pytest
import torch

import oracle
from limitador_amd import workloads as W
from limitador_amd.sharded import HipLocal, ShardedEngine, owner_of_tensor
from limitador_amd.wire import HIT_DTYPE

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [1, 2, 3, 8, 16])
@pytest.mark.parametrize("n", [1, 63, 2048, 100_003, 1_000_001])
def test_route_partition_is_a_stable_partition_by_owner(world, n):
    """The three kernels of the router's partition by owner against numpy's stable argsort; three calls in a row on one
    engine."""
    from limitador_amd.engine import Engine

    if n > 200_000 and world not in (1, 8):
        pytest.skip("the large batch on two world sizes only")
    eng = Engine(capacity_cells=1 << 12, max_batch_hits=max(1 << 17, n))
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(n * 31 + world)
    hits = W.uniform_batch(1 << 20, n, rng)
    hits["delta"] = np.arange(n) % 7
    t = torch.from_numpy(hits.view(np.int64).reshape(-1, 2).copy()).to(dev)
    loc = HipLocal(eng, dev, n, world)
    counts = torch.empty(world, dtype=torch.int32, device=dev)
    out, perm = loc.partition(t, world, 0, counts)
    torch.cuda.synchronize()
    owners = owner_of_tensor(t[:, 0].cpu(), eng.hash_seed, world).numpy()
    want_perm = np.argsort(owners, kind="stable")
    assert np.array_equal(perm.cpu().numpy(), want_perm)
    assert np.array_equal(counts.cpu().numpy(), np.bincount(owners, minlength=world))
    assert np.array_equal(out.cpu().numpy(), t.cpu().numpy()[want_perm])
    # un-permute returns per-hit bytes to ingress order
    src = torch.arange(n, dtype=torch.int64, device=dev).to(torch.uint8)
    dst = torch.zeros(n, dtype=torch.uint8, device=dev)
    loc.unpermute(src, perm, n, dst)
    torch.cuda.synchronize()
    want = np.zeros(n, dtype=np.uint8)
    want[want_perm] = (np.arange(n) % 256).astype(np.uint8)
    assert np.array_equal(dst.cpu().numpy(), want)
    for again in range(2):  # the same engine again, other hits
        hits2 = W.uniform_batch(1 << 20, n, rng)
        t2 = torch.from_numpy(hits2.view(np.int64).reshape(-1, 2).copy()).to(dev)
        out2, perm2 = loc.partition(t2, world, 0, counts)
        torch.cuda.synchronize()
        owners2 = owner_of_tensor(t2[:, 0].cpu(), eng.hash_seed, world).numpy()
        assert np.array_equal(perm2.cpu().numpy(), np.argsort(owners2, kind="stable"))
        assert np.array_equal(counts.cpu().numpy(), np.bincount(owners2, minlength=world))
    eng.close()


def test_sharded_engine_over_rccl_world_1(rccl_ready):
    import torch.distributed as dist
    from limitador_amd.engine import Engine

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n = 50_000
        eng = Engine(capacity_cells=1 << 16, max_batch_hits=2 * n)
        rows = [(20, 60)]
        eng.set_limits(rows)
        orc = oracle.OracleStorage()
        orc.set_limits(rows)
        sh = ShardedEngine(eng, dist.group.WORLD, dev, max_local_hits=n)
        rng = np.random.default_rng(3)
        now = W.NOW0_US
        for _ in range(3):
            hits = W.zipf_batch(5000, n, rng)
            t = torch.from_numpy(hits.view(np.int64).reshape(-1, 2).copy()).to(dev)
            out = torch.empty(n, dtype=torch.uint8, device=dev)
            sh.check_and_update(t, now, out)
            torch.cuda.synchronize()
            v, _, _, _ = orc.check_and_update(hits, now)
            assert np.array_equal(out.cpu().numpy(), v)
            now += 1000
        # three slices in flight (routed / applied / returned): same sequential result
        outs, want = [], []
        for step in range(7):
            hits = W.zipf_batch(5000, n - 17 * step, rng)
            t = torch.from_numpy(hits.view(np.int64).reshape(-1, 2).copy()).to(dev)
            outs.append((t, torch.empty(len(hits), dtype=torch.uint8, device=dev)))
            sh.submit(t, now, outs[-1][1])
            want.append(orc.check_and_update(hits, now)[0])
            if sh.in_flight == 3:
                assert sh.collect() == n - 17 * (step - 2)
            now += 1000
        with pytest.raises(RuntimeError):
            sh.check_and_update(t, now, outs[-1][1])  # not on a busy pipeline
        while sh.in_flight:
            sh.collect()
        torch.cuda.synchronize()
        for step in range(7):
            assert np.array_equal(outs[step][1].cpu().numpy(), want[step]), f"slice {step}"
        eng.close()
    finally:
        dist.destroy_process_group()


def test_namespace_sharded_requests_over_rccl_world_1(rccl_ready):
    """ShardedRequestEngine on the HIP engine (device matcher + general resolver behind RCCL with one
    rank) against the id-level CPU matcher + the oracle, multi-counter requests included."""
    import torch.distributed as dist
    from helpers.match_cpu import (Dictionary, compile_rows, limited_limit, match_key, match_requests,
                                   random_limits, random_requests)
    from limitador_amd.engine import Engine
    from limitador_amd.sharded import HipMatchLocal, ShardedRequestEngine
    from limitador_amd.wire import RL_SIMPLE

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        rng = np.random.default_rng(17)
        namespaces = [f"ns{i}" for i in range(5)]
        limits = random_limits(rng, namespaces)
        key_id, val_id = Dictionary(), Dictionary()
        rows, conds, ns_id = compile_rows(limits, key_id, val_id)
        eng = Engine(capacity_cells=1 << 16, max_batch_hits=1 << 16)
        eng.set_limits([(l.max_value, l.seconds) for l in limits])
        eng.set_match_table(rows, conds, len(ns_id.ids))
        orc = oracle.OracleStorage()
        orc.set_limits([(l.max_value, l.seconds) for l in limits])
        for i, l in enumerate(limits):
            if not l.variables:
                eng.add_counter(i | RL_SIMPLE, match_key(i, []))
                orc.add_counter(i | RL_SIMPLE, match_key(i, []))
        sh = ShardedRequestEngine(dist.group.WORLD, dev, HipMatchLocal(eng, dev))
        now = W.NOW0_US
        for step in range(4):
            _c, req_ns, ent_off, ent_key, ent_val, delta = random_requests(rng, 3000 + step, namespaces, ns_id, key_id, val_id)
            t = [torch.from_numpy(a.astype(np.int32)).to(dev) for a in (req_ns, ent_off, ent_key, ent_val, delta)]
            v, lim = sh.check(*t, now)
            torch.cuda.synchronize()
            hits, off = match_requests(rows, conds, req_ns, ent_off, ent_key, ent_val, delta)
            wv, wf, _r, _e = orc.check_and_update(hits, now, req_off=off)
            assert np.array_equal(v.cpu().numpy(), wv), f"step {step}"
            assert np.array_equal(lim.cpu().numpy(), limited_limit(wf, hits)), f"step {step}"
            now += 400_000
        eng.close()
    finally:
        dist.destroy_process_group()
