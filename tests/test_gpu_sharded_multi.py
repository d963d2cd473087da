"""ShardedMultiCounterEngine on the HIP engine (the phased general resolver, rl_gen_begin_device ..
rl_gen_commit_device): one, two and three ranks as threads of this process, one engine each (all on this GPU),
requests whose counters are spread over the ranks by key — against ONE sequential oracle on the concatenated slices.
Also: the driver over a torch.distributed group (RCCL, world 1), and a step one shard cannot take is refused on
every rank with nothing applied."""
import os
import socket
import threading

import numpy as np
import pytest
import torch

import oracle
from limitador_amd import workloads as W
from limitador_amd.engine import Engine
from limitador_amd.sharded import (HipGenLocal, InProcessGroup, ShardedMultiCounterEngine, ShardedTableFull, TorchTransport,
                                   owner_of_tensor)
from limitador_amd.wire import RL_SIMPLE
from test_sharded_multi_gloo import ROWS, SIMPLE, compare, expected, make_slices, run_rank

pytestmark = pytest.mark.gpu


def _engine(world, rank, seed, **kw):
    eng = Engine(capacity_cells=kw.pop("capacity_cells", 1 << 16), max_batch_hits=kw.pop("max_batch_hits", 1 << 16), **kw)
    eng.set_limits(ROWS)
    for limit, key in SIMPLE:
        if int(owner_of_tensor(torch.tensor([key]), seed, world)[0]) == rank:
            eng.add_counter(limit | RL_SIMPLE, key)
    return eng


def _check_tables(engines, orc):
    rows = np.concatenate([e.dump_cells() for e in engines])
    assert len(np.unique(rows["key"])) == len(rows)
    qual = rows[(rows["limit"] & RL_SIMPLE) == 0]
    assert len(qual) == orc.num_qualified()
    for r in qual:
        assert (int(r["value"]), int(r["expiry_us"]), int(r["limit"])) == orc.peek(int(r["key"]))
    for r in rows[(rows["limit"] & RL_SIMPLE) != 0]:
        assert (int(r["value"]), int(r["expiry_us"])) == orc.peek_simple(int(r["limit"]))


@pytest.mark.parametrize("world", [1, 2, 3])
def test_key_sharded_multi_counter_requests_on_the_hip_engine(world):
    dev = torch.device("cuda", 0)
    steps, n_req, load_steps = 6, 900, {1, 4}
    probe = Engine(capacity_cells=1 << 10, max_batch_hits=1 << 10)
    seed = probe.hash_seed
    probe.close()
    engines = [_engine(world, r, seed) for r in range(world)]
    group = InProcessGroup(world)
    data = make_slices(world, steps, n_req)
    got, errors = {}, []

    def run(r):
        try:
            sh = ShardedMultiCounterEngine(group.transport(r, dev), HipGenLocal(engines[r], dev), seed)
            got[r] = run_rank(sh, data, r, steps, load_steps, device=dev)
        except Exception as ex:
            errors.append((r, repr(ex)))

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=180)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads), "a rank is stuck at an exchange"
    want, orc = expected(world, steps, n_req, load_steps)
    assert compare(got, want, world, steps) >= 3
    _check_tables(engines, orc)
    # the engines are their own again: an ordinary call works
    for e in engines:
        assert e.stats()["live_cells"] > 0
        e.close()


def test_the_driver_over_rccl_world_1(rccl_ready):
    import torch.distributed as dist

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        steps, n_req, load_steps = 4, 700, {2}
        probe = Engine(capacity_cells=1 << 10, max_batch_hits=1 << 10)
        seed = probe.hash_seed
        probe.close()
        eng = _engine(1, 0, seed)
        sh = ShardedMultiCounterEngine(TorchTransport(dist.group.WORLD, dev), HipGenLocal(eng, dev), seed)
        got = {0: run_rank(sh, make_slices(1, steps, n_req), 0, steps, load_steps, device=dev)}
        want, orc = expected(1, steps, n_req, load_steps)
        compare(got, want, 1, steps)
        _check_tables([eng], orc)
        eng.close()
    finally:
        dist.destroy_process_group()


def test_a_step_one_shard_cannot_take_is_refused_everywhere():
    dev = torch.device("cuda", 0)
    world = 2
    probe = Engine(capacity_cells=1 << 10, max_batch_hits=1 << 10)
    seed = probe.hash_seed
    probe.close()
    engines = [_engine(world, 0, seed, capacity_cells=1 << 16), _engine(world, 1, seed, capacity_cells=1 << 9)]
    group = InProcessGroup(world)
    rng = np.random.default_rng(4)
    keys = rng.integers(1, 2**62, size=3000, dtype=np.int64)  # far more new cells than the small shard takes
    hits = np.zeros((3000, 2), dtype=np.int64)
    hits[:, 0] = keys
    hits[:, 1] = 4 | (1 << 32)  # limit 4, delta 1
    off = np.arange(0, 3001, 3, dtype=np.int64)
    results, errors = {}, []

    def run(r):
        try:
            sh = ShardedMultiCounterEngine(group.transport(r, dev), HipGenLocal(engines[r], dev), seed)
            t = torch.from_numpy(hits if r == 0 else hits[:0]).to(dev)
            o = torch.from_numpy(off if r == 0 else off[:1]).to(dev)
            try:
                sh.check(t, o, W.NOW0_US)
                results[r] = "applied"
            except ShardedTableFull:
                results[r] = "refused"
            # a step that fits still works afterwards
            small = torch.from_numpy(hits[:30] if r == 0 else hits[:0]).to(dev)
            so = torch.from_numpy(off[:11] if r == 0 else off[:1]).to(dev)
            v, _f, _r, _e = sh.check(small, so, W.NOW0_US + 1)
            results[(r, "after")] = int(v.sum().item())
        except Exception as ex:
            errors.append((r, repr(ex)))

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    assert results[0] == results[1] == "refused"
    assert results[(0, "after")] == 0
    assert sum(e.stats()["live_cells"] for e in engines) == 30 + len(SIMPLE)
    for e in engines:
        e.close()
