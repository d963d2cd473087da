"""Key-sharded multi-counter requests (limitador_amd.sharded.ShardedMultiCounterEngine) with two gloo ranks on CPU:
a request's counters are spread over both ranks by key hash, the owners compute per-hit pass flags (stand-in for the
HIP engine's phased resolver: tests/helpers/gen_model.py), the ingress ranks AND them per request, round after
round, until the admitted set is the fixpoint.  The outcome must equal ONE sequential storage fed the concatenated
slices (rank 0's requests, then rank 1's): verdicts, first_limited, remaining / expires_in, and the union of the
two tables (in_memory.rs:72-156, all-or-nothing across a request's counters: 141-153)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from helpers.gen_model import ModelGenLocal
from limitador_amd import workloads as W
from limitador_amd.sharded import InProcessGroup, ShardedMultiCounterEngine, TorchTransport, owner_of_tensor
from limitador_amd.wire import HIT_DTYPE, RL_SIMPLE

SEED = 0x5EED0A11CE
ROWS = [(10**6, 60), (400, 10), (12, 1), (40, 10), (7, 60), (3, 1), (9, 0), (2**64 - 1, 3600), (25, 10)]
SIMPLE = [(0, 40_000_000), (1, 40_000_001)]
STEP_US = [400_000, 900_000, 11_000_000, 1, 700_000, 2_000_000]


def make_slices(world, steps, n_req, seed=9):
    """[step][rank] -> (hits HIT_DTYPE, req_off int64): 0-4 counters per request, simple counters first, Zipf
    users, now and then the same counter twice in one request and a request without counters."""
    rng = np.random.default_rng(seed)
    out = []
    for s in range(steps):
        per_rank = []
        for r in range(world):
            hits, off = [], [0]
            for _ in range(n_req - 13 * r - s):
                if rng.random() < 0.04:
                    off.append(len(hits))
                    continue
                delta = 1 if rng.random() < 0.8 else int(rng.integers(0, 4))
                user = int(rng.zipf(1.5) - 1) % 70 if rng.random() < 0.7 else int(rng.integers(0, 70))
                req = []
                for lid in (0, 1):
                    if rng.random() < 0.5:
                        req.append((SIMPLE[lid][1], lid | RL_SIMPLE, delta))
                for lid in rng.permutation(np.arange(2, len(ROWS)))[: int(rng.integers(0, 4))]:
                    key = int(W.splitmix64(np.array([int(lid) * 100_003 + user], dtype=np.uint64))[0])
                    req.append((key, int(lid), delta))
                if len(req) > 2 and rng.random() < 0.1:
                    req.append(req[-1])  # the same counter twice
                hits.extend(req)
                off.append(len(hits))
            arr = np.zeros(len(hits), dtype=HIT_DTYPE)
            for i, h in enumerate(hits):
                arr[i] = h
            per_rank.append((arr, np.array(off, dtype=np.int64)))
        out.append(per_rank)
    return out


def expected(world, steps, n_req, load_steps):
    orc = oracle.OracleStorage()
    orc.set_limits(ROWS)
    for limit, key in SIMPLE:
        orc.add_counter(limit | RL_SIMPLE, key)
    data = make_slices(world, steps, n_req)
    now = W.NOW0_US
    want = []
    for s in range(steps):
        hits = np.concatenate([data[s][r][0] for r in range(world)])
        off = [0]
        for r in range(world):
            off.extend((data[s][r][1][1:] + off[-1]).tolist())
        v, f, rem, exp = orc.check_and_update(hits, now, req_off=np.array(off, dtype=np.uint32), load_counters=s in load_steps)
        per_rank, lo_r, lo_h = [], 0, 0
        for r in range(world):
            nr, nh = len(data[s][r][1]) - 1, len(data[s][r][0])
            fr = f[lo_r:lo_r + nr].astype(np.int64)
            per_rank.append((v[lo_r:lo_r + nr], np.where(fr >= 0, fr - lo_h, -1), rem[lo_h:lo_h + nh] if s in load_steps else None,
                             exp[lo_h:lo_h + nh] if s in load_steps else None))
            lo_r, lo_h = lo_r + nr, lo_h + nh
        want.append(per_rank)
        now += STEP_US[s % len(STEP_US)]
    return want, orc


def run_rank(sh, data, rank, steps, load_steps, device="cpu"):
    now = W.NOW0_US
    outs = []
    for s in range(steps):
        h, off = data[s][rank]
        t = torch.from_numpy(h.view(np.int64).reshape(-1, 2).copy()).to(device)
        v, f, rem, exp = sh.check(t, torch.from_numpy(off).to(device), now, load_counters=s in load_steps)
        outs.append((v.cpu().numpy(), f.cpu().numpy(), None if rem is None else rem.cpu().numpy().view(np.uint64),
                     None if exp is None else exp.cpu().numpy().view(np.uint64), sh.rounds))
        now += STEP_US[s % len(STEP_US)]
    return outs


def compare(got, want, world, steps):
    deep = 0
    for s in range(steps):
        for r in range(world):
            v, f, rem, exp, rounds = got[r][s]
            wv, wf, wrem, wexp = want[s][r]
            assert np.array_equal(v, wv), f"step {s} rank {r}: verdicts"
            assert np.array_equal(f, wf), f"step {s} rank {r}: first_limited"
            if wrem is not None:
                assert np.array_equal(rem, wrem), f"step {s} rank {r}: remaining"
                assert np.array_equal(exp, wexp), f"step {s} rank {r}: expires_in"
            deep = max(deep, rounds)
    return deep


def _worker(rank, world, port, steps, n_req, load_steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        local = ModelGenLocal(ROWS, [(l, k) for l, k in SIMPLE if int(owner_of_tensor(torch.tensor([k]), SEED, world)[0]) == rank])
        sh = ShardedMultiCounterEngine(TorchTransport(dist.group.WORLD, "cpu"), local, SEED)
        outs = run_rank(sh, make_slices(world, steps, n_req), rank, steps, load_steps)
        q.put((rank, outs, {k: tuple(v) for k, v in local.table.items()}))
    finally:
        dist.destroy_process_group()


def test_two_gloo_ranks_key_sharded_multi_counter_requests_match_the_sequential_reference():
    world, steps, n_req, load_steps = 2, 6, 260, {1, 4}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, world, port, steps, n_req, load_steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = {r: outs for r, outs, _t in res}
    want, orc = expected(world, steps, n_req, load_steps)
    deep = compare(got, want, world, steps)
    assert deep >= 3, "the trace should need several fixpoint rounds"
    # the owners' tables are a partition of the sequential storage's cells
    tables = {}
    for r, _o, t in res:
        for k, cell in t.items():
            assert k not in tables
            assert int(owner_of_tensor(torch.tensor([k if k < 2**63 else k - 2**64]), SEED, world)[0]) == r
            tables[k] = cell
    n_simple = len(SIMPLE)
    assert len(tables) == orc.num_qualified() + n_simple
    for k, (value, expiry, limit) in tables.items():
        if limit & RL_SIMPLE:
            assert (value, expiry) == orc.peek_simple(limit)
        else:
            assert (value, expiry, limit) == orc.peek(k)


@pytest.mark.parametrize("world", [1, 3])
def test_in_process_ranks_with_the_model(world):
    """The same driver over the in-process transport (ranks = threads), one and three ranks."""
    import threading

    steps, n_req, load_steps = 4, 150, {2}
    group = InProcessGroup(world)
    data = make_slices(world, steps, n_req)
    got, errors = {}, []

    def run(r):
        try:
            local = ModelGenLocal(ROWS, [(l, k) for l, k in SIMPLE if int(owner_of_tensor(torch.tensor([k]), SEED, world)[0]) == r])
            sh = ShardedMultiCounterEngine(group.transport(r, "cpu"), local, SEED)
            got[r] = run_rank(sh, data, r, steps, load_steps)
        except Exception as ex:
            errors.append((r, repr(ex)))

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    want, _orc = expected(world, steps, n_req, load_steps)
    compare(got, want, world, steps)
