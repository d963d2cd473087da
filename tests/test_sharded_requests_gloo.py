"""Namespace-sharded multi-counter requests (limitador_amd.sharded.ShardedRequestEngine) with two gloo
ranks on CPU: requests travel to their namespace's owner, are matched and decided there (stand-in:
the id-level CPU matcher + the oracle's general check_and_update), results travel back.  The outcome
must equal ONE sequential storage fed the concatenated slices (rank 0's, then rank 1's) — all-or-nothing
across each request's counters included (in_memory.rs:141-153)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from helpers.match_cpu import (Dictionary, compile_rows, limited_limit, match_requests, random_limits,
                               random_requests)
from limitador_amd import workloads as W
from limitador_amd.sharded import ShardedRequestEngine, namespace_owner
from limitador_amd.wire import RL_SIMPLE

NAMESPACES = [f"ns{i}" for i in range(7)]


def _setup(seed=5):
    rng = np.random.default_rng(seed)
    limits = random_limits(rng, NAMESPACES, per_ns=5)
    key_id, val_id = Dictionary(), Dictionary()
    rows, conds, ns_id = compile_rows(limits, key_id, val_id)
    return limits, rows, conds, ns_id, key_id, val_id


def _storage(limits, rows):
    orc = oracle.OracleStorage()
    orc.set_limits([(l.max_value, l.seconds) for l in limits])
    for i, l in enumerate(limits):
        if not l.variables:
            from helpers.match_cpu import match_key

            orc.add_counter(i | RL_SIMPLE, match_key(i, []))
    return orc


class OracleMatchLocal:
    def __init__(self, limits, rows, conds):
        self.rows, self.conds = rows, conds
        self.orc = _storage(limits, rows)
        self.seen_ns = set()

    def match_and_check(self, ns, ent_off, ent_key, ent_val, delta, now_us, verdict, limited):
        n = ns.shape[0]
        if not n:
            return
        self.seen_ns.update(int(x) for x in ns.tolist())
        hits, off = match_requests(self.rows, self.conds, ns.numpy(), ent_off.numpy(), ent_key.numpy(),
                                   ent_val.numpy(), delta.numpy())
        v, f, _r, _e = self.orc.check_and_update(hits, now_us, req_off=off)
        verdict.copy_(torch.from_numpy(v))
        limited.copy_(torch.from_numpy(limited_limit(f, hits)))


def _slices(world, steps, n, ns_id, key_id, val_id):
    rng = np.random.default_rng(99)
    return [[random_requests(rng, n - 13 * r, NAMESPACES, ns_id, key_id, val_id)[1:] for r in range(world)]
            for _ in range(steps)]


def _worker(rank, world, port, steps, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        limits, rows, conds, ns_id, key_id, val_id = _setup()
        local = OracleMatchLocal(limits, rows, conds)
        sh = ShardedRequestEngine(dist.group.WORLD, torch.device("cpu"), local)
        data = _slices(world, steps, n, ns_id, key_id, val_id)
        now = W.NOW0_US
        outs = []
        for s in range(steps):
            t = [torch.from_numpy(a.astype(np.int32)) for a in data[s][rank]]
            v, lim = sh.check(t[0], t[1], t[2], t[3], t[4], now)
            outs.append((v.numpy().copy(), lim.numpy().copy()))
            now += 300_000
        # a rank only ever decides requests of the namespaces it owns
        for x in local.seen_ns:
            assert int(namespace_owner(torch.tensor([x]), world)[0]) == rank
        q.put((rank, outs))
    finally:
        dist.destroy_process_group()


def test_two_rank_namespace_sharded_requests_match_the_sequential_reference():
    world, steps, n = 2, 5, 400
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, world, port, steps, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    limits, rows, conds, ns_id, key_id, val_id = _setup()
    orc = _storage(limits, rows)
    data = _slices(world, steps, n, ns_id, key_id, val_id)
    now = W.NOW0_US
    limited_total = multi = 0
    for s in range(steps):
        cat = [np.concatenate([data[s][r][k] for r in range(world)]) for k in (0, 2, 3, 4)]
        off = [0]
        for r in range(world):
            o = data[s][r][1]
            off.extend((o[1:].astype(np.int64) + off[-1] - int(o[0])).tolist())
        hits, req_off = match_requests(rows, conds, cat[0], np.array(off), cat[1], cat[2], cat[3])
        multi += int((np.diff(req_off) > 1).sum())
        v, f, _r, _e = orc.check_and_update(hits, now, req_off=req_off)
        lim = limited_limit(f, hits)
        lo = 0
        for r in range(world):
            k = len(data[s][r][0])
            assert np.array_equal(got[r][s][0], v[lo:lo + k]), f"step {s} rank {r}: verdicts"
            assert np.array_equal(got[r][s][1], lim[lo:lo + k]), f"step {s} rank {r}: limited limit"
            lo += k
        limited_total += int(v.sum())
        now += 300_000
    assert limited_total > 0 and multi > 100


def test_empty_and_entryless_slices_single_rank():
    """Edge cases of the request router in one process (gloo, world 1): an empty slice, requests
    without descriptor entries, and a slice whose requests all miss every limit."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        limits, rows, conds, ns_id, key_id, val_id = _setup()
        local = OracleMatchLocal(limits, rows, conds)
        sh = ShardedRequestEngine(dist.group.WORLD, torch.device("cpu"), local)
        i32 = lambda a: torch.tensor(a, dtype=torch.int32)  # noqa: E731
        v, lim = sh.check(i32([]), i32([0]), i32([]), i32([]), i32([]), W.NOW0_US)
        assert v.numel() == 0 and lim.numel() == 0
        # three requests, none carries an entry: only condition-less simple limits can apply
        ns = [ns_id(NAMESPACES[0]), ns_id(NAMESPACES[1]), ns_id(NAMESPACES[0])]
        v, lim = sh.check(i32(ns), i32([0, 0, 0, 0]), i32([]), i32([]), i32([1, 1, 1]), W.NOW0_US)
        hits, off = match_requests(rows, conds, np.array(ns), np.array([0, 0, 0, 0]), np.array([]), np.array([]),
                                   np.array([1, 1, 1]))
        ref = _storage(limits, rows)
        wv, wf, _r, _e = ref.check_and_update(hits, W.NOW0_US, req_off=off)
        assert np.array_equal(v.numpy(), wv) and np.array_equal(lim.numpy(), limited_limit(wf, hits))
        # a namespace id no limit knows: nothing applies, every request passes
        v, lim = sh.check(i32([len(ns_id.ids)] * 4), i32([0, 1, 2, 3, 4]), i32([0, 0, 0, 0]), i32([1, 1, 1, 1]),
                          i32([1, 1, 1, 1]), W.NOW0_US)
        assert not v.any() and (lim == -1).all()
    finally:
        dist.destroy_process_group()
