"""include/rl_sharded.h on one MI355X: the routed step behind the C ABI, (a) over the library's own RCCL
communicator with world 1, (b) with world 2 and 3 — ranks as threads of this process, each with its own engine,
exchanging through the in-process transport — against the sequential oracle on the concatenated slices."""
import threading

import numpy as np
import pytest
import torch

import oracle
from limitador_amd import sharded_abi
from limitador_amd import workloads as W
from limitador_amd.engine import Engine

pytestmark = pytest.mark.gpu

ROWS = [(20, 60), (3, 1)]


def _to_dev(hits, dev):
    return torch.from_numpy(hits.view(np.int64).reshape(-1, 2).copy()).to(dev)


def _slice(rng, n, n_keys):
    hits = W.zipf_batch(n_keys, n, rng)
    hits["limit"] = (hits["key"] % 2).astype(np.uint32)  # a key always comes with the same limit id
    hits["delta"] = 1 + (hits["key"] % 3 == 0)
    return hits


@pytest.mark.parametrize("engine_streams", ["external", "own"])
def test_routed_step_over_the_library_owned_rccl_communicator_world_1(engine_streams, monkeypatch):
    # "own": the engine keeps its two streams and is ordered by rl_engine_wait_event / rl_engine_record_event
    monkeypatch.setenv("RL_SHARDED_ENGINE_STREAMS", engine_streams)
    dev = torch.device("cuda", 0)
    n = 40_000
    eng = Engine(capacity_cells=1 << 16, max_batch_hits=2 * n)
    eng.set_limits(ROWS)
    orc = oracle.OracleStorage()
    orc.set_limits(ROWS)
    sh = sharded_abi.Sharded(eng, 1, 0, n, unique_id=sharded_abi.unique_id())
    rng = np.random.default_rng(5)
    now = W.NOW0_US
    for _ in range(3):  # blocking form
        hits = _slice(rng, n, 6000)
        t, out = _to_dev(hits, dev), torch.empty(n, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        assert sh.check_and_update(t.data_ptr(), n, now, out.data_ptr()) == n
        assert np.array_equal(out.cpu().numpy(), orc.check_and_update(hits, now)[0])
        now += 400_000  # the 1-second windows run out between slices
    outs, want = [], []
    for step in range(8):  # three slices in flight
        m = n - 131 * step
        hits = _slice(rng, m, 6000)
        t, out = _to_dev(hits, dev), torch.empty(m, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        outs.append((t, out))
        sh.submit(t.data_ptr(), m, now, out.data_ptr())
        want.append(orc.check_and_update(hits, now)[0])
        if sh.in_flight == 3:
            assert sh.collect() == n - 131 * (step - 2)
        now += 300_000
    with pytest.raises(sharded_abi.ShardedError):
        sh.check_and_update(t.data_ptr(), m, now, out.data_ptr())  # not on a busy pipeline
    while sh.in_flight:
        sh.collect()
    sh.sync()
    for step in range(8):
        assert np.array_equal(outs[step][1].cpu().numpy(), want[step]), f"slice {step}"
    sh.close()
    # the engine is its own again
    v, _, _, _ = eng.check_and_update(hits[:100], now)
    assert np.array_equal(v, orc.check_and_update(hits[:100], now)[0])
    eng.close()


@pytest.mark.parametrize("world", [2, 3])
def test_routed_step_world_n_with_the_in_process_transport(world):
    """Every rank a thread with its own engine (all on this GPU).  The owners' tables are disjoint, the
    verdicts must equal ONE sequential storage fed rank 0's slice, then rank 1's, ... per step."""
    dev = torch.device("cuda", 0)
    n, steps, n_keys = 30_000, 7, 5000
    group = sharded_abi.LocalGroup(world)
    engines = [Engine(capacity_cells=1 << 16, max_batch_hits=world * n) for _ in range(world)]
    for e in engines:
        e.set_limits(ROWS)
    ranks = [sharded_abi.Sharded(engines[r], world, r, n, transport=group.transport(r)) for r in range(world)]
    orc = oracle.OracleStorage()
    orc.set_limits(ROWS)
    rng = np.random.default_rng(11 + world)
    slices = [[_slice(rng, n - 97 * s - 13 * r, n_keys) for r in range(world)] for s in range(steps)]
    if world == 3:
        slices[2][1] = slices[2][1][:0]  # an empty ingress slice on one rank
    now0 = W.NOW0_US
    want = []
    for s in range(steps):
        now = now0 + 350_000 * s
        for r in range(world):
            want.append(orc.check_and_update(slices[s][r], now)[0] if len(slices[s][r]) else np.zeros(0, np.uint8))
    dev_in = [[_to_dev(slices[s][r], dev) if len(slices[s][r]) else torch.empty((0, 2), dtype=torch.int64, device=dev)
               for r in range(world)] for s in range(steps)]
    dev_out = [[torch.full((len(slices[s][r]),), 7, dtype=torch.uint8, device=dev) for r in range(world)] for s in range(steps)]
    torch.cuda.synchronize()
    applied = [[] for _ in range(world)]
    errors = []

    def run(r):
        try:
            sh = ranks[r]
            for s in range(steps):
                sh.submit(dev_in[s][r].data_ptr(), len(slices[s][r]), now0 + 350_000 * s, dev_out[s][r].data_ptr())
                if sh.in_flight == 3:
                    applied[r].append(sh.collect())
            while sh.in_flight:
                applied[r].append(sh.collect())
            sh.sync()
        except Exception as ex:  # a rank that dies leaves the others at the rendezvous: report, the join times out
            errors.append((r, ex))

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads), "a rank is stuck at the rendezvous"
    torch.cuda.synchronize()
    for s in range(steps):
        for r in range(world):
            assert np.array_equal(dev_out[s][r].cpu().numpy(), want[s * world + r]), f"step {s} rank {r}"
        # every hit of the step was applied by exactly one owner
        assert sum(applied[r][s] for r in range(world)) == sum(len(slices[s][r]) for r in range(world))
    # the owners' tables are a partition of the oracle's cells
    rows = np.concatenate([e.dump_cells() for e in engines])
    assert len(rows) == orc.num_qualified()
    assert len(np.unique(rows["key"])) == len(rows)
    for row in rows[::53]:
        assert (int(row["value"]), int(row["expiry_us"]), int(row["limit"])) == orc.peek(int(row["key"]))
    for sh in ranks:
        sh.close()
    for e in engines:
        e.close()
    group.close()


def test_a_slice_that_fails_on_one_rank_is_a_collective_outcome_not_a_hang():
    """a rank-local error used to leave the collective sequence (the failing rank returned before its
    exchanges, its peers waited for ever) and the slice stuck in flight.  Now: rank 1's engine takes 12 000 hits per
    batch; slice 1 routes more than that to it.  Every exchange of the slice is still issued, rank 1 answers 0xFF for
    the hits it owns and gets RL_ERR_BATCH_TOO_LARGE from THAT slice's collect, rank 0 gets a normal collect; the slices
    before and after are decided like one sequential storage that never saw the failed slice's hits on rank 1's keys."""
    dev = torch.device("cuda", 0)
    world, n, n_keys = 2, 10_000, 4000
    group = sharded_abi.LocalGroup(world)
    engines = [Engine(capacity_cells=1 << 16, max_batch_hits=(world * n if r == 0 else 12_000)) for r in range(world)]
    for e in engines:
        e.set_limits(ROWS)
    ranks = [sharded_abi.Sharded(engines[r], world, r, n, transport=group.transport(r)) for r in range(world)]
    rng = np.random.default_rng(21)
    steps = 4
    slices = [[_slice(rng, n, n_keys) for r in range(world)] for s in range(steps)]
    # slice 1: keys owned by rank 1 only, on both ingress ranks -> 20 000 hits routed to an engine that takes 12 000
    pool = W.splitmix64(np.arange(1, 60_000, dtype=np.uint64))
    own1 = np.array([k for k in pool if engines[0].owner_of(k, world) == 1][:3000], dtype=np.uint64)
    for r in range(world):
        h = slices[1][r]
        h["key"] = own1[rng.integers(0, len(own1), size=n)]
        h["limit"] = (h["key"] % 2).astype(np.uint32)
    owner = lambda hits: np.array([engines[0].owner_of(k, world) for k in hits["key"][:200]])  # noqa: E731
    orc = oracle.OracleStorage()
    orc.set_limits(ROWS)
    now0 = W.NOW0_US
    want = {}
    for s in range(steps):
        for r in range(world):
            if s == 1:
                want[(s, r)] = np.full(n, 0xFF, dtype=np.uint8)  # every hit of slice 1 is rank 1's: nothing applied
            else:
                want[(s, r)] = orc.check_and_update(slices[s][r], now0 + 1000 * s)[0]
    dev_in = [[_to_dev(slices[s][r], dev) for r in range(world)] for s in range(steps)]
    dev_out = [[torch.full((n,), 7, dtype=torch.uint8, device=dev) for r in range(world)] for s in range(steps)]
    torch.cuda.synchronize()
    outcomes = [[] for _ in range(world)]
    errors = []

    def run(r):
        try:
            sh = ranks[r]
            for s in range(steps):
                sh.submit(dev_in[s][r].data_ptr(), n, now0 + 1000 * s, dev_out[s][r].data_ptr())
                if sh.in_flight == 3:
                    outcomes[r].append(_collect(sh))
            while sh.in_flight:
                outcomes[r].append(_collect(sh))
            sh.sync()
        except Exception as ex:
            errors.append((r, ex))

    def _collect(sh):
        try:
            return ("ok", sh.collect())
        except sharded_abi.ShardedError as ex:
            return ("err", ex.code)

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads), "a rank is stuck at the rendezvous"
    torch.cuda.synchronize()
    assert [o[0] for o in outcomes[0]] == ["ok"] * steps
    assert [o[0] for o in outcomes[1]] == ["ok", "err", "ok", "ok"] and outcomes[1][1][1] == -7  # RL_ERR_BATCH_TOO_LARGE
    assert owner(slices[1][0]).min() == 1
    for s in range(steps):
        for r in range(world):
            assert np.array_equal(dev_out[s][r].cpu().numpy(), want[(s, r)]), f"slice {s} rank {r}"
    for sh in ranks:
        sh.close()
    for e in engines:
        e.close()
    group.close()
