"""include/rl_sharded.h on one MI355X: the routed step behind the C ABI, (a) over the library's own RCCL
communicator with world 1, (b) with world 2 and 3 — ranks as threads of this process, each with its own engine,
exchanging through the in-process transport — against the sequential oracle on the concatenated slices.  Single-counter
slices (pipelined) and multi-counter requests whose counters live on several ranks (rl_sharded_check_requests_device)."""
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
def test_routed_step_over_the_library_owned_rccl_communicator_world_1(engine_streams, monkeypatch, rccl_ready):
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


@pytest.mark.parametrize("world", [2, 3, 8])
def test_routed_step_world_n_with_the_in_process_transport(world):
    """Every rank a thread with its own engine (all on this GPU).  The owners' tables are disjoint, the
    verdicts must equal ONE sequential storage fed rank 0's slice, then rank 1's, ... per step.
    World 8 is bench.py --gpus 8's shape scaled down (8 engines, 16 k keys each, 16 k-hit ingress slices, three slices in
    flight) with two skewed steps: every rank's slice made of keys ONE rank owns, so that rank receives 8 x its own slice
    (the receive buffers and the engine's max_batch_hits are sized for exactly that) while seven ranks apply nothing."""
    dev = torch.device("cuda", 0)
    n, steps, n_keys = (30_000, 7, 5000) if world < 8 else (16_000, 8, 128_000)
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
    if world == 8:
        pool = W.splitmix64(np.arange(1, 200_000, dtype=np.uint64))
        for s_skew, owner in ((3, 5), (4, 0)):  # (two skewed slices in flight at once, different owners)
            own = np.array([k for k in pool[:40_000] if engines[0].owner_of(int(k), world) == owner][:2500], dtype=np.uint64)
            for r in range(world):
                h = slices[s_skew][r]
                h["key"] = own[rng.integers(0, len(own), size=len(h))]
                h["limit"] = (h["key"] % 2).astype(np.uint32)
                h["delta"] = 1 + (h["key"] % 3 == 0)
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
                if sh.in_flight == (sharded_abi.MAX_IN_FLIGHT if world != 3 else 3):  # (the full window of four, and a shallower one)
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
    if world == 8:  # the skewed steps: ONE rank applied all eight slices, the others nothing
        assert applied[5][3] == sum(len(slices[3][r]) for r in range(world)) and applied[0][3] == 0
        assert applied[0][4] == sum(len(slices[4][r]) for r in range(world)) and applied[5][4] == 0
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


# ---- multi-counter requests, counters sharded by key: rl_sharded_check_requests_device --------------------------------
class _RequestsOverTheAbi:
    """What tests/test_sharded_multi_gloo.py's run_rank drives: check(hits [n, 2] int64, req_off int64) -> verdict,
    first_limited, remaining, expires_in (torch tensors) and .rounds — here through the C ABI."""

    def __init__(self, sh, dev):
        self.sh, self.dev, self.rounds = sh, dev, 0

    def check(self, hits, req_off, now_us, load_counters=False):
        n, n_req = int(hits.shape[0]), int(req_off.shape[0]) - 1
        off = req_off.to(torch.int32).contiguous()  # (bit pattern of the u32 offsets)
        v = torch.full((max(n_req, 1),), 9, dtype=torch.uint8, device=self.dev)
        f = torch.full((max(n_req, 1),), -7, dtype=torch.int32, device=self.dev)
        rem = torch.zeros(max(n, 1), dtype=torch.int64, device=self.dev) if load_counters else None
        exp = torch.zeros(max(n, 1), dtype=torch.int64, device=self.dev) if load_counters else None
        torch.cuda.synchronize()
        self.rounds = self.sh.check_requests(hits.data_ptr() if n else None, n, off.data_ptr(), n_req, now_us, v.data_ptr(),
                                             load_counters, f.data_ptr(), rem.data_ptr() if load_counters else None,
                                             exp.data_ptr() if load_counters else None)
        return v[:n_req], f[:n_req].to(torch.int64), None if rem is None else rem[:n], None if exp is None else exp[:n]


def _multi_engine(world, rank, seed, **kw):
    from limitador_amd.sharded import owner_of_tensor
    from limitador_amd.wire import RL_SIMPLE
    from test_sharded_multi_gloo import ROWS as MROWS, SIMPLE

    eng = Engine(capacity_cells=kw.pop("capacity_cells", 1 << 16), max_batch_hits=kw.pop("max_batch_hits", 1 << 16), **kw)
    eng.set_limits(MROWS)
    for limit, key in SIMPLE:
        if int(owner_of_tensor(torch.tensor([key]), seed, world)[0]) == rank:
            eng.add_counter(limit | RL_SIMPLE, key)
    return eng


def _check_union_of_tables(engines, orc):
    from limitador_amd.wire import RL_SIMPLE

    rows = np.concatenate([e.dump_cells() for e in engines])
    assert len(np.unique(rows["key"])) == len(rows)
    qual = rows[(rows["limit"] & RL_SIMPLE) == 0]
    assert len(qual) == orc.num_qualified()
    for r in qual:
        assert (int(r["value"]), int(r["expiry_us"]), int(r["limit"])) == orc.peek(int(r["key"]))
    for r in rows[(rows["limit"] & RL_SIMPLE) != 0]:
        assert (int(r["value"]), int(r["expiry_us"])) == orc.peek_simple(int(r["limit"]))


@pytest.mark.parametrize("world", [1, 2, 3])
def test_key_sharded_multi_counter_requests_behind_the_c_abi(world, request):
    """in_memory.rs:141-153 across GPUs: a request is admitted iff all its counters — on whatever rank their keys hash
    to — take it.  World 1 over the library's own RCCL communicator; world 2 and 3 as threads with the in-process
    transport.  Against ONE sequential oracle on the concatenated slices: verdicts, first_limited, remaining /
    expires_in, the union of the tables; at least one step needs three rounds."""
    from test_sharded_multi_gloo import compare, expected, make_slices, run_rank

    if world == 1:
        request.getfixturevalue("rccl_ready")  # (world 1 runs over the library's own RCCL communicator)
    dev = torch.device("cuda", 0)
    steps, n_req, load_steps = 6, 900, {1, 4}
    probe = Engine(capacity_cells=1 << 10, max_batch_hits=1 << 10)
    seed = probe.hash_seed
    probe.close()
    engines = [_multi_engine(world, r, seed) for r in range(world)]
    group = sharded_abi.LocalGroup(world) if world > 1 else None
    uid = sharded_abi.unique_id() if world == 1 else None
    ranks = [sharded_abi.Sharded(engines[r], world, r, 8192, unique_id=uid, transport=group.transport(r) if group else None)
             for r in range(world)]
    data = make_slices(world, steps, n_req)
    got, errors = {}, []

    def run(r):
        try:
            got[r] = run_rank(_RequestsOverTheAbi(ranks[r], dev), data, r, steps, load_steps, device=dev)
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
    _check_union_of_tables(engines, orc)
    # single-counter slices still go through the same communicator afterwards
    rng = np.random.default_rng(3)
    for r in range(world if world == 1 else 0):
        hits = W.zipf_batch(500, 2000, rng)
        hits["limit"] = 2
        t, out = _to_dev(hits, dev), torch.empty(2000, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        assert ranks[r].check_and_update(t.data_ptr(), 2000, W.NOW0_US + 10**9, out.data_ptr()) == 2000
        assert np.array_equal(out.cpu().numpy(), orc.check_and_update(hits, W.NOW0_US + 10**9)[0])
    for sh in ranks:
        sh.close()
    for e in engines:
        assert e.stats()["live_cells"] > 0  # the engines are their own again
        e.close()
    if group:
        group.close()


def test_a_rank_that_owns_none_of_the_steps_counters_still_takes_every_turn_with_the_others():
    """Every counter of these steps hashes to rank 0: rank 1 routes its requests' hits away and its own owner side has NOTHING
    (a phased pass of no hits).  It still has to go through the step's exchanges, offer its veto word and close its pass on
    the same turn as the owner — also when the group of rounds was too short and another one follows (the limits are tight
    enough for three rounds), and with load_counters.  Against one sequential oracle on the concatenated slices."""
    import oracle
    from limitador_amd.sharded import owner_of_tensor
    from limitador_amd.wire import HIT_DTYPE

    dev = torch.device("cuda", 0)
    world, n_req = 2, 600
    rows = [(40, 60), (7, 60), (3, 10)]
    engines = []
    for _ in range(world):
        e = Engine(capacity_cells=1 << 14, max_batch_hits=1 << 14)
        e.set_limits(rows)
        engines.append(e)
    seed = engines[0].hash_seed
    rng = np.random.default_rng(21)
    cand = rng.integers(1, 2**62, size=4000, dtype=np.int64)
    mine = cand[owner_of_tensor(torch.from_numpy(cand), seed, world).numpy() == 0][:90]  # keys of rank 0 only
    assert len(mine) == 90
    group = sharded_abi.LocalGroup(world)
    ranks = [sharded_abi.Sharded(engines[r], world, r, 1 << 13, transport=group.transport(r)) for r in range(world)]

    def a_slice():
        k = rng.integers(1, 4, size=n_req)
        off = np.concatenate([[0], np.cumsum(k)])
        h = np.zeros(int(off[-1]), dtype=HIT_DTYPE)
        h["key"] = mine[rng.integers(0, 30, size=len(h))]
        h["delta"] = 1
        for i in range(n_req):  # counter j of a request is on limit j (a key belongs to one limit: 30 keys per limit)
            for j in range(off[i], off[i + 1]):
                lim = j - off[i]
                h["limit"][j] = lim
                h["key"][j] = mine[30 * lim + int(rng.integers(0, 30))]
        return h, off

    steps, load_steps = 4, {1, 3}
    data = [[a_slice() for _ in range(world)] for _ in range(steps)]
    got, errors = {}, []

    def run(r):
        try:
            sh = _RequestsOverTheAbi(ranks[r], dev)
            outs = []
            for s in range(steps):
                h, off = data[s][r]
                t = torch.from_numpy(h.view(np.int64).reshape(-1, 2).copy()).to(dev)
                v, f, rem, exp = sh.check(t, torch.from_numpy(off).to(dev), W.NOW0_US + 1000 * s, load_counters=s in load_steps)
                outs.append((v.cpu().numpy(), f.cpu().numpy(), None if rem is None else rem.cpu().numpy().view(np.uint64),
                             None if exp is None else exp.cpu().numpy().view(np.uint64), sh.rounds))
            got[r] = outs
        except Exception as ex:
            errors.append((r, repr(ex)))

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads), "a rank is stuck at an exchange"
    orc = oracle.OracleStorage()
    orc.set_limits(rows)
    deepest = 0
    for s in range(steps):
        hits = np.concatenate([data[s][r][0] for r in range(world)])
        off, lo_h = [0], [0]
        for r in range(world):
            off.extend((data[s][r][1][1:] + off[-1]).tolist())
            lo_h.append(lo_h[-1] + len(data[s][r][0]))
        v, f, rem, exp = orc.check_and_update(hits, W.NOW0_US + 1000 * s, req_off=np.array(off, dtype=np.uint32), load_counters=s in load_steps)
        for r in range(world):
            fr = f[r * n_req:(r + 1) * n_req].astype(np.int64)
            gv, gf, grem, gexp, rounds = got[r][s]
            assert np.array_equal(gv, v[r * n_req:(r + 1) * n_req]), f"step {s} rank {r}: verdicts"
            assert np.array_equal(gf, np.where(fr >= 0, fr - lo_h[r], -1)), f"step {s} rank {r}: first_limited"
            if s in load_steps:
                assert np.array_equal(grem, rem[lo_h[r]:lo_h[r + 1]]), f"step {s} rank {r}: remaining"
                assert np.array_equal(gexp, exp[lo_h[r]:lo_h[r + 1]]), f"step {s} rank {r}: expires_in"
            deepest = max(deepest, rounds)
    assert deepest >= 3
    assert engines[1].stats()["live_cells"] == 0 and engines[0].stats()["live_cells"] == orc.num_qualified()
    for sh in ranks:
        sh.close()
    for e in engines:
        e.close()
    group.close()


def test_a_multi_counter_step_whose_owner_side_fails_on_one_rank_is_refused_on_every_rank():
    """Rank 1's engine takes 512 hits per pass; the step sends it more.  Its rl_gen_begin_device answers RL_ERR_BATCH_TOO_LARGE
    on the HOST, before anything is enqueued — the rank keeps taking part in the step's exchanges, its veto word says "error",
    and both ranks leave the step together with nothing applied; a step that fits is served right after."""
    from limitador_amd.sharded import owner_of_tensor
    from limitador_amd.wire import HIT_DTYPE

    dev = torch.device("cuda", 0)
    world = 2
    rows = [(1000, 60)]
    engines = [Engine(capacity_cells=1 << 14, max_batch_hits=1 << 13), Engine(capacity_cells=1 << 14, max_batch_hits=512)]
    for e in engines:
        e.set_limits(rows)
    seed = engines[0].hash_seed
    group = sharded_abi.LocalGroup(world)
    ranks = [sharded_abi.Sharded(engines[r], world, r, 1 << 12, transport=group.transport(r)) for r in range(world)]
    rng = np.random.default_rng(5)
    keys = rng.integers(1, 2**62, size=3000, dtype=np.int64)
    assert int((owner_of_tensor(torch.from_numpy(keys), seed, world).numpy() == 1).sum()) > 600  # more than rank 1 takes
    h = np.zeros(3000, dtype=HIT_DTYPE)
    h["key"], h["delta"] = keys, 1
    off = np.arange(0, 3001, 3, dtype=np.int64)
    results, errors = {}, []

    def run(r):
        try:
            sh = _RequestsOverTheAbi(ranks[r], dev)
            t = torch.from_numpy((h if r == 0 else h[:0]).view(np.int64).reshape(-1, 2).copy()).to(dev)
            o = torch.from_numpy(off if r == 0 else off[:1]).to(dev)
            try:
                sh.check(t, o, W.NOW0_US)
                results[r] = "applied"
            except sharded_abi.ShardedError as ex:
                results[r] = ex.code
            small = torch.from_numpy((h[:300] if r == 0 else h[:0]).view(np.int64).reshape(-1, 2).copy()).to(dev)
            so = torch.from_numpy(off[:101] if r == 0 else off[:1]).to(dev)
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
    assert results[1] == -7 and results[0] < 0, results  # RL_ERR_BATCH_TOO_LARGE where it happened, a refusal on the other rank
    assert results[(0, "after")] == 0
    assert sum(e.stats()["live_cells"] for e in engines) == 300  # only the step that fitted was applied
    for sh in ranks:
        sh.close()
    for e in engines:
        e.close()
    group.close()


def test_a_multi_counter_step_one_shard_cannot_take_is_refused_on_every_rank():
    from test_sharded_multi_gloo import SIMPLE

    dev = torch.device("cuda", 0)
    world = 2
    probe = Engine(capacity_cells=1 << 10, max_batch_hits=1 << 10)
    seed = probe.hash_seed
    probe.close()
    engines = [_multi_engine(world, 0, seed, capacity_cells=1 << 16), _multi_engine(world, 1, seed, capacity_cells=1 << 9)]
    group = sharded_abi.LocalGroup(world)
    ranks = [sharded_abi.Sharded(engines[r], world, r, 4096, transport=group.transport(r)) for r in range(world)]
    rng = np.random.default_rng(4)
    keys = rng.integers(1, 2**62, size=3000, dtype=np.int64)  # far more new cells than the small shard takes
    hits = np.zeros((3000, 2), dtype=np.int64)
    hits[:, 0] = keys
    hits[:, 1] = 4 | (1 << 32)  # limit 4, delta 1
    off = np.arange(0, 3001, 3, dtype=np.int64)
    results, errors = {}, []

    def run(r):
        try:
            sh = _RequestsOverTheAbi(ranks[r], dev)
            t = torch.from_numpy(hits if r == 0 else hits[:0]).to(dev)
            o = torch.from_numpy(off if r == 0 else off[:1]).to(dev)
            try:
                sh.check(t, o, W.NOW0_US)
                results[r] = "applied"
            except sharded_abi.ShardedError as ex:
                results[r] = ex.code
            small = torch.from_numpy(hits[:30] if r == 0 else hits[:0]).to(dev)  # a step that fits still works afterwards
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
    assert results[0] == results[1] == -4  # RL_ERR_TABLE_FULL on both ranks, nothing applied on either
    assert results[(0, "after")] == 0
    assert sum(e.stats()["live_cells"] for e in engines) == 30 + len(SIMPLE)
    for sh in ranks:
        sh.close()
    for e in engines:
        e.close()
    group.close()


def test_a_heavy_key_on_a_cold_engine_makes_every_rank_go_round_again_not_just_its_owner():
    """The owners' sort takes at most 65 535 hits per hash bucket.  A cold engine that meets a key with 65 400 hits beside
    the bucket's ordinary share reports it (its veto word says so, gathered on the device with the others'), the key gets a
    bucket of its own, and ALL ranks begin the step again — nothing was applied by the first go.  Verdicts, first_limited
    and the union of the tables against one sequential oracle on the concatenated slices (in_memory.rs:141-153)."""
    import oracle
    from limitador_amd.wire import HIT_DTYPE

    dev = torch.device("cuda", 0)
    world, n_heavy, n_plain = 2, 32_700, 100_000
    rows = [(100_000, 60), (3, 60)]
    engines = []
    for _ in range(world):
        e = Engine(capacity_cells=1 << 19, max_batch_hits=1 << 18)
        e.set_limits(rows)
        engines.append(e)
    group = sharded_abi.LocalGroup(world)
    ranks = [sharded_abi.Sharded(engines[r], world, r, 1 << 18, transport=group.transport(r)) for r in range(world)]
    heavy = 0x1234_5678_9ABC
    rng = np.random.default_rng(11)

    def a_slice():
        # requests in random order: n_heavy of them [the heavy key, a user's key], n_plain of them [a user's key]
        with_heavy = rng.permutation(np.arange(n_heavy + n_plain) < n_heavy)
        users = W.splitmix64((rng.integers(0, 60_000, size=n_heavy + n_plain) + 7).astype(np.uint64)) & np.uint64(0x3FFFFFFFFFFFFFFF)
        off = np.concatenate([[0], np.cumsum(1 + with_heavy.astype(np.int64))])
        h = np.zeros(int(off[-1]), dtype=HIT_DTYPE)
        h["delta"] = 1
        h["key"][off[:-1][with_heavy]] = heavy  # (limit 0)
        last = off[1:] - 1
        h["key"][last] = users
        h["limit"][last] = 1
        return h, off

    steps = 2
    data = [[a_slice() for _ in range(world)] for _ in range(steps)]
    got, errors = {}, []

    def run(r):
        try:
            sh = _RequestsOverTheAbi(ranks[r], dev)
            outs = []
            for s in range(steps):
                h, off = data[s][r]
                t = torch.from_numpy(h.view(np.int64).reshape(-1, 2).copy()).to(dev)
                v, f, _rem, _exp = sh.check(t, torch.from_numpy(off).to(dev), W.NOW0_US + s)
                outs.append((v.cpu().numpy(), f.cpu().numpy()))
            got[r] = outs
        except Exception as ex:
            errors.append((r, repr(ex)))

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads), "a rank is stuck at an exchange"
    orc = oracle.OracleStorage()
    orc.set_limits(rows)
    for s in range(steps):
        hits = np.concatenate([data[s][r][0] for r in range(world)])
        off, lo_h = [0], [0]
        for r in range(world):
            off.extend((data[s][r][1][1:] + off[-1]).tolist())
            lo_h.append(lo_h[-1] + len(data[s][r][0]))
        v, f, _rem, _exp = orc.check_and_update(hits, W.NOW0_US + s, req_off=np.array(off, dtype=np.uint32))
        nr = n_heavy + n_plain
        for r in range(world):
            fr = f[r * nr:(r + 1) * nr].astype(np.int64)
            assert np.array_equal(got[r][s][0], v[r * nr:(r + 1) * nr]), f"step {s} rank {r}: verdicts"
            assert np.array_equal(got[r][s][1], np.where(fr >= 0, fr - lo_h[r], -1)), f"step {s} rank {r}: first_limited"
    rows_out = np.concatenate([e.dump_cells() for e in engines])
    assert len(np.unique(rows_out["key"])) == len(rows_out) == orc.num_qualified()
    for r in rows_out:
        assert (int(r["value"]), int(r["expiry_us"]), int(r["limit"])) == orc.peek(int(r["key"]))
    for sh in ranks:
        sh.close()
    for e in engines:
        e.close()
    group.close()


# ---- a sweep as a command of the routed pipeline: rl_sharded_sweep_submit / _collect ------------------------------------
@pytest.mark.parametrize("world", [2, 3])
def test_routed_sweeps_between_slices_in_flight(world):
    """BASELINE.json configs[4] "concurrent expiry sweep", routed: every rank sweeps its shard at the same point of the
    sequence — behind every slice submitted so far, in front of every later one — with slices in flight on both sides of it
    (three commands deep).  Against ONE sequential oracle that is swept at those points: verdicts of every slice, the sum of
    the ranks' removed cells per sweep, and the union of the tables (a swept cell that a later slice touches is created anew:
    value 0, expiry now + window, in_memory.rs:122-127)."""
    dev = torch.device("cuda", 0)
    n, steps, n_keys = 20_000, 10, 6000
    sweep_after = {1, 2, 5, 8}  # (two sweeps back to back in the window of three, too)
    group = sharded_abi.LocalGroup(world)
    engines = [Engine(capacity_cells=1 << 16, max_batch_hits=world * n) for _ in range(world)]
    for e in engines:
        e.set_limits(ROWS)
    ranks = [sharded_abi.Sharded(engines[r], world, r, n, transport=group.transport(r)) for r in range(world)]
    orc = oracle.OracleStorage()
    orc.set_limits(ROWS)
    rng = np.random.default_rng(70 + world)
    slices = [[_slice(rng, n - 53 * s - 11 * r, n_keys) for r in range(world)] for s in range(steps)]
    now_of = [W.NOW0_US + 450_000 * s for s in range(steps)]   # limit 1's one-second windows run out every other slice
    want, want_removed = [], []
    for s in range(steps):
        for r in range(world):
            want.append(orc.check_and_update(slices[s][r], now_of[s])[0])
        if s in sweep_after:
            want_removed.append(orc.sweep_expired(now_of[s] + 200_000))
    assert sum(want_removed) > 1000
    dev_in = [[_to_dev(slices[s][r], dev) for r in range(world)] for s in range(steps)]
    dev_out = [[torch.full((len(slices[s][r]),), 7, dtype=torch.uint8, device=dev) for r in range(world)] for s in range(steps)]
    torch.cuda.synchronize()
    removed = [[] for _ in range(world)]
    errors = []

    def run(r):
        try:
            sh, kinds = ranks[r], []

            def collect_one():
                if kinds.pop(0) == "sweep":
                    removed[r].append(sh.sweep_collect())
                else:
                    sh.collect()

            for s in range(steps):
                while sh.in_flight == 3:
                    collect_one()
                sh.submit(dev_in[s][r].data_ptr(), len(slices[s][r]), now_of[s], dev_out[s][r].data_ptr())
                kinds.append("slice")
                if s in sweep_after:
                    while sh.in_flight == 3:
                        collect_one()
                    sh.sweep_submit(now_of[s] + 200_000)
                    kinds.append("sweep")
                    if s == 5:  # the wrong collect for the oldest command is refused, and nothing is lost
                        if kinds[0] == "sweep":
                            with pytest.raises(sharded_abi.ShardedError):
                                sh.collect()
                        else:
                            with pytest.raises(sharded_abi.ShardedError):
                                sh.sweep_collect()
            while sh.in_flight:
                collect_one()
            sh.sync()
        except Exception as ex:
            errors.append((r, repr(ex)))

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
    assert [sum(removed[r][i] for r in range(world)) for i in range(len(want_removed))] == want_removed
    rows = np.concatenate([e.dump_cells() for e in engines])
    assert len(rows) == orc.num_qualified() and len(np.unique(rows["key"])) == len(rows)
    for row in rows[::29]:
        assert (int(row["value"]), int(row["expiry_us"]), int(row["limit"])) == orc.peek(int(row["key"]))
    for sh in ranks:
        sh.close()
    for e in engines:
        e.close()
    group.close()


@pytest.mark.parametrize("world", [2, 3])
def test_config5_trace_routed_by_key_with_sweeps_between_the_steps(world):
    """BASELINE.json configs[4] / SURVEY.md §8(d) config #5 as a ROUTED run (VERDICT r04 missing #2): 4 namespaces x 8 limits
    (2 simple + 6 qualified on 1-2 variables), windows {1, 10, 60, 3600} s, k in [1, 8] counters per request with the
    conditions as per-request applicability, 16 steps with an advancing clock that crosses the 1 s and 10 s windows — every
    request's counters spread over the ranks by key hash, the all-or-nothing rule spanning them (in_memory.rs:141-153) through
    rl_sharded_check_requests_device, and a routed sweep (rl_sharded_sweep_submit / _collect) between the steps.  Against ONE
    sequential oracle on the concatenated slices, swept at the same points: verdicts, first_limited, remaining / expires_in
    on the load_counters steps, the sweeps' removed counts, the union of the shards' tables."""
    from limitador_amd.sharded import owner_of_tensor
    from limitador_amd.wire import HIT_DTYPE, RL_SIMPLE

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(550 + world)
    windows = [1, 10, 60, 3600]
    rows, simple, limit_of = [], [], {}
    for ns in range(4):
        for j in range(8):
            lid = len(rows)
            rows.append((int(rng.integers(20, 2000)) if j == 0 else int(rng.integers(2, 60)), windows[(ns + j) % 4]))
            limit_of[(ns, j)] = lid
            if j < 2:
                simple.append((lid, 20_000_000 + lid))
    probe = Engine(capacity_cells=1 << 10, max_batch_hits=1 << 10)
    seed = probe.hash_seed
    probe.close()
    engines = []
    for r in range(world):
        e = Engine(capacity_cells=1 << 16, max_batch_hits=1 << 15)
        e.set_limits(rows)
        for lid, key in simple:  # a simple counter's cell lives on the rank its key hashes to (in_memory.rs:38-44)
            if int(owner_of_tensor(torch.tensor([key]), seed, world)[0]) == r:
                e.add_counter(lid | RL_SIMPLE, key)
        engines.append(e)
    orc = oracle.OracleStorage()
    orc.set_limits(rows)
    for lid, key in simple:
        orc.add_counter(lid | RL_SIMPLE, key)
    group = sharded_abi.LocalGroup(world)
    ranks = [sharded_abi.Sharded(engines[r], world, r, 1 << 13, transport=group.transport(r)) for r in range(world)]
    steps = 16

    def make(n_req):
        hits, off = [], [0]
        for _ in range(n_req):
            ns, user, path = int(rng.integers(0, 4)), int(rng.zipf(1.4) - 1) % 300, int(rng.integers(0, 6))
            delta = 1 if rng.random() < 0.8 else int(rng.integers(0, 5))
            applies = [j for j in range(8) if rng.random() < (0.9 if j < 2 else 0.45)] or [0]
            for j in sorted(applies, key=lambda x: (x >= 2,)):  # simple first (in_memory.rs:105,121)
                lid = limit_of[(ns, j)]
                if j < 2:
                    hits.append((20_000_000 + lid, lid | RL_SIMPLE, delta))
                else:
                    material = lid * 1_000_003 + user * 7 + (path if j % 2 else 0)
                    hits.append((int(W.splitmix64(np.array([material], dtype=np.uint64))[0]), lid, delta))
            off.append(len(hits))
        arr = np.zeros(len(hits), dtype=HIT_DTYPE)
        for i, h in enumerate(hits):
            arr[i] = h
        return arr, np.array(off, dtype=np.int64)

    data = [[make(int(rng.integers(150, 700))) for _ in range(world)] for _ in range(steps)]
    nows, sweeps, now = [], [], W.NOW0_US
    for s in range(steps):
        nows.append(now)
        now += int(rng.integers(250_000, 3_000_000))
        sweeps.append(now if s % 2 else None)
    want, want_removed = [], []
    for s in range(steps):
        hits = np.concatenate([data[s][r][0] for r in range(world)])
        off = [0]
        for r in range(world):
            off.extend((data[s][r][1][1:] + off[-1]).tolist())
        load = s % 3 == 2
        v, f, rem, exp = orc.check_and_update(hits, nows[s], req_off=np.array(off, dtype=np.uint32), load_counters=load)
        per_rank, lo_r, lo_h = [], 0, 0
        for r in range(world):
            nr, nh = len(data[s][r][1]) - 1, len(data[s][r][0])
            fr = f[lo_r:lo_r + nr].astype(np.int64)
            per_rank.append((v[lo_r:lo_r + nr], np.where(fr >= 0, fr - lo_h, -1), rem[lo_h:lo_h + nh] if load else None,
                             exp[lo_h:lo_h + nh] if load else None))
            lo_r, lo_h = lo_r + nr, lo_h + nh
        want.append(per_rank)
        if sweeps[s] is not None:
            want_removed.append(orc.sweep_expired(sweeps[s]))
    assert sum(want_removed) > 50
    got, removed, errors = {r: [] for r in range(world)}, {r: [] for r in range(world)}, []

    def run(r):
        try:
            sh = _RequestsOverTheAbi(ranks[r], dev)
            for s in range(steps):
                h, off = data[s][r]
                t = torch.from_numpy(h.view(np.int64).reshape(-1, 2).copy()).to(dev)
                load = s % 3 == 2
                v, f, rem, exp = sh.check(t, torch.from_numpy(off).to(dev), nows[s], load_counters=load)
                got[r].append((v.cpu().numpy(), f.cpu().numpy(), None if rem is None else rem.cpu().numpy().view(np.uint64),
                               None if exp is None else exp.cpu().numpy().view(np.uint64)))
                if sweeps[s] is not None:
                    ranks[r].sweep_submit(sweeps[s])
                    removed[r].append(ranks[r].sweep_collect())
        except Exception as ex:
            errors.append((r, repr(ex)))

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads), "a rank is stuck at an exchange"
    for s in range(steps):
        for r in range(world):
            v, f, rem, exp = got[r][s]
            wv, wf, wrem, wexp = want[s][r]
            assert np.array_equal(v, wv), f"step {s} rank {r}: verdicts"
            assert np.array_equal(f, wf), f"step {s} rank {r}: first_limited"
            if wrem is not None:
                assert np.array_equal(rem, wrem) and np.array_equal(exp, wexp), f"step {s} rank {r}: remaining / expires_in"
    assert [sum(removed[r][i] for r in range(world)) for i in range(len(want_removed))] == want_removed
    _check_union_of_tables(engines, orc)
    for sh in ranks:
        sh.close()
    for e in engines:
        e.close()
    group.close()
