"""Parity of the HIP engine (through the C ABI) against the CPU oracle: bit-exact verdicts,
first-limited indices and final table contents on the same seeded inputs.  Needs a MI355X."""
import numpy as np
import pytest

import oracle
import scenarios
from helpers.limiter import TestsLimiter
from limitador_amd import workloads as W
from limitador_amd.wire import CELL_ROW_DTYPE, HIT_DTYPE, RL_SIMPLE

pytestmark = pytest.mark.gpu

SEC = 1_000_000
NOW = W.NOW0_US


@pytest.fixture()
def make_engine():
    from limitador_amd.engine import Engine

    made = []

    def _make(capacity_cells=1 << 16, **kw):
        e = Engine(capacity_cells=capacity_cells, **kw)
        made.append(e)
        return e

    yield _make
    for e in made:
        e.close()


def assert_same_state(eng, orc, n_simple_expected=None):
    """Every live cell of the engine equals the oracle's cell, and the counts agree."""
    rows = eng.dump_cells()
    qual = rows[(rows["limit"] & RL_SIMPLE) == 0]
    simp = rows[(rows["limit"] & RL_SIMPLE) != 0]
    assert len(qual) == orc.num_qualified(), (len(qual), orc.num_qualified())
    assert len(np.unique(rows["key"])) == len(rows), "duplicate key in the table"
    for r in qual:
        got = orc.peek(int(r["key"]))
        assert got is not None, f"engine has key {int(r['key'])} the oracle lacks"
        assert (int(r["value"]), int(r["expiry_us"]), int(r["limit"])) == got, (r, got)
    for r in simp:
        got = orc.peek_simple(int(r["limit"]))
        assert got is not None
        assert (int(r["value"]), int(r["expiry_us"])) == got, (r, got)
    if n_simple_expected is not None:
        assert len(simp) == n_simple_expected


def pair(make_engine, rows, simple_keys=(), **kw):
    eng = make_engine(**kw)
    orc = oracle.OracleStorage()
    eng.set_limits(rows)
    orc.set_limits(rows)
    for limit, key in simple_keys:
        eng.add_counter(limit | RL_SIMPLE, key)
        orc.add_counter(limit | RL_SIMPLE, key)
    return eng, orc


def run_both(eng, orc, hits, now, **kw):
    v1, f1, r1, e1 = eng.check_and_update(hits, now, **kw)
    v2, f2, r2, e2 = orc.check_and_update(hits, now, **kw)
    assert np.array_equal(v1, v2), f"verdict mismatch at {np.nonzero(v1 != v2)[0][:10]}"
    assert np.array_equal(f1, f2), f"first_limited mismatch at {np.nonzero(f1 != f2)[0][:10]}"
    if kw.get("load_counters"):
        assert np.array_equal(r1, r2), f"remaining mismatch at {np.nonzero(r1 != r2)[0][:10]}"
        assert np.array_equal(e1, e2), f"expires_in mismatch at {np.nonzero(e1 != e2)[0][:10]}"
    return v1


# ---- the reference's own scenarios, through the engine ---------------------------------------
@pytest.mark.parametrize("mode", ["served", "one_launch_per_call", "partitioned_only"])
@pytest.mark.parametrize("scenario", scenarios.ALL, ids=lambda f: f.__name__)
def test_reference_scenarios_on_engine(make_engine, monkeypatch, scenario, mode):
    """Per-request calls (one request, 1..k counters) are answered by a lingering k_gen_serve without a launch per call
    (the default); RL_SERVE=0: one launch per call (k_gen_tiny / k_bkt_tiny); RL_TINY_MAX=0: the same scenarios through
    the partitioned kernels."""
    if mode != "served":
        monkeypatch.setenv("RL_SERVE", "0")
    if mode == "partitioned_only":
        monkeypatch.setenv("RL_TINY_MAX", "0")
    scenario(TestsLimiter(make_engine(capacity_cells=1 << 12, max_batch_hits=1 << 12)))


@pytest.mark.parametrize("load", [False, True])
def test_requests_that_list_qualified_counters_first(make_engine, load):
    """The storage walks a request's simple counters first whatever order the caller lists them in (in_memory.rs:105,121:
    two filtered loops).  Micro-batches whose requests list a qualified counter in front of a simple one — the one-launch
    kernel's wave-resident replay then takes its two-pass form, ordered batches the single walk — and one request at a
    time through the server."""
    rng = np.random.default_rng(91)
    rows = [(6, 1), (25, 10), (40, 60), (3, 60)]
    eng, orc = pair(make_engine, rows, [(2, 9_000_002), (3, 9_000_003)])
    keys = W.splitmix64(np.arange(1, 40, dtype=np.uint64))
    now = NOW
    for step in range(60):
        n_req = int(rng.integers(1, 20)) if step % 3 else 1
        hits, off = [], [0]
        for _ in range(n_req):
            d = int(rng.integers(0, 3))
            req = [(int(keys[i]), int(i % 2), d) for i in rng.integers(0, len(keys), size=int(rng.integers(0, 3)))]
            simple = [(9_000_002, 2 | RL_SIMPLE, d), (9_000_003, 3 | RL_SIMPLE, d)][: int(rng.integers(0, 3))]
            order = rng.integers(0, 3)  # qualified first / simple first / interleaved
            req = req + simple if order == 0 else simple + req if order == 1 else [x for p in zip(req, simple) for x in p] + req[len(simple):] + simple[len(req):]
            hits += req
            off.append(len(hits))
        arr = np.zeros(len(hits), dtype=HIT_DTYPE)
        for i, h in enumerate(hits):
            arr[i] = h
        run_both(eng, orc, arr, now, req_off=np.array(off, dtype=np.uint32), load_counters=load)
        now += int(rng.integers(0, SEC // 2))
    assert_same_state(eng, orc, n_simple_expected=2)


def test_per_request_calls_between_everything_else(make_engine):
    """The server (k_gen_serve) lingers on the engine's stream: every other call — batches, reads, sweeps, limits —
    must send it away first, and a per-request call after a pause longer than its linger finds it gone and starts
    another.  One request at a time against the oracle, with u64 deltas and load_counters, interleaved with the rest of
    the surface."""
    import time

    rng = np.random.default_rng(77)
    rows = [(9, 1), (40, 10), (300, 60)]
    eng, orc = pair(make_engine, rows, [(2, 9_000_009)])
    keys = W.splitmix64(np.arange(1, 60, dtype=np.uint64))
    now = NOW

    def one_request(load, big_delta=False):
        k = int(rng.integers(1, 5))
        idx = rng.integers(0, len(keys), size=k)
        hits = np.zeros(k + 1, dtype=HIT_DTYPE)
        hits[0] = (9_000_009, 2 | RL_SIMPLE, 1)  # simple counters first
        hits["key"][1:] = keys[idx]
        hits["limit"][1:] = idx % 2
        d = int(rng.integers(0, 3))
        hits["delta"] = d
        kw = {"req_off": np.array([0, k + 1], dtype=np.uint32), "load_counters": load}
        if big_delta:
            kw["req_delta"] = np.array([2**40 + d], dtype=np.uint64)
        run_both(eng, orc, hits, now, **kw)

    for step in range(300):
        one_request(load=bool(step % 3 == 0), big_delta=(step % 50 == 49))
        now += int(rng.integers(0, SEC // 3))
        if step % 40 == 7:  # a batch in between
            idx = rng.integers(0, len(keys), size=5000)
            hits = np.zeros(5000, dtype=HIT_DTYPE)
            hits["key"], hits["limit"], hits["delta"] = keys[idx], idx % 2, 1
            run_both(eng, orc, hits, now)
        if step % 40 == 17:
            assert eng.sweep_expired(now) == orc.sweep_expired(now)
        if step % 40 == 27:
            probe = np.zeros(3, dtype=HIT_DTYPE)
            probe["key"], probe["limit"], probe["delta"] = keys[:3], np.arange(3) % 2, 1
            assert np.array_equal(eng.is_within_limits(probe, now), orc.is_within_limits(probe, now))
        if step % 40 == 37:
            time.sleep(0.002)  # ten lingers: the server has left by itself
    assert_same_state(eng, orc, n_simple_expected=1)


# ---- displaced keys: long probe chains, tags that share a word with EMPTY -----------------------
@pytest.mark.parametrize("low_word", ["any", "all_ones"])
def test_dense_table_long_probe_chains(make_engine, low_word):
    """A table filled to 0.7 batch by batch: most new keys are displaced, many by more cells than one probe trip fetches
    (apply2_round reads PROBE_W tags per trip — their low words — and claims an EMPTY cell with a compare-and-swap).  With
    `all_ones` every key's low word equals EMPTY's, so that every comparison of a low word is inconclusive."""
    rng = np.random.default_rng(7)
    rows = [(9, 60), (400, 3600), (3, 1)]
    eng, orc = pair(make_engine, rows, capacity_cells=4096)
    ids = W.splitmix64(np.arange(1, 2901, dtype=np.uint64))
    keys = ids if low_word == "any" else (ids << np.uint64(32)) | np.uint64(0xFFFFFFFF)
    assert len(np.unique(keys)) == len(keys)
    key_limit = rng.integers(0, 3, size=len(keys))
    now = NOW
    known = 0
    for step in range(14):
        known = min(len(keys), known + 230)  # ~230 keys the table has not seen, among hits on the ones it has
        n = 3000 + 137 * step
        idx = rng.integers(0, known, size=n)
        idx[: known - max(0, known - 230)] = np.arange(max(0, known - 230), known)  # every new key at least once
        rng.shuffle(idx)
        hits = np.empty(n, dtype=HIT_DTYPE)
        hits["key"] = keys[idx]
        hits["limit"] = key_limit[idx]
        hits["delta"] = rng.integers(0, 3, size=n) if step % 2 else 1
        run_both(eng, orc, hits, now)
        now += int(rng.integers(0, 2 * SEC))
    assert eng.stats()["live_cells"] == known == len(keys)
    assert_same_state(eng, orc)


# ---- seeded random traces, single-counter requests -------------------------------------------
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_single_counter_batches(make_engine, seed):
    rng = np.random.default_rng(seed)
    rows = [(5, 1), (50, 10), (1000, 60), (0, 60), (2**64 - 1, 3600), (7, 0)]
    simple = [(4, 9_000_001), (0, 9_000_002)]
    eng, orc = pair(make_engine, rows, simple)
    n_keys = 300
    key_limit = rng.integers(0, 4, size=n_keys)  # each qualified key belongs to one limit
    key_limit[:10] = 5  # some zero-window keys
    now = NOW
    for step in range(25):
        n = int(rng.integers(1, 3000))
        idx = (rng.zipf(1.3, size=n) - 1) % n_keys if step % 2 else rng.integers(0, n_keys, size=n)
        hits = np.empty(n, dtype=HIT_DTYPE)
        hits["key"] = W.splitmix64(idx.astype(np.uint64))
        hits["limit"] = key_limit[idx]
        hits["delta"] = rng.integers(0, 4, size=n) if step % 3 else 1
        # sprinkle the two simple counters in
        sm = rng.random(n) < 0.05
        which = rng.integers(0, 2, size=n)
        hits["key"][sm] = np.where(which[sm] == 0, 9_000_001, 9_000_002)
        hits["limit"][sm] = np.where(which[sm] == 0, 4, 0) | RL_SIMPLE
        run_both(eng, orc, hits, now)
        now += int(rng.integers(0, 3 * SEC))
    assert_same_state(eng, orc, n_simple_expected=2)


def test_registered_host_arrays_give_the_same_answers(make_engine):
    """rl_host_register pins the caller's staging arrays in place (what a binding does with the buffers it reuses): the
    same calls, the same bytes — single-counter batches, a tiny call, a multi-counter call with load_counters — with the
    hits read from and the verdicts written into registered memory."""
    from limitador_amd.engine import EngineError

    rng = np.random.default_rng(41)
    rows = [(5, 1), (50, 10), (1000, 60)]
    eng, orc = pair(make_engine, rows, [(0, 9_000_001)])
    cap = 20_000
    hits_buf = np.zeros(cap, dtype=HIT_DTYPE)
    vout = np.full(cap, 0xEE, dtype=np.uint8)
    eng.host_register(hits_buf)
    eng.host_register(vout)
    now = NOW
    for n in (20_000, 1, 7, 12_345, 0, 3000):
        idx = rng.integers(0, 400, size=n)
        hits = hits_buf[:n]
        hits["key"] = W.splitmix64(idx.astype(np.uint64))
        hits["limit"] = idx % 3
        hits["delta"] = rng.integers(0, 3, size=n)
        req_off = None
        kw = {}
        if n == 3000:  # requests of 1..3 counters, values loaded
            cuts = np.unique(np.concatenate([[0, n], rng.integers(0, n, size=1500)])).astype(np.uint32)
            kw = {"req_off": cuts, "load_counters": True}
            # (one delta per request: in_memory.rs:75 — every hit carries its request's)
            hits["delta"] = np.repeat(rng.integers(0, 3, size=len(cuts) - 1), np.diff(cuts))
        vout[:] = 0xEE
        v1, f1, r1, e1 = eng.check_and_update(hits, now, verdict_out=vout, **kw)
        v2, f2, r2, e2 = orc.check_and_update(hits, now, **kw)
        assert v1.ctypes.data == vout.ctypes.data or n == 0
        assert np.array_equal(v1, v2) and np.array_equal(f1, f2)
        assert np.all(vout[len(v2):] == 0xEE), "nothing past the batch's verdicts is written"
        if kw:
            assert np.array_equal(r1, r2) and np.array_equal(e1, e2)
        now += SEC // 2
    eng.host_unregister(hits_buf)
    eng.host_unregister(vout)
    with pytest.raises(EngineError):
        eng.host_unregister(np.zeros(16, dtype=np.uint8))  # never registered
    run_both(eng, orc, hits_buf[:500].copy(), now)  # and pageable again
    assert_same_state(eng, orc, n_simple_expected=1)


def test_one_hot_key_nonuniform_deltas(make_engine):
    """Every hit on one cell, mixed deltas: the one-lane replay of the round (mixed deltas)."""
    rng = np.random.default_rng(7)
    eng, orc = pair(make_engine, [(5000, 60)])
    for step in range(4):
        n = 20000
        hits = np.empty(n, dtype=HIT_DTYPE)
        hits["key"] = 77
        hits["limit"] = 0
        hits["delta"] = rng.integers(0, 6, size=n)
        v = run_both(eng, orc, hits, NOW + step)
        if step == 0:
            assert 0 < v.sum() < n  # the window fills up inside the first batch
    assert_same_state(eng, orc)


def test_hot_keys_uniform_delta_saturate_mid_batch(make_engine):
    rng = np.random.default_rng(8)
    eng, orc = pair(make_engine, [(1000, 60)], capacity_cells=1 << 18, max_batch_hits=200_000)
    n = 200_000
    hits = np.empty(n, dtype=HIT_DTYPE)
    hits["key"] = W.splitmix64(rng.integers(0, 40, size=n).astype(np.uint64))
    hits["limit"] = 0
    hits["delta"] = 1
    v = run_both(eng, orc, hits, NOW)
    assert v.sum() == n - 40 * 1000
    run_both(eng, orc, hits, NOW + 1)
    run_both(eng, orc, hits, NOW + 61 * SEC)  # windows rolled over
    assert_same_state(eng, orc)


def test_wrapping_values_match_release_build_arithmetic(make_engine):
    """value + delta wraps (in_memory.rs:88,261 in a release build)."""
    big = 2**64 - 3
    eng, orc = pair(make_engine, [(2**64 - 1, 60)])
    one = np.zeros(1, dtype=HIT_DTYPE)
    one[0] = (5, 0, 2)
    cell = np.zeros(1, dtype=CELL_ROW_DTYPE)
    cell[0] = (5, 0, 0, big, NOW + 30 * SEC)
    eng.load_cells(cell)
    orc.load_cells([5], [0], [big], [NOW + 30 * SEC])
    hits = np.repeat(one, 8)
    run_both(eng, orc, hits, NOW)
    assert_same_state(eng, orc)


def test_empty_batch_and_single_hit(make_engine):
    eng, orc = pair(make_engine, [(1, 60)])
    v, f, _, _ = eng.check_and_update(np.zeros(0, dtype=HIT_DTYPE), NOW)
    assert v.shape == (0,)
    one = np.zeros(1, dtype=HIT_DTYPE)
    one[0] = (5, 0, 1)
    assert run_both(eng, orc, one, NOW)[0] == 0
    assert run_both(eng, orc, one, NOW)[0] == 1
    assert_same_state(eng, orc)


def test_maximum_batch_size(make_engine):
    n = 1 << 18
    eng, orc = pair(make_engine, [(3, 60)], capacity_cells=1 << 20, max_batch_hits=n)
    rng = np.random.default_rng(5)
    hits = W.uniform_batch(50_000, n, rng)
    run_both(eng, orc, hits, NOW)
    assert_same_state(eng, orc)
    from limitador_amd.engine import EngineError

    with pytest.raises(EngineError) as e:
        eng.check_and_update(np.zeros(n + 1, dtype=HIT_DTYPE), NOW)
    assert e.value.code == -7


# ---- error behaviour ---------------------------------------------------------------------------
def test_errors_are_loud_and_leave_the_table_usable(make_engine):
    from limitador_amd.engine import EngineError

    eng, orc = pair(make_engine, [(5, 60), (5, 60)])
    bad = np.zeros(1, dtype=HIT_DTYPE)
    bad[0] = (1, 9, 1)  # unknown limit id
    with pytest.raises(EngineError) as e:
        eng.check_and_update(bad, NOW)
    assert e.value.code == -1 and not e.value.transient
    bad[0] = (1, 0 | RL_SIMPLE, 1)  # simple counter never add_counter'ed
    with pytest.raises(EngineError) as e:
        eng.check_and_update(bad, NOW)
    assert e.value.code == -5
    ok = np.zeros(1, dtype=HIT_DTYPE)
    ok[0] = (2, 0, 1)
    run_both(eng, orc, ok, NOW)
    bad[0] = (2, 1, 1)  # same key, different limit
    with pytest.raises(EngineError) as e:
        eng.check_and_update(bad, NOW)
    assert e.value.code == -6
    run_both(eng, orc, ok, NOW)
    assert_same_state(eng, orc)


# ---- the other CounterStorage methods ------------------------------------------------------------
def test_is_within_limits_and_update_counter_parity(make_engine):
    rng = np.random.default_rng(11)
    rows = [(5, 1), (50, 10), (9, 0)]
    eng, orc = pair(make_engine, rows, [(1, 7_000_001)])
    now = NOW
    for step in range(12):
        n = int(rng.integers(1, 4000))
        idx = rng.integers(0, 200, size=n)
        hits = np.empty(n, dtype=HIT_DTYPE)
        hits["key"] = W.splitmix64(idx.astype(np.uint64))
        hits["limit"] = idx % 3
        hits["delta"] = rng.integers(0, 5, size=n)
        sm = rng.random(n) < 0.05
        hits["key"][sm] = 7_000_001
        hits["limit"][sm] = 1 | RL_SIMPLE
        assert np.array_equal(eng.is_within_limits(hits, now), orc.is_within_limits(hits, now))
        eng.update_counters(hits, now)
        orc.update_counters(hits, now)
        assert np.array_equal(eng.is_within_limits(hits, now), orc.is_within_limits(hits, now))
        now += int(rng.integers(0, 2 * SEC))
    # update_counter on a vacant simple cell creates it (in_memory.rs:60-62)
    h = np.zeros(1, dtype=HIT_DTYPE)
    h[0] = (7_000_002, 0 | RL_SIMPLE, 3)
    eng.update_counters(h, now)
    orc.update_counters(h, now)
    assert_same_state(eng, orc, n_simple_expected=2)


def test_get_delete_clear_sweep_parity(make_engine):
    rng = np.random.default_rng(12)
    rows = [(100, 1), (100, 10), (100, 60)]
    eng, orc = pair(make_engine, rows, [(2, 8_000_001)])
    hits = np.empty(3000, dtype=HIT_DTYPE)
    idx = rng.integers(0, 500, size=3000)
    hits["key"] = W.splitmix64(idx.astype(np.uint64))
    hits["limit"] = idx % 2
    hits["delta"] = 1
    run_both(eng, orc, hits, NOW)
    s = np.zeros(1, dtype=HIT_DTYPE)
    s[0] = (8_000_001, 2 | RL_SIMPLE, 1)
    run_both(eng, orc, s, NOW)

    def same_counters(limit, now):
        a = eng.get_counters(limit, now)
        b = orc.get_counters(limit, now)
        if limit & RL_SIMPLE:
            assert len(a) == len(b)
            if len(a):
                assert (int(a[0]["value"]), int(a[0]["expiry_us"])) == (int(b[0]["value"]), int(b[0]["expires_in_us"]))
            return len(a)
        ka = sorted((int(r["key"]), int(r["value"]), int(r["expiry_us"])) for r in a)
        kb = sorted((int(r["key"]), int(r["value"]), int(r["expires_in_us"])) for r in b)
        assert ka == kb
        return len(ka)

    t = NOW + SEC // 2
    assert same_counters(0, t) > 0 and same_counters(1, t) > 0 and same_counters(2 | RL_SIMPLE, t) == 1
    t = NOW + 2 * SEC  # limit 0's 1-second windows are over: hidden (in_memory.rs:168,180)
    assert same_counters(0, t) == 0 and same_counters(1, t) > 0
    assert eng.sweep_expired(t) == orc.sweep_expired(t)
    assert_same_state(eng, orc, n_simple_expected=1)
    eng.delete_counters(1)
    orc.delete_counters(1)
    assert same_counters(1, t) == 0
    eng.clear()
    orc.clear()
    assert_same_state(eng, orc, n_simple_expected=0)
    eng.compact()
    assert_same_state(eng, orc, n_simple_expected=0)
    st = eng.stats()
    assert st["tombstones"] == 0 and st["live_cells"] == orc.num_qualified()
    # the table keeps working after compaction
    run_both(eng, orc, hits, t)
    assert_same_state(eng, orc)


def test_sweep_is_an_explicit_eviction_event(make_engine):
    """SURVEY.md §7 hard part 3: dropping an expired cell is observable when the next touch is a
    denied one; replaying the sweep into the oracle keeps parity."""
    eng, orc = pair(make_engine, [(3, 1)])
    h = np.zeros(1, dtype=HIT_DTYPE)
    h[0] = (42, 0, 2)
    run_both(eng, orc, h, NOW)
    t = NOW + 5 * SEC
    assert eng.sweep_expired(t) == orc.sweep_expired(t) == 1
    h[0] = (42, 0, 9)  # denied (9 > 3) but re-creates the cell with a fresh window
    run_both(eng, orc, h, t)
    h[0] = (42, 0, 1)
    run_both(eng, orc, h, t + SEC // 2)
    assert_same_state(eng, orc)


# ---- the trait's full-width arguments: u64 deltas, one clock value per request --------------------------
@pytest.mark.parametrize("n_req", [7, 3000])
def test_u64_deltas_and_per_request_clocks(make_engine, n_req):
    """rl_check_and_update_batch_ex: req_delta carries the trait's `delta: u64` (a delta beyond 2^32 is
    Limited / added with wrapping arithmetic like in_memory.rs:259-264, never an error) and req_now_us one
    clock value per request (in_memory.rs:83).  7 requests take the one-launch kernel, 3000 the passes."""
    rng = np.random.default_rng(41 + n_req)
    rows = [(2**64 - 1, 60), (2**40, 1), (1000, 10), (5, 0)]
    simple = [(0, 9_100_001)]
    eng, orc = pair(make_engine, rows, simple)
    now = NOW
    for step in range(6):
        hits, off, deltas, nows = [], [0], [], []
        for r in range(n_req):
            k = int(rng.integers(1, 4))
            d = [1, 3, 2**33, 2**40 - 1, 2**63, 2**64 - 2][int(rng.integers(0, 6))]
            if rng.random() < 0.3:
                hits.append((9_100_001, 0 | RL_SIMPLE, min(d, 2**32 - 1)))
            for _ in range(k):
                lid = int(rng.integers(1, 4))
                key = int(W.splitmix64(np.array([lid * 1000 + int(rng.integers(0, 40))], dtype=np.uint64))[0])
                hits.append((key, lid, min(d, 2**32 - 1)))
            off.append(len(hits))
            deltas.append(d)
            if step % 2 and rng.random() < 0.2:
                now += int(rng.integers(1, 2 * SEC))
            nows.append(now)
        arr = np.zeros(len(hits), dtype=HIT_DTYPE)
        for i, h in enumerate(hits):
            arr[i] = h
        kw = dict(req_off=np.array(off, dtype=np.uint32), req_delta=np.array(deltas, dtype=np.uint64),
                  load_counters=bool(step % 3 == 0))
        if step % 2:
            kw["req_now_us"] = np.array(nows, dtype=np.uint64)
        run_both(eng, orc, arr, now, **kw)
        now += int(rng.integers(0, SEC))
    assert_same_state(eng, orc, n_simple_expected=1)


def test_single_counter_requests_with_u64_deltas(make_engine):
    rng = np.random.default_rng(43)
    eng, orc = pair(make_engine, [(2**63, 60), (100, 60)])
    keys = W.splitmix64(np.arange(1, 301, dtype=np.uint64))
    for step in range(3):
        n = 5000
        idx = rng.integers(0, 300, size=n)
        hits = np.empty(n, dtype=HIT_DTYPE)
        hits["key"], hits["limit"], hits["delta"] = keys[idx], (idx % 2).astype(np.uint32), 1
        deltas = rng.choice(np.array([1, 2**32, 2**35 + 7, 2**62], dtype=np.uint64), size=n)
        run_both(eng, orc, hits, NOW + step, req_delta=deltas)
    assert_same_state(eng, orc)


# ---- BASELINE.json configs at full size ----------------------------------------------------------
def _full_size(make_engine, n_keys, n_hits, steps, zipf, in_flight=False, nows=None, early_third=False, capacity_cells=None, timing_mode=0):
    """in_flight: the bench's entry point and geometry — device-resident batches through
    rl_check_and_update_submit_device / _collect, three in flight, table at load <= 0.30.
    nows: the clock of every batch (default NOW + 1 ms per batch: no pre-populated window ends inside the run).
    early_third: every third key of the universe carries an expiry 2.5 ms after NOW instead of 30 s, so its window ends
    INSIDE the run (atomic_expiring_value.rs:36-42,87-99: the first admitted hit after that resets value and expiry).
    capacity_cells: the table's size (default: load <= 0.30; tests/test_gpu_bench_config.py passes bench.py's 2^26)."""
    rows = [(W.MAX_VALUE, W.WINDOW_S)]
    cap = capacity_cells or 1 << (int(n_keys * 2.2 - 1).bit_length())
    eng, orc = pair(make_engine, rows, capacity_cells=cap, max_batch_hits=n_hits)
    chunk = 1 << 20
    for lo in range(0, n_keys, chunk):
        cells = W.universe_rows(n_keys, lo=lo, hi=min(n_keys, lo + chunk))
        if early_third:
            cells["expiry_us"][(np.arange(lo, lo + len(cells)) % 3) == 0] = NOW + 2500
        eng.load_cells(cells)
        orc.load_cells(cells["key"], cells["limit"], cells["value"], cells["expiry_us"])
    assert eng.stats()["live_cells"] == n_keys
    rng = np.random.default_rng(W.SEED)
    cdf = W.zipf_cdf(n_keys) if zipf else None
    if nows is None:
        nows = [NOW + 1000 * i for i in range(steps)]
    assert len(nows) == steps
    denied = 0
    batches = [W.zipf_batch(n_keys, n_hits, rng, cdf) if zipf else W.uniform_batch(n_keys, n_hits, rng) for _ in range(steps)]
    if not in_flight:
        for hits, now in zip(batches, nows):
            denied += int(run_both(eng, orc, hits, now, want_first_limited=True).sum())
    else:
        import torch

        dev = torch.device("cuda", 0)
        d_hits = [torch.from_numpy(h.view(np.int64).reshape(-1, 2).copy()).to(dev) for h in batches]
        d_verdict = [torch.empty(n_hits, dtype=torch.uint8, device=dev) for _ in batches]
        d_first = [torch.empty(n_hits, dtype=torch.int32, device=dev) for _ in batches]
        torch.cuda.synchronize()
        if timing_mode:  # bench.py's timed region: some launches carry their own start / stop events (rl_kernel_timing)
            eng.kernel_timing(timing_mode)
        pending = 0
        for i in range(steps):
            eng.submit_device(d_hits[i].data_ptr(), n_hits, nows[i], d_verdict[i].data_ptr(), d_first[i].data_ptr())
            if pending == 2:
                eng.collect()
            else:
                pending += 1
        while pending:
            eng.collect()
            pending -= 1
        torch.cuda.synchronize()
        if timing_mode:
            kt = eng.kernel_timing_read(reset=True)
            eng.kernel_timing(0)
            assert kt["launches"] >= 1
        for i, hits in enumerate(batches):
            v, f, _r, _e = orc.check_and_update(hits, nows[i])
            assert np.array_equal(d_verdict[i].cpu().numpy(), v), f"verdicts of batch {i}"
            assert np.array_equal(d_first[i].cpu().numpy(), f), f"first_limited of batch {i}"
            denied += int(v.sum())
    assert 0 < denied < steps * n_hits  # a real allow/deny mix
    # final table: EVERY cell, bit for bit (both sides sorted by key)
    rows_ = np.sort(eng.dump_cells(), order="key")
    k, li, v, e = orc.dump_qualified()
    assert len(rows_) == len(k) == n_keys
    assert np.array_equal(rows_["key"], k)
    assert np.array_equal(rows_["limit"], li)
    assert np.array_equal(rows_["value"], v)
    assert np.array_equal(rows_["expiry_us"], e)
    return eng


def test_config2_uniform_64k_on_1m_keys(make_engine):
    """BASELINE.json configs[1]: 1M keys, uniform 64k batches, fixed window, delta 1."""
    _full_size(make_engine, 1 << 20, 1 << 16, steps=20, zipf=False)


def test_config3_zipf_1m_on_10m_keys(make_engine):
    """BASELINE.json configs[2] (fixed-window semantics): 10M keys, Zipf-0.99 1M batches."""
    eng = _full_size(make_engine, 10_000_000, 1_000_000, steps=3, zipf=True)
    assert eng.stats()["hits"] == 3_000_000


def test_config3_through_the_bench_entry_point(make_engine):
    """The same workload the way bench.py drives it: device-resident batches, submit / collect with three
    batches in flight (the partition of one overlapping the replay of the one before), every verdict,
    first_limited and final cell compared."""
    eng = _full_size(make_engine, 10_000_000, 1_000_000, steps=5, zipf=True, in_flight=True)
    assert eng.stats()["hits"] == 5_000_000


def test_config3_every_window_ends_between_two_batches(make_engine):
    """configs[2] at full size with the clock jumping past the pre-populated 30 s expiry between batches 2 and 3 (VERDICT
    r03 missing #3): every cell touched from then on is read as 0 and its first admitted hit resets value AND expiry
    (atomic_expiring_value.rs:19-24,36-42,87-99: the 16-byte write-back), the saturated Zipf head is admitted again.
    Three batches in flight through the bench's entry point; every verdict, first_limited and all 10 M cells compared."""
    nows = [NOW, NOW + 1000, NOW + 31_000_000, NOW + 31_001_000, NOW + 31_002_000]
    eng = _full_size(make_engine, 10_000_000, 1_000_000, steps=5, zipf=True, in_flight=True, nows=nows)
    rows = eng.dump_cells()
    reset = rows["expiry_us"] > np.uint64(NOW + 60_000_000)  # restarted windows: now + 60 s
    assert 0.05 * len(rows) < int(reset.sum()) < 0.5 * len(rows)  # (~1.3 M distinct keys in three Zipf batches)
    assert int(rows["value"][reset].max()) <= W.MAX_VALUE


def test_config3_a_third_of_the_windows_end_inside_the_run(make_engine):
    """configs[2] at full size where every third key's window ends 2.5 ms into the run (between batches 3 and 4 of 6):
    live, expired-and-reset and never-touched expired cells side by side in every bucket; blocking calls (one kernel
    sequence per batch) this time.  All 10 M cells compared."""
    eng = _full_size(make_engine, 10_000_000, 1_000_000, steps=6, zipf=True, early_third=True)
    rows = eng.dump_cells()
    n_reset = int((rows["expiry_us"] > np.uint64(NOW + 59_000_000)).sum())
    assert n_reset > 100_000


# ---- multi-counter requests and load_counters (the general resolver) ------------------------------
def _multi_batch(rng, n_req, rows, simple_ids, n_users, max_k=6, dup_prob=0.05, zero_k_prob=0.03):
    """Requests over a small universe: simple counters first, then qualified (in_memory.rs:105,121)."""
    qual_ids = [i for i in range(len(rows)) if i not in simple_ids]
    hits, off = [], [0]
    for _ in range(n_req):
        delta = int(rng.integers(0, 4)) if rng.random() < 0.3 else 1
        req = []
        if rng.random() >= zero_k_prob:
            k = int(rng.integers(1, max_k + 1))
            user = int(rng.zipf(1.5) - 1) % n_users if rng.random() < 0.7 else int(rng.integers(0, n_users))
            chosen = rng.permutation(len(rows))[:k]
            for lid in sorted(chosen, key=lambda x: (x not in simple_ids,)):
                lid = int(lid)
                if lid in simple_ids:
                    req.append((10_000_000 + lid, lid | RL_SIMPLE, delta))
                else:
                    key = int(W.splitmix64(np.array([lid * 100_003 + user], dtype=np.uint64))[0])
                    req.append((key, lid, delta))
            if rng.random() < dup_prob and qual_ids:
                q = [h for h in req if not (h[1] & RL_SIMPLE)]
                if q:
                    req.append(q[0])  # the same counter twice in one request
        hits.extend(req)
        off.append(len(hits))
    arr = np.zeros(len(hits), dtype=HIT_DTYPE)
    for i, h in enumerate(hits):
        arr[i] = h
    return arr, np.array(off, dtype=np.uint32)


@pytest.mark.parametrize("load", [False, True], ids=["noload", "load_counters"])
@pytest.mark.parametrize("seed", [21, 22])
def test_random_multi_counter_requests(make_engine, seed, load):
    rng = np.random.default_rng(seed)
    rows = [(40, 1), (5000, 10), (3, 1), (25, 10), (200, 60), (2, 60), (9, 0), (2**64 - 1, 3600)]
    simple_ids = {0, 1}
    eng, orc = pair(make_engine, rows, [(0, 10_000_000), (1, 10_000_001)])
    now = NOW
    for step in range(14):
        hits, off = _multi_batch(rng, int(rng.integers(1, 1500)), rows, simple_ids, n_users=60)
        run_both(eng, orc, hits, now, req_off=off, load_counters=load)
        now += int(rng.integers(0, 2 * SEC))
    assert_same_state(eng, orc, n_simple_expected=2)
    st = eng.stats()
    assert st["live_cells"] == orc.num_qualified() + 2


@pytest.mark.parametrize("one_launch", [False, True], ids=["general_pipeline", "k_gen_tiny"])
@pytest.mark.parametrize("load", [False, True], ids=["noload", "load_counters"])
@pytest.mark.parametrize("seed", [23, 24, 25])
def test_random_small_multi_counter_batches(make_engine, monkeypatch, seed, load, one_launch):
    """A few requests per call (up to 64 hits) — the per-request calls of the trait: the same random
    shapes as above (simple and qualified counters, duplicates inside a request, 0-second windows, a
    limit that never limits), many calls so that windows expire and counters are created, reached,
    dropped and recreated.  Through the general pipeline (default) and through the one-launch kernel."""
    monkeypatch.setenv("RL_GEN_TINY_MAX", "64" if one_launch else "0")
    rng = np.random.default_rng(seed)
    rows = [(40, 1), (5000, 10), (3, 1), (25, 10), (200, 60), (2, 60), (9, 0), (2**64 - 1, 3600)]
    simple_ids = {0, 1}
    eng, orc = pair(make_engine, rows, [(0, 10_000_000), (1, 10_000_001)])
    now = NOW
    calls = 0
    for step in range(250):
        hits, off = _multi_batch(rng, int(rng.integers(1, 14)), rows, simple_ids, n_users=12)
        if len(hits) > 64:
            continue
        run_both(eng, orc, hits, now, req_off=off, load_counters=load)
        calls += 1
        now += int(rng.integers(0, SEC // 2))
    assert calls > 150
    assert_same_state(eng, orc, n_simple_expected=2)
    st = eng.stats()
    assert st["live_cells"] == orc.num_qualified() + 2


def test_single_counter_requests_with_load_counters(make_engine):
    rng = np.random.default_rng(31)
    eng, orc = pair(make_engine, [(50, 2), (7, 1)])
    now = NOW
    for step in range(8):
        n = 5000
        idx = (rng.zipf(1.2, size=n) - 1) % 300
        hits = np.empty(n, dtype=HIT_DTYPE)
        hits["key"] = W.splitmix64(idx.astype(np.uint64))
        hits["limit"] = idx % 2
        hits["delta"] = rng.integers(0, 3, size=n)
        run_both(eng, orc, hits, now, load_counters=True)
        now += SEC // 2
    assert_same_state(eng, orc)


def test_unreached_counters_are_not_created(make_engine):
    """Appendix A (G): a request stops at its first limited counter when !load_counters."""
    eng, orc = pair(make_engine, [(0, 60), (10, 60)])
    hits = np.zeros(4, dtype=HIT_DTYPE)
    hits[0] = (100, 0, 1)  # limited (max 0)
    hits[1] = (200, 1, 1)  # never reached -> never created
    hits[2] = (300, 1, 1)  # second request, reaches and creates 300 ...
    hits[3] = (200, 1, 1)  # ... and 200
    off = np.array([0, 2, 4], dtype=np.uint32)
    v = run_both(eng, orc, hits[:2], NOW, req_off=off[:2])
    assert v[0] == 1
    assert_same_state(eng, orc)
    assert {int(k) for k in eng.dump_cells()["key"]} == {100}
    run_both(eng, orc, hits, NOW + 1, req_off=off)
    assert_same_state(eng, orc)
    assert {int(k) for k in eng.dump_cells()["key"]} == {100, 200, 300}


def test_huge_deltas_take_the_exact_path(make_engine):
    """Deltas that could carry out of the packed 40-bit batch sum are rerouted, not truncated."""
    rng = np.random.default_rng(41)
    eng, orc = pair(make_engine, [(2**40, 60), (2**64 - 1, 60)])
    n = 3000
    idx = rng.integers(0, 20, size=n)
    hits = np.empty(n, dtype=HIT_DTYPE)
    hits["key"] = W.splitmix64(idx.astype(np.uint64))
    hits["limit"] = idx % 2
    hits["delta"] = rng.integers(2**31, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    run_both(eng, orc, hits, NOW)
    eng.update_counters(hits, NOW + 1)
    orc.update_counters(hits, NOW + 1)
    run_both(eng, orc, hits, NOW + 2)
    assert_same_state(eng, orc)


def test_config5_shape_multi_namespace_trace_replay(make_engine):
    """BASELINE.json configs[4] at test size: 4 namespaces x 8 limits (2 simple
    + 6 qualified on 1-2 variables), windows {1, 10, 60, 3600} s, conditions as per-request
    applicability, k in [1, 8] counters per request, >= 16 batches with an advancing clock (crossing the
    1 s and 10 s windows) and an rl_sweep_expired between batches, replayed into the oracle."""
    rng = np.random.default_rng(55)
    windows = [1, 10, 60, 3600]
    rows, simple, limit_of = [], [], {}
    for ns in range(4):
        for j in range(8):
            lid = len(rows)
            is_simple = j < 2
            rows.append((int(rng.integers(20, 4000)) if j == 0 else int(rng.integers(2, 60)), windows[(ns + j) % 4]))
            limit_of[(ns, j)] = lid
            if is_simple:
                simple.append((lid, 20_000_000 + lid))
    eng, orc = pair(make_engine, rows, simple, capacity_cells=1 << 17, max_batch_hits=1 << 16)
    now = NOW
    for step in range(18):
        n_req = int(rng.integers(500, 4000))
        hits, off = [], [0]
        for _ in range(n_req):
            ns = int(rng.integers(0, 4))
            user = int(rng.zipf(1.4) - 1) % 500
            path = int(rng.integers(0, 6))
            delta = 1 if rng.random() < 0.8 else int(rng.integers(0, 5))
            applies = [j for j in range(8) if rng.random() < (0.9 if j < 2 else 0.45)][:8] or [0]
            req = []
            for j in sorted(applies, key=lambda x: (x >= 2,)):  # simple first (in_memory.rs:105,121)
                lid = limit_of[(ns, j)]
                if j < 2:
                    req.append((20_000_000 + lid, lid | RL_SIMPLE, delta))
                else:  # qualified on user (odd j: user and path)
                    material = lid * 1_000_003 + user * 7 + (path if j % 2 else 0)
                    req.append((int(W.splitmix64(np.array([material], dtype=np.uint64))[0]), lid, delta))
            hits.extend(req)
            off.append(len(hits))
        arr = np.zeros(len(hits), dtype=HIT_DTYPE)
        for i, h in enumerate(hits):
            arr[i] = h
        run_both(eng, orc, arr, now, req_off=np.array(off, dtype=np.uint32), load_counters=bool(step % 3 == 2))
        now += int(rng.integers(SEC // 4, 3 * SEC))
        if step % 2:
            assert eng.sweep_expired(now) == orc.sweep_expired(now)
    assert_same_state(eng, orc, n_simple_expected=8)


def test_resize_answers_table_full_without_losing_a_counter(make_engine):
    """rl_resize: a table past its occupancy bound is grown in place (rehash into a larger one), every
    counter keeps its value and window; shrinking below twice the live cells is refused."""
    from limitador_amd.engine import EngineError

    rng = np.random.default_rng(23)
    eng, orc = pair(make_engine, [(7, 60), (3, 10)], simple_keys=[(1, 9_000_001)], capacity_cells=4096,
                    max_batch_hits=4096)
    keys = W.splitmix64(np.arange(1, 3301, dtype=np.uint64))

    def batch(lo, hi):
        h = np.empty(hi - lo, dtype=HIT_DTYPE)
        h["key"], h["limit"], h["delta"] = keys[lo:hi], 0, rng.integers(1, 4, size=hi - lo)
        return h

    run_both(eng, orc, batch(0, 1400), NOW)
    run_both(eng, orc, batch(700, 2800), NOW + 1)
    run_both(eng, orc, batch(2600, 3200), NOW + 2)  # 3201 live cells of 4096: past 3/4
    with pytest.raises(EngineError) as e:
        eng.check_and_update(batch(3200, 3300), NOW + 3)  # refused up front, nothing applied
    assert e.value.code == -4
    before = np.sort(eng.dump_cells(), order="key")
    eng.resize(1 << 14)
    assert eng.stats()["capacity_cells"] == 1 << 14
    assert np.array_equal(before, np.sort(eng.dump_cells(), order="key"))
    run_both(eng, orc, batch(3200, 3300), NOW + 3)
    run_both(eng, orc, batch(0, 3300), NOW + 4)
    h = np.zeros(1, dtype=HIT_DTYPE)
    h[0] = (9_000_001, 1 | RL_SIMPLE, 1)
    run_both(eng, orc, h, NOW + 5)
    with pytest.raises(EngineError) as e:
        eng.resize(4096)  # 3301 live cells > 4096 / 2
    assert e.value.code == -1
    eng.resize(8192)  # shrinking is fine while the table stays at most half full
    assert eng.stats()["capacity_cells"] == 8192
    run_both(eng, orc, batch(100, 2700), NOW + 61 * SEC)  # every window has rolled over
    assert_same_state(eng, orc, n_simple_expected=1)


@pytest.mark.parametrize("cap,n_keys", [(4096, 2900), (1 << 16, 44_000), (1024, 740)])
def test_compaction_in_place_keeps_every_counter_reachable(make_engine, cap, n_keys):
    """rl_compact at the same capacity closes the gaps in place (k_compact_mark / k_compact_shift): dense tables
    (long probe clusters, clusters that wrap around the end of the table), three rounds of sweep -> compact -> refill;
    after each, every live counter is found again with its value and window and the dropped ones start fresh
    (in_memory.rs:122-127), exactly as in the oracle."""
    rng = np.random.default_rng(cap)
    eng, orc = pair(make_engine, [(9, 60), (4, 2)], simple_keys=[(1, 9_000_001)], capacity_cells=cap,
                    max_batch_hits=max(n_keys, 64))
    keys = W.splitmix64(np.arange(1, n_keys + 1, dtype=np.uint64))

    def batch(idx):
        h = np.empty(len(idx), dtype=HIT_DTYPE)
        h["key"], h["limit"], h["delta"] = keys[idx], (idx % 2).astype(np.uint32), rng.integers(1, 3, size=len(idx))
        return h

    now = NOW
    for _ in range(3):
        # windows of 60 s on even keys, 2 s on odd ones: 3 s on, the odd ones are swept (half of the table -> tombstones)
        run_both(eng, orc, batch(rng.permutation(n_keys)), now)
        now += 3 * SEC
        assert eng.sweep_expired(now) == orc.sweep_expired(now)
        before = np.sort(eng.dump_cells(), order="key")
        st0 = eng.stats()
        eng.compact()
        st = eng.stats()
        assert st["tombstones"] == 0 and st["live_cells"] == len(before) and st["capacity_cells"] == cap
        assert st["rebuilds"] == st0["rebuilds"] + 1
        assert np.array_equal(before, np.sort(eng.dump_cells(), order="key"))
        assert_same_state(eng, orc, n_simple_expected=1)
        # every key again: the kept ones continue their window, the swept ones are created anew
        run_both(eng, orc, batch(rng.permutation(n_keys)), now + 1)
        assert_same_state(eng, orc, n_simple_expected=1)
        now += 58 * SEC  # the rest of the 60-second windows ends too
        assert eng.sweep_expired(now) == orc.sweep_expired(now)
        eng.compact()
        assert_same_state(eng, orc, n_simple_expected=1)


def test_auto_grow_doubles_the_table_instead_of_refusing(make_engine):
    rng = np.random.default_rng(29)
    eng, orc = pair(make_engine, [(5, 60), (2, 1)], capacity_cells=1024, max_batch_hits=4096, auto_grow=True)
    keys = W.splitmix64(np.arange(1, 20_001, dtype=np.uint64))
    now = NOW
    for step in range(12):
        idx = rng.integers(0, min(20_000, 2000 * (step + 1)), size=3000)
        h = np.empty(3000, dtype=HIT_DTYPE)
        h["key"], h["limit"], h["delta"] = keys[idx], (idx % 2).astype(np.uint32), rng.integers(0, 3, size=3000)
        run_both(eng, orc, h, now)
        now += int(rng.integers(0, SEC))
    st = eng.stats()
    assert st["capacity_cells"] >= 32768 and st["rebuilds"] >= 5
    assert_same_state(eng, orc)


def _big_multi_batch(rng, n_req, n_users, n_ns=3):
    """Vectorised: requests of 1-4 counters — the namespace's simple counter first, then qualified ones on a
    Zipf user (limit 2 + 3 * ns + j) — as (hits, req_off)."""
    ns = rng.integers(0, n_ns, size=n_req)
    user = ((rng.zipf(1.3, size=n_req) - 1) % n_users).astype(np.uint64)
    has_simple = rng.random(n_req) < 0.6
    has_q = rng.random((n_req, 3)) < np.array([0.8, 0.5, 0.3])
    delta = np.where(rng.random(n_req) < 0.85, 1, rng.integers(0, 4, size=n_req)).astype(np.uint32)
    k = has_simple.astype(np.int64) + has_q.sum(axis=1)
    off = np.zeros(n_req + 1, dtype=np.uint32)
    np.cumsum(k, out=off[1:])
    hits = np.zeros(int(off[-1]), dtype=HIT_DTYPE)
    pos = off[:-1].astype(np.int64).copy()
    sel = np.nonzero(has_simple)[0]
    hits["key"][pos[sel]] = 30_000_000 + ns[sel]
    hits["limit"][pos[sel]] = ns[sel].astype(np.uint32) | RL_SIMPLE
    hits["delta"][pos[sel]] = delta[sel]
    pos[sel] += 1
    for j in range(3):
        sel = np.nonzero(has_q[:, j])[0]
        lid = (n_ns + 3 * ns[sel] + j).astype(np.uint64)
        hits["key"][pos[sel]] = W.splitmix64(lid * np.uint64(1_000_003) + user[sel])
        hits["limit"][pos[sel]] = lid.astype(np.uint32)
        hits["delta"][pos[sel]] = delta[sel]
        pos[sel] += 1
    return hits, off


@pytest.mark.parametrize("load", [False, True], ids=["noload", "load_counters"])
@pytest.mark.parametrize("bucket_log2", [None, 8], ids=["default_buckets", "long_buckets"])
def test_large_multi_counter_batches_against_the_oracle(make_engine, monkeypatch, bucket_log2, load):
    """The general resolver at a size where its machinery is exercised: hundreds of thousands of hits per call,
    buckets beyond 512 hits (re-read path; forced longer still with RL_GEN_BUCKET_LOG2), several pieces per bucket
    and per hot key, a cold first call (overflow, promotion of the heavy keys, retry), windows that run out
    between calls, duplicates of a counter inside a request — every verdict, first_limited and cell against the
    oracle."""
    if bucket_log2 is not None:
        monkeypatch.setenv("RL_GEN_BUCKET_LOG2", str(bucket_log2))
    rng = np.random.default_rng(314 + (bucket_log2 or 0) + int(load))
    n_ns = 3
    rows = [(10**9, 60), (50_000, 10), (3000, 1)] + [(int(rng.integers(3, 400)), [1, 10, 60][j % 3]) for j in range(3 * n_ns)]
    simple = [(ns, 30_000_000 + ns) for ns in range(n_ns)]
    n_req = 120_000 if load else 250_000
    eng, orc = pair(make_engine, rows, simple, capacity_cells=1 << 20, max_batch_hits=1 << 20)
    now = NOW
    for step in range(4):
        hits, off = _big_multi_batch(rng, n_req - 1000 * step, n_users=40_000, n_ns=n_ns)
        if step == 2:  # the same counter twice in one request, a few thousand times
            dup = rng.integers(0, len(off) - 1, size=3000)
            dup = dup[(off[dup + 1] - off[dup]) >= 2]
            hits["key"][off[dup + 1] - 1] = hits["key"][off[dup + 1] - 2]
            hits["limit"][off[dup + 1] - 1] = hits["limit"][off[dup + 1] - 2]
        run_both(eng, orc, hits, now, req_off=off, load_counters=load)
        now += [SEC // 3, 2 * SEC, 11 * SEC, 1][step]
    assert_same_state(eng, orc, n_simple_expected=n_ns)
