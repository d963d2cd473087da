"""snapshot files (rl_snapshot_save / _load) and the cross-node merge (rl_merge_cells /
rl_export_local) against the oracle's restatement of CrCounterValue (cr_counter_value.rs:81-113, pinned by the
reference's vectors in tests/test_oracle_golden.py).  Needs a MI355X."""
import numpy as np
import pytest

import oracle
from limitador_amd import workloads as W
from limitador_amd.wire import CELL_ROW_DTYPE, HIT_DTYPE, RL_SIMPLE
from test_gpu_parity import NOW, SEC, assert_same_state, make_engine, pair, run_both  # noqa: F401

pytestmark = pytest.mark.gpu


def test_snapshot_file_round_trip(make_engine, tmp_path):
    """Dump to a file, load into a FRESH engine (another table size, another hash seed): the same cells, and the
    restored engine goes on deciding exactly like the oracle that never stopped."""
    rng = np.random.default_rng(51)
    rows = [(30, 60), (5, 10), (2**64 - 1, 3600), (7, 0)]
    eng, orc = pair(make_engine, rows, simple_keys=[(0, 9_000_777)], capacity_cells=1 << 14)
    keys = W.splitmix64(np.arange(1, 4001, dtype=np.uint64))

    def batch(n):
        idx = rng.integers(0, 4000, size=n)
        h = np.empty(n, dtype=HIT_DTYPE)
        h["key"], h["limit"], h["delta"] = keys[idx], (1 + idx % 3).astype(np.uint32), rng.integers(0, 3, size=n)
        sm = rng.random(n) < 0.05
        h["key"][sm], h["limit"][sm] = 9_000_777, 0 | RL_SIMPLE
        return h

    now = NOW
    for _ in range(4):
        run_both(eng, orc, batch(3000), now)
        now += SEC // 2
    path = tmp_path / "counters.rlsnap"
    eng.snapshot_save(path)
    before = np.sort(eng.dump_cells(), order="key")
    fresh = make_engine(capacity_cells=1 << 15, hash_seed=0x1234_5678_9ABC_DEF1)
    fresh.snapshot_load(path)
    assert np.array_equal(before, np.sort(fresh.dump_cells(), order="key"))
    assert fresh.stats()["live_cells"] == len(before)
    for _ in range(4):  # the limit table came with the snapshot
        run_both(fresh, orc, batch(3000), now)
        now += SEC // 2
    assert_same_state(fresh, orc, n_simple_expected=1)


def test_merge_cells_matches_cr_counter_value(make_engine):
    """Node 0 = the engine; nodes 1..3 = simulated peers (CrCounterValue per counter).  Local increments
    (update_counter = AtomicExpiringValue::update = inc_at), merges of what the peers report (their local_values),
    expiry inside the merge, counters first heard of from a peer, a peer's memory of our own value: after every
    step every counter reads like the oracle's CrCounterValue, and what the engine exports is its local part."""
    from oracle import CrCounterValue as Cr

    rng = np.random.default_rng(52)
    U64 = 2**64 - 1
    LONG, SHORT = 60 * SEC, 2 * SEC
    eng = make_engine(capacity_cells=1 << 13)
    eng.set_limits([(U64, 60), (U64, 2)])
    keys = [int(k) for k in W.splitmix64(np.arange(1, 301, dtype=np.uint64))]
    window_of = {k: (LONG if i % 4 else SHORT) for i, k in enumerate(keys)}   # a quarter of the counters: 2 s windows
    limit_of = {k: (0 if window_of[k] == LONG else 1) for k in keys}
    ours = {}                                   # key -> Cr(ourselves = 0): node 0 as the reference holds it
    restarts = [0, 0]                           # local restarts seen; of those, with a stale `others` part the restart keeps
    peers = {p: {} for p in (1, 2, 3)}          # actor -> key -> Cr(ourselves = actor)
    now = NOW

    def check():
        cells = {int(r["key"]): r for r in eng.dump_cells()}
        for k, c in ours.items():
            want = c.read_at(now)
            r = cells.get(k)
            got = 0 if r is None or int(r["expiry_us"]) <= now else int(r["value"])
            assert got == want, (k, got, want)
            if want:
                assert int(r["expiry_us"]) == c.expiry_us, (k, int(r["expiry_us"]), c.expiry_us)
        exported = {int(r["key"]): int(r["value"]) for r in eng.export_local(now)}
        for k, c in ours.items():
            if c.expiry_us > now:
                assert exported.get(k) == c.local_value, (k, exported.get(k), c.local_value)

    for step in range(40):
        op = rng.choice(["local", "peer_inc", "merge", "merge", "echo"])
        if op == "local":
            # Every counter, the 2-second windows included.  Within a window the update adds to our own part.  A LOCAL RESTART
            # of an expired window follows CrCounterValue::inc_at (cr_counter_value.rs:53-59): our own value becomes the
            # increment, the `others` of the window that ended stay until a merge resets them (:85-87,144-149) — the engine
            # does the same since round 5 (the general resolver's resolve + commit, rl_general.hpp; rounds 2-4 pinned the
            # opposite here as a deviation).
            ks = [keys[i] for i in rng.choice(len(keys), size=60, replace=False)]
            h = np.zeros(len(ks), dtype=HIT_DTYPE)
            for i, k in enumerate(ks):
                d = int(rng.integers(1, 5))
                h[i] = (k, limit_of[k], d)
                c = ours.get(k)
                if c is None:
                    c = ours[k] = Cr(0, U64, now + window_of[k])   # created by the update: (0, now + window)
                elif c.expiry_us <= now:
                    restarts[0] += 1
                    c.inc_at(0, window_of[k], now)                  # (the restart alone, to see what it keeps)
                    if c.read_at(now):
                        restarts[1] += 1
                c.inc_at(d, window_of[k], now)
            eng.update_counters(h, now)
        elif op == "peer_inc":
            p = int(rng.integers(1, 4))
            for i in rng.choice(len(keys), size=80, replace=False):
                k = keys[i]
                c = peers[p].get(k)
                if c is None:
                    c = peers[p][k] = Cr(p, U64, now + window_of[k])
                c.inc_at(int(rng.integers(1, 9)), window_of[k], now)
        elif op == "merge":
            p = int(rng.integers(1, 4))
            ks = [k for k in peers[p] if rng.random() < 0.7]
            if ks:
                rows = np.zeros(len(ks), dtype=CELL_ROW_DTYPE)
                for i, k in enumerate(ks):
                    c = peers[p][k]
                    rows[i] = (k, limit_of[k], 0, c.local_value, c.expiry_us)   # local_values(): (expiry, actor, value)
                    mine = ours.get(k)
                    incoming = Cr.from_values(c.expiry_us, {p: c.local_value})
                    if mine is None:
                        if c.expiry_us > now:   # first heard of from a peer
                            mine = ours[k] = Cr(0, U64, c.expiry_us)
                            mine.merge_at(incoming, now)
                    else:
                        mine.merge_at(incoming, now)
                eng.merge_cells(0, p, rows, now)
        else:  # a replica echoes what it remembers of OUR value: larger only after we lost state
            ks = [k for k in ours if window_of[k] == LONG and rng.random() < 0.2]
            if ks:
                rows = np.zeros(len(ks), dtype=CELL_ROW_DTYPE)
                for i, k in enumerate(ks):
                    c = ours[k]
                    remembered = c.local_value + int(rng.integers(0, 3)) - 1 if c.local_value else 0
                    rows[i] = (k, limit_of[k], 0, max(0, remembered), c.expiry_us)
                    c.merge_at(Cr.from_values(c.expiry_us, {0: max(0, remembered)}), now)
                eng.merge_cells(0, 0, rows, now)
        check()
        now += int(rng.choice([0, 1000, SEC // 3, SEC]))
    assert len(ours) > 200
    assert restarts[0] >= 10 and restarts[1] >= 3, restarts   # local restarts that kept a stale peer part were exercised


def test_a_local_window_restart_keeps_the_stale_peer_part_like_cr_counter_value(make_engine):
    """CrCounterValue::inc_at (cr_counter_value.rs:53-59) walked number by number — the case rounds 2-4 pinned as a
    DEVIATION: a window that is restarted by a LOCAL update stores the increment into our own value and leaves `others`
    alone, so the next read is increment + what the peers had contributed to the OLD window, and a later report of that peer
    only counts if it is larger than the stale figure (:96-110).  The engine now does exactly that (from its first
    rl_merge_cells on its counters go through the general resolver, whose per-cell resolve reads the peers' part of the
    window that ended and whose commit moves their entries on to the new window)."""
    from oracle import CrCounterValue as Cr

    U64 = 2**64 - 1
    W2 = 2 * SEC
    eng = make_engine(capacity_cells=1 << 10)
    eng.set_limits([(U64, 2)])
    k = 0xABCDEF
    ref = Cr(0, U64, NOW + W2)

    def eng_read(now):
        r = {int(x["key"]): x for x in eng.dump_cells()}.get(k)
        return 0 if r is None or int(r["expiry_us"]) <= now else int(r["value"])

    def one(key, limit, v, exp):
        row = np.zeros(1, dtype=CELL_ROW_DTYPE)
        row[0] = (key, limit, 0, v, exp)
        return row

    def hit(d):
        h = np.zeros(1, dtype=HIT_DTYPE)
        h[0] = (k, 0, d)
        return h

    def own(now):
        return {int(r["key"]): int(r["value"]) for r in eng.export_local(now)}.get(k)

    t0 = NOW
    # peer 1 reports 5 for the window that ends at t0 + 2 s; then 3 local hits: both sides read 8
    eng.merge_cells(0, 1, one(k, 0, 5, t0 + W2), t0)
    ref.merge_at(Cr.from_values(t0 + W2, {1: 5}), t0)
    eng.update_counters(hit(3), t0)
    ref.inc_at(3, W2, t0)
    assert eng_read(t0) == ref.read_at(t0) == 8
    # the window expires: both read 0
    t1 = t0 + 3 * SEC
    assert eng_read(t1) == ref.read_at(t1) == 0
    # a LOCAL update restarts it: both still add peer 1's 5 of the old window
    eng.update_counters(hit(2), t1)
    ref.inc_at(2, W2, t1)
    assert ref.read_at(t1) == eng_read(t1) == 7
    # peer 1 reports 1 for ITS new window: the larger stale 5 stays on both sides; then 9: only the excess counts
    eng.merge_cells(0, 1, one(k, 0, 1, t1 + W2), t1)
    ref.merge_at(Cr.from_values(t1 + W2, {1: 1}), t1)
    assert ref.read_at(t1) == eng_read(t1) == 7
    eng.merge_cells(0, 1, one(k, 0, 9, t1 + W2), t1)
    ref.merge_at(Cr.from_values(t1 + W2, {1: 9}), t1)
    assert ref.read_at(t1) == eng_read(t1) == 11
    # what each side would send to its peers as "our own part" is the same: 2
    assert ref.local_value == own(t1) == 2
    # once the restarted window has expired too, a merge resets both (:85-87): the stale parts are gone
    t2 = t1 + 3 * SEC
    eng.merge_cells(0, 1, one(k, 0, 4, t2 + W2), t2)
    ref.merge_at(Cr.from_values(t2 + W2, {1: 4}), t2)
    assert eng_read(t2) == ref.read_at(t2) == 4
    assert ref.local_value == own(t2) == 0


def test_admission_after_a_local_restart_counts_the_stale_peer_part(make_engine):
    """The same rule where it decides verdicts (distributed/mod.rs:93-168 is in_memory.rs's check_and_update over
    CrCounterValue::read / inc_at): max 10, a peer has contributed 6 to the window that ended.  Five hits of delta 1 in ONE
    batch after the expiry: the first reads 0 and restarts the window, the next read 1 + 6, 2 + 6, 3 + 6 — and the fifth,
    4 + 6 + 1 > 10, is limited.  Then `remaining` / `expires_in` of a load_counters request, and a request of two counters
    of which only one has a peer part.  The engine's pipelined hot-path entry refuses an engine with peer state."""
    import ctypes as C

    from limitador_amd.engine import EngineError
    from oracle import CrCounterValue as Cr

    eng = make_engine(capacity_cells=1 << 10)
    eng.set_limits([(10, 2), (100, 2)])
    k, k2 = 0x5151, 0x5252
    W2 = 2 * SEC

    def rows(*items):
        r = np.zeros(len(items), dtype=CELL_ROW_DTYPE)
        for i, it in enumerate(items):
            r[i] = it
        return r

    def hits(*items):
        h = np.zeros(len(items), dtype=HIT_DTYPE)
        for i, it in enumerate(items):
            h[i] = it
        return h

    t0 = NOW
    eng.merge_cells(0, 1, rows((k, 0, 0, 6, t0 + W2)), t0)
    ref = Cr(0, 10, t0 + W2)
    ref.merge_at(Cr.from_values(t0 + W2, {1: 6}), t0)
    eng.update_counters(hits((k, 0, 2), (k2, 1, 1)), t0)
    ref.inc_at(2, W2, t0)
    t1 = t0 + 3 * SEC
    v, f, _, _ = eng.check_and_update(hits(*[(k, 0, 1)] * 5), t1)
    want = []
    for _ in range(5):  # in_memory.rs:259-264 over read_at, then inc_at for the admitted ones
        ok = ref.read_at(t1) + 1 <= 10
        want.append(0 if ok else 1)
        if ok:
            ref.inc_at(1, W2, t1)
    assert list(v) == want == [0, 0, 0, 0, 1]
    cell = {int(r["key"]): r for r in eng.dump_cells()}[k]
    assert int(cell["value"]) == ref.read_at(t1) == 10 and int(cell["expiry_us"]) == t1 + W2
    # load_counters on the saturated counter + a counter without peers, as ONE request: limited by the first, nothing applied
    v, f, rem, exp = eng.check_and_update(hits((k, 0, 1), (k2, 1, 1)), t1 + 1, req_off=np.array([0, 2], dtype=np.uint32), load_counters=True)
    assert list(v) == [1] and list(f) == [0] and int(rem[0]) == 0 and int(exp[0]) == W2 - 1
    assert int({int(r["key"]): r for r in eng.dump_cells()}[k]["value"]) == 10
    with pytest.raises(EngineError):
        eng.submit_device(C.c_void_p(8), 1, t1, C.c_void_p(8))
