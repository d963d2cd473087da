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
    ours = {}                                   # key -> Cr(ourselves = 0): node 0 as the reference would hold it
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
            # long-window counters only: a LOCAL restart of a window forgets the peers' part here, the reference's
            # distributed storage keeps it (stated deviation, rl_engine.h); within a window both add to our own part
            ks = [keys[i] for i in rng.choice(len(keys), size=60, replace=False) if window_of[keys[i]] == LONG]
            h = np.zeros(len(ks), dtype=HIT_DTYPE)
            for i, k in enumerate(ks):
                d = int(rng.integers(1, 5))
                h[i] = (k, limit_of[k], d)
                c = ours.get(k)
                if c is None:
                    c = ours[k] = Cr(0, U64, now + window_of[k])   # created by the update: (0, now + window)
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
