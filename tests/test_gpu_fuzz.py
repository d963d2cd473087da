"""Randomised operation sequences over the whole C ABI (single- and multi-counter checks, load_counters,
update_counter, is_within_limits, get/delete/clear, sweep, compact, submit/collect) on small tables so
that probing, tombstones, compaction, hot-set churn and tiny batches all get exercised — every step
against the CPU oracle, final tables compared.  Needs a MI355X."""
import numpy as np
import pytest

from limitador_amd import workloads as W
from limitador_amd.wire import HIT_DTYPE, RL_SIMPLE
from test_gpu_parity import NOW, SEC, assert_same_state, make_engine, pair, run_both  # noqa: F401

pytestmark = pytest.mark.gpu


def _hits(rng, n, n_keys, key_limit, simple, p_simple=0.05, deltas=None):
    idx = (rng.zipf(1.3, size=n) - 1) % n_keys if rng.random() < 0.5 else rng.integers(0, n_keys, size=n)
    h = np.empty(n, dtype=HIT_DTYPE)
    h["key"] = W.splitmix64(idx.astype(np.uint64))
    h["limit"] = key_limit[idx]
    h["delta"] = rng.integers(0, 4, size=n) if deltas is None else deltas
    if simple:
        sm = rng.random(n) < p_simple
        which = rng.integers(0, len(simple), size=n)
        for q, (lid, key) in enumerate(simple):
            m = sm & (which == q)
            h["key"][m], h["limit"][m] = key, lid | RL_SIMPLE
    return h


@pytest.mark.parametrize("seed", range(12))
def test_random_operation_sequences(make_engine, seed):
    rng = np.random.default_rng(1000 + seed)
    n_limits = int(rng.integers(2, 9))
    rows = [(int(rng.integers(0, 60)) if rng.random() < 0.8 else 2**64 - 1, int(rng.choice([0, 1, 2, 10, 60])))
            for _ in range(n_limits)]
    n_simple = int(rng.integers(0, min(3, n_limits) + 1))
    simple = [(lid, 5_000_000 + lid) for lid in range(n_simple)]
    qual_ids = list(range(n_simple, n_limits)) or [0]
    cap = int(rng.choice([1 << 12, 1 << 13, 1 << 15]))
    eng, orc = pair(make_engine, rows, simple, capacity_cells=cap, max_batch_hits=1 << 14)
    n_keys = int(rng.integers(5, cap // 8))
    key_limit = np.array([qual_ids[i % len(qual_ids)] for i in range(n_keys)], dtype=np.uint32)
    if not n_simple and 0 not in qual_ids:
        qual_ids = [0]
    now = NOW
    for step in range(40):
        op = rng.choice(["k1", "k1", "k1", "multi", "load", "update", "within", "sweep", "delete", "clear", "compact", "get"],
                        p=[.22, .1, .1, .14, .1, .08, .08, .06, .03, .03, .03, .03])
        n = int(rng.choice([1, 3, 64, 500, 513, 2000, 6000]))
        if op == "k1":
            run_both(eng, orc, _hits(rng, n, n_keys, key_limit, simple, deltas=1 if rng.random() < 0.5 else None), now)
        elif op in ("multi", "load"):
            h = _hits(rng, n, n_keys, key_limit, simple, p_simple=0.0)
            cuts = np.sort(rng.integers(0, n + 1, size=max(1, n // 3)))
            off = np.concatenate([[0], cuts, [n]]).astype(np.uint32)
            # one delta per request; simple counters first inside a request
            hs, offs = [], [0]
            for a, b in zip(off[:-1], off[1:]):
                req = [(int(k), int(l), 0) for k, l in zip(h["key"][a:b], h["limit"][a:b])]
                if simple and rng.random() < 0.5:
                    lid, key = simple[int(rng.integers(0, len(simple)))]
                    req.insert(0, (key, lid | RL_SIMPLE, 0))
                d = int(rng.integers(0, 4))
                hs += [(k, l, d) for k, l, _ in req]
                offs.append(len(hs))
            arr = np.array(hs, dtype=HIT_DTYPE) if hs else np.zeros(0, dtype=HIT_DTYPE)
            run_both(eng, orc, arr, now, req_off=np.array(offs, dtype=np.uint32), load_counters=(op == "load"))
        elif op == "update":
            h = _hits(rng, n, n_keys, key_limit, simple)
            eng.update_counters(h, now)
            orc.update_counters(h, now)
        elif op == "within":
            h = _hits(rng, n, n_keys, key_limit, simple)
            assert np.array_equal(eng.is_within_limits(h, now), orc.is_within_limits(h, now))
        elif op == "sweep":
            assert eng.sweep_expired(now) == orc.sweep_expired(now)
        elif op == "delete":
            lid = int(rng.integers(0, n_limits))
            wire = lid | (RL_SIMPLE if lid < n_simple else 0)
            eng.delete_counters(wire)
            orc.delete_counters(wire)
            if lid < n_simple:  # the reference re-adds a limit before using it again (add_counter)
                eng.add_counter(wire, 5_000_000 + lid)
                orc.add_counter(wire, 5_000_000 + lid)
        elif op == "clear":
            eng.clear()
            orc.clear()
            for lid, key in simple:
                eng.add_counter(lid | RL_SIMPLE, key)
                orc.add_counter(lid | RL_SIMPLE, key)
        elif op == "compact":
            eng.compact()
        elif op == "get":
            lid = int(rng.integers(0, n_limits))
            wire = lid | (RL_SIMPLE if lid < n_simple else 0)
            a, b = eng.get_counters(wire, now), orc.get_counters(wire, now)
            assert len(a) == len(b)
        now += int(rng.choice([0, 1, 1000, SEC // 2, 3 * SEC]))
    assert_same_state(eng, orc, n_simple_expected=n_simple)
