"""Committed golden traces (tests/golden/*.npz, made by tests/golden/make_golden.py).
CPU: the oracle must keep reproducing them (it may not drift from what the reference's vectors pinned).
GPU: the HIP engine reproduces them through the C ABI with no oracle in the loop."""
import glob
import os

import numpy as np
import pytest

from limitador_amd.wire import RL_SIMPLE

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def replay(storage, t, final_rows):
    """storage: oracle.OracleStorage or limitador_amd.engine.Engine (same method names)."""
    storage.set_limits([tuple(int(x) for x in r) for r in t["limit_rows"]])
    for limit, key in t["simple"]:
        storage.add_counter(int(limit) | RL_SIMPLE, int(key))
    for i in range(int(t["n_events"])):
        kind = str(t[f"e{i}_kind"])
        if kind == "check":
            load = bool(t[f"e{i}_load"])
            req_off = t[f"e{i}_req_off"] if f"e{i}_req_off" in t else None
            v, f, r, e = storage.check_and_update(t[f"e{i}_hits"], int(t[f"e{i}_now"]), req_off=req_off,
                                                  load_counters=load)
            assert np.array_equal(v, t[f"e{i}_verdict"]), f"event {i}: verdicts"
            assert np.array_equal(f, t[f"e{i}_first"]), f"event {i}: first_limited"
            if load:
                assert np.array_equal(r, t[f"e{i}_remaining"]), f"event {i}: remaining"
                assert np.array_equal(e, t[f"e{i}_expires"]), f"event {i}: expires_in"
        elif kind == "update":
            storage.update_counters(t[f"e{i}_hits"], int(t[f"e{i}_now"]))
        elif kind == "within":
            assert np.array_equal(storage.is_within_limits(t[f"e{i}_hits"], int(t[f"e{i}_now"])), t[f"e{i}_within"])
        elif kind == "sweep":
            assert storage.sweep_expired(int(t[f"e{i}_now"])) == int(t[f"e{i}_removed"])
        elif kind == "delete":
            storage.delete_counters(int(t[f"e{i}_limit"]))
        elif kind == "clear":
            storage.clear()
    got = final_rows(storage)
    want = t["final"]
    assert got.shape == want.shape and np.array_equal(got, want), "final table differs"


def _oracle_final(t):
    def f(orc):
        rows = []
        keys = set()
        for i in range(int(t["n_events"])):
            if f"e{i}_hits" in t and str(t[f"e{i}_kind"]) in ("check", "update"):
                h = t[f"e{i}_hits"]
                keys.update(int(k) for k in h["key"][(h["limit"] & RL_SIMPLE) == 0])
        for k in sorted(keys):
            got = orc.peek(k)
            if got is not None:
                rows.append((k, got[2], got[0], got[1]))
        for limit, key in t["simple"]:
            got = orc.peek_simple(int(limit) | RL_SIMPLE)
            if got is not None:
                rows.append((int(key), int(limit) | RL_SIMPLE, got[0], got[1]))
        return np.array(rows, dtype=np.uint64).reshape(-1, 4)
    return f


def _engine_final(eng):
    rows = eng.dump_cells()
    q = rows[(rows["limit"] & RL_SIMPLE) == 0]
    s = rows[(rows["limit"] & RL_SIMPLE) != 0]
    q = q[np.argsort(q["key"])]
    s = s[np.argsort(s["key"])]
    out = [(int(r["key"]), int(r["limit"]), int(r["value"]), int(r["expiry_us"])) for r in q]
    out += [(int(r["key"]), int(r["limit"]), int(r["value"]), int(r["expiry_us"])) for r in s]
    return np.array(out, dtype=np.uint64).reshape(-1, 4)


def test_golden_traces_exist():
    assert len(GOLDEN) >= 3


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_the_golden_traces(path):
    import oracle

    t = np.load(path)
    orc = oracle.OracleStorage()
    try:
        replay(orc, t, _oracle_final(t))
    finally:
        orc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_engine_reproduces_the_golden_traces(path):
    from limitador_amd.engine import Engine

    t = np.load(path)
    with Engine(capacity_cells=1 << 14, max_batch_hits=1 << 14) as eng:
        replay(eng, t, _engine_final)
