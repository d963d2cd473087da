"""Pins the CPU oracle against the reference's own vectors for this path.

Cell-level vectors: limitador/src/storage/atomic_expiring_value.rs:175-245.
Storage / facade vectors: tests/scenarios.py (each cites its reference test).
"""
import ctypes as C

import numpy as np
import pytest

import oracle
import scenarios
from helpers.limiter import TestsLimiter

SEC = 1_000_000
NOW = 1_700_000_000 * SEC


def cell(value, expiry):
    return oracle.Cell(value, expiry)


# atomic_expiring_value.rs:181-186
def test_returns_value_when_valid():
    L = oracle.lib()
    assert L.lo_cell_value_at(C.byref(cell(42, NOW)), NOW - 1 * SEC) == 42


# :188-193
def test_returns_default_when_expired():
    L = oracle.lib()
    assert L.lo_cell_value_at(C.byref(cell(42, NOW - 1 * SEC)), NOW) == 0


# :195-200 — expiry == now is expired
def test_returns_default_on_expiry():
    L = oracle.lib()
    assert L.lo_cell_value_at(C.byref(cell(42, NOW)), NOW) == 0


# :202-208
def test_updates_when_valid():
    L = oracle.lib()
    c = cell(42, NOW + 1 * SEC)
    L.lo_cell_update(C.byref(c), 3, 10 * SEC, NOW)
    assert L.lo_cell_value_at(C.byref(c), NOW - 1 * SEC) == 45


# :210-217
def test_updates_when_expired():
    L = oracle.lib()
    c = cell(42, NOW)
    assert L.lo_cell_ttl_us(C.byref(c), NOW) == 0
    L.lo_cell_update(C.byref(c), 3, 10 * SEC, NOW)
    assert L.lo_cell_value_at(C.byref(c), NOW - 1 * SEC) == 3
    assert c.expiry_us == NOW + 10 * SEC


# :219-237 — the two threads' updates in either order give 2 or 3
def test_overlapping_updates_either_order():
    L = oracle.lib()
    for order in ((0, 1), (1, 0)):
        c = cell(42, NOW + 10 * SEC)
        ops = [(1, 1 * SEC, NOW), (2, 1 * SEC, NOW + 11 * SEC)]
        for i in order:
            L.lo_cell_update(C.byref(c), *ops[i])
        assert c.value in (2, 3)


# :239-244
def test_size_of_struct():
    assert C.sizeof(oracle.Cell) == 16


@pytest.mark.parametrize("scenario", scenarios.ALL, ids=lambda f: f.__name__)
def test_reference_scenarios_on_oracle(scenario):
    st = oracle.OracleStorage()
    try:
        scenario(TestsLimiter(st))
    finally:
        st.close()


# ---- edge cases of SURVEY.md Appendix A, stated directly on the storage ---------------------

def _st(rows):
    st = oracle.OracleStorage()
    st.set_limits(rows)
    return st


def _hit(key, limit, delta=1):
    h = np.zeros(1, dtype=oracle.HIT_DTYPE)
    h[0] = (key, limit, delta)
    return h


# (B) delta > max on an empty qualified cell: limited, but the cell now exists as (0, now+w)
def test_denied_hit_still_creates_the_qualified_cell():
    st = _st([(10, 60)])
    v, first, _, _ = st.check_and_update(_hit(7, 0, 11), NOW)
    assert v[0] == 1 and first[0] == 0
    assert st.peek(7) == (0, NOW + 60 * SEC, 0)


# pre-created simple cell is (0, EPOCH): first admitted hit opens the window
def test_simple_cell_default_and_first_hit():
    st = _st([(5, 10)])
    st.add_counter(0 | oracle.SIMPLE_FLAG)
    assert st.peek_simple(0) == (0, 0)
    v, _, rem, exp = st.check_and_update(_hit(1, 0 | oracle.SIMPLE_FLAG), NOW, load_counters=True)
    assert v[0] == 0 and rem[0] == 4 and exp[0] == 0  # expires_in is read BEFORE the update
    assert st.peek_simple(0) == (1, NOW + 10 * SEC)


# simple counter without add_counter: the reference unwraps None (in_memory.rs:107)
def test_missing_simple_cell_is_an_error():
    st = _st([(5, 10)])
    with pytest.raises(oracle.OracleError) as e:
        st.check_and_update(_hit(1, 0 | oracle.SIMPLE_FLAG), NOW)
    assert e.value.code == oracle.LO_ERR_MISSING_SIMPLE


# (G) early return leaves later qualified counters uncreated when !load_counters
def test_early_return_skips_later_qualified_counters():
    st = _st([(0, 60), (10, 60)])
    hits = np.zeros(2, dtype=oracle.HIT_DTYPE)
    hits[0] = (100, 0, 1)
    hits[1] = (200, 1, 1)
    v, first, _, _ = st.check_and_update(hits, NOW, req_off=np.array([0, 2], dtype=np.uint32))
    assert v[0] == 1 and first[0] == 0
    assert st.peek(100) is not None and st.peek(200) is None
    # with load_counters every counter is visited (and created)
    st2 = _st([(0, 60), (10, 60)])
    v, first, rem, _ = st2.check_and_update(hits, NOW, req_off=np.array([0, 2], dtype=np.uint32), load_counters=True)
    assert v[0] == 1 and first[0] == 0 and list(rem) == [0, 9]
    assert st2.peek(200) == (0, NOW + 60 * SEC, 1)


# all-or-nothing: a limited request updates nothing
def test_limited_request_updates_nothing():
    st = _st([(10, 60), (1, 60)])
    hits = np.zeros(2, dtype=oracle.HIT_DTYPE)
    hits[0] = (1, 0, 2)
    hits[1] = (2, 1, 2)
    v, first, _, _ = st.check_and_update(hits, NOW, req_off=np.array([0, 2], dtype=np.uint32))
    assert v[0] == 1 and first[0] == 1
    assert st.peek(1)[0] == 0 and st.peek(2)[0] == 0


# clear() empties only the simple cells (in_memory.rs:198-201)
def test_clear_keeps_qualified_cells():
    st = _st([(5, 10), (5, 10)])
    st.add_counter(0 | oracle.SIMPLE_FLAG)
    st.update_counters(_hit(9, 1), NOW)
    st.clear()
    assert st.peek_simple(0) is None
    assert st.peek(9) == (1, NOW + 10 * SEC, 1)


# update_counter on a vacant simple cell creates (delta, now+w) (in_memory.rs:60-62)
def test_update_counter_creates_simple_cell():
    st = _st([(5, 10)])
    st.update_counters(_hit(1, 0 | oracle.SIMPLE_FLAG, 3), NOW)
    assert st.peek_simple(0) == (3, NOW + 10 * SEC)


# window rollover: expired cell keeps its stale state until the first ADMITTED update
def test_expired_cell_resets_only_on_admitted_update():
    st = _st([(3, 1)])
    st.update_counters(_hit(5, 0, 3), NOW)
    t2 = NOW + 2 * SEC
    v, _, _, _ = st.check_and_update(_hit(5, 0, 4), t2)  # 0 + 4 > 3: denied
    assert v[0] == 1 and st.peek(5) == (3, NOW + 1 * SEC, 0)  # untouched
    v, _, _, _ = st.check_and_update(_hit(5, 0, 2), t2)
    assert v[0] == 0 and st.peek(5) == (2, t2 + 1 * SEC, 0)


# ---- CrCounterValue (cr_counter_value.rs:176-303): the vectors behind rl_merge_cells ------------------------------
def test_cr_counter_value_vectors():
    """limitador/src/storage/distributed/cr_counter_value.rs:180-303, with explicit clocks (T = now, window 1 s)."""
    from oracle import CrCounterValue as Cr

    T, W, U64 = 1_700_000_000_000_000, 1_000_000, 2**64 - 1
    a = Cr(1, U64, T + W)                                  # local_increments_are_readable :180-187
    a.inc_at(3, W, T)
    assert a.read_at(T) == 3
    a.inc_at(2, W, T)
    assert a.read_at(T) == 5
    a = Cr(1, U64, T + W)                                  # local_increments_expire :189-198
    a.inc_at(3, W, T)
    assert a.read_at(T) == 3
    a.inc_at(2, W, T + W)
    assert a.read_at(T + W) == 2
    a = Cr(1, U64, T + W)                                  # other_increments_are_readable :200-207
    a.inc_actor_at(2, 3, W, T)
    assert a.read_at(T) == 3
    a.inc_actor_at(2, 2, W, T)
    assert a.read_at(T) == 5
    a = Cr(1, U64, T + W)                                  # other_increments_expire :209-218
    a.inc_actor_at(2, 3, W, T)
    a.inc_actor_at(2, 2, W, T + W)
    assert a.read_at(T + W) == 2

    def ab():
        x, y = Cr(1, U64, T + W), Cr(2, U64, T + W)
        x.inc_at(3, W, T)
        y.inc_at(2, W, T)
        return x, y

    a, b = ab()                                            # merges :220-229
    a.merge_at(b, T)
    assert a.read_at(T) == 5
    a, b = ab()                                            # merges_symetric :231-240
    b.merge_at(a, T)
    assert b.read_at(T) == 5
    a, b = ab()                                            # merges_overrides_with_larger_value :242-252
    b.inc_actor_at(1, 2, W, T)  # older value!
    b.merge_at(a, T)            # merges the 3
    assert b.read_at(T) == 5
    a, b = ab()                                            # merges_ignore_lesser_values :254-264
    b.inc_actor_at(1, 5, W, T)  # newer value!
    b.merge_at(a, T)            # ignores the 3 and keeps its own 5 for a
    assert b.read_at(T) == 7
    a = Cr(1, U64, T + 0)                                  # merge_ignores_expired_sets :266-275
    a.inc_at(3, 0, T)
    b = Cr(2, U64, T + W)
    b.inc_at(2, W, T)
    b.merge_at(a, T)
    assert b.read_at(T) == 2
    a = Cr(1, U64, T + 0)                                  # merge_ignores_expired_sets_symmetric :277-286
    a.inc_at(3, 0, T)
    b = Cr(2, U64, T + W)
    b.inc_at(2, W, T)
    a.merge_at(b, T)
    assert a.read_at(T) == 2
    a, b = Cr(1, U64, T + W), Cr(2, U64, T + 200_000)      # merge_uses_earliest_expiry :288-302
    a.inc_at(3, W, T)
    b.inc_at(2, W, T)
    a.merge_at(b, T)
    assert a.expiry_us - T <= 200_000
