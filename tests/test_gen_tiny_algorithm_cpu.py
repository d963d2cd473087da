"""The ALGORITHM of k_gen_tiny (limitador_amd/csrc/rl_general.hpp: resolve / create every hit's cell up
front, one sequential replay of the requests over copies of the cells, write back, drop the cells this
call created that no request reached) restated in Python and run against the CPU oracle on the request
shapes of the GPU stress test (tests/test_gpu_parity.py::test_random_small_multi_counter_batches).
No GPU: this pins the algorithm's equivalence with in_memory.rs:72-156; the kernel itself is compared
with the oracle by the GPU test."""
import numpy as np
import pytest

import oracle
from limitador_amd import workloads as W
from limitador_amd.wire import HIT_DTYPE, RL_SIMPLE

NOW, SEC, M64 = 1_700_000_000_000_000, 1_000_000, (1 << 64) - 1
ROWS = [(40, 1), (5000, 10), (3, 1), (25, 10), (200, 60), (2, 60), (9, 0), (2**64 - 1, 3600)]
SIMPLE_IDS = {0, 1}


def multi_batch(rng, n_req, n_users, max_k=6, dup_prob=0.05, zero_k_prob=0.03):
    """Requests over a small universe: simple counters first, then qualified (in_memory.rs:105,121)."""
    hits, off = [], [0]
    for _ in range(n_req):
        delta = int(rng.integers(0, 4)) if rng.random() < 0.3 else 1
        req = []
        if rng.random() >= zero_k_prob:
            k = int(rng.integers(1, max_k + 1))
            user = int(rng.zipf(1.5) - 1) % n_users if rng.random() < 0.7 else int(rng.integers(0, n_users))
            for lid in sorted(rng.permutation(len(ROWS))[:k], key=lambda x: (x not in SIMPLE_IDS,)):
                lid = int(lid)
                if lid in SIMPLE_IDS:
                    req.append((10_000_000 + lid, lid | RL_SIMPLE, delta))
                else:
                    req.append((int(W.splitmix64(np.array([lid * 100_003 + user], dtype=np.uint64))[0]), lid, delta))
            q = [h for h in req if not (h[1] & RL_SIMPLE)]
            if rng.random() < dup_prob and q:
                req.append(q[0])  # the same counter twice in one request
        hits.extend(req)
        off.append(len(hits))
    arr = np.zeros(len(hits), dtype=HIT_DTYPE)
    for i, h in enumerate(hits):
        arr[i] = h
    return arr, np.array(off, dtype=np.uint32)


def gen_tiny(table, hits, off, now, load):
    """table: key -> [value, expiry_us]; mirrors the kernel's three phases."""
    n = len(hits)
    created = set()
    for i in range(n):  # phase 1: every hit's cell exists (qualified ones are created: in_memory.rs:122-127)
        k, lim = int(hits["key"][i]), int(hits["limit"][i])
        if k not in table:
            assert not (lim & RL_SIMPLE), "a simple counter must pre-exist (in_memory.rs:106-107)"
            table[k] = [0, now + ROWS[lim][1] * SEC]
            created.add(k)
    cells = {int(k): {"v": table[int(k)][0], "e": table[int(k)][1], "dirty": False, "reached": False} for k in hits["key"]}
    verdict, first_limited = [], []
    remaining, expires_in = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
    for r in range(len(off) - 1):  # phase 2: one sequential replay
        b, e_ = int(off[r]), int(off[r + 1])
        first, stopped = -1, False
        for qualified_pass in (False, True):
            if stopped:
                break
            for j in range(b, e_):
                lim = int(hits["limit"][j])
                if bool(lim & RL_SIMPLE) == qualified_pass:
                    continue
                c = cells[int(hits["key"][j])]
                c["reached"] = True
                value = 0 if c["e"] <= now else c["v"]
                total = (value + int(hits["delta"][j])) & M64
                max_value = ROWS[lim & ~RL_SIMPLE][0]
                within = total <= max_value
                if load:
                    remaining[j] = max_value - total if within else 0
                    if first < 0 and not within:
                        first = j
                    expires_in[j] = c["e"] - now if c["e"] > now else 0
                elif not within:
                    first, stopped = j, True
                    break
        if first < 0:
            for qualified_pass in (False, True):
                for j in range(b, e_):
                    lim = int(hits["limit"][j])
                    if bool(lim & RL_SIMPLE) == qualified_pass:
                        continue
                    c = cells[int(hits["key"][j])]
                    if c["e"] <= now:
                        c["e"], c["v"] = now + ROWS[lim & ~RL_SIMPLE][1] * SEC, int(hits["delta"][j])
                    else:
                        c["v"] = (c["v"] + int(hits["delta"][j])) & M64
                    c["dirty"] = True
        verdict.append(0 if first < 0 else 1)
        first_limited.append(first)
    for k, c in cells.items():  # phase 3
        if c["dirty"]:
            table[k] = [c["v"], c["e"]]
        if not load and k in created and not c["reached"]:
            del table[k]
    return np.array(verdict, dtype=np.uint8), np.array(first_limited, dtype=np.int32), remaining, expires_in


@pytest.mark.parametrize("load", [False, True], ids=["noload", "load_counters"])
@pytest.mark.parametrize("seed", [23, 24, 25, 26])
def test_resolve_replay_writeback_equals_the_reference(seed, load):
    rng = np.random.default_rng(seed)
    orc = oracle.OracleStorage()
    orc.set_limits(ROWS)
    table = {}
    for lid in SIMPLE_IDS:
        orc.add_counter(lid | RL_SIMPLE, 10_000_000 + lid)
        table[10_000_000 + lid] = [0, 0]  # add_counter: (0, UNIX_EPOCH), in_memory.rs:38-44
    now, calls = NOW, 0
    for _ in range(250):
        hits, off = multi_batch(rng, int(rng.integers(1, 14)), n_users=12)
        if len(hits) > 64:
            continue
        v2, f2, r2, e2 = orc.check_and_update(hits, now, req_off=off, load_counters=load)
        v1, f1, r1, e1 = gen_tiny(table, hits, off, now, load)
        assert np.array_equal(v1, v2) and np.array_equal(f1, f2), f"call {calls}"
        if load:
            assert np.array_equal(r1, r2) and np.array_equal(e1, e2), f"call {calls}"
        calls += 1
        now += int(rng.integers(0, SEC // 2))
    assert calls > 150
    # the surviving cells are the oracle's
    assert len(table) - len(SIMPLE_IDS) == orc.num_qualified()
    for k, (v, e) in table.items():
        if k >= 10_000_000 and k < 10_000_010:
            continue
        assert (v, e) == orc.peek(k)[:2], k
