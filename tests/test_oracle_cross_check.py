"""A second, independent restatement of the reference path — pure-Python loops written straight from
SURVEY.md Appendix A / in_memory.rs:72-156, atomic_expiring_value.rs:19-46,76-99 — cross-checked against
the C oracle on seeded traces (small cases only: Python loops).  Two restatements written separately
from the same reference lines and agreeing bit for bit is what backs the C oracle between the points
the reference's own vectors pin (tests/test_oracle_golden.py)."""
import numpy as np
import pytest

import oracle
from limitador_amd import workloads as W
from limitador_amd.wire import HIT_DTYPE, RL_SIMPLE

M64 = (1 << 64) - 1
SEC = 1_000_000


class PyStorage:
    def __init__(self, rows):
        self.rows = rows  # [(max_value, seconds)]
        self.simple = {}  # limit id -> [value, expiry]
        self.qualified = {}  # key -> [value, expiry]

    @staticmethod
    def read(cell, t):  # atomic_expiring_value.rs:19-24,76-79
        return 0 if cell[1] <= t else cell[0]

    @staticmethod
    def bump(cell, d, w_us, t):  # :36-42,87-99
        if cell[1] <= t:
            cell[1] = t + w_us
            cell[0] = d
        else:
            cell[0] = (cell[0] + d) & M64

    def add_counter(self, limit):  # in_memory.rs:38-44
        self.simple.setdefault(limit, [0, 0])

    def check_and_update(self, ctrs, d, load, t):  # in_memory.rs:72-156; ctrs: [(key, limit|SIMPLE)]
        first, touched, rem, exp = None, [], [None] * len(ctrs), [None] * len(ctrs)
        order = [i for i, c in enumerate(ctrs) if c[1] & RL_SIMPLE] + [i for i, c in enumerate(ctrs) if not c[1] & RL_SIMPLE]
        for i in order:
            key, lim = ctrs[i]
            mx, secs = self.rows[lim & ~RL_SIMPLE]
            if lim & RL_SIMPLE:
                cell = self.simple[lim & ~RL_SIMPLE]
            else:
                cell = self.qualified.setdefault(key, [0, t + secs * SEC])  # :122-127, before the verdict
            v = (self.read(cell, t) + d) & M64
            if load:
                rem[i] = mx - v if v <= mx else 0  # checked_sub().unwrap_or_default()
                if first is None and v > mx:
                    first = i
            if v > mx and not load:
                return True, i, rem, exp  # :109-113,129-133
            if load:
                exp[i] = max(cell[1] - t, 0)
            touched.append((cell, secs * SEC))
        if first is not None:
            return True, first, rem, exp
        for cell, w in touched:
            self.bump(cell, d, w, t)
        return False, None, rem, exp


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
@pytest.mark.parametrize("load", [False, True])
def test_c_oracle_agrees_with_the_python_restatement(seed, load):
    rng = np.random.default_rng(seed)
    rows = [(5, 1), (50, 10), (3, 0), (2**64 - 1, 60), (0, 60), (20, 60)]
    simple_ids = [4, 5]
    orc = oracle.OracleStorage()
    orc.set_limits(rows)
    py = PyStorage(rows)
    for lid in simple_ids:
        orc.add_counter(lid | RL_SIMPLE, 0)
        py.add_counter(lid)
    now = W.NOW0_US
    for step in range(12):
        n_req = int(rng.integers(1, 250))
        hits, off, reqs = [], [0], []
        for _ in range(n_req):
            k = int(rng.integers(0, 5))
            d = int(rng.integers(0, 4))
            cs = []
            for lid in simple_ids:
                if len(cs) < k and rng.random() < 0.4:
                    cs.append((7_000_000 + lid, lid | RL_SIMPLE))
            while len(cs) < k:
                u = int(rng.integers(0, 25))
                lid = u % 4
                cs.append((int(W.splitmix64(np.array([u], dtype=np.uint64))[0]), lid))
            reqs.append((cs, d))
            hits += [(key, lim, d) for key, lim in cs]
            off.append(len(hits))
        arr = np.array(hits, dtype=HIT_DTYPE) if hits else np.zeros(0, dtype=HIT_DTYPE)
        v, f, r, e = orc.check_and_update(arr, now, req_off=np.array(off, dtype=np.uint32), load_counters=load)
        for q, (cs, d) in enumerate(reqs):
            if not cs:
                assert v[q] == 0 and f[q] == -1
                continue
            limited, first, rem, exp = py.check_and_update(cs, d, load, now)
            assert bool(v[q]) == limited, (step, q)
            assert (int(f[q]) - off[q] if limited else -1) == (first if limited else -1), (step, q)
            if load:
                assert [int(x) for x in r[off[q]:off[q + 1]]] == rem
                assert [int(x) for x in e[off[q]:off[q + 1]]] == exp
        now += int(rng.integers(0, 2 * SEC))
    for key, cell in py.qualified.items():
        got = orc.peek(key)
        assert got is not None and (got[0], got[1]) == (cell[0], cell[1]), key
    assert orc.num_qualified() == len(py.qualified)
    orc.close()
