"""The ALGORITHM of the bucketed single-counter path (limitador_amd/csrc/rl_bucket.hpp + rl_apply.hpp) restated in
Python, decision rule by decision rule, and run against the CPU oracle — no GPU.  What is pinned here is the exactness
argument of DESIGN.md §3.1, independent of the kernels (which the `-m gpu` tests compare with the same oracle):

  * stable partition by key hash; hot keys get a bucket of their own (any set is valid);
  * a bucket is replayed in rounds of R hits with the cells it touches cached (LDS): per round and key
        run + sum <= max            -> every hit admitted, whatever the order
        run + delta > max           -> this hit denied, whatever the order
        uniform delta               -> admitted iff trace-order rank < (max - run) / delta
        anything else               -> the key's hits replayed one by one with the reference's wrapping add
    (mixed deltas, deltas >= 2^23, sums that wrap, 0-second windows);
  * the cached cells are written back and dropped between rounds when too many are live (only entries with >= 16 hits
    survive a rebuild), so a key met again is read again — after its write-back;
  * hot buckets: with one delta value, a window, a value < 2^62 and a cell of the same limit id the verdict is
    `position < (max - value) / delta`, the update is applied once; otherwise the bucket is replayed;
  * verdicts are prefilled "admitted" and only denials are stored (sparse output).

Reference: in_memory.rs:72-156 called once per request (k = 1), atomic_expiring_value.rs:19-24,36-42,87-99."""
import numpy as np
import pytest

import oracle
from limitador_amd import workloads as W
from limitador_amd.wire import HIT_DTYPE, RL_SIMPLE

NOW, SEC, M64 = 1_700_000_000_000_000, 1_000_000, (1 << 64) - 1
SEED = 0x9E3779B97F4A7C15
R, E = 256, 512                 # hits per round, LDS cells (Apply2Lds<1, 9>)
KEEP = E * 3 // 4 - R           # rebuild before a round if more cells are live
EF_HOT_MIN = 16                 # hits that let a cell survive a rebuild
BIG_DELTA = 1 << 23             # AP2_BIG_DELTA
HOT_CHUNK = 1024
STATS = {}  # which branches the traces took (asserted below: the test is only worth something if they all ran)


def took(what, n=1):
    STATS[what] = STATS.get(what, 0) + n


def fmix64(x):
    x ^= x >> 33
    x = (x * 0xFF51AFD7ED558CCD) & M64
    x ^= x >> 33
    x = (x * 0xC4CEB9FE1A85EC53) & M64
    x ^= x >> 33
    return x


class Table:
    """key -> [value, expiry_us, limit]: the counter table (probe sequences are not modelled: identity is exact)."""

    def __init__(self, rows):
        self.rows = rows  # (max_value, window_s)
        self.cells = {}

    def window_us(self, limit):
        return self.rows[limit & ~RL_SIMPLE][1] * SEC

    def max_value(self, limit):
        return self.rows[limit & ~RL_SIMPLE][0]


def replay_bucket(t, hits, idxs, now, verdict, promote=None):
    """apply2_bucket: the hits `idxs` (trace order) of one bucket, in rounds of R, over cached cells."""
    lds = {}  # key -> dict(run, expired, dirty, limit, count, bad)

    def commit(rebuild):
        nonlocal lds
        keep = {}
        for k, c in lds.items():
            if c["dirty"]:
                cell = t.cells[k]
                cell[0] = c["run"]
                if c["expired"]:  # update_if_expired, atomic_expiring_value.rs:87-99
                    cell[1] = now + t.window_us(c["limit"])
                    if t.window_us(c["limit"]) != 0:
                        c["expired"] = False
                c["dirty"] = False
            if rebuild and c["count"] >= EF_HOT_MIN and not c["bad"]:
                keep[k] = c
        if not rebuild:
            if promote is not None:
                promote.update({k: c["count"] for k, c in lds.items()})
            return
        took("rebuild")
        took("cells kept by a rebuild", len(keep) if len(keep) <= KEEP else 0)
        lds = keep if len(keep) <= KEEP else {}

    for r0 in range(0, len(idxs), R):
        if r0 and len(lds) > KEEP:
            commit(True)
        rnd = idxs[r0:r0 + R]
        # ---- A: aggregate per key (one LDS atomic per hit: count, sum of the deltas below 2^23, largest delta) ----
        agg = {}
        for i in rnd:
            k, d = int(hits["key"][i]), int(hits["delta"][i])
            a = agg.setdefault(k, {"cnt": 0, "sum": 0, "dmax": 0, "hits": []})
            a["cnt"] += 1
            a["sum"] += d if d < BIG_DELTA else 0
            a["dmax"] = max(a["dmax"], d)
            a["hits"].append(i)
        # ---- B: a key new to the cache: read (or create) its cell -----------------------------------------------
        for k, a in agg.items():
            if k in lds:
                took("key met again in a later round, still cached")
                continue
            lim = int(hits["limit"][a["hits"][0]])
            if k not in t.cells:
                took("cell created")
                assert not (lim & RL_SIMPLE), "a simple counter must pre-exist (in_memory.rs:106-107)"
                t.cells[k] = [0, now + t.window_us(lim), lim]  # in_memory.rs:122-127: created BEFORE the verdict
            value, expiry, cl = t.cells[k]
            expired = expiry <= now
            lds[k] = {"run": 0 if expired else value, "expired": expired, "dirty": False, "limit": cl, "count": 0,
                      "bad": False}
        # ---- C: verdicts ------------------------------------------------------------------------------------------
        slow_keys = set()
        for k, a in agg.items():
            c = lds[k]
            mx, win = t.max_value(c["limit"]), t.window_us(c["limit"])
            run, s, cnt, dm = c["run"], a["sum"], a["cnt"], a["dmax"]
            ovf = run + s > M64
            if win == 0 or ovf or dm >= BIG_DELTA:
                took("slow: 0-second window" if win == 0 else ("slow: sum wraps" if ovf else "slow: delta >= 2^23"))
                slow_keys.add(k)
                continue
            tot = run + s
            uniform = s == cnt * dm
            rank = 0
            for i in a["hits"]:
                d = int(hits["delta"][i])
                if tot <= mx:
                    v = 0
                elif run + d > mx:
                    v = 1
                elif uniform:
                    took("decided by rank")
                    v = 0 if rank < (mx - run) // d else 1
                else:
                    took("slow: mixed deltas across the limit")
                    slow_keys.add(k)
                    break
                if v:
                    verdict[i] = 1  # sparse output: only denials are stored
                rank += 1
        # ---- slow keys: replayed hit by hit, in trace order, wrapping add (in_memory.rs:88) ---------------------
        for i in rnd:
            k = int(hits["key"][i])
            if k not in slow_keys:
                continue
            c = lds[k]
            mx, win = t.max_value(c["limit"]), t.window_us(c["limit"])
            d = int(hits["delta"][i])
            cur = 0 if win == 0 else c["run"]
            tot = (cur + d) & M64
            adm = tot <= mx
            if adm:
                c["run"] = d if win == 0 else tot
                c["dirty"] = True
                if win == 0:
                    c["expired"] = True
            verdict[i] = 0 if adm else 1  # (overwrites what a non-slow lane of the same key may have stored: same value)
        # ---- D: fold the round into `run` -------------------------------------------------------------------------
        for k, a in agg.items():
            c = lds[k]
            if k not in slow_keys:
                mx = t.max_value(c["limit"])
                run, s, cnt, dm = c["run"], a["sum"], a["cnt"], a["dmax"]
                if run + s <= mx:
                    c["run"] = run + s
                    c["dirty"] = True
                elif s == cnt * dm and run + dm <= mx:
                    n_adm = min(cnt, (mx - run) // dm)
                    if n_adm:
                        c["run"] = run + n_adm * dm
                        c["dirty"] = True
            c["count"] += a["cnt"]
    commit(False)


def hot_bucket(t, hits, idxs, now, verdict):
    """A bucket of ONE key: chunks decided from positions when the cell allows it (apply2_hot_chunk_self), else
    replayed; the update is applied once, after every chunk has read the unchanged cell."""
    k, lim = int(hits["key"][idxs[0]]), int(hits["limit"][idxs[0]])
    deltas = hits["delta"][idxs]
    uniform = int(deltas.min()) == int(deltas.max())
    if not uniform:  # mixed deltas never own chunks
        took("hot bucket replayed: mixed deltas")
        return replay_bucket(t, hits, idxs, now, verdict)
    d = int(deltas[0])
    found = k in t.cells
    value, expiry, cl = t.cells[k] if found else (0, 0, lim)
    expired = found and expiry <= now
    s = value if (found and not expired) else 0
    win, mx = t.window_us(lim), t.max_value(lim)
    fast = win != 0 and s < (1 << 62) and (not found or cl == lim) and (found or not (lim & RL_SIMPLE))
    if not fast:
        took("hot bucket replayed: cell state")
        return replay_bucket(t, hits, idxs, now, verdict)
    took("hot bucket decided from positions")
    if expired:
        took("hot bucket resets the window")
    if not found:
        took("hot bucket creates the cell")
    room = 0 if s > mx else ((mx - s) // d if d else M64)
    for c0 in range(0, len(idxs), HOT_CHUNK):  # any order, any number of workgroups: position only
        for pos in range(c0, min(c0 + HOT_CHUNK, len(idxs))):
            if pos >= room:
                verdict[idxs[pos]] = 1
    # the last chunk to arrive applies AtomicExpiringValue::update for the admitted hits
    n_adm = min(len(idxs), room)
    if not found:
        t.cells[k] = [0, now + win, lim]  # first touch creates the cell, verdict or not
        expired = False
    if n_adm:
        t.cells[k][0] = s + n_adm * d
        if expired:
            t.cells[k][1] = now + win


def bucketed_batch(t, hits, now, hot_set, bk_log2):
    n = len(hits)
    verdict = np.zeros(n, dtype=np.uint8)  # k_bkt_hist: "admitted" is the default answer
    hot_index = {int(k): j for j, k in enumerate(hot_set)}
    buckets, hot_buckets = {}, {}
    for i in range(n):  # stable partition: trace order is kept inside every bucket
        k = int(hits["key"][i])
        if k in hot_index:
            hot_buckets.setdefault(hot_index[k], []).append(i)
        else:
            b = fmix64(k ^ SEED) >> (64 - bk_log2) if bk_log2 else 0
            buckets.setdefault(b, []).append(i)
    counts = {}
    for b in sorted(buckets, key=lambda b: -len(buckets[b])):  # any bucket order: buckets share no cell
        replay_bucket(t, hits, buckets[b], now, verdict, promote=counts)
    for j in sorted(hot_buckets, reverse=True):
        hot_bucket(t, hits, hot_buckets[j], now, verdict)
        counts[int(hot_set[j])] = len(hot_buckets[j])
    return verdict, counts


ROWS = [(5, 1), (50, 10), (1000, 60), (0, 60), (M64, 3600), (7, 0), (M64 - 3, 60), (300, 60)]
SIMPLE = [(7, 9_000_001)]


def make_batch(rng, n, keys, key_limit, mode):
    idx = (rng.zipf(1.25, size=n) - 1) % len(keys) if mode % 2 else rng.integers(0, len(keys), size=n)
    h = np.empty(n, dtype=HIT_DTYPE)
    h["key"], h["limit"] = keys[idx], key_limit[idx]
    if mode == 0:
        h["delta"] = 1
    elif mode == 1:
        h["delta"] = rng.integers(0, 4, size=n)
    elif mode == 2:
        h["delta"] = np.where(rng.random(n) < 0.01, BIG_DELTA + rng.integers(0, 9, size=n), 2)
    else:
        h["delta"] = 3
    simple = rng.random(n) < 0.1  # the simple counter every request of a namespace hits
    h["key"][simple], h["limit"][simple] = SIMPLE[0][1], SIMPLE[0][0] | RL_SIMPLE
    if mode == 1:
        h["delta"][simple] = 1
    return h


@pytest.mark.parametrize("seed", [41, 42, 43])
def test_bucketed_replay_equals_the_reference(seed):
    rng = np.random.default_rng(seed)
    orc = oracle.OracleStorage()
    orc.set_limits(ROWS)
    t = Table(ROWS)
    for lim, key in SIMPLE:
        orc.add_counter(lim | RL_SIMPLE, key)
        t.cells[key] = [0, 0, lim | RL_SIMPLE]  # add_counter: (0, UNIX_EPOCH), in_memory.rs:38-44
    n_keys = 1500
    keys = W.splitmix64(np.arange(1, n_keys + 1, dtype=np.uint64))
    key_limit = rng.integers(0, 7, size=n_keys).astype(np.uint32)
    key_limit[:12] = [2, 2, 7, 7, 5, 6, 4, 1, 2, 7, 3, 0]  # the Zipf head on every kind of limit
    # a value next to 2^64 on the wrapping limit (release-build arithmetic, in_memory.rs:88,261)
    wrap = [int(k) for k, l in zip(keys, key_limit) if l == 6][:3]
    orc.load_cells(wrap, [6] * len(wrap), [M64 - 20] * len(wrap), [NOW + 30 * SEC] * len(wrap))
    for k in wrap:
        t.cells[k] = [M64 - 20, NOW + 30 * SEC, 6]
    # ... and on the limit whose maximum is 2^64 - 1: a round's SUM wraps, the 21st hit is admitted again at 0
    top = int(keys[6])
    orc.load_cells([top], [4], [M64 - 20], [NOW + 3000 * SEC])
    t.cells[top] = [M64 - 20, NOW + 3000 * SEC, 4]
    now, hot = NOW, np.array([], dtype=np.uint64)
    for step in range(14):
        n = int(rng.integers(300, 6000))
        h = make_batch(rng, n, keys, key_limit, step % 4)
        v2, _f, _r, _e = orc.check_and_update(h, now)
        bk_log2 = int(rng.integers(0, 4))  # 1..8 buckets: long buckets, several rounds, rebuilds
        v1, counts = bucketed_batch(t, h, now, hot, bk_log2)
        assert np.array_equal(v1, v2), f"step {step}: first mismatch at {np.nonzero(v1 != v2)[0][:5]}"
        # the next batch's hot set: any set is valid — here the keys that took >= 40 hits, sometimes a stale one
        if step % 5 != 4:
            hot = np.array([k for k, c in counts.items() if c >= 40][:64], dtype=np.uint64)
        now += int(rng.integers(0, 2 * SEC)) if step % 6 else 61 * SEC
        if step == 6:  # an eviction event: the hot keys' cells are gone, the hot path has to create them
            n_removed = orc.sweep_expired(now)
            gone = [k for k, c in t.cells.items() if not (c[2] & RL_SIMPLE) and c[1] <= now]
            for k in gone:
                del t.cells[k]
            assert n_removed == len(gone) > 0
            hot = keys[:10].copy()  # (any set is valid: the Zipf head, whose cells have just been evicted)
    # every cell equals the oracle's
    assert len([k for k in t.cells if k != SIMPLE[0][1]]) == orc.num_qualified()
    for k, (value, expiry, lim) in t.cells.items():
        if k == SIMPLE[0][1]:
            assert (value, expiry) == orc.peek_simple(lim)[:2]
        else:
            assert (value, expiry, lim) == orc.peek(k), k


def test_one_key_fills_the_batch():
    """Position == rank: a single hot key, uniform delta, decided from positions in any chunk order; then the same
    key through a hash bucket (not in the hot set) and with mixed deltas."""
    orc = oracle.OracleStorage()
    rows = [(2500, 60)]
    orc.set_limits(rows)
    t = Table(rows)
    key = int(W.splitmix64(np.array([77], dtype=np.uint64))[0])
    for step, (n, delta, hot) in enumerate([(3000, 1, False), (3000, 1, True), (900, 2, True), (700, None, True),
                                             (4000, 3, False)]):
        h = np.empty(n, dtype=HIT_DTYPE)
        h["key"], h["limit"] = key, 0
        h["delta"] = delta if delta is not None else np.arange(n) % 3
        now = NOW + step * 61 * SEC if step in (2, 4) else NOW + step
        v2, _f, _r, _e = orc.check_and_update(h, now)
        v1, _c = bucketed_batch(t, h, now, np.array([key] if hot else [], dtype=np.uint64), 2)
        assert np.array_equal(v1, v2), step
    assert tuple(t.cells[key]) == orc.peek(key)


def test_every_branch_of_the_restatement_ran():
    """(runs after the traces above: pytest keeps file order)"""
    for what in ("rebuild", "cells kept by a rebuild", "key met again in a later round, still cached", "cell created",
                 "slow: 0-second window", "slow: sum wraps", "slow: delta >= 2^23", "slow: mixed deltas across the limit",
                 "decided by rank", "hot bucket replayed: mixed deltas", "hot bucket replayed: cell state",
                 "hot bucket decided from positions", "hot bucket resets the window", "hot bucket creates the cell"):
        assert STATS.get(what, 0) > 0, f"no trace took the branch: {what} ({STATS})"
