"""The index arithmetic of the stable partition (limitador_amd/csrc/rl_bucket.hpp: k_bkt_hist -> k_bkt_scan ->
k_bkt_scatter) restated thread by thread in Python and checked against numpy — no GPU.

  * k_bkt_scan: a workgroup owns 32 columns of the (tiles x columns) count matrix, its 32 thread groups split the tiles;
    up to 8 tiles per group the counts stay in registers (one read), above that they are read twice.  Result: every
    column holds its exclusive prefix over the tiles, `total` the column sums (the maximum for the attribute columns).
  * k_bkt_scatter: wave w of a tile's workgroup owns STEPS x 64 consecutive hits and walks them in 64-hit steps; the rank
    of a hit inside (wave, bucket) = the wave's private counter + its position among the lanes of the step that carry
    the same bucket (a ballot match per bucket-id bit); waves are offset by the exclusive prefix of their counters.
    Result: a STABLE partition — inside every bucket the hits keep their trace order — which is what makes a hit's
    position in its bucket its trace-order rank.

The kernels themselves are compared with the oracle by the `-m gpu` tests; this file pins the arithmetic they share."""
import numpy as np
import pytest

SC_REG = 8       # tiles per thread group kept in registers (rl_bucket.hpp)
PT_WAVES = 16    # waves per partition workgroup
LANES = 64


def scan_kernel(hist, nbt):
    """k_bkt_scan over hist[ntiles][nrow] (in place) -> total[nrow]; columns >= nbt are reduced with max."""
    ntiles, nrow = hist.shape
    total = np.zeros(nrow, dtype=np.uint32)
    n_wg = -(-nrow // 32)
    per = (ntiles + 31) // 32
    cached = per <= SC_REG
    for wg in range(n_wg):
        s_part = np.zeros((32, 32), dtype=np.uint64)
        regs = {}
        for tid in range(1024):  # ---- first half: up to the barrier
            cl, g = tid & 31, tid >> 5
            c = wg * 32 + cl
            is_max = c >= nbt
            t_lo = g * per if g * per < ntiles else ntiles
            t_hi = t_lo + per if t_lo + per < ntiles else ntiles
            s = 0
            if c < nrow:
                if cached:
                    v8 = []
                    for q in range(SC_REG):
                        t = t_lo + q
                        tc = t if t < t_hi else ntiles - 1  # the unconditional load of a valid row
                        v = int(hist[tc, c])
                        v8.append(v if t < t_hi else 0)
                    for v in v8:
                        s = max(s, v) if is_max else s + v
                    regs[tid] = v8
                else:
                    for t in range(t_lo, t_hi):
                        v = int(hist[t, c])
                        s = max(s, v) if is_max else s + v
            s_part[g, cl] = s
        new = hist.copy()
        for tid in range(1024):  # ---- second half
            cl, g = tid & 31, tid >> 5
            c = wg * 32 + cl
            if c >= nrow:
                continue
            is_max = c >= nbt
            t_lo = g * per if g * per < ntiles else ntiles
            t_hi = t_lo + per if t_lo + per < ntiles else ntiles
            run = all_ = 0
            for gg in range(32):
                x = int(s_part[gg, cl])
                if gg < g:
                    run += x
                all_ = max(all_, x) if is_max else all_ + x
            if not is_max:
                if cached:
                    for q in range(SC_REG):
                        t = t_lo + q
                        if t < t_hi:
                            new[t, c] = run
                        run += regs[tid][q]
                else:
                    for t in range(t_lo, t_hi):
                        v = int(hist[t, c])
                        new[t, c] = run
                        run += v
            if g == 0:
                total[c] = all_
        hist[:] = new
    return total


@pytest.mark.parametrize("ntiles", [1, 2, 31, 32, 33, 64, 245, 256, 257, 300, 520])
def test_scan_gives_every_column_its_exclusive_prefix_over_the_tiles(ntiles):
    rng = np.random.default_rng(ntiles)
    nbt, n_attr = 70, 11  # 70 count columns + 11 attribute columns (maximum); 81 columns = three workgroups, one ragged
    hist = rng.integers(0, 50, size=(ntiles, nbt + n_attr)).astype(np.uint32)
    hist[:, 3] = 0  # an empty bucket
    want_prefix = np.cumsum(hist[:, :nbt], axis=0, dtype=np.uint64) - hist[:, :nbt]
    want_total = hist[:, :nbt].sum(axis=0)
    want_max = hist[:, nbt:].max(axis=0)
    attr_before = hist[:, nbt:].copy()
    total = scan_kernel(hist, nbt)
    assert np.array_equal(hist[:, :nbt], want_prefix.astype(np.uint32))
    assert np.array_equal(total[:nbt], want_total)
    assert np.array_equal(total[nbt:], want_max)
    assert np.array_equal(hist[:, nbt:], attr_before), "the attribute columns are reduced, not rewritten"


def match_digit(d, valid):
    """Lanes of the wave whose `d` equals mine, among `valid` lanes: one ballot per bit (rl_bucket.hpp)."""
    nbits = max(int(d.max()).bit_length(), 1)
    vmask = sum(1 << l for l in range(LANES) if valid[l])
    m = [vmask] * LANES
    for b in range(nbits):
        bm = sum(1 << l for l in range(LANES) if (int(d[l]) >> b) & 1)
        for l in range(LANES):
            m[l] &= bm if (int(d[l]) >> b) & 1 else ~bm
    return m


def scatter_tile(bucket, tile_base, n, steps, base_of_tile, out):
    """One workgroup of k_bkt_scatter: hits [tile_base, tile_base + PT_WAVES*steps*64) -> out[dst] = hit index."""
    nbt = len(base_of_tile)
    s_cnt = np.zeros((PT_WAVES, nbt), dtype=np.int64)
    rank = np.zeros((PT_WAVES, steps, LANES), dtype=np.int64)
    dig = np.zeros((PT_WAVES, steps, LANES), dtype=np.int64)
    for w in range(PT_WAVES):  # waves run in any order: their counters are private
        wbase = tile_base + w * LANES * steps
        for u in range(steps):
            i = wbase + u * LANES + np.arange(LANES)
            ok = i < n
            d = np.where(ok, bucket[np.minimum(i, n - 1)], 0)
            m = match_digit(d, ok)
            for lane in range(LANES):
                if not ok[lane]:
                    continue
                lt = (1 << lane) - 1
                c = s_cnt[w, d[lane]]
                rank[w, u, lane] = c + bin(m[lane] & lt).count("1")
                dig[w, u, lane] = d[lane]
            for lane in range(LANES):  # the first lane of every group adds the group's size (after all lanes have read)
                if ok[lane] and (m[lane] & ((1 << lane) - 1)) == 0:
                    s_cnt[w, d[lane]] += bin(m[lane]).count("1")
    woff = np.cumsum(s_cnt, axis=0) - s_cnt  # exclusive prefix over the waves
    for w in range(PT_WAVES):
        wbase = tile_base + w * LANES * steps
        for u in range(steps):
            for lane in range(LANES):
                i = wbase + u * LANES + lane
                if i < n:
                    d = dig[w, u, lane]
                    out[base_of_tile[d] + woff[w, d] + rank[w, u, lane]] = i


@pytest.mark.parametrize("n,nb,steps,skew", [(5000, 8, 1, False), (9000, 37, 4, False), (4096 * 2 + 17, 5, 4, True),
                                             (1000, 3, 1, True)])
def test_wave_private_counters_and_ballot_ranks_give_a_stable_partition(n, nb, steps, skew):
    rng = np.random.default_rng(n)
    bucket = rng.integers(0, nb, size=n)
    if skew:  # most hits in one bucket: long runs of equal ids inside a step
        bucket[rng.random(n) < 0.7] = 1
    tile = PT_WAVES * steps * LANES
    ntiles = -(-n // tile)
    # k_bkt_hist + k_bkt_scan: where each (tile, bucket) run starts
    hist = np.zeros((ntiles, nb), dtype=np.int64)
    for t in range(ntiles):
        hist[t] = np.bincount(bucket[t * tile:(t + 1) * tile], minlength=nb)
    totals = hist.sum(axis=0)
    bucket_lo = np.cumsum(totals) - totals
    prefix = np.cumsum(hist, axis=0) - hist
    out = np.full(n, -1, dtype=np.int64)
    for t in rng.permutation(ntiles):  # tiles in any order
        scatter_tile(bucket, t * tile, n, steps, bucket_lo + prefix[t], out)
    assert np.array_equal(out, np.argsort(bucket, kind="stable")), "every bucket keeps its hits in trace order"


# ---- k_bkt_part_l (rl_part.hpp): the hot table probed a group at a time, the counters bumped without waits ----------------
HOT_SLOTS, HS_GROUPS = 2048, 512
FREE = 0xFFFF


def _fmix64(x):
    x &= (1 << 64) - 1
    x ^= x >> 33
    x = (x * 0xff51afd7ed558ccd) & ((1 << 64) - 1)
    x ^= x >> 33
    x = (x * 0xc4ceb9fe1a85ec53) & ((1 << 64) - 1)
    x ^= x >> 33
    return x


def _hs_group(hh):
    return (hh >> 8) & (HS_GROUPS - 1)


def _hs_fp(hh):
    return (hh >> 20) & 31


def build_hot_table(keys, deny, seed, order):
    """Every key of the set inserts itself (any interleaving: `order`): from the first slot of its 4-slot-aligned group on,
    the first slot that is free or holds the same key — a key listed twice keeps the smaller index."""
    slots = [FREE] * HOT_SLOTS
    for i in order:
        hh = _fmix64(keys[i] ^ seed)
        mine = i | (0x200 if deny[i] else 0) | (_hs_fp(hh) << 10)
        s = _hs_group(hh) * 4
        while True:
            x = slots[s]
            if x == FREE:
                slots[s] = mine
                break
            if keys[x & 0x1FF] == keys[i]:
                if (x & 0x1FF) > i:
                    slots[s] = mine
                break
            s = (s + 1) & (HOT_SLOTS - 1)
    return slots


def lookup_plain(slots, keys, key, seed):
    hh = _fmix64(key ^ seed)
    q = _hs_group(hh) * 4
    while True:
        x = slots[q]
        if x == FREE:
            return None
        if keys[x & 0x1FF] == key:
            return x & 0x3FF
        q = (q + 1) & (HOT_SLOTS - 1)


def lookup_group(slots, keys, key, seed):
    """The kernel's fast path: ONE read of the group; the first entry that is free or carries the key's fingerprint
    decides.  Returns (answer, took_the_slow_path)."""
    hh = _fmix64(key ^ seed)
    g, fp = _hs_group(hh) * 4, _hs_fp(hh)
    c = 0xFFFE
    for e in reversed(slots[g:g + 4]):  # walked from the last entry to the first: the first deciding entry wins
        c = FREE if e == FREE else (e if (e >> 10) == fp else c)
    if c == FREE:
        return None, False
    if c != 0xFFFE and keys[c & 0x1FF] == key:
        return c & 0x3FF, False
    return lookup_plain(slots, keys, key, seed), True


@pytest.mark.parametrize("n_hot,dups", [(0, 0), (1, 0), (375, 0), (512, 0), (512, 40), (300, 25)])
def test_the_group_probed_hot_table_answers_like_the_plain_probe_whatever_the_insertion_order(n_hot, dups):
    rng = np.random.default_rng(1000 + n_hot + dups)
    seed = int(rng.integers(0, 1 << 63))
    keys = [int(k) for k in rng.integers(1, 1 << 62, size=n_hot, dtype=np.uint64)]
    for _ in range(dups):  # a key listed twice (a stale or arbitrary set is valid: rl_bucket.hpp)
        a, b = rng.integers(0, n_hot, size=2)
        keys[int(a)] = keys[int(b)]
    # clustered keys too: many keys of ONE group, so that groups overflow into their neighbours
    if n_hot >= 300:
        base_g = None
        k, filled = 1 << 40, 0
        while filled < 11:
            hh = _fmix64(k ^ seed)
            if base_g is None:
                base_g = _hs_group(hh)
            if _hs_group(hh) == base_g and k not in keys:
                keys[filled] = k
                filled += 1
            k += 1
    deny = [bool(b) for b in rng.integers(0, 2, size=n_hot)]
    want = {}
    for i, k in enumerate(keys):
        want.setdefault(k, i)  # the smaller index
    tables = [build_hot_table(keys, deny, seed, rng.permutation(n_hot)) for _ in range(3)]
    probes = list(dict.fromkeys(keys)) + [int(k) for k in rng.integers(1, 1 << 62, size=4000, dtype=np.uint64)]
    slow = 0
    for key in probes:
        answers = set()
        for slots in tables:
            a, s = lookup_group(slots, keys, key, seed)
            slow += s
            assert a == lookup_plain(slots, keys, key, seed)
            answers.add(None if a is None else a & 0x1FF)
        assert len(answers) == 1, "two workgroups (two insertion orders) would send one key to two bins"
        (a,) = answers
        assert a == want.get(key)
        if a is not None:
            for slots in tables:
                full, _ = lookup_group(slots, keys, key, seed)
                assert bool(full & 0x200) == deny[a]
    assert slow < 0.2 * 3 * len(probes)  # the fast path is the common one


def test_counters_read_before_the_leaders_add_give_the_same_ranks_as_read_modify_write():
    """k_bkt_part_l's phase 4: per 64-hit step every lane READS its (wave, bin) counter, then the lowest lane of every
    group of equal bins ADDS the group's size — no lane waits for its read before the add is issued, because the add's
    operand does not depend on it and LDS operations of one wave execute in issue order.  The ranks equal the stable
    partition's: counter before the step + lanes of the same bin below me."""
    rng = np.random.default_rng(7)
    for steps, nbins in [(8, 1536), (8, 40), (4, 3)]:
        d = rng.integers(0, nbins, size=(steps, LANES))
        d[:, ::7] = 5 % nbins  # a hot bin in every step
        cnt = np.zeros(nbins, dtype=np.int64)
        ranks = np.zeros((steps, LANES), dtype=np.int64)
        for u in range(steps):
            pre = cnt[d[u]].copy()               # the step's loads (issued first)
            for lane in range(LANES):
                same = np.nonzero(d[u] == d[u, lane])[0]
                below = int((same < lane).sum())
                ranks[u, lane] = pre[lane] + below
                if below == 0:                   # the group's leader: one add, the group's size
                    cnt[d[u, lane]] += len(same)
        seen = {}
        for u in range(steps):
            for lane in range(LANES):
                b = int(d[u, lane])
                assert ranks[u, lane] == seen.get(b, 0)
                seen[b] = seen.get(b, 0) + 1
