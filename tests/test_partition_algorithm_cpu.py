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
