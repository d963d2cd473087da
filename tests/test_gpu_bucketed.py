"""Parity cases aimed at the code paths of the bucketed hot path (limitador_amd/csrc/rl_bucket.hpp):
long hash buckets (window pass, deferred ring, LDS rebuilds), hot-key buckets (decided from their
position; replayed when they do not fit that form), hot-set promotion / demotion, and the fallbacks.
Every case is bit-exact against the CPU oracle.  Needs a MI355X."""
import os

import numpy as np
import pytest

from limitador_amd import workloads as W
from limitador_amd.wire import CELL_ROW_DTYPE, HIT_DTYPE, RL_SIMPLE
from test_gpu_parity import NOW, SEC, assert_same_state, make_engine, pair, run_both  # noqa: F401

pytestmark = pytest.mark.gpu


# (the two-stream pipeline holds a replay back across one more submit by default since round 5 — RL_DEFER2, first run and seen
# green in gpurun_out/r13a; the form before it stays in the suite as "two_streams_replays_not_held_back")
_FORMS = ["engine_default", "two_streams", "two_streams_replays_not_held_back"]


@pytest.fixture(autouse=True, params=_FORMS)
def pipeline_form(request, monkeypatch):
    """Engines as small as these tests' run the fused form by default (one stream, one launch per step); every test here
    also runs on the two-stream pipeline that engines for large batches use — where a replay whose partition is still running
    is held back across one more submit until the host sees the partition complete (rl_engine::pend_old, the default) — and
    on that pipeline without the second hold (RL_DEFER2=0)."""
    if request.param != "engine_default":
        monkeypatch.setenv("RL_FUSE", "0")
    if request.param == "two_streams_replays_not_held_back":
        monkeypatch.setenv("RL_DEFER2", "0")

SEED = 0x9E3779B97F4A7C15  # Engine's default hash_seed
M64 = (1 << 64) - 1


def fmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x ^ (x >> np.uint64(33))
        x = x * np.uint64(0xFF51AFD7ED558CCD)
        x = x ^ (x >> np.uint64(33))
        x = x * np.uint64(0xC4CEB9FE1A85EC53)
        x = x ^ (x >> np.uint64(33))
    return x


def bucket_log2(n_hits):
    """rl_engine.hip run_check_k1_bucketed: buckets for a batch of n_hits."""
    per = -(-n_hits // 384)
    cap = 11 if n_hits > (2 << 20) else 10  # (RL_BUCKET_LOG2 default: 1024 buckets up to 2 M hits)
    return min(cap, max(0, int(np.ceil(np.log2(per))) if per > 1 else 0))


def keys_in_bucket(n_keys, n_hits, bucket=0, start=1):
    """n_keys distinct keys that all fall into hash bucket `bucket` for a batch of n_hits."""
    b = bucket_log2(n_hits)
    out = []
    x = start
    while len(out) < n_keys:
        cand = np.arange(x, x + 200_000, dtype=np.uint64)
        k = W.splitmix64(cand)
        if b:
            k = k[(fmix64(k ^ np.uint64(SEED)) >> np.uint64(64 - b)) == bucket]
        out.extend(int(v) for v in k[: n_keys - len(out)])
        x += 200_000
    return np.array(out, dtype=np.uint64)


def make_hits(keys, limits, deltas):
    h = np.empty(len(keys), dtype=HIT_DTYPE)
    h["key"] = keys
    h["limit"] = limits
    h["delta"] = deltas
    return h


@pytest.mark.parametrize("seed,n_keys", [(1, 40), (2, 900), (3, 6000)])
def test_one_long_bucket_windows_ring_and_rebuilds(make_engine, seed, n_keys):
    """Every hit of the batch lands in ONE hash bucket: 40 keys (all in LDS, window pass denies the
    saturated ones), 900 keys (LDS rebuilds) and 6000 keys (rebuilds every round)."""
    rng = np.random.default_rng(seed)
    n = 30_000
    rows = [(7, 60), (300, 60), (2**64 - 1, 60), (3, 0), (40, 1)]
    eng, orc = pair(make_engine, rows, max_batch_hits=n)
    keys = keys_in_bucket(n_keys, n)
    key_limit = rng.integers(0, 5, size=n_keys)
    now = NOW
    for step in range(5):
        idx = (rng.zipf(1.2, size=n) - 1) % n_keys if step % 2 == 0 else rng.integers(0, n_keys, size=n)
        deltas = 1 if step < 2 else rng.integers(0, 5, size=n)
        run_both(eng, orc, make_hits(keys[idx], key_limit[idx], deltas), now)
        now += int(rng.integers(0, 2 * SEC))
    assert_same_state(eng, orc)


def test_hot_keys_are_promoted_decided_by_position_and_demoted(make_engine):
    rng = np.random.default_rng(11)
    n = 120_000
    rows = [(5000, 60), (100, 60), (10**9, 60), (50, 0)]
    eng, orc = pair(make_engine, rows, max_batch_hits=n, capacity_cells=1 << 18)
    n_bg = 50_000
    bg = W.splitmix64(np.arange(1000, 1000 + n_bg, dtype=np.uint64))
    hot = W.splitmix64(np.arange(10, 18, dtype=np.uint64))  # 8 keys, ~7500 hits each
    hot_limit = np.array([0, 0, 1, 1, 2, 2, 3, 0])

    def batch(hot_share, deltas_fn):
        is_hot = rng.random(n) < hot_share
        hi = rng.integers(0, len(hot), size=n)
        bi = rng.integers(0, n_bg, size=n)
        keys = np.where(is_hot, hot[hi], bg[bi])
        limits = np.where(is_hot, hot_limit[hi], bi % 3)
        return make_hits(keys, limits, deltas_fn(n, is_hot))

    ones = lambda n_, h_: np.ones(n_, dtype=np.uint32)  # noqa: E731
    now = NOW
    run_both(eng, orc, batch(0.5, ones), now)  # the hot keys sit in hash buckets: long buckets, promotion
    for step in range(3):  # now in hot buckets, uniform delta: decided by position
        now += 1000
        run_both(eng, orc, batch(0.5, ones), now)
    # mixed deltas on the hot keys: the hot buckets are replayed
    now += 1000
    run_both(eng, orc, batch(0.5, lambda n_, h_: rng.integers(0, 4, size=n_).astype(np.uint32)), now)
    # uniform but larger delta, after the windows rolled over (contested inside the batch)
    now += 61 * SEC
    run_both(eng, orc, batch(0.5, lambda n_, h_: np.full(n_, 3, dtype=np.uint32)), now)
    # delta 0 everywhere (admitted iff value <= max, and still refreshes an expired window)
    now += 61 * SEC
    run_both(eng, orc, batch(0.5, lambda n_, h_: np.zeros(n_, dtype=np.uint32)), now)
    # the hot keys' cells are evicted: the hot path has to create them (in_memory.rs:122-127)
    now += 61 * SEC
    assert eng.sweep_expired(now) == orc.sweep_expired(now)
    run_both(eng, orc, batch(0.5, ones), now)
    # traffic moves away: the hot set empties again
    for step in range(2):
        now += 1000
        run_both(eng, orc, batch(0.0, ones), now)
    now += 1000
    run_both(eng, orc, batch(0.5, ones), now)
    assert_same_state(eng, orc)


def test_more_hot_keys_than_hot_buckets(make_engine):
    rng = np.random.default_rng(12)
    n = 400_000
    eng, orc = pair(make_engine, [(900, 60)], max_batch_hits=n, capacity_cells=1 << 19)
    keys = W.splitmix64(np.arange(1, 701, dtype=np.uint64))  # 700 keys x ~570 hits: all above the bar, 512 slots
    now = NOW
    for step in range(3):
        run_both(eng, orc, make_hits(keys[rng.integers(0, 700, size=n)], 0, 1), now)
        now += 1000
    assert_same_state(eng, orc)


def test_hot_simple_counter_and_wrapping_hot_value(make_engine):
    """A simple limit every request hits is the canonical hot key; and a hot key whose value sits
    next to 2^64 has to be replayed with the reference's wrapping add."""
    rng = np.random.default_rng(13)
    n = 50_000
    rows = [(10_000, 60), (2**64 - 1, 60), (5, 60)]
    eng, orc = pair(make_engine, rows, simple_keys=[(0, 7_000_001)], max_batch_hits=n)
    cell = np.zeros(1, dtype=CELL_ROW_DTYPE)
    cell[0] = (4242, 1, 0, 2**64 - 40, NOW + 30 * SEC)
    eng.load_cells(cell)
    orc.load_cells([4242], [1], [2**64 - 40], [NOW + 30 * SEC])
    bg = W.splitmix64(np.arange(500, 900, dtype=np.uint64))
    now = NOW
    for step in range(4):
        which = rng.random(n)
        keys = np.where(which < 0.4, 7_000_001, np.where(which < 0.7, 4242, bg[rng.integers(0, 400, size=n)]))
        limits = np.where(which < 0.4, 0 | RL_SIMPLE, np.where(which < 0.7, 1, 2)).astype(np.uint32)
        run_both(eng, orc, make_hits(keys, limits, 1), now)
        now += 1000
    assert_same_state(eng, orc, n_simple_expected=1)


def test_hot_key_with_two_limit_ids_is_reported(make_engine):
    from limitador_amd.engine import EngineError

    n = 20_000
    eng, orc = pair(make_engine, [(10**6, 60), (10**6, 60)], max_batch_hits=n)
    good = make_hits(np.full(n, 99, dtype=np.uint64), 0, 1)
    run_both(eng, orc, good, NOW)       # promoted
    run_both(eng, orc, good, NOW + 1)   # decided by position
    bad = good.copy()
    bad["limit"][n // 2] = 1
    with pytest.raises(EngineError) as e:
        eng.check_and_update(bad, NOW + 2)
    assert e.value.code == -6


def test_large_limit_tables_take_the_same_path(make_engine):
    """k_bkt_apply reads limit rows from global memory: no row limit (the first cut kept the table in LDS
    and sent tables of more than 512 rows through the first-generation pipeline)."""
    rng = np.random.default_rng(14)
    rows = [(int(rng.integers(1, 50)), 60) for _ in range(700)]
    eng, orc = pair(make_engine, rows, max_limits=1024)
    idx = rng.integers(0, 5000, size=20_000)
    hits = make_hits(W.splitmix64(idx.astype(np.uint64)), idx % 700, 1)
    run_both(eng, orc, hits, NOW)
    run_both(eng, orc, hits, NOW + 1)
    assert_same_state(eng, orc)
    assert eng.stats()["ordered_batches"] == 0


@pytest.mark.parametrize("n", [1, 63, 64, 65, 383, 385, 511, 513, 1023, 1024, 1025, 2049, 4097, 8191,
                               262_144, 262_145, 266_241])  # 1024-hit tiles up to 256 of them, then 4096-hit tiles
@pytest.mark.parametrize("one_launch", [True, False], ids=["tiny_path_on", "partitioned_only"])
def test_batch_sizes_around_the_tile_and_round_boundaries(make_engine, monkeypatch, n, one_launch):
    """Batches of at most 1024 hits take the one-launch path (k_bkt_tiny); RL_TINY_MAX=0 sends them
    through the four partitioned kernels like every larger batch."""
    if not one_launch:
        if n > 1025:
            pytest.skip("the one-launch path only concerns batches of up to 1024 hits")
        monkeypatch.setenv("RL_TINY_MAX", "0")
    rng = np.random.default_rng(n)
    eng, orc = pair(make_engine, [(3, 60), (1, 0)], capacity_cells=1 << 16 if n < 100_000 else 1 << 19,
                    max_batch_hits=max(n, 8192))
    keys = W.splitmix64(rng.integers(0, max(2, n // 3), size=n).astype(np.uint64))
    hits = make_hits(keys, (keys % np.uint64(2)).astype(np.uint32), rng.integers(0, 3, size=n))
    run_both(eng, orc, hits, NOW)
    run_both(eng, orc, hits, NOW + 5)
    assert_same_state(eng, orc)


def test_submit_collect_keeps_the_sequential_contract(make_engine):
    """Three batches in flight (rl_check_and_update_submit_device / _collect; the partition of one overlaps
    k_bkt_apply of the one before) on overlapping keys: the result is the reference applied to batch 0, then
    batch 1, ...; other entry points answer BUSY."""
    import torch

    from limitador_amd.engine import EngineError

    rng = np.random.default_rng(21)
    n = 60_000
    eng, orc = pair(make_engine, [(40, 60), (900, 60)], max_batch_hits=n, capacity_cells=1 << 17)
    dev = torch.device("cuda", 0)
    batches, expect = [], []
    now = NOW
    for step in range(6):
        idx = (rng.zipf(1.2, size=n) - 1) % 20_000
        h = make_hits(W.splitmix64(idx.astype(np.uint64)), (idx % 2).astype(np.uint32), 1)
        v, _f, _r, _e = orc.check_and_update(h, now + step)
        expect.append(v)
        batches.append(torch.from_numpy(h.view(np.int64).reshape(-1, 2).copy()).to(dev))
    out = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(6)]
    torch.cuda.synchronize()
    eng.submit_device(batches[0].data_ptr(), n, now, out[0].data_ptr())
    eng.submit_device(batches[1].data_ptr(), n, now + 1, out[1].data_ptr())
    for step in range(2, 6):
        eng.submit_device(batches[step].data_ptr(), n, now + step, out[step].data_ptr())
        if step == 3:
            with pytest.raises(EngineError) as e:  # three in flight: a fourth submit, or any other call, is refused
                eng.submit_device(batches[0].data_ptr(), n, now, out[0].data_ptr())
            assert e.value.code == -9 and e.value.transient
            with pytest.raises(EngineError) as e:
                eng.dump_cells()
            assert e.value.code == -9
        eng.collect()
    eng.collect()
    eng.collect()
    with pytest.raises(EngineError):
        eng.collect()  # nothing left
    torch.cuda.synchronize()
    for step in range(6):
        assert np.array_equal(out[step].cpu().numpy(), expect[step]), f"batch {step}"
    assert_same_state(eng, orc)


def test_a_batch_that_does_not_fit_is_refused_before_anything_is_applied(make_engine):
    """All-or-nothing under RL_ERR_TABLE_FULL: the batch's NEW keys are counted exactly (k_bkt_count_new) when
    the cheap bound (every hit a new key) does not fit; a batch that would fill the table is refused with
    the table untouched, one that fits — however many hits it carries — is applied in full."""
    from limitador_amd.engine import EngineError

    rng = np.random.default_rng(31)
    eng, orc = pair(make_engine, [(50, 60)], capacity_cells=4096, max_batch_hits=1 << 16)
    keys = W.splitmix64(np.arange(1, 5001, dtype=np.uint64))
    # 40 000 hits on 2 000 keys into an empty 4 096-cell table: the cheap bound fails, the exact count fits
    run_both(eng, orc, make_hits(keys[rng.integers(0, 2000, size=40_000)], 0, 1), NOW)
    assert eng.stats()["live_cells"] == 2000
    before = np.sort(eng.dump_cells(), order="key")
    # 30 000 hits that would bring 2 000 more keys: 4 000 > 15/16 x 4 096 -> refused, nothing applied
    big = make_hits(keys[rng.integers(0, 4000, size=30_000)], 0, 1)
    assert len(np.unique(big["key"])) == 4000
    with pytest.raises(EngineError) as e:
        eng.check_and_update(big, NOW + 1)
    assert e.value.code == -4
    assert np.array_equal(before, np.sort(eng.dump_cells(), order="key"))
    assert eng.stats()["live_cells"] == 2000
    # the engine is still usable, and a batch that fits goes through: 1 000 new keys -> 3 000 live (<= 3/4)
    run_both(eng, orc, make_hits(keys[rng.integers(0, 3000, size=30_000)], 0, 1), NOW + 2)
    run_both(eng, orc, make_hits(keys[rng.integers(0, 3000, size=3000)], 0, 1), NOW + 3)
    eng.resize(1 << 14)
    run_both(eng, orc, big, NOW + 4)
    assert_same_state(eng, orc)


def test_soak_of_the_two_stream_pipeline_with_batches_of_every_size(make_engine):
    """150 batches through submit / collect, three in flight, sizes from one hit to 400 k (the one-launch path, the
    small-tile partition, the full one, in any succession), Zipf keys with a shifting head (promotion and demotion
    of hot keys while batches are in flight), mixed deltas now and then, windows that run out: every verdict and the
    final table against the oracle.  The partition of batch k+1 runs beside k_bkt_apply of batch k on rotating
    buffers — this is the test that would see a buffer reused too early."""
    import torch

    rng = np.random.default_rng(77)
    n_max = 400_000
    rows = [(60, 60), (2000, 2), (7, 1)]
    eng, orc = pair(make_engine, rows, max_batch_hits=n_max, capacity_cells=1 << 19)
    dev = torch.device("cuda", 0)
    sizes = [1, 7, 300, 1024, 1025, 5000, 70_000, 262_144, 262_145, n_max]
    plan = [int(sizes[int(rng.integers(0, len(sizes)))] if rng.random() < 0.6 else rng.integers(1, n_max)) for _ in range(150)]
    now = NOW
    in_engine, later = [], []
    done = 0
    for step, n in enumerate(plan):
        shift = (step // 25) * 37_000  # the popular keys move on every 25 batches
        idx = ((rng.zipf(1.15, size=n) - 1) % 150_000 + shift) % 200_000
        delta = 1 if step % 7 else rng.integers(0, 4, size=n).astype(np.uint32)
        h = make_hits(W.splitmix64(idx.astype(np.uint64)), (idx % 3).astype(np.uint32), delta)
        want = orc.check_and_update(h, now)[0]
        t = torch.from_numpy(h.view(np.int64).reshape(-1, 2).copy()).to(dev)
        out = torch.full((n,), 9, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        eng.submit_device(t.data_ptr(), n, now, out.data_ptr())
        in_engine.append((t, out, want, step))
        if len(in_engine) == 3:
            eng.collect()
            rec = in_engine.pop(0)
            if rec[3] % 5 == 0:  # (checking every batch at once would drain the pipeline every time)
                torch.cuda.synchronize()
                assert np.array_equal(rec[1].cpu().numpy(), rec[2]), f"batch {rec[3]} ({plan[rec[3]]} hits)"
                done += 1
            else:
                later.append(rec)
        now += int(rng.choice([0, 1, 1000, 400_000, 1_100_000]))
    for rec in in_engine:
        eng.collect()
        later.append(rec)
    torch.cuda.synchronize()
    for _t, out, want, s in later:
        assert np.array_equal(out.cpu().numpy(), want), f"batch {s} ({plan[s]} hits)"
        done += 1
    assert done == len(plan)
    assert_same_state(eng, orc)


def test_sweeps_between_batches_in_flight_and_reads_beside_them(make_engine):
    """BASELINE.json configs[4]: "mixed TTLs with concurrent expiry sweep".  rl_sweep_expired_submit is a command of the
    pipeline — three commands in flight, batches and sweeps interleaved, nothing drained in between — and the result is the
    same events replayed into the oracle in submission order (the sweep is an explicit eviction event there).  The
    read-only calls (is_within_limits, get_counters) are answered while batches are in flight: they see every batch
    submitted before them."""
    import ctypes as C

    import torch

    rng = np.random.default_rng(77)
    rows = [(40, 1), (25, 2), (10**6, 60), (3, 0)]
    eng, orc = pair(make_engine, rows, max_batch_hits=20_000, capacity_cells=1 << 16)
    dev = torch.device("cuda", 0)
    n = 20_000
    now = NOW
    pend = []  # ("batch", expected verdicts, device verdict tensor) | ("sweep", expected count)

    def collect_one():
        kind, want, got = pend.pop(0)
        if kind == "batch":
            eng.collect()
            torch.cuda.synchronize()
            assert np.array_equal(got.cpu().numpy(), want)
        else:
            assert eng.sweep_expired_collect() == want

    keep = []
    for step in range(24):
        if step % 3 == 2:
            if len(pend) == 3:
                collect_one()
            eng.sweep_expired_submit(now)
            pend.append(("sweep", orc.sweep_expired(now), None))
        else:
            idx = rng.integers(0, 6000, size=n)
            h = make_hits(W.splitmix64(idx.astype(np.uint64)), (idx % 4).astype(np.uint32), rng.integers(1, 3, size=n).astype(np.uint32))
            v, _f, _r, _e = orc.check_and_update(h, now)
            d_hits = torch.from_numpy(h.view(np.int64).reshape(-1, 2).copy()).to(dev)
            d_v = torch.full((n,), 0xCC, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            keep.append((d_hits, d_v))
            if len(pend) == 3:
                collect_one()
            eng.submit_device(d_hits.data_ptr(), n, now, d_v.data_ptr())
            pend.append(("batch", v, d_v))
        if step % 5 == 4:  # reads beside the commands in flight: as of everything submitted so far
            probe = make_hits(W.splitmix64(np.arange(0, 300, dtype=np.uint64)), (np.arange(300) % 4).astype(np.uint32), 1)
            assert np.array_equal(eng.is_within_limits(probe, now), orc.is_within_limits(probe, now))
            assert eng.count_counters(0, now) == len(orc.get_counters(0, now))
        now += int(rng.choice([1000, 300_000, 700_000]))
    while pend:
        collect_one()
    assert_same_state(eng, orc)
