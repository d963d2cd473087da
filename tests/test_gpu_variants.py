"""The build variants of the bucketed hot path, each bit-exact against the CPU oracle on a trace that takes every
branch of the hot-key code (promotion, decided by position, replayed, created by the hot path, demoted):

  default            an engine for batches of <= 256 k hits: RL_FUSE=1 below (larger engines: RL_FUSE=0)
  RL_FUSE=0          two streams: k_bkt_part on a stream of its own, the replay (k_bkt_step) enqueued one submit late
                     (PendingApply); 16-bit limit ids in LDS, default verdicts written by the partition
  RL_DEFER_APPLY=0   the replay enqueued at submit, behind a wait for its partition
  RL_PIPE_DEPTH=2    the partition of batch p waits for the replay of batch p - 2
  RL_APPLY2_CFG=1    32-bit limit ids in LDS (what an engine with more than 32768 limit rows takes by itself)
  RL_OVERLAP=0       k_bkt_part and the replay on one stream
  RL_PART_COMPACT=0  k_bkt_part (1024 threads) for 4096-hit tiles instead of k_bkt_part_c (512 threads, half the LDS)
  RL_APPLY_EVENTS=0  the replay as a plain launch; the partition stream ordered by what the host has collected
  RL_DEFER2=0        the default (1) holds a replay whose partition is still running when the next batch is submitted back across
                     that submit and sends it out from the collect's spin, without a wait command; 0 = the form before
  RL_FUSE=1          one stream, one k_bkt_step launch per step: the replay of batch j and, beside it, the partition of batch
                     j + 1 as a role of 256-thread workgroups

and the output contract of the sparse verdict stores: every byte of verdict[] / every word of first_limited[] of a
batch is defined when it is collected, whatever the buffers held before.  Needs a MI355X."""
import os

import numpy as np
import pytest

from limitador_amd import workloads as W
from limitador_amd.wire import CELL_ROW_DTYPE, RL_SIMPLE
from test_gpu_bucketed import make_hits
from test_gpu_parity import NOW, SEC, assert_same_state, make_engine, pair, run_both  # noqa: F401

pytestmark = pytest.mark.gpu

# (an engine this small is fused by default: the two-stream variants say RL_FUSE=0)
TWO = {"RL_FUSE": "0"}
VARIANTS = [{}, TWO, {**TWO, "RL_DEFER_APPLY": "0"}, {**TWO, "RL_APPLY2_CFG": "1"}, {"RL_FUSE": "1"}, {**TWO, "RL_OVERLAP": "0"},
            {"RL_FUSE": "1", "RL_DEFER_APPLY": "0"}, {**TWO, "RL_DEFER_APPLY": "0", "RL_PIPE_DEPTH": "2", "RL_APPLY2_CFG": "2"},
            {**TWO, "RL_APPLY_EVENTS": "0"}, {**TWO, "RL_PART_COMPACT": "0"}]
VARIANTS.append({**TWO, "RL_DEFER2": "0"})  # (the default since round 5 is 1: gpurun_out/r13a)


def hot_trace(eng, orc, rng, n=60_000):
    """Background keys + 6 hot keys on limits with different maxima, a 0-second window, a simple counter every request
    of a namespace hits, and a value next to 2^64."""
    bg = W.splitmix64(np.arange(1000, 21_000, dtype=np.uint64))
    hot = W.splitmix64(np.arange(10, 16, dtype=np.uint64))
    hot_limit = np.array([0, 0, 1, 2, 3, 1])
    cell = np.zeros(1, dtype=CELL_ROW_DTYPE)
    cell[0] = (int(hot[5]), 1, 0, 2**64 - 25, NOW + 30 * SEC)  # wraps inside the first batch
    eng.load_cells(cell)
    orc.load_cells([int(hot[5])], [1], [2**64 - 25], [NOW + 30 * SEC])

    def batch(hot_share, deltas):
        which = rng.random(n)
        is_hot = which < hot_share
        is_simple = (~is_hot) & (which < hot_share + 0.15)
        hi = rng.integers(0, len(hot), size=n)
        bi = rng.integers(0, len(bg), size=n)
        keys = np.where(is_hot, hot[hi], np.where(is_simple, 7_000_001, bg[bi]))
        limits = np.where(is_hot, hot_limit[hi], np.where(is_simple, 4 | RL_SIMPLE, bi % 3)).astype(np.uint32)
        return make_hits(keys, limits, deltas(n))

    ones = lambda m: np.ones(m, dtype=np.uint32)  # noqa: E731
    now = NOW
    for step in range(4):  # promotion (depth 3: the set of batch p is used by batch p + 3), then decided by position
        run_both(eng, orc, batch(0.5, ones), now)
        now += 1000
    run_both(eng, orc, batch(0.5, lambda m: rng.integers(0, 4, size=m).astype(np.uint32)), now)  # mixed deltas: replayed
    now += 61 * SEC  # windows rolled over: the first admitted hit of a hot bucket resets the window
    run_both(eng, orc, batch(0.5, lambda m: np.full(m, 3, dtype=np.uint32)), now)
    now += 61 * SEC
    assert eng.sweep_expired(now) == orc.sweep_expired(now)  # the hot keys' cells are gone: the hot path creates them
    run_both(eng, orc, batch(0.5, ones), now)
    for step in range(4):  # traffic moves away, and back
        now += 1000
        run_both(eng, orc, batch(0.0 if step < 3 else 0.5, ones), now)
    assert_same_state(eng, orc, n_simple_expected=1)


@pytest.mark.parametrize("env", VARIANTS, ids=lambda e: "+".join(f"{k[3:]}={v}" for k, v in e.items()) or "default")
def test_hot_path_variants_against_the_oracle(make_engine, monkeypatch, env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rows = [(5000, 60), (100, 60), (10**9, 60), (50, 0), (20_000, 60)]
    eng, orc = pair(make_engine, rows, simple_keys=[(4, 7_000_001)], max_batch_hits=60_000, capacity_cells=1 << 17)
    hot_trace(eng, orc, np.random.default_rng(31))


def test_more_than_32768_limit_rows_take_the_wide_kernel(make_engine):
    """Limit ids do not fit the 16-bit LDS field: the engine picks the 32-bit instantiation of k_bkt_apply by itself."""
    rng = np.random.default_rng(32)
    n_rows = 40_000
    rows = [(int(m), 60) for m in rng.integers(1, 40, size=n_rows)]
    eng, orc = pair(make_engine, rows, max_limits=n_rows, max_batch_hits=50_000, capacity_cells=1 << 17)
    idx = rng.integers(0, 9000, size=50_000)
    lim = (idx * 4 + 3) % n_rows  # ids on both sides of 32768, one id per key
    hits = make_hits(W.splitmix64(idx.astype(np.uint64)), lim.astype(np.uint32), 1)
    for step in range(3):
        run_both(eng, orc, hits, NOW + step)
    assert int(lim.max()) > 32768
    assert_same_state(eng, orc)


@pytest.mark.parametrize("n", [700, 30_000])
def test_every_output_of_a_batch_is_defined_whatever_the_buffers_held(make_engine, n):
    """k_bkt_part writes "admitted" for every request and k_bkt_apply stores only the denials (k_bkt_tiny, the
    one-launch path of the 700-hit batch, stores every verdict): poisoned output buffers come back exact."""
    import ctypes as C

    import torch

    rng = np.random.default_rng(33)
    eng, orc = pair(make_engine, [(3, 60), (200, 60)], max_batch_hits=n, capacity_cells=1 << 16)
    dev = torch.device("cuda", 0)
    now = NOW
    for step in range(4):
        idx = (rng.zipf(1.3, size=n) - 1) % 5000
        h = make_hits(W.splitmix64(idx.astype(np.uint64)), (idx % 2).astype(np.uint32), 1)
        v, f, _r, _e = orc.check_and_update(h, now)
        d_hits = torch.from_numpy(h.view(np.int64).reshape(-1, 2).copy()).to(dev)
        d_v = torch.full((n,), 0xCC, dtype=torch.uint8, device=dev)
        d_f = torch.full((n,), 0x5A5A5A5A, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        eng.submit_device(d_hits.data_ptr(), n, now, d_v.data_ptr(), C.c_void_p(d_f.data_ptr()))
        eng.collect()
        torch.cuda.synchronize()
        assert np.array_equal(d_v.cpu().numpy(), v), f"batch {step}"
        assert np.array_equal(d_f.cpu().numpy(), f), f"batch {step}"
        now += 1000
    assert_same_state(eng, orc)


@pytest.mark.parametrize("compact", ["1", "0"])
def test_both_shapes_of_the_partition_on_4096_hit_tiles(make_engine, monkeypatch, compact):
    """Batches of more than 256 x 1024 hits take 4096-hit tiles: k_bkt_part_c (512 threads), k_bkt_part (1024 threads,
    RL_PART_COMPACT=0).  300 000 Zipf hits per batch (74 tiles, the last one ragged), hot keys with mixed
    limits, four batches so that the hot set is in use: every verdict and the final table against the oracle."""
    monkeypatch.setenv("RL_PART_COMPACT", compact)
    rng = np.random.default_rng(41)
    n, n_keys = 300_000, 200_000
    eng, orc = pair(make_engine, [(900, 60), (40, 60), (10**6, 60)], max_batch_hits=n, capacity_cells=1 << 20)
    now = NOW
    for step in range(4):
        idx = (rng.zipf(1.15, size=n) - 1) % n_keys
        h = make_hits(W.splitmix64(idx.astype(np.uint64)), (idx % 3).astype(np.uint32), 1 + (idx % 5 == 0).astype(np.uint32))
        run_both(eng, orc, h, now)
        now += 1000
    assert_same_state(eng, orc)


@pytest.mark.parametrize("compact", ["1", "0"])
def test_the_partition_kernels_agree_on_crowded_hot_sets_simple_counters_and_refusals(make_engine, monkeypatch, compact):
    """Both shapes of the partition kernel, each against the oracle (and therefore against each other):
    a FULL hot set (500+ keys over the threshold), simple counters in the batch (the cell must pre-exist,
    in_memory.rs:106-107), hot keys whose hits stop matching the set's
    predicted delta / limit, a ragged last tile, and the three refusals of the validation pass — unknown limit id, reserved
    key, missing simple cell — which must leave the table untouched and name the same error."""
    from limitador_amd.engine import EngineError

    monkeypatch.setenv("RL_PART_COMPACT", compact)
    rng = np.random.default_rng(77)
    n = 330_001  # 81 tiles of 4096, the last one ragged
    rows = [(700, 60), (30, 60), (10**7, 60), (5, 1), (10**9, 60)]
    eng, orc = pair(make_engine, rows, simple_keys=[(4, 9_100_004)], max_batch_hits=n, capacity_cells=1 << 20)
    hot_keys = W.splitmix64(np.arange(50_000, 50_560, dtype=np.uint64))  # 560 keys x ~330 hits: more than fit the set
    cold = W.splitmix64(np.arange(100_000, 400_000, dtype=np.uint64))
    now = NOW

    def batch(step):
        r = rng.random(n)
        hk = rng.integers(0, len(hot_keys), size=n)
        ck = rng.integers(0, len(cold), size=n)
        is_hot = r < 0.55
        is_simple = (~is_hot) & (r < 0.60)
        keys = np.where(is_hot, hot_keys[hk], np.where(is_simple, 9_100_004, cold[ck]))
        limits = np.where(is_hot, hk % 3, np.where(is_simple, 4 | RL_SIMPLE, 2 + ck % 2)).astype(np.uint32)  # (one id per key)
        delta = np.ones(n, dtype=np.uint32)
        if step >= 4:  # some hot keys now arrive with another delta than the set predicts: their buckets are replayed
            delta[is_hot & (hk % 7 == 0)] = 2
        return make_hits(keys, limits, delta)

    for step in range(7):
        run_both(eng, orc, batch(step), now)
        now += 400_000 if step == 5 else 1000  # (limit 3's one-second windows run out along the way)
    assert_same_state(eng, orc, n_simple_expected=1)
    before = np.sort(eng.dump_cells(), order="key")
    good = batch(0)
    for what, code in (("limit", -1), ("reserved", -1), ("missing_simple", -5)):
        bad = good.copy()
        at = int(rng.integers(0, n))
        if what == "limit":
            bad["limit"][at] = 77
        elif what == "reserved":
            bad["key"][at] = 2**64 - 1
        else:
            bad["key"][at], bad["limit"][at] = 9_100_099, 2 | RL_SIMPLE
        with pytest.raises(EngineError) as e:
            eng.check_and_update(bad, now)
        assert e.value.code == code, what
        assert np.array_equal(np.sort(eng.dump_cells(), order="key"), before), f"a refused batch ({what}) touched the table"
    run_both(eng, orc, good, now)
    assert_same_state(eng, orc, n_simple_expected=1)
