"""Host logic of the synthetic workload generators (no GPU)."""
import numpy as np
import torch

from limitador_amd import workloads as W


def test_splitmix_numpy_matches_torch():
    idx = np.array([0, 1, 2, 12345678, 2**40 + 17, 2**63 - 1], dtype=np.uint64)
    a = W.splitmix64(idx)
    b = W._t_splitmix64(torch.from_numpy(idx.astype(np.int64))).numpy().astype(np.uint64)
    assert (a == b).all()
    # published splitmix64 first output for seed 0
    assert int(W.splitmix64(np.array([0], dtype=np.uint64))[0]) == 0xE220A8397B1DCDAF


def test_universe_rows_numpy_matches_torch():
    n = 5000
    a = W.universe_rows(n)
    b = W.torch_universe_rows(n, "cpu").numpy()
    assert (a["key"] == b[:, 0].astype(np.uint64)).all()
    assert (a["value"] == b[:, 2].astype(np.uint64)).all()
    assert (a["expiry_us"] == b[:, 3].astype(np.uint64)).all()
    assert a["value"].max() <= W.MAX_VALUE


def test_zipf_batch_is_skewed_and_in_universe():
    n = 100_000
    rng = np.random.default_rng(W.SEED)
    hits = W.zipf_batch(n, 50_000, rng)
    keys = set(W.splitmix64(np.arange(n, dtype=np.uint64)).tolist())
    assert set(hits["key"].tolist()) <= keys
    _, counts = np.unique(hits["key"], return_counts=True)
    assert counts.max() > 0.03 * 50_000  # the hottest key carries several percent of the batch


def test_torch_batch_layout_is_rl_hit():
    g = torch.Generator().manual_seed(1)
    t = W.torch_batch(1000, 64, "cpu", g, limit=3, delta=2)
    raw = t.numpy().view(W.HIT_DTYPE).reshape(-1)
    assert (raw["limit"] == 3).all() and (raw["delta"] == 2).all()
