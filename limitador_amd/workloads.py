"""Synthetic workloads of BASELINE.json's configs, numpy and torch flavours.

config #2  1 GPU, 1 048 576 keys, uniform 65 536-hit batches, max 1000 / 60 s, delta 1
config #3  1 GPU, 10 000 000 keys, Zipf(0.99) 1 000 000-hit batches, max 1000 / 60 s, delta 1
           (fixed-window semantics: the reference has no sliding window)

Keys: key(i) = splitmix64(i); the table is pre-populated with value(i) = splitmix64(i ^ seed) % 1001
and expiry = now + 30 s.  Zipf rank r (0-based) maps to key index (r * PERM_MULT) mod N so that hot
keys are not neighbours in index space.
"""
import numpy as np

from .wire import CELL_ROW_DTYPE, HIT_DTYPE

SEED = 42  # limitador/benches/bench.rs:21
PERM_MULT = 6364137  # coprime with 10^7 and with 2^k
MAX_VALUE = 1000
WINDOW_S = 60
NOW0_US = 1_700_000_000_000_000
M64 = (1 << 64) - 1


def splitmix64(x):
    """Vectorised splitmix64 finaliser of (x + golden) over uint64 arrays."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    # keep clear of the two reserved tags
    return np.minimum(z, np.uint64(0xFFFFFFFFFFFFFFFD))


def universe_rows(n_keys, now_us=NOW0_US, limit=0, seed=SEED, lo=0, hi=None):
    """rl_cell_row array pre-populating keys [lo, hi) of an n_keys universe."""
    hi = n_keys if hi is None else hi
    idx = np.arange(lo, hi, dtype=np.uint64)
    rows = np.zeros(hi - lo, dtype=CELL_ROW_DTYPE)
    rows["key"] = splitmix64(idx)
    rows["limit"] = limit
    rows["value"] = splitmix64(idx ^ np.uint64(seed)) % np.uint64(MAX_VALUE + 1)
    rows["expiry_us"] = now_us + 30_000_000
    return rows


def zipf_cdf(n_keys, s=0.99):
    w = np.arange(1, n_keys + 1, dtype=np.float64) ** (-s)
    c = np.cumsum(w)
    return c / c[-1]


def uniform_batch(n_keys, n_hits, rng, limit=0, delta=1):
    idx = rng.integers(0, n_keys, size=n_hits, dtype=np.uint64)
    hits = np.empty(n_hits, dtype=HIT_DTYPE)
    hits["key"] = splitmix64(idx)
    hits["limit"] = limit
    hits["delta"] = delta
    return hits


def zipf_batch(n_keys, n_hits, rng, cdf=None, limit=0, delta=1):
    cdf = zipf_cdf(n_keys) if cdf is None else cdf
    ranks = np.searchsorted(cdf, rng.random(n_hits), side="left").astype(np.uint64)
    ranks = np.minimum(ranks, np.uint64(n_keys - 1))
    idx = (ranks * np.uint64(PERM_MULT)) % np.uint64(n_keys)
    hits = np.empty(n_hits, dtype=HIT_DTYPE)
    hits["key"] = splitmix64(idx)
    hits["limit"] = limit
    hits["delta"] = delta
    return hits


# ---- torch (device-resident) flavours for bench.py ------------------------------------------

def _t_splitmix64(x):
    """splitmix64 on int64 torch tensors (two's-complement wrap == uint64 wrap)."""
    import torch

    def lsr(v, k):  # logical shift right on int64
        return (v >> k) & ((1 << (64 - k)) - 1)

    def c(v):  # uint64 constant as int64
        return v - (1 << 64) if v >= (1 << 63) else v

    z = x + c(0x9E3779B97F4A7C15)
    z = (z ^ lsr(z, 30)) * c(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * c(0x94D049BB133111EB)
    z = z ^ lsr(z, 31)
    # clamp the two reserved tags (-1, -2 as int64)
    return torch.where(z >= -2, torch.where(z < 0, torch.full_like(z, -3), z), z)


def torch_universe_rows(n_keys, device, now_us=NOW0_US, limit=0, seed=SEED, keep=None, lo=0, hi=None):
    """[n,4] int64 tensor laid out as rl_cell_row for keys [lo, hi) of the universe; keep(key_tensor)->bool mask selects a shard."""
    import torch

    idx = torch.arange(lo, n_keys if hi is None else hi, dtype=torch.int64, device=device)
    keys = _t_splitmix64(idx)
    if keep is not None:
        m = keep(keys)
        idx, keys = idx[m], keys[m]
    vals = _t_splitmix64(idx ^ seed)
    # unsigned modulo of an int64 bit pattern: split into high/low halves
    v_hi = (vals >> 32) & 0xFFFFFFFF
    v_lo = vals & 0xFFFFFFFF
    vals = ((v_hi % (MAX_VALUE + 1)) * ((1 << 32) % (MAX_VALUE + 1)) + v_lo % (MAX_VALUE + 1)) % (MAX_VALUE + 1)
    rows = torch.empty((keys.shape[0], 4), dtype=torch.int64, device=device)
    rows[:, 0] = keys
    rows[:, 1] = limit  # limit in the low 32 bits, reserved = 0
    rows[:, 2] = vals
    rows[:, 3] = now_us + 30_000_000
    return rows


def torch_zipf_cdf(n_keys, device, s=0.99):
    import torch

    w = torch.arange(1, n_keys + 1, dtype=torch.float64, device=device) ** (-s)
    c = torch.cumsum(w, 0)
    return c / c[-1]


def torch_batch(n_keys, n_hits, device, gen, cdf=None, limit=0, delta=1):
    """[n,2] int64 tensor laid out as rl_hit (key | limit + delta<<32). cdf=None -> uniform."""
    import torch

    if cdf is None:
        idx = torch.randint(0, n_keys, (n_hits,), dtype=torch.int64, device=device, generator=gen)
    else:
        u = torch.rand(n_hits, dtype=torch.float64, device=device, generator=gen)
        ranks = torch.searchsorted(cdf, u).clamp_(max=n_keys - 1)
        idx = (ranks * PERM_MULT) % n_keys
    hits = torch.empty((n_hits, 2), dtype=torch.int64, device=device)
    hits[:, 0] = _t_splitmix64(idx)
    hits[:, 1] = limit | (delta << 32)
    return hits
