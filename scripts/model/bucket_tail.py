#!/usr/bin/env python3
"""CPU model of the replay's tail (DESIGN.md §8 "Left" 2): how long are the hash buckets of a configs[2] batch, and how
many 256-hit rounds does the slowest workgroup of k_bkt_step walk, for a given number of hash buckets / hot buckets /
round size?  No GPU: the batch is the bench's own generator (limitador_amd/workloads.py), the bucket of a key is the top
bits of a 64-bit mix of it (what bucket_of_hash does with fmix64(key ^ seed)), the hot set is "the keys with the most
hits in the batch before, at most HOT of them, at least THR hits" (what the promotion rules converge to).

A workgroup of the replay costs ~6.5 us + ~5.3 us per round (RL_APPLY_TRACE phase stamps); every
workgroup is resident at once, so the kernel's span is its slowest workgroup.

usage: python scripts/model/bucket_tail.py [--batches 4]
"""
import argparse
import sys, os

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from limitador_amd import workloads as W  # noqa: E402


def mix(x):
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xFF51AFD7ED558CCD)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xC4CEB9FE1A85EC53)
        x ^= x >> np.uint64(33)
    return x


def bucket_of(keys, nb):
    h = mix(keys ^ np.uint64(0x1234567))
    # top bits for a power of two, multiply-shift otherwise
    return ((h >> np.uint64(32)) * np.uint64(nb)) >> np.uint64(32)


def model(prev, cur, nb, hot_max, thr, rnd):
    pk, pc = np.unique(prev["key"], return_counts=True)
    order = np.argsort(-pc)
    hot = pk[order[:hot_max]]
    hot = hot[pc[order[:hot_max]] >= thr]
    is_hot = np.isin(cur["key"], hot)
    cold = cur["key"][~is_hot]
    b = bucket_of(cold, nb).astype(np.int64)
    lens = np.bincount(b, minlength=nb)
    rounds = -(-lens // rnd)
    t = 6.5 + 5.3 * rounds * (rnd / 256.0) ** 0.5  # (a wider round costs more than a 256-hit one, less than proportionally)
    return dict(hot_keys=len(hot), hot_frac=float(is_hot.mean()), mean_len=float(lens.mean()), max_len=int(lens.max()),
                mean_rounds=float(rounds.mean()), max_rounds=int(rounds.max()), mean_us=float(t.mean()), span_us=float(t.max()),
                hist=np.bincount(rounds, minlength=8)[:8].tolist())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=3)
    ap.add_argument("--keys", type=int, default=10_000_000)
    ap.add_argument("--hits", type=int, default=1_000_000)
    a = ap.parse_args()
    rng = np.random.default_rng(W.SEED)
    cdf = W.zipf_cdf(a.keys)
    bs = [W.zipf_batch(a.keys, a.hits, rng, cdf) for _ in range(a.batches + 1)]
    print("nb hot thr round | hot keys, hot frac | bucket mean / max | rounds mean / max | model us mean / span | workgroups by rounds 0..7")
    for nb, hot_max, thr, rnd in [(1024, 512, 100, 256), (1024, 512, 40, 256), (1024, 1024, 40, 256), (1024, 2048, 20, 256),
                                  (1280, 512, 100, 256), (1536, 512, 100, 256), (2048, 512, 100, 256),
                                  (1280, 1024, 40, 256), (1536, 1024, 40, 256),
                                  (1024, 512, 100, 320), (1024, 512, 100, 384), (1024, 512, 100, 512), (768, 512, 100, 512),
                                  (1024, 1024, 40, 384)]:
        rs = [model(bs[i], bs[i + 1], nb, hot_max, thr, rnd) for i in range(a.batches)]
        m = {k: np.mean([r[k] for r in rs]) for k in rs[0] if k != "hist"}
        hist = np.mean([r["hist"] for r in rs], axis=0).round().astype(int).tolist()
        print(f"{nb:5d} {hot_max:5d} {thr:4d} {rnd:4d} | {m['hot_keys']:6.0f} {m['hot_frac']:.3f} | {m['mean_len']:6.0f} {m['max_len']:6.0f} | "
              f"{m['mean_rounds']:.2f} {m['max_rounds']:.1f} | {m['mean_us']:5.1f} {m['span_us']:5.1f} | {hist}")


if __name__ == "__main__":
    main()
