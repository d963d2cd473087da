"""Offline look at a raw RL_APPLY_TRACE_FILE dump: per-workgroup phase times of k_bkt_apply against the bucket's hits.
usage: apply_trace.py <file>"""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.float64)
n_wg, n_hot_wgs = int(raw[0, 0]), int(raw[0, 1])
raw = raw[1:]
tp = raw[n_wg:]
tp = tp[tp[:, 6] > 0]  # the partition role's workgroups (next batch), if any ran in this launch
t = raw[:n_wg]
t = t[t[:, 5] > 0]
th = t[len(t) - n_hot_wgs:]  # the workgroups that walk the hot work items
t = t[:len(t) - n_hot_wgs]
_t0 = min(t[:, 0].min(), th[:, 0].min())
hs, he, hn = (th[:, 0] - _t0) / 100, (th[:, 5] - _t0) / 100, th[:, 7]
print(f"{len(th)} hot workgroups: start p50 {np.median(hs):.1f} max {hs.max():.1f}; end p50 {np.median(he):.1f} max {he.max():.1f}; "
      f"items per workgroup mean {hn.mean():.1f} max {hn.max():.0f}; time per item {((he - hs).sum() / max(1, hn.sum())):.2f} us")
if len(tp):
    ps = (tp[:, 0] - _t0) / 100
    names = ("records requested + LDS init", "hot table", "walk 1", "scan", "walk 2", "flags")
    print(f"{len(tp)} partition workgroups: start p50 {np.median(ps):.1f} max {ps.max():.1f}; end p50 {np.median((tp[:, 6] - _t0) / 100):.1f} max {((tp[:, 6] - _t0) / 100).max():.1f}")
    for k, nm in enumerate(names):
        v = (tp[:, k + 1] - tp[:, k]) / 100
        print(f"   {nm:30s} mean {v.mean():6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f} us")
t0 = t[:, 0].min()
start = (t[:, 0] - t0) / 100
view = (np.where(t[:, 1] > 0, t[:, 1], t[:, 0]) - t[:, 0]) / 100
bucket = (t[:, 2] - np.where(t[:, 1] > 0, t[:, 1], t[:, 0])) / 100
hot = (t[:, 3] - t[:, 2]) / 100
end = (t[:, 5] - t0) / 100
total = (t[:, 5] - t[:, 0]) / 100
hits = t[:, 6]
print(f"{len(t)} workgroups; span {end.max():.1f} us; start: max {start.max():.1f} us")
for name, v in (("start", start), ("view", view), ("bucket", bucket), ("hot", hot), ("total", total), ("end", end), ("hits", hits)):
    q = np.percentile(v, [0, 10, 50, 90, 99, 100])
    print(f"{name:7s} min {q[0]:7.1f}  p10 {q[1]:7.1f}  p50 {q[2]:7.1f}  p90 {q[3]:7.1f}  p99 {q[4]:7.1f}  max {q[5]:7.1f}")
rounds = np.ceil(hits / 256)
for r in np.unique(rounds):
    m = rounds == r
    print(f"rounds {int(r)}: {m.sum():4d} workgroups, bucket time mean {bucket[m].mean():6.1f} p90 {np.percentile(bucket[m], 90):6.1f} max {bucket[m].max():6.1f}; hot mean {hot[m].mean():5.1f}")
worst = np.argsort(-total)[:12]
print("slowest workgroups: (index, start, view, bucket, hot, total, hits, hot bucket+1)")
for i in worst:
    print(f"  {i:5d} {start[i]:6.1f} {view[i]:6.1f} {bucket[i]:6.1f} {hot[i]:6.1f} {total[i]:6.1f} {int(hits[i]):6d} {int(t[i, 7]):4d}")
print("corr(total, hits) =", np.corrcoef(total, hits)[0, 1], " corr(hot, total) =", np.corrcoef(hot, total)[0, 1])
