"""PCIe-inclusive rate of the host-buffer entry point (rl_check_and_update_batch: 16 B/hit in, 1 B/hit
verdict + 4 B/hit first_limited out) on the bench workload — the number DESIGN.md §5 quotes next to
`value`, never `value` itself.  Needs a MI355X."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from limitador_amd import workloads as W  # noqa: E402
from limitador_amd.engine import Engine  # noqa: E402

n_keys, n_hits, steps = 10_000_000, 1_000_000, 10
eng = Engine(capacity_cells=1 << 27, max_batch_hits=n_hits)
eng.set_limits([(W.MAX_VALUE, W.WINDOW_S)])
rows = W.universe_rows(n_keys)
for lo in range(0, n_keys, 1 << 20):
    eng.load_cells(rows[lo:lo + (1 << 20)])
rng = np.random.default_rng(W.SEED)
cdf = W.zipf_cdf(n_keys)
batches = [W.zipf_batch(n_keys, n_hits, rng, cdf) for _ in range(steps + 2)]
now = W.NOW0_US
for b in batches[:2]:
    eng.check_and_update(b, now)
    now += 1000
t0 = time.perf_counter()
for b in batches[2:]:
    eng.check_and_update(b, now)
    now += 1000
dt = time.perf_counter() - t0
print(json.dumps({"entry": "rl_check_and_update_batch (host buffers, pageable numpy arrays)",
                  "decisions_per_s": steps * n_hits / dt, "ms_per_batch": dt / steps * 1e3,
                  "bytes_over_pcie_per_hit": 21}))
eng.close()
