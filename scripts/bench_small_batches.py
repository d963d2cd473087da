#!/usr/bin/env python
"""Pipelined single-counter batches of several sizes on a 1 M-key table (BASELINE.json configs[1] is the 64 k-hit row):
microseconds per batch, three in flight.  usage: python scripts/bench_small_batches.py [sizes...]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from limitador_amd import workloads as W  # noqa: E402
from limitador_amd.engine import Engine  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [4096, 16384, 65536, 262144]
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev)
gen.manual_seed(W.SEED)
out = {}
for n in sizes:
    e = Engine(capacity_cells=1 << 22, max_batch_hits=max(n, 1 << 16), device=0)
    e.set_limits([(W.MAX_VALUE, W.WINDOW_S)])
    rows = W.torch_universe_rows(1 << 20, dev)
    torch.cuda.synchronize()
    e.load_cells_device(rows.data_ptr(), rows.shape[0])
    b = [W.torch_batch(1 << 20, n, dev, gen, None) for _ in range(50)]
    v = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(4)]
    now = W.NOW0_US

    def run(steps):
        global now
        torch.cuda.synchronize()
        pending = 0
        t0 = time.perf_counter()
        for i in range(steps):
            e.submit_device(b[i % 50].data_ptr(), n, now, v[i & 3].data_ptr())
            now += 1000
            if pending == 2:
                e.collect()
            else:
                pending += 1
        while pending:
            e.collect()
            pending -= 1
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(50)
    dt = run(500)
    out[n] = round(dt / 500 * 1e6, 2)
    e.close()
print(json.dumps({"what": "us per pipelined batch, uniform keys on a 1 M-key table", "env": {k: v for k, v in os.environ.items() if k.startswith("RL_")}, "us_per_batch": out}))
