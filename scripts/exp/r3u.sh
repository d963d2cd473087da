#!/bin/bash
set -u
out=$PWD/gpurun_out/r3u; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
for env in "X=1" "RL_STAGE_CHUNK=0" "RL_STAGE_CHUNK=131072" "RL_STAGE_CHUNK=524288" "RL_STAGE_THREADS=15" "RL_STAGE_THREADS=3"; do
env $env python - <<'PY'
import os, time, numpy as np
from limitador_amd import workloads as W
from limitador_amd.engine import Engine
eng = Engine(capacity_cells=1<<22, max_batch_hits=1<<20)
eng.set_limits([(W.MAX_VALUE, W.WINDOW_S)])
rng = np.random.default_rng(1)
hits = W.uniform_batch(1_000_000, 1_000_000, rng)
now = W.NOW0_US
for _ in range(3): eng.check_and_update(hits, now); now += 1000
t0 = time.perf_counter(); K = 10
for _ in range(K): eng.check_and_update(hits, now, want_first_limited=False) if "want_first_limited" in eng.check_and_update.__code__.co_varnames else eng.check_and_update(hits, now); now += 1000
dt = (time.perf_counter() - t0) / K
print({k: v for k, v in os.environ.items() if k.startswith("RL_")}, round(dt*1e3, 3), "ms per 1M-hit call", round(1e6/dt/1e9, 2), "G/s")
PY
done
