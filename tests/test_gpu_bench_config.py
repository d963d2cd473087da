"""Parity on EXACTLY what the driver times: bench.py's headline runs the RELEASE library
(limitador_amd/lib/, no -DRL_EXPERIMENT) on a 2^26-cell table — 10 M keys at load 0.15 — through
rl_check_and_update_submit_device / _collect, three batches in flight, with rl_kernel_timing(3) on (every fourth batch's
launches carry their own events).  Probe chains, bucket occupancy and the hot-set threshold all differ from the 2^25-cell
table tests/test_gpu_parity.py::test_config3_* use on the experiment build, so the same three traces are repeated here
at the bench's geometry, every verdict / first_limited / all 10 M cells compared with the oracle
(in_memory.rs:72-156; window restarts: atomic_expiring_value.rs:36-42,87-99).

This module only runs in a process that loads the release build: tests/test_gpu_release_lib.py starts it
(LIMITADOR_AMD_LIB=release); in the suite's own process (experiment build) it is skipped."""
import os

import numpy as np
import pytest

from limitador_amd import workloads as W
from test_gpu_parity import NOW, _full_size, make_engine  # noqa: F401  (make_engine is the fixture)

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("LIMITADOR_AMD_LIB", "") == "exp",
                                 reason="the bench's configuration is checked on the release build (tests/test_gpu_release_lib.py)")]

BENCH_CAP = 1 << 26  # bench.py: cap = 2^ceil(log2(10 M * 4.4))
N_KEYS, N_HITS = 10_000_000, 1_000_000


def _release_build_loaded():
    from limitador_amd import build as b

    assert not b.EXPERIMENT and b.ENGINE_SO.endswith("limitador_amd/lib/librl_engine.so")
    blob = open("/proc/self/maps").read()
    assert "lib/exp/librl_engine.so" not in blob


def test_bench_table_size_plain_trace(make_engine):
    """configs[2] as bench.py runs it: 8 Zipf-0.99 batches, the clock advancing 1 ms per batch."""
    eng = _full_size(make_engine, N_KEYS, N_HITS, steps=8, zipf=True, in_flight=True, capacity_cells=BENCH_CAP, timing_mode=3)
    _release_build_loaded()
    st = eng.stats()
    assert st["capacity_cells"] == BENCH_CAP and st["hits"] == 8 * N_HITS


def test_bench_table_size_every_window_ends_between_two_batches(make_engine):
    """secondary.headline_with_expiry's shape: the clock jumps past every pre-populated expiry inside the run."""
    nows = [NOW, NOW + 1000, NOW + 31_000_000, NOW + 31_001_000, NOW + 31_002_000, NOW + 92_000_000]
    eng = _full_size(make_engine, N_KEYS, N_HITS, steps=6, zipf=True, in_flight=True, nows=nows, capacity_cells=BENCH_CAP,
                     timing_mode=3)
    rows = eng.dump_cells()
    reset = rows["expiry_us"] > np.uint64(NOW + 60_000_000)
    assert 0.05 * len(rows) < int(reset.sum()) < 0.6 * len(rows)
    assert int(rows["value"][reset].max()) <= W.MAX_VALUE


def test_bench_table_size_a_third_of_the_windows_end_inside_the_run(make_engine):
    """Live, expired-and-reset and never-touched expired cells side by side in every bucket, three batches in flight."""
    eng = _full_size(make_engine, N_KEYS, N_HITS, steps=6, zipf=True, in_flight=True, early_third=True,
                     capacity_cells=BENCH_CAP, timing_mode=3)
    rows = eng.dump_cells()
    assert int((rows["expiry_us"] > np.uint64(NOW + 59_000_000)).sum()) > 100_000
