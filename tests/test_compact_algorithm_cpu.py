"""The in-place compaction's algorithm without a GPU: scripts/model/compact_in_place.py restates k_compact_bounds +
k_compact_seg (limitador_amd/csrc/rl_kernels.hpp) slot for slot — segments between EMPTY slots of the unmodified table, waves
in ANY order, 64-slot steps through a read-ahead window, the free-slot bitmask of a cluster's first 64 slots, the slow path for
the rest — and checks what rl_compact promises: no tombstone left, exactly the live keys remain, every key reachable from
its home slot by linear probing.  (The kernel itself is compared with the oracle by tests/test_gpu_parity.py::
test_compaction_in_place_keeps_every_counter_reachable and the fuzz suite.)"""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("compact_model", os.path.join(ROOT, "scripts", "model", "compact_in_place.py"))
model = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(model)


@pytest.mark.parametrize("block", range(8))
def test_random_tables(block):
    """Capacities 64 .. 2048, loads 0.2 .. 0.85, any share of the keys deleted, keys inserted after the deletions (clusters
    that grew over tombstones), clustered homes (clusters of hundreds of slots, clusters that wrap around the table's end)."""
    for seed in range(block * 50, block * 50 + 50):
        model.case(seed)


def test_dense_table_with_every_other_key_deleted():
    model.run(1024, 860, 430, 7)
    model.run(2048, 1700, 1699, 8)  # one survivor
    model.run(256, 200, 0, 9)  # nothing to do
