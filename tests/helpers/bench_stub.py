"""A CPU stand-in for bench.py's GpuPlatform (tests/test_bench_sharded_cpu.py): bench.main() — the routed run's phases,
watchdog, fall-back agreement, timing protocol and JSON line — runs unchanged at world 2 over gloo; the three device-side
operations of a shard are numpy + the CPU oracle (the GPU versions are covered by tests/test_gpu_sharded*.py)."""
import ctypes

import numpy as np
import torch

import oracle
from limitador_amd.sharded import ShardedEngine, owner_of_tensor
from limitador_amd.wire import HIT_DTYPE

SEED = 0x9E3779B97F4A7C15


class StubEngine:
    """What bench.main() calls on an Engine, backed by the oracle."""

    def __init__(self, capacity_cells, max_batch_hits):
        self.capacity_cells = capacity_cells
        self.max_batch_hits = max_batch_hits
        self.hash_seed = SEED
        self.orc = oracle.OracleStorage()
        self.cells_loaded = 0
        self.hits = 0
        self.batches = 0

    def set_limits(self, rows):
        self.orc.set_limits(rows)

    def load_cells_device(self, ptr, n):
        if n == 0:
            return
        raw = np.ctypeslib.as_array((ctypes.c_int64 * (n * 4)).from_address(ptr)).reshape(n, 4).copy()
        self.orc.load_cells(raw[:, 0].view(np.uint64), (raw[:, 1] & 0xFFFFFFFF).astype(np.uint32), raw[:, 2].view(np.uint64),
                            raw[:, 3].view(np.uint64))
        self.cells_loaded += n

    def kernel_timing(self, mode):
        pass

    def kernel_timing_read(self, reset=False):
        return {"launches": 0, "ms": {}}

    def stats(self):
        return {"hits": self.hits, "batches": self.batches, "ordered_hits": 0, "capacity_cells": self.capacity_cells}

    def close(self):
        self.orc.close()


class OracleLocal:
    """HipLocal's contract (limitador_amd/sharded.py) on the CPU."""

    def __init__(self, eng):
        self.eng = eng

    def partition(self, hits, world, slot, counts_out):
        owners = owner_of_tensor(hits[:, 0], self.eng.hash_seed, world).numpy()
        perm = np.argsort(owners, kind="stable")
        counts_out.copy_(torch.from_numpy(np.bincount(owners, minlength=world).astype(np.int32)))
        return hits[torch.from_numpy(perm)].contiguous(), torch.from_numpy(perm.astype(np.int32))

    def check(self, hits, n, now_us, verdict):
        if n == 0:
            return False
        h = hits.numpy().view(HIT_DTYPE).reshape(-1)
        v, _, _, _ = self.eng.orc.check_and_update(h, now_us)
        verdict.copy_(torch.from_numpy(v))
        self.eng.hits += n
        self.eng.batches += 1
        return False

    def unpermute(self, src, perm, n, dst):
        dst[perm.long()] = src


class CpuPlatform:
    dist_backend = "gloo"

    def __init__(self, abi="missing", slow_init_s=0.0):
        """abi: what the C-ABI router's bring-up does — "missing" (rank 0 cannot make an id), "one_rank_fails" (rank 1's
        create raises: every rank must fall back together)."""
        self.device = torch.device("cpu")
        self.abi = abi
        self.slow_init_s = slow_init_s
        self.warmed = False

    def sync(self):
        pass

    def make_engine(self, capacity_cells, max_batch_hits):
        return StubEngine(capacity_cells, max_batch_hits)

    def warm_collectives(self, wd):
        import time

        wd.kick("stub warm-up", limit_s=self.slow_init_s + 30)
        time.sleep(self.slow_init_s)
        self.warmed = True
        return True, f"stub, {self.slow_init_s} s"

    def make_torch_sharded(self, eng, group, max_local_hits):
        return ShardedEngine(None, group, self.device, max_local_hits=max_local_hits, local=OracleLocal(eng))

    def abi_unique_id(self):
        if self.abi == "missing":
            raise RuntimeError("no librccl on this box (stub)")
        return bytes(range(1, 129))

    def make_abi_sharded(self, eng, world, rank, max_slice_hits, unique_id):
        assert unique_id == bytes(range(1, 129)), "the id rank 0 made must reach every rank"
        if rank == 1:
            raise RuntimeError("ncclCommInitRank failed (stub)")

        class _Up:
            closed = False

            def close(self):
                _Up.closed = True

        return _Up()
