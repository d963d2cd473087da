"""bench.py's routed main() end to end at world 2 over gloo, on the CPU (VERDICT r05 "next" #2d).

The engine is a stand-in (tests/helpers/bench_stub.py: numpy + the oracle); everything else is bench.py's own code: the
progress-based watchdog and its phases, the chunked pre-population of each rank's shard, the bring-up of the C-ABI router
with the every-rank-falls-back agreement, the torch.distributed router (limitador_amd/sharded.py), the barrier-bracketed
timed region with max-over-ranks, and the ONE JSON line of rank 0."""
import io
import json
import os
import socket
import sys
from contextlib import redirect_stdout

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, argv, abi, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    from helpers.bench_stub import CpuPlatform

    plat = CpuPlatform(abi=abi)
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main(argv, platform=plat)
    q.put((rank, buf.getvalue(), plat.warmed))


@pytest.mark.parametrize("impl,abi", [("torch", "missing"), ("abi", "missing"), ("abi", "one_rank_fails")],
                         ids=["torch_router", "abi_router_no_id", "abi_router_one_rank_fails"])
def test_bench_routed_main_runs_end_to_end_at_world_2(impl, abi):
    world = 2
    argv = ["--gpus", str(world), "--steps", "4", "--warmup", "2", "--keys", "6000", "--batch", "1500", "--secondary", "0",
            "--cpu-seconds", "0", "--sharded-impl", impl, "--stall-seconds", "60", "--init-seconds", "120"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, argv, abi, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, out, warmed = q.get(timeout=240)
        got[r] = (out, warmed)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1][0].strip() == "", "only rank 0 prints"
    lines = [ln for ln in got[0][0].splitlines() if ln.strip()]
    assert len(lines) == 1, "ONE JSON line"
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["steps"] == 4 and line["warmup"] == 2
    assert line["scaling"] == "weak" and line["unit"] == "decisions/s" and line["higher_is_better"] is True
    # whole-job throughput: every rank's slice counts
    assert line["value"] == pytest.approx(1500 * world * 4 / (line["ms_per_step"] * 4 / 1e3), rel=1e-6)
    par = line["config"]["parallelism"]
    assert "hash-sharded x2" in par
    if impl == "abi":
        assert "fell back from the C-ABI router" in par and "(torch.distributed)" in par
    bring = line["config"]["bringup_s"]
    for ph in ("init_process_group", "first collective", "engine", "pre-populate", "batches", "warm-up steps", "timed region"):
        assert ph in bring, ph
    assert got[0][1] and got[1][1], "the collectives' warm-up ran on every rank before the rendezvous"
    # each rank loaded about half of the 2 x 6000 keys (its own shard, found by filtering pieces of the universe)
    assert 0.4 * 6000 * world / 2 < bring["cells_loaded_rank0"] < 1.6 * 6000 * world / 2
    assert line["cpu_baseline"] is None  # N > 1: rank 0 does not time the CPU leg


def test_bench_prepopulation_covers_the_universe_once_across_ranks():
    """Every key of the N x keys universe is loaded by exactly one rank (chunked walk, each rank starting at its own N-th)."""
    import numpy as np
    import torch

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers.bench_stub import SEED
    from limitador_amd import workloads as W
    from limitador_amd.sharded import owner_mask

    world, keys = 3, 5000
    total = world * keys
    seen = []
    for rank in range(world):
        piece = 4096  # (bench.py walks pieces of 4 M; the walk is the same)
        n_pieces = (total + piece - 1) // piece
        first = (rank * n_pieces) // world
        mine = []
        for j in range(n_pieces):
            lo = ((first + j) % n_pieces) * piece
            rows = W.torch_universe_rows(total, torch.device("cpu"), keep=lambda k: owner_mask(k, SEED, world, rank), lo=lo,
                                         hi=min(total, lo + piece))
            mine.append(rows[:, 0].numpy().view(np.uint64))
        seen.append(np.concatenate(mine))
    allk = np.concatenate(seen)
    assert len(allk) == total and len(np.unique(allk)) == total
    assert np.array_equal(np.sort(allk), np.sort(W.universe_rows(total)["key"]))
