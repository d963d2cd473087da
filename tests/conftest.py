import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The test-suite walks every engine mode (stream forms, kernel shapes, pipeline depths) by environment switches that only
# the -DRL_EXPERIMENT build of the libraries reads (limitador_amd/lib/exp/, limitador_amd/build.py); the release
# build — what bench.py, smoke() and a host link — is run by tests/test_gpu_release_lib.py in a process of its own
# (LIMITADOR_AMD_LIB=release there).  Must be set before limitador_amd is imported.
os.environ.setdefault("LIMITADOR_AMD_LIB", "exp")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Tests marked `gpu` need a MI355X: on a box without one they are SKIPPED (with the reason), not failed —
    `pytest tests` then reads the same as `pytest tests -m "not gpu"`.  The engine has no CPU path to fall back to
    (rl_engine_create answers RL_ERR_NO_DEVICE)."""
    if not any("gpu" in item.keywords for item in items):
        return
    try:
        import torch

        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="needs a MI355X (no HIP device here; the engine has no CPU fallback)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    import oracle

    oracle.build()


@pytest.fixture(scope="session")
def engine_lib():
    """The built HIP library; building is `__graft_entry__.build()`'s job, tests only load it."""
    from limitador_amd import _lib

    return _lib.load()


_RCCL_PROBE = r"""
import os, sys, socket, datetime
import torch, torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=200))
t = torch.ones(1024, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
dist.destroy_process_group(); print("rccl ok")
"""


@pytest.fixture(scope="session")
def rccl_ready():
    """The first RCCL initialisation on a fresh box maps a few hundred MB of device code (librccl.so) and has been seen to
    take minutes — once long enough to run a whole suite into its time limit (gpurun_out/r13a/suite_1.log).  Tests that
    bring up a communicator ask for this fixture first: one world-1 initialisation + all-reduce in a PROCESS OF ITS OWN
    (so that a stuck one can be killed), twice if need be; the box's page cache is warm afterwards.  If RCCL cannot be
    brought up at all the tests that need it are skipped WITH that reason — the routed step is also covered without RCCL
    (in-process and multi-process transports)."""
    import subprocess

    last = ""
    for attempt in range(2):
        try:
            p = subprocess.run([sys.executable, "-c", _RCCL_PROBE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                               timeout=240)
            last = p.stdout[-400:]
            if p.returncode == 0 and "rccl ok" in p.stdout:
                return True
        except subprocess.TimeoutExpired:
            last = f"attempt {attempt + 1}: no world-1 communicator within 240 s"
    pytest.skip(f"RCCL cannot be initialised on this box ({last.strip()[-200:]}): tests over an RCCL communicator are skipped")
