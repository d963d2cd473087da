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
