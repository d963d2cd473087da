"""The device-fault class, bounded: GPU AddressSanitizer is not available on this pool, so the experiment
build can put RED ZONES around every device array of the engine (RL_REDZONE=1: 4 KB of a known pattern on either side, checked
when the block is freed — rl_engine_destroy at the latest — and a damaged one ends the process with the block's size and the
offset).  The parity suites of every path that runs the general resolver, the hot path, the matcher and the wire decoder are run
again under it, in a process of their own: they must still pass, i.e. no kernel stored a byte within 4 KB outside any array on
any of their traces."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = ["tests/test_gpu_parity.py", "tests/test_gpu_bucketed.py", "tests/test_gpu_general_variants.py", "tests/test_gpu_match.py",
          "tests/test_gpu_rls_e2e.py", "tests/test_gpu_kuadrant.py", "tests/test_gpu_merge.py", "tests/test_gpu_sharded_multi.py"]


def test_red_zones_catch_a_store_behind_an_array():
    """The mechanism itself: a block whose zone is written over is reported (rl_debug_redzones), by size and offset."""
    import ctypes as C

    import torch

    from limitador_amd import _lib
    from limitador_amd.engine import Engine

    lib = _lib.load()
    if not hasattr(lib, "rl_debug_redzones"):
        pytest.skip("release build: no red zones")
    os.environ["RL_REDZONE"] = "1"
    try:
        eng = Engine(capacity_cells=1 << 10, max_batch_hits=1 << 10)
    finally:
        del os.environ["RL_REDZONE"]
    lib.rl_debug_redzones.argtypes = [C.POINTER(C.c_uint32), C.c_char_p, C.c_uint32]
    n, msg = C.c_uint32(0), C.create_string_buffer(200)
    assert lib.rl_debug_redzones(C.byref(n), msg, 200) == 0 and n.value > 10
    eng.close()  # (intact zones: the blocks are freed quietly)


def test_the_parity_suites_leave_every_red_zone_intact():
    files = [f for f in SUITES if os.path.exists(os.path.join(ROOT, f))]
    env = dict(os.environ, RL_REDZONE="3", LIMITADOR_AMD_LIB="exp")  # (3: the arrays themselves start as the pattern too)
    r = subprocess.run([sys.executable, "-m", "pytest", *files, "-m", "gpu", "-x", "-q", "--timeout", "300", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout[-1500:] + r.stderr[-1500:])
    assert "RL_REDZONE:" not in r.stderr, tail
    assert r.returncode == 0, tail
    assert " passed" in r.stdout, tail
