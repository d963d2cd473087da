"""The RELEASE build of the libraries (limitador_amd/lib/, no -DRL_EXPERIMENT) in a process of its own: the rest of the
suite loads the experiment build so that it can walk every engine mode by environment switches (tests/conftest.py).
Same sources, same kernels; this file proves the shipped binaries pass the reference's scenarios, the smoke check and a
pipelined trace too.  Needs a MI355X."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, timeout):
    env = dict(os.environ, LIMITADOR_AMD_LIB="release")
    for k in list(env):  # a release library ignores them anyway; keep the child's environment clean
        if k.startswith(("RL_", "RLI_")):
            del env[k]
    p = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=timeout, text=True)
    assert p.returncode == 0, p.stdout[-4000:]
    return p.stdout


def test_smoke_on_the_release_build():
    out = _run(["-c", "import __graft_entry__ as g; g.smoke()"], 300)
    assert "smoke ok" in out


def test_reference_scenarios_and_the_pipelined_hot_path_on_the_release_build():
    out = _run(["-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_parity.py", "tests/test_gpu_bucketed.py", "-k",
                "served or three_batches_in_flight or config2 or sweeps_between"], 600)
    assert " passed" in out and "lib/exp" not in out


def test_the_bench_configuration_at_full_size_on_the_release_build():
    """bench.py's headline geometry — 10 M keys in a 2^26-cell table, 1 M-hit Zipf batches, three in flight, timed
    launches — plain and with both window-expiry variants, all 10 M cells compared (tests/test_gpu_bench_config.py)."""
    out = _run(["-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_bench_config.py"], 900)
    assert "3 passed" in out and "skipped" not in out.split("\n")[-2]
