"""The C ABI used from C++ with no Python in the loop: tests/cpp/abi_harness.cpp is compiled against
include/rl_engine.h + librl_engine.so (and the C oracle as the checker) and run on the MI355X.  It drives
check_and_update (CSR requests, load_counters, u64 deltas), is_within_limits, update_counter, sweep, delete,
get_counters, dump, a snapshot file round trip into a second engine, and the error path."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_drives_the_c_abi_against_the_c_oracle(tmp_path):
    import oracle
    from limitador_amd import build as b

    oracle.build()
    b.build_engine()
    exe = str(tmp_path / "abi_harness")
    lib, orc = os.path.join(ROOT, "limitador_amd", "lib"), os.path.join(ROOT, "oracle")
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", os.path.join(ROOT, "tests", "cpp", "abi_harness.cpp"),
           "-I" + os.path.join(ROOT, "include"), "-I" + orc, "-L" + lib, "-lrl_engine", "-L" + orc, "-llimitador_oracle",
           "-Wl,-rpath," + lib, "-Wl,-rpath," + orc, "-o", exe]
    subprocess.run(cmd, check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    sys.stdout.write(r.stdout)
    sys.stderr.write(r.stderr)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "abi_harness ok" in r.stdout
