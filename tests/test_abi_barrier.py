"""No C++ exception crosses the C ABI: the reference's errors are values
(storage/mod.rs:312-339) and its in-memory path never fails (in_memory.rs:72-156), so a host that links these libraries
must get a status back where the C++ behind an entry point throws — std::bad_alloc from a vector that a bogus size
reached, std::length_error, anything — instead of std::terminate taking the process down (SIGABRT).

Every entry point ends in the same barrier (limitador_amd/csrc/rl_abi_guard.h); each library exports a self-test that
throws behind it.  Runs without a GPU: in a child process, so that a missing barrier shows up as the child's SIGABRT
and not as the end of the pytest session."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RL_ERR_NOMEM, RL_ERR_INTERNAL = -8, -11

CHILD = r"""
import sys
sys.path.insert(0, %r)
from limitador_amd import _lib, host_storage, ingest, sharded_abi
eng = _lib.load()
tests = [("rl_abi_selftest", eng.rl_abi_selftest)]
tests.append(("rls_abi_selftest", host_storage.load().rls_abi_selftest))
ingest._lib()
tests.append(("rli_abi_selftest", ingest.SYMBOLS["rli_abi_selftest"]))
sharded_abi.load()
tests.append(("rl_sharded_abi_selftest", sharded_abi.SYMBOLS["rl_sharded_abi_selftest"]))
for name, fn in tests:
    for kind in (0, 1, 2, 3, 4) + ((5,) if name == "rl_abi_selftest" else ()):
        rc = fn(kind)
        msg = eng.rl_last_internal_error().decode()
        print(name, kind, rc, msg, flush=True)
print("survived")
"""


@pytest.mark.parametrize("flavour", ["exp", "release"])
def test_exceptions_stop_at_the_c_abi(flavour):
    env = dict(os.environ, LIMITADOR_AMD_LIB=flavour)
    p = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, f"child died with {p.returncode} (an exception crossed the boundary?):\n{p.stdout[-3000:]}"
    assert "survived" in p.stdout
    rows = [l.split(" ", 3) for l in p.stdout.splitlines() if "_selftest " in l]
    assert len(rows) == 5 * 4 + 1
    for name, kind, rc, msg in rows:
        kind, rc = int(kind), int(rc)
        if kind == 0:
            assert rc == 0
        elif kind in (1, 4):  # std::bad_alloc thrown / a vector of 2^58 elements: out of memory is a VALUE
            assert rc == RL_ERR_NOMEM, (name, kind, rc, msg)
            assert name in msg and "bad_alloc" in msg
        else:  # std::length_error, a non-std object, reserve() past max_size
            assert rc == RL_ERR_INTERNAL, (name, kind, rc, msg)
            assert name in msg


def test_every_allocating_entry_point_ends_in_the_barrier():
    """The sources themselves: inside the extern "C" block of each of the four files, every non-static `int32_t r…(`
    definition with a body of its own is a function-try-block closed by RL_ABI_CATCH (one-line bodies that only compare
    integers or read a field are exempt)."""
    files = ["limitador_amd/csrc/rl_engine.hip", "limitador_amd/csrc/host/gpu_counter_storage.cpp",
             "limitador_amd/csrc/host/ingest.cpp", "limitador_amd/csrc/host/rl_sharded.cpp"]
    total = 0
    for f in files:
        lines = open(os.path.join(ROOT, f)).read().split("\n")
        start = next(i for i, l in enumerate(lines) if l.startswith('extern "C" {'))
        end = max(i for i, l in enumerate(lines) if l.startswith('}  // extern "C"'))
        i = start
        while i < end:
            m = re.match(r"^int32_t (r[a-z0-9_]+)\(", lines[i])
            if m and m.group(1) != "rl_abi_caught":  # (the barrier's own landing pad: snprintf into a fixed buffer)
                j = i
                while not re.search(r"[{};]\s*$", lines[j]):
                    j += 1
                sig_end = lines[j].rstrip()
                if sig_end.endswith("{"):
                    assert sig_end.endswith("try {"), f"{f}:{j + 1}: {m.group(1)} is not behind the barrier"
                    k = j + 1
                    while not lines[k].startswith("}"):
                        k += 1
                    assert lines[k].startswith("} RL_ABI_CATCH"), f"{f}:{k + 1}: {m.group(1)} does not end in RL_ABI_CATCH"
                    total += 1
                    i = k
                else:
                    assert sig_end.endswith("}") and len(lines[i]) < 140, f"{f}:{i + 1}: unexpected shape"
            i += 1
    assert total >= 95
