"""The C-ABI library loads and exports every symbol include/rl_engine.h declares (no GPU,
no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "rl_engine.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rl_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_counter_storage_surface():
    syms = _header_symbols()
    for must in ("rl_check_and_update_batch", "rl_is_within_limits_batch", "rl_update_counter_batch",
                 "rl_add_counter", "rl_get_counters", "rl_delete_counters", "rl_clear", "rl_sweep_expired"):
        assert must in syms


def test_library_exports_every_declared_symbol(engine_lib):
    from limitador_amd import _lib

    for name in _header_symbols():
        assert hasattr(engine_lib, name), f"{name} declared in rl_engine.h but not exported"
        assert name in _lib.SYMBOLS, f"{name} has no ctypes signature in limitador_amd/_lib.py"


def test_wire_structs_match_header_sizes():
    import ctypes as C

    from limitador_amd import _lib, wire

    assert wire.HIT_DTYPE.itemsize == 16
    assert wire.CELL_ROW_DTYPE.itemsize == 32
    assert wire.LIMIT_ROW_DTYPE.itemsize == 16
    assert C.sizeof(_lib.RlConfig) == 32
    assert C.sizeof(_lib.RlStats) == 72


def test_engine_refuses_to_run_without_a_gpu(engine_lib):
    """No CPU fallback: on a box without a HIP device the engine fails loudly."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from limitador_amd.engine import Engine, EngineError

    with pytest.raises(EngineError) as e:
        Engine(capacity_cells=1024)
    assert e.value.code == -3


def test_owner_of_is_a_pure_function(engine_lib):
    seen = {engine_lib.rl_owner_of(k * 0x9E3779B97F4A7C15 % (1 << 64), 7, 8) for k in range(1000)}
    assert seen == set(range(8))
    assert engine_lib.rl_owner_of(12345, 7, 1) == 0


def _storage_header_symbols():
    src = open(os.path.join(ROOT, "include", "rl_storage.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rls_[a-z0-9_]+)\s*\(", src)) - {"rls_emit_fn"})


def test_host_mirror_library_exports_every_declared_symbol(engine_lib):
    """include/rl_storage.h (C view of the C++ CounterStorage mirror) against librl_storage.so."""
    from limitador_amd import host_storage

    so = host_storage.load()
    syms = _storage_header_symbols()
    assert "rls_check_and_update" in syms and "rls_batcher_check_and_update" in syms
    for name in syms:
        assert hasattr(so, name), f"{name} declared in rl_storage.h but not exported"
        assert name in host_storage.SYMBOLS, f"{name} has no ctypes signature in limitador_amd/host_storage.py"


def test_ingest_library_exports_every_declared_symbol(engine_lib):
    """include/rl_ingest.h (host-side ingest of the device matcher) against librl_storage.so."""
    from limitador_amd import host_storage, ingest

    so = host_storage.load()
    src = open(os.path.join(ROOT, "include", "rl_ingest.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    syms = sorted(set(re.findall(r"\b(rli_[a-z0-9_]+)\s*\(", src)))
    assert "rli_add_limit" in syms and "rli_check" in syms and len(syms) >= 20
    ingest._lib()
    for name in syms:
        assert hasattr(so, name), f"{name} declared in rl_ingest.h but not exported"
        assert name in ingest.SYMBOLS, f"{name} has no ctypes signature in limitador_amd/ingest.py"


def test_sharded_library_exports_every_symbol_of_rl_sharded_h():
    """include/rl_sharded.h (the routed multi-GPU step): loads without a GPU, every entry has a ctypes signature."""
    from limitador_amd import sharded_abi

    so = sharded_abi.load()
    src = open(os.path.join(ROOT, "include", "rl_sharded.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(rl_(?:sharded|local_group)_[a-z0-9_]+)\s*\(", src)))
    assert len(names) >= 14
    for name in names:
        assert hasattr(so, name), f"{name} declared in rl_sharded.h but not exported"
        assert name in sharded_abi.SYMBOLS, f"{name} has no ctypes signature in limitador_amd/sharded_abi.py"
    # the in-process transport is plain host code: a group can be made and asked for a rank's transport
    g = sharded_abi.LocalGroup(2)
    t = g.transport(1)
    assert t.ctx and t.exchange
    with pytest.raises(sharded_abi.ShardedError):
        g.transport(2)
    g.close()


DOCUMENTED_SWITCHES = {"RL_FUSE", "RL_SERVE", "RL_TINY_MAX", "RL_STREAM", "RLI_THREADS"}


def _switch_names(path):
    blob = open(path, "rb").read()
    return {m.decode() for m in re.findall(rb"(?<![A-Za-z0-9_])(RLI?_[A-Z][A-Z0-9_]+)\x00", blob)}


def test_release_libraries_read_only_the_documented_switches():
    """shapes, budgets, priorities and diagnostics are compiled behind -DRL_EXPERIMENT; the libraries a
    host links against name at most the documented handful of environment variables (include/rl_engine.h,
    include/rl_ingest.h) — the experiment build (limitador_amd/lib/exp/, what this suite loads) names the rest."""
    from limitador_amd import build as b

    seen = set()
    for so in ("librl_engine.so", "librl_storage.so", "librl_sharded.so"):
        path = os.path.join(b.RELEASE_LIBDIR, so)
        assert os.path.exists(path), f"{path}: run __graft_entry__.build()"
        names = _switch_names(path)
        assert names <= DOCUMENTED_SWITCHES, f"{so} names undocumented switches: {sorted(names - DOCUMENTED_SWITCHES)}"
        seen |= names
    assert len(seen) <= 6
    exp = _switch_names(os.path.join(b.EXP_LIBDIR, "librl_engine.so"))
    assert "RL_PIPE_DEPTH" in exp and "RL_APPLY_TRACE" in exp, "the experiment build lost its switches"
    assert b.EXPERIMENT, "tests/conftest.py selects the experiment build for the suite"
