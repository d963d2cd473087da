"""ctypes loader of lib/librl_engine.so.  Fails loudly: there is no fallback implementation."""
import ctypes as C
import os

from .build import ENGINE_SO


class EngineLibraryMissing(ImportError):
    pass


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ENGINE_SO):
        raise EngineLibraryMissing(
            f"{ENGINE_SO} is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). limitador_amd has no CPU fallback."
        )
    # torch ships its own libamdhip64 (same SONAME). If torch is (or will be) in this process it
    # must be loaded first so that both share ONE HIP runtime.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing, the engine itself does not need it
        pass
    lib = C.CDLL(ENGINE_SO, mode=C.RTLD_GLOBAL)
    _declare(lib)
    _lib = lib
    return lib


# Every symbol include/rl_engine.h declares; tests/test_abi.py checks the list against the header.
SYMBOLS = {}


def _sig(lib, name, restype, argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
    SYMBOLS[name] = fn


class RlConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_batch_hits", C.c_uint32), ("capacity_cells", C.c_uint64),
                ("max_limits", C.c_uint32), ("flags", C.c_uint32), ("hash_seed", C.c_uint64)]


class RlStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("capacity_cells", "live_cells", "tombstones", "batches", "hits",
                                           "ordered_hits", "ordered_batches", "probe_steps", "rebuilds")]


def _declare(lib):
    p = C.c_void_p
    u8p, u32p, i32p, u64p = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_int32),
                             C.POINTER(C.c_uint64))
    _sig(lib, "rl_engine_create", C.c_int32, [C.POINTER(RlConfig), C.POINTER(p)])
    _sig(lib, "rl_engine_destroy", None, [p])
    _sig(lib, "rl_last_error", C.c_char_p, [p])
    _sig(lib, "rl_status_is_transient", C.c_int32, [C.c_int32])
    _sig(lib, "rl_stats", C.c_int32, [p, C.POINTER(RlStats)])
    _sig(lib, "rl_limits_set", C.c_int32, [p, C.c_uint32, p, C.c_uint32])
    _sig(lib, "rl_add_counter", C.c_int32, [p, C.c_uint32, C.c_uint64])
    _sig(lib, "rl_check_and_update_batch", C.c_int32,
         [p, p, C.c_uint32, p, C.c_uint32, C.c_uint64, C.c_int32, p, p, p, p])
    _sig(lib, "rl_check_and_update_batch_ex", C.c_int32,
         [p, p, C.c_uint32, p, C.c_uint32, p, p, C.c_uint64, C.c_int32, p, p, p, p])
    _sig(lib, "rl_check_and_update_batch_device", C.c_int32,
         [p, p, C.c_uint32, p, C.c_uint32, C.c_uint64, C.c_int32, p, p, p, p])
    _sig(lib, "rl_check_and_update_submit_device", C.c_int32, [p, p, C.c_uint32, C.c_uint64, p, p])
    _sig(lib, "rl_check_and_update_collect", C.c_int32, [p])
    _sig(lib, "rl_check_and_update_submit_device_ev", C.c_int32, [p, p, C.c_uint32, C.c_uint64, p, p, p])
    _sig(lib, "rl_engine_flush", C.c_int32, [p])
    _sig(lib, "rl_is_within_limits_batch", C.c_int32, [p, p, C.c_uint32, C.c_uint64, p])
    _sig(lib, "rl_update_counter_batch", C.c_int32, [p, p, C.c_uint32, C.c_uint64])
    _sig(lib, "rl_is_within_limits_batch_ex", C.c_int32, [p, p, C.c_uint32, p, C.c_uint64, p])
    _sig(lib, "rl_update_counter_batch_ex", C.c_int32, [p, p, C.c_uint32, p, C.c_uint64])
    _sig(lib, "rl_get_counters", C.c_int32, [p, C.c_uint32, C.c_uint64, p, C.c_uint64, u64p])
    _sig(lib, "rl_delete_counters", C.c_int32, [p, C.c_uint32])
    _sig(lib, "rl_clear", C.c_int32, [p])
    _sig(lib, "rl_sweep_expired", C.c_int32, [p, C.c_uint64, u64p])
    _sig(lib, "rl_sweep_expired_rows", C.c_int32, [p, C.c_uint64, p, C.c_uint64, u64p])
    _sig(lib, "rl_sweep_expired_submit", C.c_int32, [p, C.c_uint64])
    _sig(lib, "rl_sweep_expired_collect", C.c_int32, [p, u64p])
    _sig(lib, "rl_compact", C.c_int32, [p])
    _sig(lib, "rl_resize", C.c_int32, [p, C.c_uint64])
    _sig(lib, "rl_load_cells", C.c_int32, [p, p, C.c_uint64])
    _sig(lib, "rl_load_cells_device", C.c_int32, [p, p, C.c_uint64])
    _sig(lib, "rl_dump_cells", C.c_int32, [p, p, C.c_uint64, u64p])
    _sig(lib, "rl_snapshot_save", C.c_int32, [p, C.c_char_p])
    _sig(lib, "rl_snapshot_load", C.c_int32, [p, C.c_char_p])
    _sig(lib, "rl_merge_cells", C.c_int32, [p, C.c_uint32, C.c_uint32, p, C.c_uint64, C.c_uint64])
    _sig(lib, "rl_export_local", C.c_int32, [p, C.c_uint64, p, C.c_uint64, u64p])
    _sig(lib, "rl_match_table_set", C.c_int32, [p, p, C.c_uint32, p, C.c_uint32, C.c_uint32])
    _sig(lib, "rl_match_table_set_ex", C.c_int32, [p, p, C.c_uint32, p, C.c_uint32, C.c_uint32, p, C.c_uint32])
    _sig(lib, "rl_match_key", C.c_uint64, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32])
    _sig(lib, "rl_match_and_check_batch", C.c_int32,
         [p, p, p, p, p, p, C.c_uint32, C.c_uint64, C.c_int32, p, p, p, p, C.c_uint32, u32p, p, p])
    _sig(lib, "rl_match_and_check_batch_device", C.c_int32,
         [p, p, p, p, p, p, C.c_uint32, C.c_uint64, C.c_int32, p, p, u32p])
    _sig(lib, "rl_wire_table_set", C.c_int32, [p, p, C.c_uint32, p, C.c_uint32, p, C.c_uint32, p, C.c_uint32, p, C.c_uint32, p])
    _sig(lib, "rl_host_staging", C.c_int32, [p, C.c_uint32, C.c_uint64, C.POINTER(p)])
    _sig(lib, "rl_wire_match_and_check_batch", C.c_int32,
         [p, p, p, C.c_uint32, C.c_uint64, C.c_int32, p, p, p, p, p, C.c_uint32, p, p, p, p])
    _sig(lib, "rl_match_batch_op", C.c_int32, [p, C.c_int32, p, p, p, p, p, C.c_uint32, C.c_uint64, p, p])
    _sig(lib, "rl_wire_match_batch_op", C.c_int32, [p, C.c_int32, p, p, C.c_uint32, C.c_uint64, p, p, p, p])
    _sig(lib, "rl_resp_table_set", C.c_int32, [p, p, C.c_uint32, p, C.c_uint32])
    _sig(lib, "rl_match_serve_batch", C.c_int32, [p, p, p, p, p, p, C.c_uint32, C.c_uint64, C.c_uint32, p, C.POINTER(p), C.POINTER(p)])
    _sig(lib, "rl_wire_serve_batch", C.c_int32, [p, p, p, C.c_uint32, C.c_uint64, C.c_uint32, p, p, C.POINTER(p), C.POINTER(p), p])
    _sig(lib, "rl_resp_table_ready", C.c_int32, [p])
    _sig(lib, "rl_gen_set_async", C.c_int32, [p, C.c_int32])
    _sig(lib, "rl_serve_wait", C.c_int32, [p, C.c_uint64])
    _sig(lib, "rl_wire_serve_batch_set", C.c_int32, [p, C.c_uint32, p, p, C.c_uint32, C.c_uint64, C.c_uint32, p, p, C.POINTER(p),
                                                   C.POINTER(p), p])
    _sig(lib, "rl_serve_wait_set", C.c_int32, [p, C.c_uint32, C.c_uint64])
    _sig(lib, "rl_owner_of", C.c_uint32, [C.c_uint64, C.c_uint64, C.c_uint32])
    _sig(lib, "rl_route_partition_device", C.c_int32, [p, p, C.c_uint32, C.c_uint32, p, p, p])
    _sig(lib, "rl_unpermute_u8_device", C.c_int32, [p, p, p, C.c_uint32, p])
    _sig(lib, "rl_engine_stream", p, [p])
    _sig(lib, "rl_engine_set_stream", C.c_int32, [p, p, C.c_int32])
    _sig(lib, "rl_engine_wait_event", C.c_int32, [p, p])
    _sig(lib, "rl_gen_begin_device", C.c_int32, [p, p, p, C.c_uint32, C.c_uint64, C.c_int32])
    _sig(lib, "rl_gen_round_device", C.c_int32, [p, p, p, p, p])
    _sig(lib, "rl_gen_count_device", C.c_int32, [p, p, u32p, u64p])
    _sig(lib, "rl_gen_count_async_device", C.c_int32, [p, p, p, p])
    _sig(lib, "rl_gen_commit_gated_device", C.c_int32, [p, p, p, C.c_uint32, C.c_uint32, u32p])
    _sig(lib, "rl_gen_commit_device", C.c_int32, [p])
    _sig(lib, "rl_gen_abort", C.c_int32, [p])
    _sig(lib, "rl_engine_record_event", C.c_int32, [p, p])
    _sig(lib, "rl_engine_info", C.c_int32, [p, i32p, u32p])
    _sig(lib, "rl_route_partition_stream", C.c_int32, [p, p, p, C.c_uint32, C.c_uint32, p, p, p])
    _sig(lib, "rl_unpermute_u8_stream", C.c_int32, [p, p, p, p, C.c_uint32, p])
    _sig(lib, "rl_req_ids_stream", C.c_int32, [p, p, p, C.c_uint32, C.c_uint32, C.c_uint32, p, p, p])
    _sig(lib, "rl_req_round_stream", C.c_int32, [p, p, p, p, p, p, C.c_uint32, C.c_uint32, C.c_int32, p, p, p, p, p, p])
    _sig(lib, "rl_req_reached_stream", C.c_int32, [p, p, p, p, p, C.c_uint32, p])
    _sig(lib, "rl_unpermute_u64_stream", C.c_int32, [p, p, p, p, C.c_uint32, p])
    _sig(lib, "rl_copy_segments_stream", C.c_int32, [p, p, p, C.c_uint32])
    _sig(lib, "rl_host_register", C.c_int32, [p, C.c_void_p, C.c_uint64])
    _sig(lib, "rl_host_unregister", C.c_int32, [p, C.c_void_p])
    _sig(lib, "rl_kernel_timing", C.c_int32, [p, C.c_int32])
    _sig(lib, "rl_kernel_timing_read", C.c_int32, [p, C.POINTER(C.c_double), u64p, C.c_int32])
    _sig(lib, "rl_last_internal_error", C.c_char_p, [])
    _sig(lib, "rl_abi_selftest", C.c_int32, [C.c_int32])
    _sig(lib, "rl_abi_caught", C.c_int32, [C.c_char_p, C.c_char_p, C.c_int32])
    del u8p, u32p, i32p
