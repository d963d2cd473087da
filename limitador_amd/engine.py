"""Python face of the C ABI in include/rl_engine.h (ctypes; numpy arrays as host buffers).

Every method maps 1:1 onto an ``rl_*`` entry point; the reference method each one stands for is
cited in the header.  Nothing here computes a verdict: all decisions come from the HIP kernels.
"""
import ctypes as C

import numpy as np

from . import _lib
from .wire import CELL_ROW_DTYPE, HIT_DTYPE, LIMIT_ROW_DTYPE, MATCH_COND_DTYPE, MATCH_LIMIT_DTYPE

RL_OK = 0
ERR_NAMES = {
    -1: "RL_ERR_INVALID", -2: "RL_ERR_DEVICE", -3: "RL_ERR_NO_DEVICE", -4: "RL_ERR_TABLE_FULL",
    -5: "RL_ERR_MISSING_SIMPLE", -6: "RL_ERR_KEY_LIMIT", -7: "RL_ERR_BATCH_TOO_LARGE", -8: "RL_ERR_NOMEM",
    -9: "RL_ERR_BUSY", -10: "RL_ERR_KEY_COLLISION", -11: "RL_ERR_INTERNAL",
}


class EngineError(RuntimeError):
    """Mirror of StorageErr (limitador/src/storage/mod.rs:312-339): msg + transient flag."""

    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code
        self.msg = msg
        self.transient = bool(_lib.load().rl_status_is_transient(code))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Engine:
    """One counter table on one MI355X (InMemoryStorage::new, in_memory.rs:205-212)."""

    def __init__(self, capacity_cells, max_batch_hits=1 << 20, max_limits=1024, device=0,
                 hash_seed=0x9E3779B97F4A7C15, auto_grow=False):
        self._lib = _lib.load()
        cfg = _lib.RlConfig(device=device, max_batch_hits=max_batch_hits, capacity_cells=capacity_cells,
                            max_limits=max_limits, flags=1 if auto_grow else 0, hash_seed=hash_seed)
        h = C.c_void_p()
        rc = self._lib.rl_engine_create(C.byref(cfg), C.byref(h))
        if rc != RL_OK:
            raise EngineError(rc, "rl_engine_create failed (no MI355X visible?)" if rc == -3 else "rl_engine_create failed")
        self._h = h
        self._dependents = []  # communicators built on this engine (sharded_abi.Sharded): closed BEFORE it, whatever order the
        #                        garbage collector picks — rl_sharded_destroy hands the engine its own streams back
        self.max_batch_hits = max_batch_hits
        self.hash_seed = hash_seed
        self.device = device

    # -- plumbing ----------------------------------------------------------------------------
    def close(self):
        for dep in list(getattr(self, "_dependents", ())):
            dep.close()
        if getattr(self, "_h", None):
            self._lib.rl_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != RL_OK:
            raise EngineError(rc, self._lib.rl_last_error(self._h).decode())

    @property
    def handle(self):
        return self._h

    @property
    def stream(self):
        """The engine's hipStream_t as an integer (for torch.cuda.ExternalStream)."""
        return self._lib.rl_engine_stream(self._h) or 0

    def set_stream(self, stream):
        """Launch on the caller's hipStream_t (an int, e.g. torch.cuda.current_stream().cuda_stream — 0 is
        the default stream); None returns to the engine's own stream."""
        if stream is None:
            self._check(self._lib.rl_engine_set_stream(self._h, None, 0))
        else:
            self._check(self._lib.rl_engine_set_stream(self._h, C.c_void_p(int(stream)), 1))

    def stats(self):
        s = _lib.RlStats()
        self._check(self._lib.rl_stats(self._h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in s._fields_}

    # -- limits ------------------------------------------------------------------------------
    def set_limits(self, rows, first=0):
        """rows: iterable of (max_value, seconds) or a LIMIT_ROW_DTYPE array."""
        arr = np.asarray(rows, dtype=LIMIT_ROW_DTYPE) if isinstance(rows, np.ndarray) else np.array(
            [tuple(r) for r in rows], dtype=LIMIT_ROW_DTYPE)
        self._check(self._lib.rl_limits_set(self._h, first, _ptr(arr), arr.shape[0]))

    def add_counter(self, limit, key):
        self._check(self._lib.rl_add_counter(self._h, int(limit), int(key)))

    # -- hot path ----------------------------------------------------------------------------
    def host_register(self, array):
        """Pin a numpy array the caller reuses for host-buffer calls (rl_host_register); undo with host_unregister
        before the array is freed."""
        self._check(self._lib.rl_host_register(self._h, array.ctypes.data, array.nbytes))

    def host_unregister(self, array):
        self._check(self._lib.rl_host_unregister(self._h, array.ctypes.data))

    def check_and_update(self, hits, now_us, req_off=None, load_counters=False, want_first_limited=True,
                         req_delta=None, req_now_us=None, verdict_out=None):
        """CounterStorage::check_and_update for a batch.  Returns (verdict u8[n_req],
        first_limited i32[n_req] | None, remaining u64[n_hits] | None, expires_in_us | None).
        req_delta: per-request u64 deltas (the trait's `delta: u64`); req_now_us: per-request clock values."""
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        n_hits = hits.shape[0]
        if req_off is not None:
            req_off = np.ascontiguousarray(req_off, dtype=np.uint32)
            n_req = req_off.shape[0] - 1
        else:
            n_req = n_hits
        if req_delta is not None:
            req_delta = np.ascontiguousarray(req_delta, dtype=np.uint64)
            assert req_delta.shape[0] == n_req
        if req_now_us is not None:
            req_now_us = np.ascontiguousarray(req_now_us, dtype=np.uint64)
            assert req_now_us.shape[0] == n_req
        if verdict_out is not None:  # (a caller-owned, possibly registered, result array)
            assert verdict_out.dtype == np.uint8 and verdict_out.flags.c_contiguous and verdict_out.shape[0] >= n_req
            verdict = verdict_out[:n_req]
        else:
            verdict = np.empty(n_req, dtype=np.uint8)
        first = np.empty(n_req, dtype=np.int32) if want_first_limited else None
        remaining = np.zeros(n_hits, dtype=np.uint64) if load_counters else None
        expires = np.zeros(n_hits, dtype=np.uint64) if load_counters else None
        self._check(self._lib.rl_check_and_update_batch_ex(
            self._h, _ptr(hits), n_hits, _ptr(req_off), n_req, _ptr(req_delta), _ptr(req_now_us), int(now_us),
            int(bool(load_counters)), _ptr(verdict), _ptr(first), _ptr(remaining), _ptr(expires)))
        return verdict, first, remaining, expires

    def check_and_update_device(self, d_hits, n_hits, now_us, d_verdict, d_req_off=None, n_req=None,
                                load_counters=False, d_first_limited=None, d_remaining=None, d_expires=None):
        """Same with raw device pointers (ints).  Blocks until the batch is applied."""
        self._check(self._lib.rl_check_and_update_batch_device(
            self._h, d_hits, n_hits, d_req_off, n_hits if n_req is None else n_req, int(now_us),
            int(bool(load_counters)), d_verdict, d_first_limited, d_remaining, d_expires))

    def submit_device(self, d_hits, n_hits, now_us, d_verdict, d_first_limited=None):
        """Enqueue a single-counter batch (raw device pointers) without waiting; at most three in flight."""
        self._check(self._lib.rl_check_and_update_submit_device(self._h, d_hits, n_hits, int(now_us), d_verdict,
                                                                d_first_limited))

    def collect(self):
        """Wait for the oldest submitted batch; raises its error, if any."""
        self._check(self._lib.rl_check_and_update_collect(self._h))

    def is_within_limits(self, hits, now_us):
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        out = np.empty(hits.shape[0], dtype=np.uint8)
        self._check(self._lib.rl_is_within_limits_batch(self._h, _ptr(hits), hits.shape[0], int(now_us), _ptr(out)))
        return out

    def update_counters(self, hits, now_us):
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        self._check(self._lib.rl_update_counter_batch(self._h, _ptr(hits), hits.shape[0], int(now_us)))

    # -- rest of the CounterStorage surface -----------------------------------------------------
    def get_counters(self, limit, now_us, cap=None):
        n = C.c_uint64(0)
        if cap is None:
            self._check(self._lib.rl_get_counters(self._h, int(limit), int(now_us), None, 0, C.byref(n)))
            cap = n.value
        out = np.empty(cap, dtype=CELL_ROW_DTYPE)
        self._check(self._lib.rl_get_counters(self._h, int(limit), int(now_us), _ptr(out), cap, C.byref(n)))
        return out[: min(cap, n.value)]

    def count_counters(self, limit, now_us):
        """Number of rows rl_get_counters would return (counted on the device, nothing copied)."""
        n = C.c_uint64(0)
        self._check(self._lib.rl_get_counters(self._h, int(limit), int(now_us), None, 0, C.byref(n)))
        return n.value

    def delete_counters(self, limit):
        self._check(self._lib.rl_delete_counters(self._h, int(limit)))

    def clear(self):
        self._check(self._lib.rl_clear(self._h))

    def sweep_expired(self, now_us):
        n = C.c_uint64(0)
        self._check(self._lib.rl_sweep_expired(self._h, int(now_us), C.byref(n)))
        return n.value

    def sweep_expired_submit(self, now_us):
        """The sweep as a stream-ordered command between the batches in flight (takes one of the three slots)."""
        self._check(self._lib.rl_sweep_expired_submit(self._h, int(now_us)))

    def sweep_expired_collect(self):
        n = C.c_uint64(0)
        self._check(self._lib.rl_sweep_expired_collect(self._h, C.byref(n)))
        return n.value

    def compact(self):
        self._check(self._lib.rl_compact(self._h))

    def resize(self, capacity_cells):
        """Rehash into a table of capacity_cells (rounded up to a power of two)."""
        self._check(self._lib.rl_resize(self._h, int(capacity_cells)))

    def load_cells(self, rows):
        rows = np.ascontiguousarray(rows, dtype=CELL_ROW_DTYPE)
        self._check(self._lib.rl_load_cells(self._h, _ptr(rows), rows.shape[0]))

    def load_cells_device(self, d_rows, n):
        self._check(self._lib.rl_load_cells_device(self._h, d_rows, n))

    def dump_cells(self):
        n = C.c_uint64(0)
        self._check(self._lib.rl_dump_cells(self._h, None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=CELL_ROW_DTYPE)
        self._check(self._lib.rl_dump_cells(self._h, _ptr(out), out.shape[0], C.byref(n)))
        return out[: min(out.shape[0], n.value)]

    # -- snapshot files, cross-node merge ----------------------------------------------------------
    def snapshot_save(self, path):
        self._check(self._lib.rl_snapshot_save(self._h, str(path).encode()))

    def snapshot_load(self, path):
        self._check(self._lib.rl_snapshot_load(self._h, str(path).encode()))

    def merge_cells(self, self_actor, actor, rows, now_us):
        """CrCounterValue::merge_at for what one remote actor reports (rows: CELL_ROW_DTYPE, value = its own part)."""
        rows = np.ascontiguousarray(rows, dtype=CELL_ROW_DTYPE)
        self._check(self._lib.rl_merge_cells(self._h, int(self_actor), int(actor), _ptr(rows), rows.shape[0], int(now_us)))

    def export_local(self, now_us):
        n = C.c_uint64(0)
        self._check(self._lib.rl_export_local(self._h, int(now_us), None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=CELL_ROW_DTYPE)
        self._check(self._lib.rl_export_local(self._h, int(now_us), _ptr(out), out.shape[0], C.byref(n)))
        return out[: min(out.shape[0], n.value)]

    # -- upstream of the trait: limit matching + key derivation on the device ---------------------
    def set_match_table(self, limits, conds, n_namespaces):
        """limits: MATCH_LIMIT_DTYPE array sorted by ns; conds: MATCH_COND_DTYPE array."""
        limits = np.ascontiguousarray(limits, dtype=MATCH_LIMIT_DTYPE)
        conds = np.ascontiguousarray(conds, dtype=MATCH_COND_DTYPE)
        self._check(self._lib.rl_match_table_set(self._h, _ptr(limits), limits.shape[0], _ptr(conds), conds.shape[0],
                                                 int(n_namespaces)))

    def match_key(self, limit_id, values=()):
        v = list(values) + [0, 0]
        return int(self._lib.rl_match_key(int(limit_id), len(values), int(v[0]), int(v[1])))

    def match_and_check(self, req_ns, ent_off, ent_key, ent_val, req_delta, now_us, load_counters=False, hits_cap=None):
        """counters_that_apply + check_and_update for a batch of dictionary-encoded requests.
        -> dict(verdict, limited_limit, req_off, hits, remaining, expires_in_us)"""
        req_ns = np.ascontiguousarray(req_ns, dtype=np.uint32)
        ent_off = np.ascontiguousarray(ent_off, dtype=np.uint32)
        ent_key = np.ascontiguousarray(ent_key, dtype=np.uint32)
        ent_val = np.ascontiguousarray(ent_val, dtype=np.uint32)
        req_delta = np.ascontiguousarray(req_delta, dtype=np.uint32)
        n_req = req_ns.shape[0]
        cap = self.max_batch_hits if hits_cap is None else hits_cap
        verdict = np.empty(n_req, dtype=np.uint8)
        limited = np.empty(n_req, dtype=np.int32)
        req_off = np.empty(n_req + 1, dtype=np.uint32)
        hits = np.empty(cap, dtype=HIT_DTYPE)
        rem = np.zeros(cap, dtype=np.uint64)
        exp = np.zeros(cap, dtype=np.uint64)
        n_hits = C.c_uint32(0)
        self._check(self._lib.rl_match_and_check_batch(
            self._h, _ptr(req_ns), _ptr(ent_off), _ptr(ent_key), _ptr(ent_val), _ptr(req_delta), n_req, int(now_us),
            int(bool(load_counters)), _ptr(verdict), _ptr(limited), _ptr(req_off), _ptr(hits), cap, C.byref(n_hits),
            _ptr(rem), _ptr(exp)))
        n = n_hits.value
        return {"verdict": verdict, "limited_limit": limited, "req_off": req_off, "hits": hits[:n],
                "remaining": rem[:n] if load_counters else None, "expires_in_us": exp[:n] if load_counters else None}

    def match_op(self, op, req_ns, ent_off, ent_key, ent_val, req_delta, now_us):
        """rl_match_batch_op: op 0 check_rate_limited_and_update, 1 is_rate_limited (read-only), 2 update_counters — the
        RateLimiter methods behind the matcher (lib.rs:362-464).  -> (verdict, limited_limit)"""
        req_ns = np.ascontiguousarray(req_ns, dtype=np.uint32)
        ent_off = np.ascontiguousarray(ent_off, dtype=np.uint32)
        ent_key = np.ascontiguousarray(ent_key, dtype=np.uint32)
        ent_val = np.ascontiguousarray(ent_val, dtype=np.uint32)
        req_delta = np.ascontiguousarray(req_delta, dtype=np.uint32)
        n_req = req_ns.shape[0]
        verdict = np.empty(n_req, dtype=np.uint8)
        limited = np.empty(n_req, dtype=np.int32)
        self._check(self._lib.rl_match_batch_op(self._h, int(op), _ptr(req_ns), _ptr(ent_off), _ptr(ent_key), _ptr(ent_val),
                                                _ptr(req_delta), n_req, int(now_us), _ptr(verdict), _ptr(limited)))
        return verdict, limited

    def match_and_check_device(self, d_req_ns, d_ent_off, d_ent_key, d_ent_val, d_req_delta, n_req, now_us,
                               d_verdict, d_limited_limit, load_counters=False):
        """rl_match_and_check_batch_device: request arrays, verdict and limited_limit are device pointers.
        -> number of counters derived."""
        n_hits = C.c_uint32(0)
        self._check(self._lib.rl_match_and_check_batch_device(
            self._h, d_req_ns, d_ent_off, d_ent_key, d_ent_val, d_req_delta, int(n_req), int(now_us),
            int(bool(load_counters)), d_verdict, d_limited_limit, C.byref(n_hits)))
        return n_hits.value

    # -- the general resolver in phases (admission decided by the caller; raw device pointers) ------
    def gen_begin(self, d_hits, d_req_id, n_hits, now_us, load_counters=False):
        self._check(self._lib.rl_gen_begin_device(self._h, d_hits, d_req_id, int(n_hits), int(now_us), int(bool(load_counters))))

    def gen_round(self, d_admitted, d_pass, d_remaining=None, d_expires=None):
        self._check(self._lib.rl_gen_round_device(self._h, d_admitted, d_pass, d_remaining, d_expires))

    def gen_count(self, d_reached):
        """-> (cells the pass would create, cells the table still takes)"""
        n_new, room = C.c_uint32(0), C.c_uint64(0)
        self._check(self._lib.rl_gen_count_device(self._h, d_reached, C.byref(n_new), C.byref(room)))
        return n_new.value, room.value

    def gen_commit(self):
        self._check(self._lib.rl_gen_commit_device(self._h))

    def gen_abort(self):
        self._check(self._lib.rl_gen_abort(self._h))

    # -- routing helpers (multi-GPU) -------------------------------------------------------------
    def owner_of(self, key, world):
        return self._lib.rl_owner_of(int(key), self.hash_seed, int(world))

    def route_partition_device(self, d_hits, n_hits, world, d_out, d_perm, d_counts):
        self._check(self._lib.rl_route_partition_device(self._h, d_hits, n_hits, world, d_out, d_perm, d_counts))

    def unpermute_u8_device(self, d_src, d_perm, n, d_dst):
        self._check(self._lib.rl_unpermute_u8_device(self._h, d_src, d_perm, n, d_dst))

    # -- measurement -----------------------------------------------------------------------------
    def kernel_timing(self, mode=1):
        """0 off, 1 both kernels of the hot path (k_bkt_part, k_bkt_apply) on every batch, 2 k_bkt_apply only,
        3 both kernels of every fourth batch (rl_engine.h)."""
        self._check(self._lib.rl_kernel_timing(self._h, int(mode)))

    TIMING_SLOTS = ("part", "apply_gap", "part_slack", "apply", "reserved4", "reserved5", "reserved6",
                    "reserved7")  # RL_T_* of include/rl_engine.h

    def kernel_timing_read(self, reset=True):
        """{"ms": {slot: accumulated milliseconds}, "launches": timed batches} since the last reset."""
        ms = (C.c_double * len(self.TIMING_SLOTS))()
        n = C.c_uint64()
        self._check(self._lib.rl_kernel_timing_read(self._h, ms, C.byref(n), int(bool(reset))))
        return {"ms": dict(zip(self.TIMING_SLOTS, list(ms))), "launches": n.value}
