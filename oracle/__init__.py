"""ctypes binding of the CPU oracle (oracle/limitador_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg — never by anything under limitador_amd/.  See limitador_oracle.h for what
it restates and how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "liblimitador_oracle.so")

HIT_DTYPE = np.dtype([("key", "<u8"), ("limit", "<u4"), ("delta", "<u4")], align=True)
LIMIT_ROW_DTYPE = np.dtype([("max_value", "<u8"), ("seconds", "<u8")], align=True)
ROW_DTYPE = np.dtype([("key", "<u8"), ("limit", "<u4"), ("qualified", "<u4"), ("value", "<u8"),
                      ("expires_in_us", "<u8")], align=True)
SIMPLE_FLAG = 0x80000000

LO_OK, LO_LIMITED, LO_ERR_MISSING_SIMPLE = 0, 1, -2


class Cell(C.Structure):
    _fields_ = [("value", C.c_uint64), ("expiry_us", C.c_uint64)]


class Counter(C.Structure):
    _fields_ = [("key", C.c_uint64), ("limit", C.c_uint32), ("qualified", C.c_uint32),
                ("max_value", C.c_uint64), ("seconds", C.c_uint64), ("remaining", C.c_uint64),
                ("expires_in_us", C.c_uint64), ("has_remaining", C.c_uint32), ("has_expires_in", C.c_uint32)]


_lib = None


def build(force=False):
    src = [os.path.join(_DIR, "limitador_oracle.c"), os.path.join(_DIR, "limitador_oracle.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(s) <= os.path.getmtime(_SO) for s in src)):
        return _SO
    subprocess.run(["make", "-C", _DIR] + (["-B"] if force else []), check=True, stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        p, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
        L.lo_cell_value_at.restype = u64
        L.lo_cell_value_at.argtypes = [C.POINTER(Cell), u64]
        L.lo_cell_update.restype = u64
        L.lo_cell_update.argtypes = [C.POINTER(Cell), u64, u64, u64]
        L.lo_cell_ttl_us.restype = u64
        L.lo_cell_ttl_us.argtypes = [C.POINTER(Cell), u64]
        L.lo_storage_new.restype = p
        L.lo_storage_free.argtypes = [p]
        L.lo_is_within_limits.argtypes = [p, C.POINTER(Counter), u64, u64, C.POINTER(i32)]
        L.lo_add_counter.argtypes = [p, u32, i32]
        L.lo_update_counter.argtypes = [p, C.POINTER(Counter), u64, u64]
        L.lo_check_and_update.argtypes = [p, C.POINTER(Counter), C.c_size_t, u64, i32, u64, C.POINTER(C.c_int64)]
        L.lo_get_counters.restype = C.c_size_t
        L.lo_get_counters.argtypes = [p, u32, i32, u64, p, C.c_size_t]
        L.lo_delete_counters_of_limit.argtypes = [p, u32, i32]
        L.lo_clear.argtypes = [p]
        L.lo_evict.argtypes = [p, u64]
        L.lo_sweep_expired.restype = C.c_size_t
        L.lo_sweep_expired.argtypes = [p, u64]
        L.lo_num_qualified.restype = C.c_size_t
        L.lo_num_qualified.argtypes = [p]
        L.lo_peek_qualified.argtypes = [p, u64, C.POINTER(Cell), C.POINTER(u32)]
        L.lo_peek_simple.argtypes = [p, u32, C.POINTER(Cell)]
        L.lo_dump_qualified.restype = C.c_size_t
        L.lo_dump_qualified.argtypes = [p, p, p, p, p, C.c_size_t]
        L.lo_load_qualified.argtypes = [p, p, p, p, p, C.c_size_t]
        L.lo_check_and_update_batch.argtypes = [p, p, C.c_size_t, p, C.c_size_t, p, C.c_size_t, u64, i32, p, p, p, p]
        L.lo_check_and_update_batch_ex.argtypes = [p, p, C.c_size_t, p, C.c_size_t, p, C.c_size_t, p, p, u64, i32, p, p,
                                                   p, p]
        L.lo_is_within_limits_batch.argtypes = [p, p, C.c_size_t, p, C.c_size_t, u64, p]
        L.lo_cr_new.restype = p
        L.lo_cr_new.argtypes = [u32, u64, u64]
        L.lo_cr_from_values.restype = p
        L.lo_cr_from_values.argtypes = [u64, p, p, C.c_size_t]
        L.lo_cr_free.argtypes = [p]
        for name in ("lo_cr_expiry_us", "lo_cr_local_value"):
            getattr(L, name).restype = u64
            getattr(L, name).argtypes = [p]
        L.lo_cr_read_at.restype = u64
        L.lo_cr_read_at.argtypes = [p, u64]
        L.lo_cr_inc_at.argtypes = [p, u64, u64, u64]
        L.lo_cr_inc_actor_at.argtypes = [p, u32, u64, u64, u64]
        L.lo_cr_merge_at.argtypes = [p, p, u64]
        L.lo_bench_sharded.restype = C.c_double
        L.lo_bench_sharded.argtypes = [p, C.c_size_t, p, C.c_size_t, p, p, C.c_size_t, C.c_size_t, u64]
        L.lo_update_counter_batch.argtypes = [p, p, C.c_size_t, p, C.c_size_t, u64]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(f"oracle error {code}")
        self.code = code


class OracleStorage:
    """InMemoryStorage restated on the CPU, driven in the engine's wire format."""

    def __init__(self):
        self.L = lib()
        self.h = self.L.lo_storage_new()
        self.limits = np.zeros(0, dtype=LIMIT_ROW_DTYPE)

    def close(self):
        if self.h:
            self.L.lo_storage_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # limits table (request-side attributes)
    def set_limits(self, rows, first=0):
        rows = np.array([tuple(r) for r in rows], dtype=LIMIT_ROW_DTYPE) if not isinstance(rows, np.ndarray) else rows
        need = first + rows.shape[0]
        if self.limits.shape[0] < need:
            new = np.zeros(need, dtype=LIMIT_ROW_DTYPE)
            new[: self.limits.shape[0]] = self.limits
            self.limits = new
        self.limits[first:need] = rows

    def add_counter(self, limit, key=0):
        simple = bool(limit & SIMPLE_FLAG)
        rc = self.L.lo_add_counter(self.h, limit & ~SIMPLE_FLAG, 0 if simple else 1)
        if rc < 0:
            raise OracleError(rc)

    def check_and_update(self, hits, now_us, req_off=None, load_counters=False, want_first_limited=True,
                         req_delta=None, req_now_us=None):
        """req_delta: per-request u64 deltas (in_memory.rs:75) replacing the 32-bit wire field;
        req_now_us: per-request clock values replacing now_us."""
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
        verdict = np.empty(n_req, dtype=np.uint8)
        first = np.empty(n_req, dtype=np.int32)
        remaining = np.zeros(n_hits, dtype=np.uint64) if load_counters else None
        expires = np.zeros(n_hits, dtype=np.uint64) if load_counters else None
        rc = self.L.lo_check_and_update_batch_ex(self.h, _ptr(self.limits), self.limits.shape[0], _ptr(hits), n_hits,
                                                 _ptr(req_off), n_req, _ptr(req_delta), _ptr(req_now_us), int(now_us),
                                                 int(bool(load_counters)), _ptr(verdict), _ptr(first),
                                                 _ptr(remaining), _ptr(expires))
        if rc < 0:
            raise OracleError(rc)
        return verdict, (first if want_first_limited else None), remaining, expires

    def is_within_limits(self, hits, now_us):
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        out = np.empty(hits.shape[0], dtype=np.uint8)
        rc = self.L.lo_is_within_limits_batch(self.h, _ptr(self.limits), self.limits.shape[0], _ptr(hits),
                                              hits.shape[0], int(now_us), _ptr(out))
        if rc < 0:
            raise OracleError(rc)
        return out

    def update_counters(self, hits, now_us):
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        rc = self.L.lo_update_counter_batch(self.h, _ptr(self.limits), self.limits.shape[0], _ptr(hits),
                                            hits.shape[0], int(now_us))
        if rc < 0:
            raise OracleError(rc)

    def get_counters(self, limit, now_us):
        simple = bool(limit & SIMPLE_FLAG)
        n = self.L.lo_get_counters(self.h, limit & ~SIMPLE_FLAG, 0 if simple else 1, int(now_us), None, 0)
        out = np.empty(n, dtype=ROW_DTYPE)
        self.L.lo_get_counters(self.h, limit & ~SIMPLE_FLAG, 0 if simple else 1, int(now_us), _ptr(out), n)
        return out

    def delete_counters(self, limit):
        simple = bool(limit & SIMPLE_FLAG)
        self.L.lo_delete_counters_of_limit(self.h, limit & ~SIMPLE_FLAG, 0 if simple else 1)

    def clear(self):
        self.L.lo_clear(self.h)

    def sweep_expired(self, now_us):
        return self.L.lo_sweep_expired(self.h, int(now_us))

    def load_cells(self, keys, limits, values, expiries):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        limits = np.ascontiguousarray(limits, dtype=np.uint32)
        values = np.ascontiguousarray(values, dtype=np.uint64)
        expiries = np.ascontiguousarray(expiries, dtype=np.uint64)
        rc = self.L.lo_load_qualified(self.h, _ptr(keys), _ptr(limits), _ptr(values), _ptr(expiries), keys.shape[0])
        if rc < 0:
            raise OracleError(rc)

    def num_qualified(self):
        return self.L.lo_num_qualified(self.h)

    def peek(self, key):
        c = Cell()
        lim = C.c_uint32()
        if not self.L.lo_peek_qualified(self.h, int(key), C.byref(c), C.byref(lim)):
            return None
        return c.value, c.expiry_us, lim.value

    def dump_qualified(self):
        """(keys, limits, values, expiries) of every qualified cell, sorted by key."""
        n = self.num_qualified()
        k, li = np.empty(n, dtype=np.uint64), np.empty(n, dtype=np.uint32)
        v, e = np.empty(n, dtype=np.uint64), np.empty(n, dtype=np.uint64)
        got = self.L.lo_dump_qualified(self.h, _ptr(k), _ptr(li), _ptr(v), _ptr(e), n)
        assert got == n
        o = np.argsort(k, kind="stable")
        return k[o], li[o], v[o], e[o]

    def peek_simple(self, limit):
        c = Cell()
        if not self.L.lo_peek_simple(self.h, int(limit) & ~SIMPLE_FLAG, C.byref(c)):
            return None
        return c.value, c.expiry_us


def bench_sharded(shards, parts, reps, now0_us):
    """Replay single-counter batches over hash-sharded OracleStorages with one C thread per shard
    (lo_bench_sharded).  parts[d][t] = hits of distinct batch d owned by shard t.  -> wall seconds."""
    n_shards, n_distinct = len(shards), len(parts)
    handles = (C.c_void_p * n_shards)(*[s.h for s in shards])
    flat = [np.ascontiguousarray(parts[d][t], dtype=HIT_DTYPE) for d in range(n_distinct) for t in range(n_shards)]
    ptrs = (C.c_void_p * len(flat))(*[a.ctypes.data for a in flat])
    sizes = (C.c_size_t * len(flat))(*[a.shape[0] for a in flat])
    limits = shards[0].limits
    sec = lib().lo_bench_sharded(handles, n_shards, _ptr(limits), limits.shape[0], ptrs, sizes, n_distinct, reps,
                                 int(now0_us))
    if sec < 0:
        raise OracleError(int(sec))
    return sec


class CrCounterValue:
    """CrCounterValue<u32> (cr_counter_value.rs) with explicit clocks (microseconds)."""

    def __init__(self, ourselves, max_value, expiry_us, _h=None):
        self.L = lib()
        self.h = _h if _h is not None else self.L.lo_cr_new(int(ourselves), int(max_value), int(expiry_us))

    @classmethod
    def from_values(cls, expiry_us, values):
        """From<(SystemTime, BTreeMap<A, u64>)>: values = {actor: value}"""
        actors = np.array(list(values.keys()), dtype=np.uint32)
        vals = np.array(list(values.values()), dtype=np.uint64)
        return cls(0, 0, 0, _h=lib().lo_cr_from_values(int(expiry_us), _ptr(actors), _ptr(vals), len(actors)))

    def __del__(self):
        try:
            if self.h:
                self.L.lo_cr_free(self.h)
                self.h = None
        except Exception:
            pass

    def read_at(self, when_us):
        return self.L.lo_cr_read_at(self.h, int(when_us))

    def inc_at(self, increment, window_us, when_us):
        self.L.lo_cr_inc_at(self.h, int(increment), int(window_us), int(when_us))

    def inc_actor_at(self, actor, increment, window_us, when_us):
        self.L.lo_cr_inc_actor_at(self.h, int(actor), int(increment), int(window_us), int(when_us))

    def merge_at(self, other, when_us):
        self.L.lo_cr_merge_at(self.h, other.h, int(when_us))

    @property
    def expiry_us(self):
        return self.L.lo_cr_expiry_us(self.h)

    @property
    def local_value(self):
        return self.L.lo_cr_local_value(self.h)
