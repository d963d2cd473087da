"""ctypes face of the C++ host mirror of `CounterStorage` (include/rl_storage.h,
limitador_amd/csrc/host/).  Speaks the reference's model — Limit / Counter made of strings — and
leaves all interning, ordering and result mapping to the C++ side; used by the tests to replay the
reference's scenarios through the same code a Rust `GpuStorage` would replace."""
import ctypes as C
import os
import threading

from . import _lib
from .build import STORAGE_SO
from .engine import ERR_NAMES


class RlsLimit(C.Structure):
    _fields_ = [("namespace_", C.c_char_p), ("max_value", C.c_uint64), ("seconds", C.c_uint64),
                ("conditions", C.POINTER(C.c_char_p)), ("n_conditions", C.c_uint32),
                ("variables", C.POINTER(C.c_char_p)), ("n_variables", C.c_uint32), ("name", C.c_char_p)]


class RlsCounter(C.Structure):
    _fields_ = [("limit", RlsLimit), ("var_names", C.POINTER(C.c_char_p)), ("var_values", C.POINTER(C.c_char_p)),
                ("n_vars", C.c_uint32), ("has_remaining", C.c_uint32), ("remaining", C.c_uint64),
                ("expires_in_us", C.c_uint64), ("has_expires_in", C.c_uint32), ("reserved", C.c_uint32)]


EMIT_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_uint32,
                      C.c_uint64, C.c_uint64)

SYMBOLS = {}
_so = None


def load():
    global _so
    if _so is not None:
        return _so
    _lib.load()  # librl_engine.so first (and torch's HIP runtime before it)
    if not os.path.exists(STORAGE_SO):
        raise _lib.EngineLibraryMissing(f"{STORAGE_SO} is not built; run __graft_entry__.build()")
    so = C.CDLL(STORAGE_SO, mode=C.RTLD_GLOBAL)
    p = C.c_void_p

    def sig(name, restype, argtypes):
        fn = getattr(so, name)
        fn.restype, fn.argtypes = restype, argtypes
        SYMBOLS[name] = fn

    sig("rls_storage_create", C.c_int32, [C.c_uint64, C.c_uint32, C.c_int32, C.POINTER(p)])
    sig("rls_storage_destroy", None, [p])
    sig("rls_last_error", C.c_char_p, [p])
    sig("rls_set_clock", None, [p, C.c_uint64])
    sig("rls_is_within_limits", C.c_int32, [p, C.POINTER(RlsCounter), C.c_uint64, C.POINTER(C.c_int32)])
    sig("rls_add_counter", C.c_int32, [p, C.POINTER(RlsLimit)])
    sig("rls_update_counter", C.c_int32, [p, C.POINTER(RlsCounter), C.c_uint64])
    sig("rls_check_and_update", C.c_int32, [p, C.POINTER(RlsCounter), C.c_uint32, C.c_uint64, C.c_int32,
                                            C.POINTER(C.c_int32), C.POINTER(C.c_int32)])
    sig("rls_get_counters", C.c_int32, [p, C.POINTER(RlsLimit), C.c_uint32, EMIT_FN, p])
    sig("rls_delete_counters", C.c_int32, [p, C.POINTER(RlsLimit), C.c_uint32])
    sig("rls_clear", C.c_int32, [p])
    sig("rls_sweep_expired", C.c_int32, [p, C.POINTER(C.c_uint64)])
    sig("rls_abi_selftest", C.c_int32, [C.c_int32])
    sig("rls_set_sweep_after", None, [p, C.c_uint64])
    sig("rls_interned_counters", C.c_uint64, [p])
    sig("rls_check_and_update_repeat", C.c_int32, [p, C.POINTER(RlsCounter), C.c_uint32, C.c_uint64, C.c_uint32,
                                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)])
    sig("rls_batcher_create", C.c_int32, [p, C.c_uint32, C.c_uint32, C.POINTER(p)])
    sig("rls_batcher_destroy", None, [p])
    sig("rls_batcher_check_and_update", C.c_int32, [p, C.POINTER(RlsCounter), C.c_uint32, C.c_uint64, C.c_int32,
                                                    C.POINTER(C.c_int32), C.POINTER(C.c_int32)])
    sig("rls_batcher_stats", None, [p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)])
    _so = so
    return so


class StorageError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


def _strs(items):
    arr = (C.c_char_p * max(1, len(items)))()
    for i, s in enumerate(items):
        arr[i] = s.encode()
    return arr


def c_limit(namespace, max_value, seconds, conditions=(), variables=(), name=None):
    """-> (RlsLimit, keepalive)"""
    conds, vars_ = _strs(list(conditions)), _strs(list(variables))
    lim = RlsLimit(namespace.encode(), max_value, seconds, conds, len(conditions), vars_, len(variables),
                   name.encode() if name is not None else None)
    return lim, (conds, vars_)


def c_counter(limit_args, set_variables):
    """set_variables: iterable of (name, value).  -> (RlsCounter, keepalive)"""
    lim, keep = c_limit(*limit_args)
    names, values = _strs([k for k, _ in set_variables]), _strs([v for _, v in set_variables])
    c = RlsCounter(lim, names, values, len(set_variables), 0, 0, 0, 0, 0)
    return c, (keep, names, values)


class HostStorage:
    """GpuCounterStorage (C++), method for method the reference's `trait CounterStorage`."""

    def __init__(self, capacity_cells=1 << 16, max_batch_hits=1 << 16, device=0):
        self._so = load()
        h = C.c_void_p()
        rc = self._so.rls_storage_create(capacity_cells, max_batch_hits, device, C.byref(h))
        if rc:
            raise StorageError(rc, "rls_storage_create failed (no MI355X visible?)" if rc == -3 else "create failed")
        self._h = h
        self._batcher = None
        self._batcher_lock = threading.Lock()

    def close(self):
        if self._batcher:
            self._so.rls_batcher_destroy(self._batcher)
            self._batcher = None
        if getattr(self, "_h", None):
            self._so.rls_storage_destroy(self._h)
            self._h = None

    def _check(self, rc):
        if rc:
            raise StorageError(rc, self._so.rls_last_error(self._h).decode())

    def set_clock(self, now_us):
        self._so.rls_set_clock(self._h, int(now_us))

    def is_within_limits(self, limit_args, set_variables, delta):
        c, _keep = c_counter(limit_args, set_variables)
        w = C.c_int32()
        self._check(self._so.rls_is_within_limits(self._h, C.byref(c), delta, C.byref(w)))
        return bool(w.value)

    def add_counter(self, limit_args):
        lim, _keep = c_limit(*limit_args)
        self._check(self._so.rls_add_counter(self._h, C.byref(lim)))

    def update_counter(self, limit_args, set_variables, delta):
        c, _keep = c_counter(limit_args, set_variables)
        self._check(self._so.rls_update_counter(self._h, C.byref(c), delta))

    def check_and_update(self, counters, delta, load_counters, batched=False):
        """counters: list of (limit_args, set_variables).  -> (limited, limited_idx, [(remaining, expires_in_us)])"""
        arr = (RlsCounter * max(1, len(counters)))()
        keep = []
        for i, (la, sv) in enumerate(counters):
            arr[i], k = c_counter(la, sv)
            keep.append(k)
        limited, idx = C.c_int32(), C.c_int32(-1)
        if batched:
            with self._batcher_lock:
                if self._batcher is None:
                    b = C.c_void_p()
                    self._check(self._so.rls_batcher_create(self._h, 64, 200, C.byref(b)))
                    self._batcher = b
            fn, h = self._so.rls_batcher_check_and_update, self._batcher
        else:
            fn, h = self._so.rls_check_and_update, self._h
        self._check(fn(h, arr, len(counters), delta, int(load_counters), C.byref(limited), C.byref(idx)))
        loaded = [(arr[i].remaining if arr[i].has_remaining else None,
                   arr[i].expires_in_us if arr[i].has_expires_in else None) for i in range(len(counters))]
        return bool(limited.value), idx.value, loaded

    def check_and_update_repeat(self, counters, delta, iterations):
        """`iterations` sequential check_and_update calls timed in native code -> (seconds, calls limited)."""
        arr = (RlsCounter * max(1, len(counters)))()
        keep = []
        for i, (la, sv) in enumerate(counters):
            arr[i], k = c_counter(la, sv)
            keep.append(k)
        ns, lim = C.c_uint64(), C.c_uint32()
        self._check(self._so.rls_check_and_update_repeat(self._h, arr, len(counters), delta, iterations, C.byref(ns),
                                                         C.byref(lim)))
        return ns.value * 1e-9, lim.value

    def get_counters(self, limits):
        """limits: list of limit_args -> [(limit index, {name: value}, remaining, expires_in_us)]"""
        arr = (RlsLimit * max(1, len(limits)))()
        keep = []
        for i, la in enumerate(limits):
            arr[i], k = c_limit(*la)
            keep.append(k)
        out = []

        def emit(_user, li, names, values, n, remaining, expires):
            out.append((li, {names[i].decode(): values[i].decode() for i in range(n)}, remaining, expires))

        cb = EMIT_FN(emit)
        self._check(self._so.rls_get_counters(self._h, arr, len(limits), cb, None))
        return out

    def delete_counters(self, limits):
        arr = (RlsLimit * max(1, len(limits)))()
        keep = []
        for i, la in enumerate(limits):
            arr[i], k = c_limit(*la)
            keep.append(k)
        self._check(self._so.rls_delete_counters(self._h, arr, len(limits)))

    def clear(self):
        self._check(self._so.rls_clear(self._h))

    def sweep_expired(self):
        n = C.c_uint64()
        self._check(self._so.rls_sweep_expired(self._h, C.byref(n)))
        return n.value

    def set_sweep_after(self, n_new_counters):
        self._so.rls_set_sweep_after(self._h, int(n_new_counters))

    def interned_counters(self):
        return self._so.rls_interned_counters(self._h)

    def batcher_stats(self):
        b, r = C.c_uint64(), C.c_uint64()
        self._so.rls_batcher_stats(self._batcher, C.byref(b), C.byref(r))
        return b.value, r.value
