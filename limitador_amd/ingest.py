"""ctypes view of the C++ host-side ingest (include/rl_ingest.h, limitador_amd/csrc/host/ingest.cpp):
Limits -> compiled match table, string requests -> dictionary-encoded request arrays, for the
on-device matcher (Engine.match_and_check / rl_match_and_check_batch)."""
import ctypes as C

import numpy as np

from . import host_storage
from .wire import LIMIT_ROW_DTYPE, MATCH_COND_DTYPE, MATCH_LIMIT_DTYPE

HOST_ONLY = -100
UNKNOWN_DOMAIN = -101
# rl_engine.h RL_OP_*: ShouldRateLimit; the Kuadrant service's CheckRateLimit / Report (envoy_rls/kuadrant_service.rs:27-184)
OP_CHECK_AND_UPDATE, OP_CHECK, OP_UPDATE = 0, 1, 2
SYMBOLS = {}


def _lib():
    so = host_storage.load()  # librl_storage.so carries the ingest as well
    if SYMBOLS:
        return so
    p, cp, u32, u64, i32, i64 = C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_int64
    strs = C.POINTER(C.c_char_p)
    sig = {
        "rli_create": (i32, [C.POINTER(p)]),
        "rli_destroy": (None, [p]),
        "rli_last_error": (cp, [p]),
        "rli_add_limit": (i32, [p, cp, u64, u64, strs, u32, strs, u32]),
        "rli_compile": (i32, [p]),
        "rli_n_limits": (u32, [p]),
        "rli_n_conds": (u32, [p]),
        "rli_n_namespaces": (u32, [p]),
        "rli_limit_rows": (p, [p]),
        "rli_match_limits": (p, [p]),
        "rli_match_conds": (p, [p]),
        "rli_install": (i32, [p, p]),
        "rli_batch_clear": (None, [p]),
        "rli_batch_add": (i32, [p, cp, strs, strs, u32, u32]),
        "rli_batch_add_descriptors": (i32, [p, cp, u32, C.POINTER(u32), strs, strs, u32]),
        "rli_batch_add_rls": (i32, [p, C.c_char_p, u32]),
        "rli_set_binding": (i32, [p, i32]),
        "rli_set_limit_name": (i32, [p, u32, cp]),
        "rli_serve_batch": (i32, [p, p, C.POINTER(C.c_char_p), C.POINTER(u32), u32, u64, i32, C.POINTER(C.c_uint8), u32,
                                  C.POINTER(u32), C.POINTER(i32)]),
        "rli_serve_batch_op": (i32, [p, p, i32, C.POINTER(C.c_char_p), C.POINTER(u32), u32, u64, C.POINTER(C.c_uint8), u32,
                                     C.POINTER(u32), C.POINTER(i32)]),
        "rli_frontend_check_rate_limit": (i32, [p, C.c_char_p, u32, C.POINTER(C.c_uint8), u32, C.POINTER(u32)]),
        "rli_frontend_report": (i32, [p, C.c_char_p, u32, C.POINTER(C.c_uint8), u32, C.POINTER(u32)]),
        "rli_frontend_create": (i32, [p, p, u32, u32, i32, C.POINTER(p)]),
        "rli_frontend_destroy": (None, [p]),
        "rli_frontend_set_clock": (None, [p, u64]),
        "rli_frontend_should_rate_limit": (i32, [p, C.c_char_p, u32, C.POINTER(C.c_uint8), u32, C.POINTER(u32)]),
        "rli_frontend_stats": (None, [p, C.POINTER(u64), C.POINTER(u64)]),
        "rli_frontend_windows": (u64, [p]),
        "rli_abi_selftest": (C.c_int32, [C.c_int32]),
        "rli_set_value_cap": (i32, [p, u32]),
        "rli_set_key_mode": (i32, [p, i32]),
        "rli_set_hash_key": (i32, [p, u64, u64]),
        "rli_hash_key": (i32, [p, C.POINTER(u64)]),
        "rli_counter_key": (i32, [p, u32, strs, C.POINTER(u32), u32, C.POINTER(u64), C.POINTER(u32)]),
        "rli_batch_n_requests": (u32, [p]),
        "rli_batch_n_entries": (u32, [p]),
        "rli_batch_req_ns": (p, [p]),
        "rli_batch_req_delta": (p, [p]),
        "rli_batch_ent_off": (p, [p]),
        "rli_batch_ent_key": (p, [p]),
        "rli_batch_ent_val": (p, [p]),
        "rli_check": (i32, [p, p, u64, C.POINTER(C.c_uint8), C.POINTER(i32)]),
        "rli_rls_response": (u32, [i32, C.POINTER(C.c_uint8)]),
        "rli_key_id": (i64, [p, cp]),
        "rli_value_id": (i64, [p, cp]),
        "rli_namespace_id": (i64, [p, cp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(so, name)
        fn.restype, fn.argtypes = res, args
        SYMBOLS[name] = fn
    return so


def _strs(items):
    arr = (C.c_char_p * max(1, len(items)))()
    for i, s in enumerate(items):
        arr[i] = s.encode()
    return arr


def _view(ptr, n, dtype):
    if not ptr or n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_uint8 * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).copy()


class IngestError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rl_ingest {code}: {msg}")
        self.code = code


class Ingest:
    def __init__(self, binding="descriptors", value_cap=None, keys="exact", hash_key=None):
        """binding: what the caller's Context binds — "descriptors" (the transports: only the list `descriptors`)
        or "root" (library callers, Context::from(HashMap): every key a root variable).
        keys: "exact" (host dictionaries, packed ids) or "hashed" (rli_set_key_mode RLI_KEYS_HASHED: the device decodes
        the messages and keys counters by a hash of their canonical key bytes; serve_batch only).
        hash_key: (k0, k1), the 128-bit secret of the hashed mode (include/rl_keyhash.h); default: drawn by rli_create."""
        self._so = _lib()
        h = C.c_void_p()
        rc = SYMBOLS["rli_create"](C.byref(h))
        if rc:
            raise IngestError(rc, "rli_create failed")
        self._h = h
        self._dependents = []  # frontends built on this ingest: closed before it (see Engine.close)
        self._check(SYMBOLS["rli_set_binding"](self._h, {"descriptors": 0, "root": 1}[binding]))
        if value_cap is not None:
            self._check(SYMBOLS["rli_set_value_cap"](self._h, int(value_cap)))
        self.keys = keys
        self._check(SYMBOLS["rli_set_key_mode"](self._h, {"exact": 0, "hashed": 1}[keys]))
        if hash_key is not None:
            self._check(SYMBOLS["rli_set_hash_key"](self._h, int(hash_key[0]), int(hash_key[1])))

    @property
    def hash_key(self):
        """(k0, k1): what another ingest in front of the same table must be given (rli_hash_key)."""
        out = (C.c_uint64 * 2)()
        self._check(SYMBOLS["rli_hash_key"](self._h, out))
        return int(out[0]), int(out[1])

    def close(self):
        for dep in list(getattr(self, "_dependents", ())):
            dep.close()
        if getattr(self, "_h", None):
            SYMBOLS["rli_destroy"](self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise IngestError(rc, SYMBOLS["rli_last_error"](self._h).decode())
        return rc

    def add_limit(self, namespace, max_value, seconds, conditions=(), variables=()):
        """-> limit id, or HOST_ONLY if the limit needs CEL the device matcher does not evaluate."""
        rc = SYMBOLS["rli_add_limit"](self._h, namespace.encode(), int(max_value), int(seconds), _strs(list(conditions)),
                                      len(conditions), _strs(list(variables)), len(variables))
        if rc == HOST_ONLY:
            return HOST_ONLY
        return self._check(rc)

    def counter_key(self, limit_id, values=()):
        """(key, check word) the hashed mode gives the counter of `limit_id` with these variable values (bytes or str,
        in variable-name order)."""
        vals = [v.encode() if isinstance(v, str) else bytes(v) for v in values]
        arr = (C.c_char_p * max(1, len(vals)))(*vals) if vals else (C.c_char_p * 1)()
        lens = (C.c_uint32 * max(1, len(vals)))(*[len(v) for v in vals])
        key, chk = C.c_uint64(0), C.c_uint32(0)
        self._check(SYMBOLS["rli_counter_key"](self._h, int(limit_id), arr, lens, len(vals), C.byref(key), C.byref(chk)))
        return key.value, chk.value

    def set_limit_name(self, limit_id, name):
        self._check(SYMBOLS["rli_set_limit_name"](self._h, int(limit_id), None if name is None else name.encode()))

    def serve_batch(self, engine, messages, now_us, with_headers=False, stride=1024):
        """ShouldRateLimit for a batch of serialized RateLimitRequests -> (status per request, response bytes per
        request): 0 OK, 1 OVER_LIMIT, UNKNOWN_DOMAIN, HOST_ONLY, or a negative error for a malformed message."""
        n = len(messages)
        msgs = (C.c_char_p * max(1, n))(*[bytes(m) for m in messages])
        lens = (C.c_uint32 * max(1, n))(*[len(m) for m in messages])
        out = (C.c_uint8 * (max(1, n) * stride))()
        out_len = (C.c_uint32 * max(1, n))()
        status = (C.c_int32 * max(1, n))()
        self._check(SYMBOLS["rli_serve_batch"](self._h, engine._h, msgs, lens, n, int(now_us), int(bool(with_headers)), out,
                                               stride, out_len, status))
        raw = bytes(out)
        return [status[i] for i in range(n)], [raw[i * stride:i * stride + out_len[i]] for i in range(n)]

    def serve_batch_op(self, engine, op, messages, now_us, stride=64):
        """rli_serve_batch_op: OP_CHECK = Kuadrant CheckRateLimit (is_rate_limited with delta 1, nothing written),
        OP_UPDATE = Report (update_counters with hits_addend), OP_CHECK_AND_UPDATE = serve_batch without headers.
        -> (status per request, response bytes per request)."""
        n = len(messages)
        msgs = (C.c_char_p * max(1, n))(*[bytes(m) for m in messages])
        lens = (C.c_uint32 * max(1, n))(*[len(m) for m in messages])
        out = (C.c_uint8 * (max(1, n) * stride))()
        out_len = (C.c_uint32 * max(1, n))()
        status = (C.c_int32 * max(1, n))()
        self._check(SYMBOLS["rli_serve_batch_op"](self._h, engine._h, int(op), msgs, lens, n, int(now_us), out, stride, out_len,
                                                  status))
        raw = bytes(out)
        return [status[i] for i in range(n)], [raw[i * stride:i * stride + out_len[i]] for i in range(n)]

    def prepare_batch(self, messages, stride=1024):
        """The ctypes arrays of serve_batch built once (benchmarks: what is timed afterwards is the C call alone)."""
        n = len(messages)
        return {"n": n, "stride": stride, "keep": [bytes(m) for m in messages],
                "msgs": (C.c_char_p * max(1, n))(*[bytes(m) for m in messages]),
                "lens": (C.c_uint32 * max(1, n))(*[len(m) for m in messages]),
                "out": (C.c_uint8 * (max(1, n) * stride))(), "out_len": (C.c_uint32 * max(1, n))(),
                "status": (C.c_int32 * max(1, n))()}

    def serve_prepared(self, engine, prep, now_us, with_headers=False):
        """rli_serve_batch on a prepared batch; the results stay in prep["status"], prep["out"], prep["out_len"]."""
        self._check(SYMBOLS["rli_serve_batch"](self._h, engine._h, prep["msgs"], prep["lens"], prep["n"], int(now_us),
                                               int(bool(with_headers)), prep["out"], prep["stride"], prep["out_len"], prep["status"]))

    def serve_prepared_op(self, engine, prep, op, now_us):
        """rli_serve_batch_op on a prepared batch (OP_CHECK / OP_UPDATE / OP_CHECK_AND_UPDATE; no headers)."""
        self._check(SYMBOLS["rli_serve_batch_op"](self._h, engine._h, int(op), prep["msgs"], prep["lens"], prep["n"], int(now_us),
                                                  prep["out"], prep["stride"], prep["out_len"], prep["status"]))

    def compile(self):
        self._check(SYMBOLS["rli_compile"](self._h))
        n, nc = SYMBOLS["rli_n_limits"](self._h), SYMBOLS["rli_n_conds"](self._h)
        return {"limit_rows": _view(SYMBOLS["rli_limit_rows"](self._h), n, LIMIT_ROW_DTYPE),
                "limits": _view(SYMBOLS["rli_match_limits"](self._h), n, MATCH_LIMIT_DTYPE),
                "conds": _view(SYMBOLS["rli_match_conds"](self._h), nc, MATCH_COND_DTYPE),
                "n_namespaces": SYMBOLS["rli_n_namespaces"](self._h)}

    def install(self, engine):
        self._check(SYMBOLS["rli_install"](self._h, engine._h))

    def batch_clear(self):
        SYMBOLS["rli_batch_clear"](self._h)

    def batch_add(self, namespace, entries, delta=1):
        """entries: the (key, value) pairs of descriptors[0], in order."""
        keys, vals = [k for k, _ in entries], [v for _, v in entries]
        rc = SYMBOLS["rli_batch_add"](self._h, namespace.encode(), _strs(keys), _strs(vals), len(keys), int(delta))
        return HOST_ONLY if rc == HOST_ONLY else self._check(rc)  # HOST_ONLY: the value dictionary is at its cap

    def batch_add_descriptors(self, namespace, descriptors, delta=1):
        """descriptors: the request's whole descriptor list, each a list of (key, value) pairs
        (envoy_rls/server.rs:121-128: one map per descriptor)."""
        keys = [k for d in descriptors for k, _ in d]
        vals = [v for d in descriptors for _, v in d]
        off = (C.c_uint32 * (len(descriptors) + 1))()
        for i, d in enumerate(descriptors):
            off[i + 1] = off[i] + len(d)
        rc = SYMBOLS["rli_batch_add_descriptors"](self._h, namespace.encode(), len(descriptors), off, _strs(keys), _strs(vals),
                                                  int(delta))
        return HOST_ONLY if rc == HOST_ONLY else self._check(rc)

    def batch_add_rls(self, message):
        """message: one serialized envoy.service.ratelimit.v3.RateLimitRequest.
        -> request index, or UNKNOWN_DOMAIN (the reference answers Code::Unknown)."""
        rc = SYMBOLS["rli_batch_add_rls"](self._h, bytes(message), len(message))
        if rc in (UNKNOWN_DOMAIN, HOST_ONLY):
            return rc
        return self._check(rc)

    def batch(self):
        n, m = SYMBOLS["rli_batch_n_requests"](self._h), SYMBOLS["rli_batch_n_entries"](self._h)
        u32 = np.uint32
        return {"req_ns": _view(SYMBOLS["rli_batch_req_ns"](self._h), n, u32),
                "req_delta": _view(SYMBOLS["rli_batch_req_delta"](self._h), n, u32),
                "ent_off": _view(SYMBOLS["rli_batch_ent_off"](self._h), n + 1, u32),
                "ent_key": _view(SYMBOLS["rli_batch_ent_key"](self._h), m, u32),
                "ent_val": _view(SYMBOLS["rli_batch_ent_val"](self._h), m, u32)}

    def check(self, engine, now_us):
        n = SYMBOLS["rli_batch_n_requests"](self._h)
        verdict = np.zeros(n, dtype=np.uint8)
        limited = np.full(n, -1, dtype=np.int32)
        self._check(SYMBOLS["rli_check"](self._h, engine._h, int(now_us), verdict.ctypes.data_as(C.POINTER(C.c_uint8)),
                                         limited.ctypes.data_as(C.POINTER(C.c_int32))))
        return verdict, limited

    @staticmethod
    def rls_response(verdict):
        """Serialized RateLimitResponse for a verdict (0 OK, 1 OVER_LIMIT, UNKNOWN_DOMAIN -> UNKNOWN)."""
        _lib()
        out = (C.c_uint8 * 2)()
        n = SYMBOLS["rli_rls_response"](int(verdict), out)
        return bytes(out[:n])

    def key_id(self, s):
        return SYMBOLS["rli_key_id"](self._h, s.encode())

    def value_id(self, s):
        return SYMBOLS["rli_value_id"](self._h, s.encode())

    def namespace_id(self, s):
        return SYMBOLS["rli_namespace_id"](self._h, s.encode())



class Frontend:
    """The micro-batcher of the wire path (rli_frontend_*): thread-safe, blocking should_rate_limit."""

    def __init__(self, ingest, engine, max_batch=256, max_delay_us=200, with_headers=False):
        _lib()
        self._ingest, self._engine = ingest, engine  # keep alive
        h = C.c_void_p()
        rc = SYMBOLS["rli_frontend_create"](ingest._h, engine._h, max_batch, max_delay_us, int(bool(with_headers)), C.byref(h))
        if rc:
            raise IngestError(rc, "rli_frontend_create failed")
        self._h = h
        # (its workers call into both: whichever of the three the garbage collector finalizes first, the frontend goes first)
        engine._dependents.append(self)
        ingest._dependents.append(self)

    def set_clock(self, now_us):
        SYMBOLS["rli_frontend_set_clock"](self._h, int(now_us))

    def should_rate_limit(self, message):
        """-> (status, response bytes)"""
        out = (C.c_uint8 * 1024)()
        n = C.c_uint32()
        rc = SYMBOLS["rli_frontend_should_rate_limit"](self._h, bytes(message), len(message), out, 1024, C.byref(n))
        return rc, bytes(out[: n.value])

    def _call(self, name, message):
        out = (C.c_uint8 * 1024)()
        n = C.c_uint32()
        rc = SYMBOLS[name](self._h, bytes(message), len(message), out, 1024, C.byref(n))
        return rc, bytes(out[: n.value])

    def check_rate_limit(self, message):
        """Kuadrant CheckRateLimit -> (status, response bytes)"""
        return self._call("rli_frontend_check_rate_limit", message)

    def report(self, message):
        """Kuadrant Report -> (status, response bytes)"""
        return self._call("rli_frontend_report", message)

    def stats(self):
        b, r = C.c_uint64(), C.c_uint64()
        SYMBOLS["rli_frontend_stats"](self._h, C.byref(b), C.byref(r))
        return b.value, r.value

    def windows(self):
        """Windows served: each waited at most max_delay_us once, whatever mix of methods it held."""
        return SYMBOLS["rli_frontend_windows"](self._h)

    def close(self):
        if getattr(self, "_h", None):
            SYMBOLS["rli_frontend_destroy"](self._h)
            self._h = None
            for owner in (self._engine, self._ingest):
                if self in owner._dependents:
                    owner._dependents.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
