"""ctypes view of include/rl_sharded.h (limitador_amd/lib/librl_sharded.so): the routed multi-GPU step behind
the C ABI — what a non-Python host calls.  `limitador_amd/sharded.py` is the torch.distributed driver of the same
protocol; this one needs no torch at all (RCCL communicator owned by the library, or the in-process transport)."""
import ctypes as C
import os

from . import _lib
from .build import SHARDED_SO

UNIQUE_ID_BYTES = 128
MAX_IN_FLIGHT = 4  # RL_SHARDED_MAX_IN_FLIGHT: slices a caller may keep in flight (submit(k); collect(k - 3))
SYMBOLS = {}
_so = None


class RlTransport(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("exchange", C.c_void_p)]


def load():
    global _so
    if _so is not None:
        return _so
    _lib.load()  # librl_engine.so first (and torch's HIP runtime before it, see _lib.load)
    if not os.path.exists(SHARDED_SO):
        raise _lib.EngineLibraryMissing(f"{SHARDED_SO} is not built. Run `python -c 'import __graft_entry__ as g; g.build()'`.")
    so = C.CDLL(SHARDED_SO, mode=C.RTLD_GLOBAL)
    p, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
    u8p = C.POINTER(C.c_uint8)
    sig = {
        "rl_sharded_unique_id": (i32, [u8p]),
        "rl_sharded_create_rccl": (i32, [p, u32, u32, u8p, u32, C.POINTER(p)]),
        "rl_sharded_create": (i32, [p, u32, u32, C.POINTER(RlTransport), u32, C.POINTER(p)]),
        "rl_sharded_destroy": (None, [p]),
        "rl_sharded_last_error": (C.c_char_p, [p]),
        "rl_sharded_submit_device": (i32, [p, p, u32, u64, p]),
        "rl_sharded_collect": (i32, [p, C.POINTER(u32)]),
        "rl_sharded_check_and_update_device": (i32, [p, p, u32, u64, p, C.POINTER(u32)]),
        "rl_sharded_sweep_submit": (i32, [p, u64]),
        "rl_sharded_sweep_collect": (i32, [p, C.POINTER(u64)]),
        "rl_sharded_check_requests_device": (i32, [p, p, u32, p, u32, u64, i32, p, p, p, p, C.POINTER(u32)]),
        "rl_sharded_stream": (p, [p]),
        "rl_sharded_sync": (i32, [p]),
        "rl_sharded_in_flight": (u32, [p]),
        "rl_sharded_abi_selftest": (C.c_int32, [C.c_int32]),
        "rl_local_group_create": (i32, [u32, C.POINTER(p)]),
        "rl_local_group_destroy": (None, [p]),
        "rl_local_group_transport": (i32, [p, u32, C.POINTER(RlTransport)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(so, name)
        fn.restype, fn.argtypes = res, args
        SYMBOLS[name] = fn
    _so = so
    return so


class ShardedError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rl_sharded {code}: {msg}")
        self.code = code


def unique_id():
    """ncclGetUniqueId as bytes: made on one rank, handed to every rank by the host."""
    load()
    buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
    rc = SYMBOLS["rl_sharded_unique_id"](buf)
    if rc:
        raise ShardedError(rc, "rl_sharded_unique_id failed")
    return bytes(buf)


class LocalGroup:
    """The in-process transport: ranks are threads of this process (rl_local_group_*)."""

    def __init__(self, world):
        load()
        self.world = world
        h = C.c_void_p()
        rc = SYMBOLS["rl_local_group_create"](world, C.byref(h))
        if rc:
            raise ShardedError(rc, "rl_local_group_create failed")
        self._h = h

    def transport(self, rank):
        t = RlTransport()
        rc = SYMBOLS["rl_local_group_transport"](self._h, rank, C.byref(t))
        if rc:
            raise ShardedError(rc, "rl_local_group_transport failed")
        return t

    def close(self):
        if self._h:
            SYMBOLS["rl_local_group_destroy"](self._h)
            self._h = None


class Sharded:
    """One rank of the routed step.  unique_id: bytes from unique_id() -> RCCL; transport: an RlTransport."""

    def __init__(self, engine, world, rank, max_slice_hits, unique_id=None, transport=None):
        load()
        self._engine = engine  # keep alive
        h = C.c_void_p()
        if transport is not None:
            self._transport = transport
            rc = SYMBOLS["rl_sharded_create"](engine._h, world, rank, C.byref(transport), max_slice_hits, C.byref(h))
        else:
            if unique_id is None or len(unique_id) != UNIQUE_ID_BYTES:
                raise ValueError("unique_id: the 128 bytes of sharded_abi.unique_id()")
            buf = (C.c_uint8 * UNIQUE_ID_BYTES)(*unique_id)
            rc = SYMBOLS["rl_sharded_create_rccl"](engine._h, world, rank, buf, max_slice_hits, C.byref(h))
        if rc:
            raise ShardedError(rc, "rl_sharded_create failed (see stderr)")
        self._h = h
        self.world, self.rank = world, rank
        engine._dependents.append(self)  # (the engine closes what is built on it first: see Engine.close)

    def _check(self, rc):
        if rc:
            raise ShardedError(rc, SYMBOLS["rl_sharded_last_error"](self._h).decode())

    def submit(self, d_hits, n_hits, now_us, d_verdict):
        """Raw device pointers; both buffers stay untouched until the matching collect()."""
        self._check(SYMBOLS["rl_sharded_submit_device"](self._h, d_hits, n_hits, int(now_us), d_verdict))

    def collect(self):
        """-> hits this rank applied for the oldest slice (its verdicts are ordered on stream(); sync() waits)."""
        n = C.c_uint32()
        self._check(SYMBOLS["rl_sharded_collect"](self._h, C.byref(n)))
        return n.value

    def sweep_submit(self, now_us):
        """This rank's share of a routed sweep, behind every slice submitted so far (every rank calls it at the same point)."""
        self._check(SYMBOLS["rl_sharded_sweep_submit"](self._h, int(now_us)))

    def sweep_collect(self):
        """-> cells this rank's shard dropped (the oldest command in flight must be the sweep)."""
        n = C.c_uint64()
        self._check(SYMBOLS["rl_sharded_sweep_collect"](self._h, C.byref(n)))
        return n.value

    def check_and_update(self, d_hits, n_hits, now_us, d_verdict):
        n = C.c_uint32()
        self._check(SYMBOLS["rl_sharded_check_and_update_device"](self._h, d_hits, n_hits, int(now_us), d_verdict, C.byref(n)))
        return n.value

    def check_requests(self, d_hits, n_hits, d_req_off, n_req, now_us, d_verdict, load_counters=False, d_first_limited=None,
                       d_remaining=None, d_expires=None):
        """Multi-counter requests, counters sharded by key (rl_sharded_check_requests_device): raw device pointers, blocking,
        every rank calls it for every step.  -> rounds the fixpoint took."""
        rounds = C.c_uint32()
        self._check(SYMBOLS["rl_sharded_check_requests_device"](self._h, d_hits, n_hits, d_req_off, n_req, int(now_us),
                                                                int(bool(load_counters)), d_verdict, d_first_limited, d_remaining,
                                                                d_expires, C.byref(rounds)))
        return rounds.value

    def sync(self):
        self._check(SYMBOLS["rl_sharded_sync"](self._h))

    @property
    def stream(self):
        return SYMBOLS["rl_sharded_stream"](self._h)

    @property
    def in_flight(self):
        return SYMBOLS["rl_sharded_in_flight"](self._h)

    def close(self):
        if getattr(self, "_h", None):
            SYMBOLS["rl_sharded_destroy"](self._h)
            self._h = None
            if self in self._engine._dependents:
                self._engine._dependents.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
