"""A MULTI-PROCESS rl_transport for the C router (include/rl_sharded.h), test infrastructure only.

`librl_sharded.so` drives its exchanges through a two-function table (rl_transport): the library's own RCCL communicator,
the in-process one (ranks = threads), or whatever the host supplies.  This one is supplied from Python: every rank is a
PROCESS of its own (its own HIP context, its own engine, its own copy of the router's state machine), the bytes travel
device -> host -> a gloo send / recv between the processes -> device.  It exists so that the C router — not only the
Python driver of the same step, limitador_amd/sharded.py — has run with its ranks in separate processes before the first
8-GPU launch (VERDICT r04 "missing" #1 / next #5b): a rank that issues its exchanges in another order than its peers, or
sizes a segment differently, fails HERE (a size check, or a recv that never completes and hits the timeout), on one GPU.
RCCL itself cannot be used for that on a one-GPU box: it refuses two ranks on one device.

worker(): `python -m helpers.proc_transport <rank> <world> <port> <outdir>` — the routed single-counter step, three slices
in flight, the same seeded slices in every process; the verdicts go to <outdir>/rank<r>.npz for the parent to compare."""
import ctypes as C
import os
import sys

import numpy as np


class RlXfer(C.Structure):
    _fields_ = [("send", C.c_void_p), ("recv", C.c_void_p), ("send_off", C.POINTER(C.c_uint64)),
                ("send_cnt", C.POINTER(C.c_uint64)), ("recv_off", C.POINTER(C.c_uint64)), ("recv_cnt", C.POINTER(C.c_uint64))]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(RlXfer), C.c_uint32, C.c_void_p)


class ProcTransport:
    """exchange(): my send buffers are complete once `stream` has drained; per peer ONE message with the segments of all
    xfers back to back (both sides know every size: my send_cnt[p] is p's recv_cnt[me])."""

    def __init__(self, rank, world):
        import torch
        import torch.distributed as dist

        from limitador_amd import sharded_abi

        self.rank, self.world, self.dist, self.torch = rank, world, dist, torch
        self.hip = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)  # (the copy torch has mapped)
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.calls = 0
        self._fn = EXCHANGE_FN(self._exchange)  # keep the callback alive
        self.transport = sharded_abi.RlTransport()
        self.transport.ctx = None
        self.transport.exchange = C.cast(self._fn, C.c_void_p)

    def _exchange(self, _ctx, xs, n, stream):
        try:
            torch, dist, me = self.torch, self.dist, self.rank
            if self.hip.hipStreamSynchronize(stream) != 0:
                return -2
            self.calls += 1
            reqs, inbox = [], {}
            for p in range(self.world):
                if p == me:
                    continue
                n_in = sum(int(xs[k].recv_cnt[p]) for k in range(n))
                n_out = sum(int(xs[k].send_cnt[p]) for k in range(n))
                out = torch.empty(n_out + 8, dtype=torch.uint8)
                out[:8] = torch.from_numpy(np.array([n_out], dtype=np.uint64).view(np.uint8))  # header: what I think I send
                pos = 8
                for k in range(n):
                    c = int(xs[k].send_cnt[p])
                    if c:
                        if self.hip.hipMemcpy(out.data_ptr() + pos, xs[k].send + int(xs[k].send_off[p]), c, 2) != 0:
                            return -2
                        pos += c
                inbox[p] = torch.empty(n_in + 8, dtype=torch.uint8)
                reqs.append(dist.irecv(inbox[p], src=p))
                reqs.append(dist.isend(out, dst=p))
            for k in range(n):  # what I send to myself: device to device
                c = int(xs[k].send_cnt[me])
                if c != int(xs[k].recv_cnt[me]):
                    return -1
                if c and self.hip.hipMemcpy(xs[k].recv + int(xs[k].recv_off[me]), xs[k].send + int(xs[k].send_off[me]), c, 3) != 0:
                    return -2
            for r in reqs:
                r.wait()
            for p, buf in inbox.items():
                said = int(buf[:8].numpy().view(np.uint64)[0])
                if said != buf.numel() - 8:
                    sys.stderr.write(f"rank {me}: peer {p} sent {said} bytes where {buf.numel() - 8} were expected (exchange {self.calls})\n")
                    return -1  # the ranks disagree on a segment's size: the router's sequence is out of step
                pos = 8
                for k in range(n):
                    c = int(xs[k].recv_cnt[p])
                    if c:
                        if self.hip.hipMemcpy(xs[k].recv + int(xs[k].recv_off[p]), buf.data_ptr() + pos, c, 1) != 0:
                            return -2
                        pos += c
            return 0
        except Exception as ex:  # noqa: BLE001  (an exception must not unwind through the C caller)
            sys.stderr.write(f"rank {self.rank}: transport failed: {ex!r}\n")
            return -2


def slices_for(seed, steps, world, n, n_keys):
    """The same slices in every process (and in the parent, for the oracle)."""
    from limitador_amd import workloads as W

    rng = np.random.default_rng(seed)
    out = []
    for s in range(steps):
        row = []
        for r in range(world):
            h = W.zipf_batch(n_keys, n - 61 * s - 17 * r, rng)
            h["limit"] = (h["key"] % 2).astype(np.uint32)
            h["delta"] = 1 + (h["key"] % 3 == 0)
            row.append(h)
        out.append(row)
    return out


ROWS = [(20, 60), (3, 1)]
STEPS, N, N_KEYS, SEED = 9, 24_000, 5000, 4242


def worker(rank, world, port, outdir):
    import datetime

    import torch
    import torch.distributed as dist

    from limitador_amd import sharded_abi
    from limitador_amd import workloads as W
    from limitador_amd.engine import Engine

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    eng = Engine(capacity_cells=1 << 16, max_batch_hits=world * N)
    eng.set_limits(ROWS)
    tr = ProcTransport(rank, world)
    sh = sharded_abi.Sharded(eng, world, rank, N, transport=tr.transport)
    slices = slices_for(SEED, STEPS, world, N, N_KEYS)
    d_in = [torch.from_numpy(slices[s][rank].view(np.int64).reshape(-1, 2).copy()).to(dev) for s in range(STEPS)]
    d_out = [torch.full((len(slices[s][rank]),), 7, dtype=torch.uint8, device=dev) for s in range(STEPS)]
    torch.cuda.synchronize()
    applied = []
    for s in range(STEPS):
        sh.submit(d_in[s].data_ptr(), len(slices[s][rank]), W.NOW0_US + 350_000 * s, d_out[s].data_ptr())
        if sh.in_flight == sharded_abi.MAX_IN_FLIGHT:
            applied.append(sh.collect())
    while sh.in_flight:
        applied.append(sh.collect())
    sh.sync()
    torch.cuda.synchronize()
    cells = eng.dump_cells()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), applied=np.array(applied), cells=cells, exchanges=tr.calls,
             **{f"v{s}": d_out[s].cpu().numpy() for s in range(STEPS)})
    sh.close()
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    worker(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
