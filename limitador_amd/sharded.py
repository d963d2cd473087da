"""Hash-sharded counter storage across the GPUs of one node (one process per GPU).

The key space is partitioned by ``owner_of(key)``; every GPU holds a private table for its share
and there is no replication and no cross-GPU atomic.  One step of the data path:

    ingress slice --stable partition by owner--> [to 0 | to 1 | ...]          (HIP, rl_route.hpp)
    counts all-to-all (8 x 8 ints)                                            (RCCL)
    descriptor all-to-all: 16-byte rl_hit records to their owners             (RCCL over xGMI)
    local check_and_update on the owner, in global trace order                (HIP engine)
    verdict all-to-all: 1 byte per hit back to the ingress rank               (RCCL)
    un-permute to ingress order                                               (HIP)

Global trace order is "rank 0's slice, then rank 1's, ..."; because the partition is stable and
all_to_all lays the received segments out by source rank, the batch an owner sees is exactly
that order restricted to its keys, so the sharded result is bit-identical to the sequential
reference applied to the concatenated slices (tests/test_sharded_gloo.py checks this with two
gloo processes; the driver runs the RCCL path on 2/4/8 GPUs).

Requests that touch several counters (multi-limit namespaces: all-or-nothing across the request's
counters, in_memory.rs:141-153) are routed by NAMESPACE instead (ShardedRequestEngine below): every
limit and counter of a namespace lives on one GPU, so a request's counters are never split and the
owner runs the on-device matcher + the general resolver on "the global trace restricted to its
namespaces".

This is a different job from the reference's only multi-node mechanism (CRDT replication over
gRPC, limitador/src/storage/distributed/): that one replicates counters, this one routes requests.
"""
import torch
import torch.distributed as dist


def _lsr(v, k):
    return (v >> k) & ((1 << (64 - k)) - 1)


def _c(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def owner_of_tensor(keys, hash_seed, world):
    """owner_of() of rl_cell.hpp on an int64 tensor of key bit patterns (two's-complement wrap)."""
    x = keys ^ _c(hash_seed)
    x = x ^ _lsr(x, 33)
    x = x * _c(0xFF51AFD7ED558CCD)
    x = x ^ _lsr(x, 33)
    x = x * _c(0xC4CEB9FE1A85EC53)
    x = x ^ _lsr(x, 33)
    return ((x & 0xFFFFFFFF) * world) >> 32


def owner_mask(keys, hash_seed, world, rank):
    return owner_of_tensor(keys, hash_seed, world) == rank


SLOTS = 3  # ingress slices in flight per rank (buffer sets)


class HipLocal:
    """The three device-side operations of a shard, on the HIP engine (raw pointers).  The engine is
    put on torch's current stream, so its kernels, the RCCL collectives and torch's own ops are ordered
    by the stream and none of the three operations blocks the host.  One set of routing buffers per
    slice in flight (ShardedEngine.submit / collect)."""

    def __init__(self, engine, device, max_local_hits, world):
        self.engine = engine
        self.device = device
        self.sorted_hits = [torch.empty((max_local_hits, 2), dtype=torch.int64, device=device) for _ in range(SLOTS)]
        self.perm = [torch.empty(max_local_hits, dtype=torch.int32, device=device) for _ in range(SLOTS)]
        engine.set_stream(torch.cuda.current_stream(device).cuda_stream)

    def partition(self, hits, world, slot, counts_out):
        """Stable partition of `hits` by owner -> (sorted hits, permutation); hits per owner into counts_out."""
        n = hits.shape[0]
        self.engine.route_partition_device(hits.data_ptr(), n, world, self.sorted_hits[slot].data_ptr(),
                                           self.perm[slot].data_ptr(), counts_out.data_ptr())
        return self.sorted_hits[slot][:n], self.perm[slot][:n]

    def check(self, hits, n, now_us, verdict):
        """Enqueue the local batch; -> True if finish() has something to wait for."""
        if n:
            self.engine.submit_device(hits.data_ptr(), n, now_us, verdict.data_ptr())
        return bool(n)

    def unpermute(self, src, perm, n, dst):
        self.engine.unpermute_u8_device(src.data_ptr(), perm.data_ptr(), n, dst.data_ptr())

    def finish(self):
        """Status of the OLDEST batch enqueued by check() (raises the engine's error, if any)."""
        self.engine.collect()


ROUTED, APPLIED, RETURNED = 1, 2, 3


class ShardedEngine:
    """submit() takes one ingress slice; collect() finishes the oldest one: its verdicts are then in
    ingress order in its verdict_out.  A slice goes through three stages, each only ENQUEUED on the
    stream (nothing here waits for the device except the size read, see below):

        ROUTED    stable partition by owner, size exchange, sizes copied to pinned host memory
        APPLIED   descriptor all-to-all (needs the sizes on the host), local check_and_update
        RETURNED  verdict all-to-all back to the ingress ranks, un-permute to ingress order

    Up to three slices are in flight, and submit() of slice i enqueues, in this order,
        route(i),  return(i-2),  exchange+apply(i-1)
    so the sizes of slice i-1 were copied long before the host needs them (their copy sits in front of
    the local batch of slice i-2 in the stream): the host does not drain the device to read 2 x world
    integers, and collect(i-2) leaves the local batch of slice i-1 queued behind it.  A blocking
    check_and_update() is submit + collect on an empty pipeline.  Slices are applied in submission
    order and every rank issues the same sequence of collectives."""

    def __init__(self, engine, group, device, max_local_hits, local=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = torch.device(device)
        self.local = local if local is not None else HipLocal(engine, device, max_local_hits, self.world)
        self.max_recv = getattr(engine, "max_batch_hits", None)
        cap = self.max_recv or 2 * max_local_hits
        dev = self.device
        self._recv_hits = [torch.empty((cap, 2), dtype=torch.int64, device=dev) for _ in range(SLOTS)]
        self._recv_verdict = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(SLOTS)]
        self._sorted_verdict = [torch.empty(max_local_hits, dtype=torch.uint8, device=dev) for _ in range(SLOTS)]
        # row 0: hits this rank sends to each owner, row 1: hits it receives from each rank
        self._counts = [torch.empty((2, self.world), dtype=torch.int32, device=dev) for _ in range(SLOTS)]
        self._on_gpu = dev.type == "cuda"
        if self._on_gpu:
            self._counts_host = [torch.empty((2, self.world), dtype=torch.int32, pin_memory=True) for _ in range(SLOTS)]
            self._counts_ready = [torch.cuda.Event() for _ in range(SLOTS)]
        self._pending = []  # oldest first: one dict per slice in flight
        self._seq = 0

    # -- the three stages of a slice ---------------------------------------------------------------
    def _route(self, hits, now_us, verdict_out):
        slot = self._seq % SLOTS
        self._seq += 1
        cnt = self._counts[slot]
        sorted_hits, perm = self.local.partition(hits, self.world, slot, cnt[0])
        dist.all_to_all_single(cnt[1], cnt[0], group=self.group)
        if self._on_gpu:
            self._counts_host[slot].copy_(cnt, non_blocking=True)
            self._counts_ready[slot].record()
        return {"slot": slot, "n": hits.shape[0], "now": now_us, "sorted": sorted_hits, "perm": perm,
                "out": verdict_out, "stage": ROUTED, "n_recv": 0, "waits": False}

    def _apply(self, p):
        slot = p["slot"]
        if self._on_gpu:
            self._counts_ready[slot].synchronize()
            send, recv = self._counts_host[slot].tolist()
        else:
            send, recv = self._counts[slot].tolist()
        n_recv = sum(recv)
        if n_recv > self._recv_hits[slot].shape[0]:
            raise RuntimeError(f"rank {self.rank}: {n_recv} routed hits exceed the receive buffer "
                               f"({self._recv_hits[slot].shape[0]}); raise max_batch_hits")
        rh = self._recv_hits[slot][:n_recv]
        dist.all_to_all_single(rh, p["sorted"], output_split_sizes=recv, input_split_sizes=send, group=self.group)
        rv = self._recv_verdict[slot][:n_recv]
        p["waits"] = bool(self.local.check(rh, n_recv, p["now"], rv))
        p.update(send=send, recv=recv, n_recv=n_recv, rv=rv, stage=APPLIED)

    def _return(self, p):
        sv = self._sorted_verdict[p["slot"]][:p["n"]]
        dist.all_to_all_single(sv, p["rv"], output_split_sizes=p["send"], input_split_sizes=p["recv"], group=self.group)
        self.local.unpermute(sv, p["perm"], p["n"], p["out"])
        p["stage"] = RETURNED

    # -- the pipeline ---------------------------------------------------------------------------------
    def submit(self, hits, now_us, verdict_out):
        """hits: [n,2] int64 tensor laid out as rl_hit; verdict_out: uint8[n] (ingress order), valid after
        the matching collect().  Both must stay alive and untouched until then."""
        if len(self._pending) >= SLOTS:
            raise RuntimeError(f"{SLOTS} slices are already in flight: collect() first")
        older = list(self._pending)
        self._pending.append(self._route(hits, now_us, verdict_out))
        for p in older:
            if p["stage"] == APPLIED:
                self._return(p)
        for p in older:
            if p["stage"] == ROUTED:
                self._apply(p)

    def collect(self):
        """Finish the oldest slice in flight: its verdicts are in its verdict_out (device-ordered on the
        current stream).  -> hits this rank applied for it."""
        if not self._pending:
            raise RuntimeError("nothing in flight")
        p = self._pending[0]
        if p["stage"] == ROUTED:
            self._apply(p)
        if p["stage"] == APPLIED:
            self._return(p)
        self._pending.pop(0)
        if p["waits"] and hasattr(self.local, "finish"):
            self.local.finish()
        return p["n_recv"]

    @property
    def in_flight(self):
        return len(self._pending)

    def check_and_update(self, hits, now_us, verdict_out):
        """One slice, blocking until it is applied (submit + collect on an empty pipeline)."""
        if self._pending:
            raise RuntimeError("slices in flight: collect() them first")
        self.submit(hits, now_us, verdict_out)
        return self.collect()


# ---------------------------------------------------------------------------------------------------
# Requests with several counters: sharded by namespace
# ---------------------------------------------------------------------------------------------------
def namespace_owner(ns, world):
    """Owner shard of a namespace id (int tensor) — a 32-bit mix, the same on every rank."""
    x = (ns.to(torch.int64) * 0x9E3779B1) & 0xFFFFFFFF
    x = x ^ (x >> 15)
    x = (x * 0x85EBCA6B) & 0xFFFFFFFF
    x = x ^ (x >> 13)
    return (x * world) >> 32


class HipMatchLocal:
    """Owner-side work on the HIP engine: counters_that_apply + check_and_update for the received
    requests (rl_match_and_check_batch_device).  The engine's match table holds ALL limits; a rank
    only ever receives requests of the namespaces it owns."""

    def __init__(self, engine, device):
        self.engine = engine
        engine.set_stream(torch.cuda.current_stream(device).cuda_stream)

    def match_and_check(self, ns, ent_off, ent_key, ent_val, delta, now_us, verdict, limited):
        n = ns.shape[0]
        if n:
            self.engine.match_and_check_device(ns.data_ptr(), ent_off.data_ptr(), ent_key.data_ptr(),
                                               ent_val.data_ptr(), delta.data_ptr(), n, now_us,
                                               verdict.data_ptr(), limited.data_ptr())


class ShardedRequestEngine:
    """check(): one ingress slice of dictionary-encoded requests (namespace id, delta, CSR descriptor
    entries — the input of rl_match_and_check_batch) -> verdict and limited-limit id per request, in
    ingress order.  Requests travel to the owner of their namespace (stable partition, so the owner
    sees "rank 0's slice, then rank 1's, ..." restricted to its namespaces = the sequential reference on
    the concatenated slices), are matched and decided there, and the two result words travel back.
    Blocking: the general resolver behind the matcher has host round trips of its own."""

    def __init__(self, group, device, local):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = torch.device(device)
        self.local = local

    def check(self, req_ns, ent_off, ent_key, ent_val, req_delta, now_us):
        """int32 tensors on self.device: req_ns[n], req_delta[n], ent_off[n+1], ent_key[m], ent_val[m]."""
        dev, world = self.device, self.world
        n = req_ns.shape[0]
        i64 = torch.int64
        owner = namespace_owner(req_ns, world)
        perm = torch.sort(owner, stable=True).indices
        ne = (ent_off[1:] - ent_off[:-1]).to(i64)
        send_cnt = torch.zeros((world, 2), dtype=i64, device=dev)
        send_cnt[:, 0] = torch.bincount(owner, minlength=world)
        send_cnt[:, 1].index_add_(0, owner, ne)
        recv_cnt = torch.empty_like(send_cnt)
        dist.all_to_all_single(recv_cnt, send_cnt, group=self.group)
        # the slice in owner order: request words, then every request's entries, contiguous again
        ne_s = ne[perm]
        reqs = torch.stack([req_ns.to(torch.int32), req_delta.to(torch.int32), ne.to(torch.int32)], dim=1)[perm].contiguous()
        m = int(ent_key.shape[0])
        first_s = ent_off[:-1].to(i64)[perm] - (torch.cumsum(ne_s, 0) - ne_s)
        src = torch.repeat_interleave(first_s, ne_s, output_size=m) + torch.arange(m, dtype=i64, device=dev)
        ents = torch.stack([ent_key.to(torch.int32), ent_val.to(torch.int32)], dim=1)[src].contiguous()
        sc, rc = send_cnt.tolist(), recv_cnt.tolist()
        send_req, send_ent = [c[0] for c in sc], [c[1] for c in sc]
        recv_req, recv_ent = [c[0] for c in rc], [c[1] for c in rc]
        n_recv, m_recv = sum(recv_req), sum(recv_ent)
        r_reqs = torch.empty((n_recv, 3), dtype=torch.int32, device=dev)
        r_ents = torch.empty((m_recv, 2), dtype=torch.int32, device=dev)
        dist.all_to_all_single(r_reqs, reqs, output_split_sizes=recv_req, input_split_sizes=send_req, group=self.group)
        dist.all_to_all_single(r_ents, ents, output_split_sizes=recv_ent, input_split_sizes=send_ent, group=self.group)
        # owner side
        r_off = torch.zeros(n_recv + 1, dtype=torch.int32, device=dev)
        r_off[1:] = torch.cumsum(r_reqs[:, 2], 0)
        r_verdict = torch.zeros(n_recv, dtype=torch.uint8, device=dev)
        r_limited = torch.full((n_recv,), -1, dtype=torch.int32, device=dev)
        self.local.match_and_check(r_reqs[:, 0].contiguous(), r_off, r_ents[:, 0].contiguous(),
                                   r_ents[:, 1].contiguous(), r_reqs[:, 1].contiguous(), now_us, r_verdict, r_limited)
        # back to the ingress ranks, then to ingress order
        s_verdict = torch.empty(n, dtype=torch.uint8, device=dev)
        s_limited = torch.empty(n, dtype=torch.int32, device=dev)
        dist.all_to_all_single(s_verdict, r_verdict, output_split_sizes=send_req, input_split_sizes=recv_req, group=self.group)
        dist.all_to_all_single(s_limited, r_limited, output_split_sizes=send_req, input_split_sizes=recv_req, group=self.group)
        verdict = torch.empty_like(s_verdict)
        limited = torch.empty_like(s_limited)
        verdict[perm] = s_verdict
        limited[perm] = s_limited
        return verdict, limited


# ---------------------------------------------------------------------------------------------------
# Requests with several counters, sharded by KEY: a request's counters live on several GPUs
# ---------------------------------------------------------------------------------------------------
class TorchTransport:
    """The exchanges of ShardedMultiCounterEngine over a torch.distributed group (RCCL on GPUs, gloo on CPU)."""

    def __init__(self, group, device):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = torch.device(device)

    def all_gather_int(self, value):
        parts = [torch.zeros(1, dtype=torch.int64, device=self.device) for _ in range(self.world)]
        dist.all_gather(parts, torch.tensor([int(value)], dtype=torch.int64, device=self.device), group=self.group)
        return [int(p.item()) for p in parts]

    def all_reduce_max(self, value):
        t = torch.tensor([int(value)], dtype=torch.int64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def all_to_all_v(self, send, send_counts, recv_counts):
        out = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        dist.all_to_all_single(out, send.contiguous(), output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts),
                               group=self.group)
        return out


class InProcessGroup:
    """Ranks as threads of one process (tests on a single GPU, or a host that drives several GPUs itself): a
    mailbox per exchange and a barrier on either side of it."""

    def __init__(self, world):
        import threading

        self.world = world
        self._barrier = threading.Barrier(world)
        self._box = [None] * world

    def transport(self, rank, device):
        return _InProcessTransport(self, rank, device)


class _InProcessTransport:
    def __init__(self, group, rank, device):
        self.g, self.rank, self.world, self.device = group, rank, group.world, torch.device(device)

    def _swap(self, item):
        self.g._box[self.rank] = item
        self.g._barrier.wait()
        got = list(self.g._box)
        self.g._barrier.wait()
        return got

    def all_gather_int(self, value):
        return [int(v) for v in self._swap(int(value))]

    def all_reduce_max(self, value):
        return max(int(v) for v in self._swap(int(value)))

    def all_to_all_v(self, send, send_counts, recv_counts):
        if send.is_cuda:
            torch.cuda.current_stream(send.device).synchronize()  # the peers read my rows from another thread
        offs = [0]
        for c in send_counts:
            offs.append(offs[-1] + int(c))
        got = self._swap((send, offs))
        parts = [got[p][0][got[p][1][self.rank]:got[p][1][self.rank + 1]] for p in range(self.world)]
        assert [int(p.shape[0]) for p in parts] == [int(c) for c in recv_counts]
        out = torch.cat(parts) if parts else send[:0]
        if out.is_cuda:
            torch.cuda.current_stream(out.device).synchronize()
        self.g._barrier.wait()  # every rank has copied: the send buffers may change again
        return out


class HipGenLocal:
    """Owner-side work of ShardedMultiCounterEngine on the HIP engine: the general resolver in phases
    (rl_gen_begin_device .. rl_gen_commit_device).  The engine works on its own stream and every call returns with
    its results complete; the tensors handed in were produced on torch's stream, hence the synchronise."""

    def __init__(self, engine, device):
        self.engine = engine
        self.device = torch.device(device)
        self.n = 0
        self.load = False

    def _ptr(self, t):
        return t.data_ptr() if t is not None and t.numel() else None

    def begin(self, hits, req_id, now_us, load):
        self.n, self.load = int(hits.shape[0]), bool(load)
        dev = self.device
        self._keep = (hits.contiguous(), req_id.contiguous())
        self.pass_ = torch.empty(max(self.n, 1), dtype=torch.uint8, device=dev)
        self.rem = torch.zeros(max(self.n, 1), dtype=torch.int64, device=dev) if load else None
        self.exp = torch.zeros(max(self.n, 1), dtype=torch.int64, device=dev) if load else None
        torch.cuda.synchronize(dev)
        self.engine.gen_begin(self._ptr(self._keep[0]), self._ptr(self._keep[1]), self.n, now_us, load)

    def round(self, admitted):
        """admitted: uint8[n] or None (all) -> pass flags uint8[n]"""
        adm = admitted.contiguous() if admitted is not None else None
        torch.cuda.synchronize(self.device)
        self.engine.gen_round(self._ptr(adm), self._ptr(self.pass_), self._ptr(self.rem), self._ptr(self.exp))
        return self.pass_[: self.n]

    def count(self, reached):
        r = reached.contiguous() if reached is not None else None
        torch.cuda.synchronize(self.device)
        return self.engine.gen_count(self._ptr(r))

    def loaded(self):
        return self.rem[: self.n], self.exp[: self.n]

    def commit(self):
        self.engine.gen_commit()

    def abort(self):
        self.engine.gen_abort()


class ShardedTableFull(RuntimeError):
    pass


class ShardedMultiCounterEngine:
    """check(): one ingress slice of requests with SEVERAL counters each, the counters sharded by key like the
    single-counter path — so a request's counters live on several GPUs, and the all-or-nothing rule of
    InMemoryStorage::check_and_update (in_memory.rs:141-153) spans them.  SURVEY.md §8(e) "k > 1":

        hits -> owners (stable partition by owner, exchanged with the id of their request)
        owners: sort by cell, read the cells                                       (rl_gen_begin_device)
        repeat  owners: per hit "fits on top of the admitted hits before it"       (rl_gen_round_device)
                flags back to the ingress ranks; per request AND; admitted bits out to the owners again
        until no rank saw the admitted set change                                  (Jacobi rounds: the unique fixpoint)
        walks' ends -> owners; cells to create, room: all ranks fit or none does   (rl_gen_count_device)
        owners commit                                                              (rl_gen_commit_device)

    Global trace order is "rank 0's requests, then rank 1's, ..." per call; the owners see that order restricted to
    their keys, so verdicts, first_limited, remaining / expires_in and the tables equal ONE sequential storage fed
    the concatenated slices (tests/test_sharded_multi_gloo.py: 2 gloo ranks with a CPU stand-in for the owner-side
    phases; tests/test_gpu_sharded_multi.py: the HIP engine, world 1 over RCCL and world 2-3 as threads).
    Two exchanges of one byte per hit per round: namespace sharding (ShardedRequestEngine) is cheaper whenever the
    namespaces balance; this one has no unit of distribution coarser than a key.  Every rank calls check() for
    every slice."""

    def __init__(self, transport, local, hash_seed):
        self.t = transport
        self.local = local
        self.hash_seed = int(hash_seed)
        self.rounds = 0

    def check(self, hits, req_off, now_us, load_counters=False):
        """hits: [n, 2] int64 rows laid out as rl_hit, simple counters first inside a request; req_off: int64[n_req + 1].
        -> (verdict uint8[n_req], first_limited int64[n_req] (index into hits, -1), remaining, expires_in_us (per hit,
        int64 bit patterns; None unless load_counters))"""
        T, W = self.t, self.t.world
        dev = hits.device
        i64 = torch.int64
        n, n_req = int(hits.shape[0]), int(req_off.shape[0]) - 1
        req_off = req_off.to(i64)
        all_req = T.all_gather_int(n_req)
        if sum(all_req) >= 2**32:
            raise ValueError("more than 2^32 requests in one step")
        base = sum(all_req[: T.rank])
        lens = req_off[1:] - req_off[:-1]
        req_of_hit = torch.repeat_interleave(torch.arange(n_req, dtype=i64, device=dev), lens)
        req_id = (req_of_hit + base).to(torch.int32)  # (bit pattern of the u32 id)
        owner = owner_of_tensor(hits[:, 0], self.hash_seed, W) if n else torch.zeros(0, dtype=i64, device=dev)
        order = torch.sort(owner, stable=True).indices
        send = torch.bincount(owner, minlength=W).tolist() if n else [0] * W
        recv = [int(x) for x in T.all_to_all_v(torch.tensor(send, dtype=i64, device=dev), [1] * W, [1] * W).tolist()]
        r_hits = T.all_to_all_v(hits[order], send, recv)
        r_req = T.all_to_all_v(req_id[order], send, recv)
        err = None
        try:
            self.local.begin(r_hits, r_req, now_us, load_counters)
        except Exception as ex:  # a malformed slice on one owner refuses the step on every rank
            err = ex
        if T.all_reduce_max(1 if err is not None else 0):
            if err is None:
                self.local.abort()
                raise RuntimeError("another rank refused the step")
            raise err
        zero_front = torch.zeros(1, dtype=i64, device=dev)
        admitted_recv, adm_prev = None, None
        rounds = 0
        while True:
            pass_recv = self.local.round(admitted_recv)
            rounds += 1
            pass_home = torch.empty(n, dtype=torch.uint8, device=dev)
            pass_home[order] = T.all_to_all_v(pass_recv, recv, send)
            fail = pass_home == 0
            c = torch.cat([zero_front, torch.cumsum(fail.to(i64), 0)])
            adm_req = (c[req_off[1:]] - c[req_off[:-1]]) == 0
            changed = (not bool(adm_req.all())) if adm_prev is None else bool((adm_req != adm_prev).any())
            if not T.all_reduce_max(1 if changed else 0):
                break  # the admitted set of this round is the one the flags were computed with: the fixpoint
            adm_prev = adm_req
            adm_hit = adm_req[req_of_hit].to(torch.uint8)
            admitted_recv = T.all_to_all_v(adm_hit[order], send, recv)
        self.rounds = rounds
        verdict = (~adm_req).to(torch.uint8)
        pos = torch.arange(n, dtype=i64, device=dev)
        first = torch.full((n_req,), n, dtype=i64, device=dev)
        if n:
            first.scatter_reduce_(0, req_of_hit, torch.where(fail, pos, torch.full_like(pos, n)), "amin")
        first_limited = torch.where(first < n, first, torch.full_like(first, -1))
        reached_recv = None
        if not load_counters:
            # the walk stops at its first limited counter (in_memory.rs:109-113,129-133)
            stop = torch.where(first < n, first + 1, req_off[1:])
            reached = (pos < stop[req_of_hit]).to(torch.uint8)
            reached_recv = T.all_to_all_v(reached[order], send, recv)
        n_new, room = self.local.count(reached_recv)
        if T.all_reduce_max(1 if n_new > room else 0):
            self.local.abort()
            raise ShardedTableFull("refused, nothing applied on any rank: a shard cannot take the cells the step creates")
        self.local.commit()
        rem = exp = None
        if load_counters:
            l_rem, l_exp = self.local.loaded()
            rem = torch.empty(n, dtype=i64, device=dev)
            exp = torch.empty(n, dtype=i64, device=dev)
            rem[order] = T.all_to_all_v(l_rem, recv, send)
            exp[order] = T.all_to_all_v(l_exp, recv, send)
        return verdict, first_limited, rem, exp
