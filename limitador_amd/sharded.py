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
