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


class HipLocal:
    """The three device-side operations of a shard, on the HIP engine (raw pointers).  The engine is
    put on torch's current stream, so its kernels, the RCCL collectives and torch's own ops are ordered
    by the stream and none of the three operations blocks the host."""

    def __init__(self, engine, device, max_local_hits, world):
        self.engine = engine
        self.device = device
        self.sorted_hits = torch.empty((max_local_hits, 2), dtype=torch.int64, device=device)
        self.perm = torch.empty(max_local_hits, dtype=torch.int32, device=device)
        self.counts = torch.empty(world, dtype=torch.int32, device=device)
        engine.set_stream(torch.cuda.current_stream(device).cuda_stream)
        self._pending = False

    def partition(self, hits, world):
        n = hits.shape[0]
        self.engine.route_partition_device(hits.data_ptr(), n, world, self.sorted_hits.data_ptr(),
                                           self.perm.data_ptr(), self.counts.data_ptr())
        return self.sorted_hits[:n], self.perm[:n], self.counts

    def check(self, hits, n, now_us, verdict):
        if n:
            self.engine.submit_device(hits.data_ptr(), n, now_us, verdict.data_ptr())
            self._pending = True

    def unpermute(self, src, perm, n, dst):
        self.engine.unpermute_u8_device(src.data_ptr(), perm.data_ptr(), n, dst.data_ptr())

    def finish(self):
        """Status of the batch submitted by check() (raises the engine's error, if any)."""
        if self._pending:
            self._pending = False
            self.engine.collect()


class ShardedEngine:
    def __init__(self, engine, group, device, max_local_hits, local=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = device
        self.local = local if local is not None else HipLocal(engine, device, max_local_hits, self.world)
        self.max_recv = getattr(engine, "max_batch_hits", None)
        cap = self.max_recv or 2 * max_local_hits
        self._recv_hits = torch.empty((cap, 2), dtype=torch.int64, device=device)
        self._recv_verdict = torch.empty(cap, dtype=torch.uint8, device=device)
        self._sorted_verdict = torch.empty(max_local_hits, dtype=torch.uint8, device=device)
        self._recv_counts = torch.empty(self.world, dtype=torch.int32, device=device)

    def check_and_update(self, hits, now_us, verdict_out):
        """hits: [n,2] int64 tensor laid out as rl_hit; verdict_out: uint8[n] (ingress order)."""
        n = hits.shape[0]
        sorted_hits, perm, counts = self.local.partition(hits, self.world)
        # who sends how much to whom
        dist.all_to_all_single(self._recv_counts, counts, group=self.group)
        send = counts.tolist()
        recv = self._recv_counts.tolist()
        n_recv = sum(recv)
        if n_recv > self._recv_hits.shape[0]:
            raise RuntimeError(f"rank {self.rank}: {n_recv} routed hits exceed the receive buffer "
                               f"({self._recv_hits.shape[0]}); raise max_batch_hits")
        rh = self._recv_hits[:n_recv]
        dist.all_to_all_single(rh, sorted_hits, output_split_sizes=recv, input_split_sizes=send, group=self.group)
        rv = self._recv_verdict[:n_recv]
        self.local.check(rh, n_recv, now_us, rv)
        sv = self._sorted_verdict[:n]
        dist.all_to_all_single(sv, rv, output_split_sizes=send, input_split_sizes=recv, group=self.group)
        # the batch's status is read while the verdicts travel back; routing helpers do not block
        if hasattr(self.local, "finish"):
            self.local.finish()
        self.local.unpermute(sv, perm, n, verdict_out)
        return n_recv
