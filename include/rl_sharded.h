/*
 * rl_sharded.h — the multi-GPU step of the counter engine behind a C ABI.
 *
 * One process (or thread) per GPU, one rl_engine per rank holding the cells of the keys it owns
 * (owner = rl_owner_of(key, seed, world): hash partition, no replication, no cross-GPU atomics).
 * A rank hands in its INGRESS slice of single-counter requests; the step routes every hit to its owner,
 * applies `InMemoryStorage::check_and_update` (limitador/src/storage/in_memory.rs:72-156) there and
 * returns the verdicts in ingress order.  The partition by owner is stable and the exchange lays the
 * received hits out by source rank, so an owner applies "rank 0's slice, then rank 1's, ..." restricted
 * to its keys: the result is bit-identical to the sequential reference on the concatenated slices.
 * The reference has no counterpart: its in-memory storage is one process.
 *
 * Per slice and rank, on the device, nothing but enqueues on the host side:
 *     route     stable partition by owner (k_route_*), hits-per-owner counts
 *     exchange  GROUP A: counts to/from every peer  (+ the verdicts of the slice two back: one launch)
 *               GROUP B: 16-byte hit records to their owners (sizes known from GROUP A)
 *     apply     the local batch on the engine (rl_check_and_update_submit_device)
 *     return    verdict bytes back to the ingress ranks (rides in GROUP A of the slice two ahead, or
 *               alone when the pipeline drains), un-permute to ingress order
 * Up to four slices are in flight (RL_SHARDED_MAX_IN_FLIGHT: the newest one routed, up to three on the engine — applied /
 * returned); a caller that keeps the window full (`submit(k); collect(k - 3)`) lets the host enqueue a slice ahead of the
 * device.  Routing and exchanges run on one
 * stream, the engine's batches on another.  submit(i) enqueues
 * GROUP B of slice i-1 and its local batch, then route(i) and GROUP A(i): the engine never waits for an
 * exchange that is itself waiting for the engine.
 *
 * EVERY rank must issue the same sequence of rl_sharded_submit_device / rl_sharded_collect calls (they
 * contain collectives), with the same now_us for the same slice.
 *
 * Failure is an outcome of the collective step, not an exit from it.  A slice that fails on ONE rank — more hits
 * routed to it than its engine takes in one batch (RL_ERR_BATCH_TOO_LARGE), a full table (RL_ERR_TABLE_FULL), a
 * malformed hit — still has every one of its exchanges issued by that rank (the receive buffers hold world x
 * max_slice_hits records, so any skew can be received): its peers never wait for a send that does not come.  The
 * failing rank answers 0xFF for every hit it owns in that slice (what an ingress rank then finds in d_verdict for
 * those requests: neither 0 nor 1), applies nothing of it, and gets the error from rl_sharded_collect of THAT slice;
 * the slices behind it proceed.
 *
 * Transport: RCCL (ncclCommInitRank + grouped ncclSend / ncclRecv over xGMI) when the communicator is
 * created from a unique id; or any rl_transport the host supplies — `rl_local_group` below is an
 * in-process one (ranks = threads of one process, device-to-device copies), which is also how the
 * world-2 tests run on a single GPU.
 */
#ifndef RL_SHARDED_H
#define RL_SHARDED_H

#include <stdint.h>

#include "rl_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rl_sharded rl_sharded;

/* One segment of a grouped all-to-all: peer p gets send_cnt[p] bytes from send + send_off[p] and
 * delivers recv_cnt[p] bytes to recv + recv_off[p] (device pointers; offsets / counts host arrays of
 * `world` entries, in bytes).  Peers agree on the sizes: my send_cnt[p] is p's recv_cnt[me]. */
typedef struct {
    const void *send;
    void *recv;
    const uint64_t *send_off, *send_cnt, *recv_off, *recv_cnt;
} rl_xfer;

/* exchange(): enqueue all segments as ONE grouped operation on `stream` (hipStream_t) and return;
 * 0 or a negative rl_status.  Called by every rank with the same number of segments. */
typedef struct {
    void *ctx;
    int32_t (*exchange)(void *ctx, const rl_xfer *xfers, uint32_t n_xfers, void *stream);
} rl_transport;

#define RL_UNIQUE_ID_BYTES 128
/* ncclGetUniqueId: call on one rank, hand the bytes to all ranks by whatever means the host has. */
int32_t rl_sharded_unique_id(uint8_t id[RL_UNIQUE_ID_BYTES]);

/* max_slice_hits: largest ingress slice of this rank.  The engine's max_batch_hits bounds what this rank can APPLY for
 * one slice (all ranks' slices may hash here): create it with world x max_slice_hits to rule the failure above out.
 * RCCL is bound at run time, to the librccl.so the process has already mapped if there is one (a host that uses
 * torch.distributed): one process, one RCCL; two mapped copies are refused.
 * The engine is switched to the communicator's apply stream (rl_engine_set_stream) until rl_sharded_destroy
 * and must not be used directly while slices are in flight.  Destroy the communicator BEFORE its engine:
 * rl_sharded_destroy hands the engine its own streams back. */
int32_t rl_sharded_create_rccl(rl_engine *e, uint32_t world, uint32_t rank, const uint8_t id[RL_UNIQUE_ID_BYTES],
                               uint32_t max_slice_hits, rl_sharded **out);
int32_t rl_sharded_create(rl_engine *e, uint32_t world, uint32_t rank, const rl_transport *t,
                          uint32_t max_slice_hits, rl_sharded **out);
void rl_sharded_destroy(rl_sharded *s);
const char *rl_sharded_last_error(const rl_sharded *s);

/* Enqueue one ingress slice (device pointers; d_hits and d_verdict stay untouched until the matching
 * collect).  RL_ERR_BUSY with RL_SHARDED_MAX_IN_FLIGHT slices in flight. */
#define RL_SHARDED_MAX_IN_FLIGHT 4
int32_t rl_sharded_submit_device(rl_sharded *s, const rl_hit *d_hits, uint32_t n_hits, uint64_t now_us,
                                 uint8_t *d_verdict);
/* Finish the OLDEST slice: its verdicts are in its d_verdict once the exchange stream has been
 * synchronised (rl_sharded_stream) — or use rl_sharded_sync.  -> the status of the local batch this
 * rank applied for that slice; *n_applied = hits it applied. */
int32_t rl_sharded_collect(rl_sharded *s, uint32_t *n_applied);
/* A sweep of expired counters as a command of the routed pipeline: every rank calls it at the same point of its sequence of
 * submits; its shard is swept behind every slice submitted so far and in front of every later one — the point a sequential
 * storage would be swept at — without draining the slices in flight (it takes one of the engine's three in-flight places — RL_ERR_BUSY with more than two commands in flight — and is
 * collected in order with rl_sharded_sweep_collect; *n_removed = cells THIS rank's shard dropped).  The sweep itself is
 * rl_sweep_expired_submit of rl_engine.h (qualified cells with expiry <= now_us; no reference analogue: an explicit eviction
 * event, replayed into the oracle by the tests). */
int32_t rl_sharded_sweep_submit(rl_sharded *s, uint64_t now_us);
int32_t rl_sharded_sweep_collect(rl_sharded *s, uint64_t *n_removed);
/* submit + collect + synchronise on an empty pipeline. */
int32_t rl_sharded_check_and_update_device(rl_sharded *s, const rl_hit *d_hits, uint32_t n_hits, uint64_t now_us,
                                           uint8_t *d_verdict, uint32_t *n_applied);
/* Requests with SEVERAL counters each, the counters sharded by key like everything else: a request's counters live on
 * several GPUs and the all-or-nothing rule of check_and_update (in_memory.rs:141-153) spans them (SURVEY.md §8e "k > 1").
 *     hits -> owners (stable partition by owner; the id of its request travels with every hit)
 *     owners: sort by cell, read the cells                                            (rl_gen_begin_device)
 *     repeat  owners: per hit "fits on top of the admitted hits before it"            (rl_gen_round_device)
 *             flags back to the ingress ranks; per request AND; admitted bits out to the owners again
 *     until no rank saw the admitted set change (Jacobi rounds from "all admitted": the unique fixpoint)
 *     the walks' ends -> owners; cells to create against room: all ranks fit or none does   (rl_gen_count_device)
 *     owners commit                                                                   (rl_gen_commit_device)
 * Global trace order per call is "rank 0's requests, then rank 1's, ...", so verdicts, first_limited, remaining /
 * expires_in and the union of the tables equal ONE sequential storage fed the concatenated slices.
 * d_hits: the slice's counters, request after request (CSR d_req_off[n_req + 1]; simple counters first inside a request,
 * every hit carrying its request's delta); d_first_limited[r] = index into d_hits of the request's first limited counter
 * or -1 (may be NULL); d_remaining / d_expires_in_us per hit with load_counters (else may be NULL).  n_hits and n_req
 * <= max_slice_hits.  BLOCKING, on an empty pipeline, called by EVERY rank for every step (it contains collectives: two
 * exchanges of one byte per hit and one word per rank each round).  A step one rank cannot take — a malformed hit, a full
 * table, no memory for the step's arrays — is refused on every rank with nothing applied anywhere; the rank at fault gets
 * the reason.  (A HIP or transport error in the MIDDLE of a step is not an outcome of the input: the ranks are no longer
 * in step and the communicator is to be destroyed.) */
int32_t rl_sharded_check_requests_device(rl_sharded *s, const rl_hit *d_hits, uint32_t n_hits, const uint32_t *d_req_off,
                                         uint32_t n_req, uint64_t now_us, int32_t load_counters, uint8_t *d_verdict,
                                         int32_t *d_first_limited, uint64_t *d_remaining, uint64_t *d_expires_in_us,
                                         uint32_t *rounds);
/* The stream the verdicts are ordered on (hipStream_t), and a host wait on it. */
void *rl_sharded_stream(rl_sharded *s);
int32_t rl_sharded_sync(rl_sharded *s);
uint32_t rl_sharded_in_flight(const rl_sharded *s);
/* exception barrier self-test of THIS library (rl_engine.h: rl_abi_selftest; kinds 1-4) */
int32_t rl_sharded_abi_selftest(int32_t kind);

/* In-process transport: `world` ranks in one process (one thread per rank; any mix of devices with peer
 * access, or all on one device).  exchange() is a rendezvous: it waits for every rank's send buffers,
 * copies device-to-device and returns when all ranks have read — a correctness transport, not a fast one. */
typedef struct rl_local_group rl_local_group;
int32_t rl_local_group_create(uint32_t world, rl_local_group **out);
void rl_local_group_destroy(rl_local_group *g);
int32_t rl_local_group_transport(rl_local_group *g, uint32_t rank, rl_transport *out);

#ifdef __cplusplus
}
#endif
#endif /* RL_SHARDED_H */
