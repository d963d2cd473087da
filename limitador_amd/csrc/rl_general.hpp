// rl_general.hpp — the general, exact form of check_and_update for a batch: multi-counter requests
// (all-or-nothing across the counters of a request, in_memory.rs:141-153), load_counters (remaining /
// expires_in of every counter, in_memory.rs:87-95,114-116), u64 per-request deltas (in_memory.rs:75), and
// update_counter (in_memory.rs:47-69: the same walk without the limit test).
//
// Sequential definition: request i is decided against the table state left by requests < i.  With A the
// set of admitted requests, hit h of request i on cell c reads
//     v_h = value_at(c, now) + SUM{ delta_j : j < i, j in A, j touches c }        (wrapping u64)
// and passes iff v_h + delta_i <= max_h; request i is admitted iff all its hits pass.  A is the unique
// fixpoint of that map (induction on i), and iterating from "everything admitted" fixes at least one
// more request of the trace prefix per round, so the rounds below end with exactly the sequential
// answer (SURVEY.md §7 hard part 1).
//
// Bucketed form (second generation; the first sorted all hits with a device radix sort, re-read every
// cell from the table in every round and went back to the host after each one):
//
//   k_bkt_hist / scan / scatter   (rl_bucket.hpp) the batch's hits, stably partitioned by key hash; hot keys
//                   in buckets of their own
//   k_gen_sort      per hash bucket (one workgroup, like k_bkt_apply): a two-pass stable counting sort of its hits
//                   BY CELL — afterwards the hits of one cell are one contiguous segment in trace order.  The usual
//                   bucket (<= 512 hits) is read once; its distinct cells are resolved with ONE probe chain each
//                   (SegInfo: the cell's state before the batch), all chains side by side.  Nothing is created yet.
//                   A bucket over GS_LONG_MAX hits or GS_E distinct cells sets `overflow` and promotes its heavy keys.
//   k_gen_admit (+ _fold), k_gen_piece_sum, k_gen_round   one fixpoint round over PIECES of GS_MAX consecutive sorted
//                   positions (any bucket, any cell): "is my request admitted so far" (AND of its hits' pass flags
//                   of the previous round) -> segmented exclusive scan of the admitted deltas, the carry of the cell
//                   that is open at the piece's start folded from the pieces before (k_gen_piece_sum) -> pass flag,
//                   remaining / expires_in per hit.  k_gen_admit also detects the fixpoint: a round whose admitted
//                   set equals the previous one does not run, and everything enqueued behind it returns at once:
//                   the host enqueues a few rounds blind and reads ONE status block.
//   k_gen_final     per request: verdict, first limited counter (in_memory.rs:90-99,141-143), the limit id to
//                   report, and where the walk stopped
//   k_gen_reach     new cells some walk got to (in_memory.rs:109-113,129-133: a request stops at its first limited
//                   counter unless load_counters, so later cells are not created)
//   k_gen_count     how many cells the batch creates — checked against the table BEFORE anything is applied
//   k_gen_commit    per cell: AtomicExpiringValue::update for the admitted hits (atomic_expiring_value.rs:
//                   36-42,87-99), creation of the reached new cells (in_memory.rs:122-127 / :51-62) — only if the
//                   device saw convergence, no error and room (all-or-nothing)
#pragma once
#include "rl_bucket.hpp"

namespace rl {

// hit index -> request index
__global__ __launch_bounds__(256) void k_gen_hit_req(const u32* __restrict__ req_off, u32 n_req,
                                                     u32* __restrict__ hit_req) {
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_req) return;
    const u32 b = req_off[r], e = req_off[r + 1];
    for (u32 q = b; q < e; ++q) hit_req[q] = r;
}

constexpr int GS_BLOCK = 256;
constexpr int GS_WAVES = GS_BLOCK / 64;
constexpr int GS_MAX = 512;           // hits of one piece of k_gen_round (2 per thread: 1024 -> 0.590, 512 -> 0.574, 256 -> 0.599 ms per bench_match call)
constexpr int GS_LONG_MAX = 65535;    // hits of one hash bucket k_gen_sort takes (per-wave counters are 16 bits)
constexpr int GS_E_LOG2 = 11;
constexpr int GS_E = 1 << GS_E_LOG2;  // LDS cells of the in-bucket sort (load <= 1/2)
constexpr int GS_HOT_BLOCKS = 512;    // extra workgroups that walk the hot buckets' chunks
constexpr int GEN_ROUNDS_MAX = 30;    // rounds the status block has a `changed` word for

constexpr u32 SF_EXPIRED0 = 1u;  // the cell was expired before the batch: the first admitted hit reopens the window
constexpr u32 SF_NEW = 2u;       // no cell yet: created at commit if a request's walk reaches it
constexpr u32 SF_ZEROWIN = 4u;   // 0-second window: expired at every read
constexpr u32 SF_BAD = 8u;

struct SHit {  // one hit of the cell-sorted batch
    u32 seg;    // segment = position of the cell's first hit in the sorted batch
    u32 req;    // request (relative to the pass)
    u32 idx;    // hit index in the caller's batch (relative to the pass)
    u32 delta;  // wire delta (the request's u64 delta, if given, is read from req_delta)
};
static_assert(sizeof(SHit) == 16, "one dwordx4 per record");
struct SegInfo {  // a cell's state before the batch
    u64 s;      // value_at(now)
    u64 ttl0;   // ttl(now) before the batch (window for a cell the batch creates)
    u32 slot;   // SLOT_INVALID: no cell yet
    u32 limit;  // limit id | SIMPLE
    u32 flags;  // SF_*
    u32 len;    // hits of the segment
};
static_assert(sizeof(SegInfo) == 32, "SegInfo");
struct SegTot {  // what the admitted hits of the last round add to the cell
    u64 sum;
    u64 last;  // delta of the last admitted hit (the value a 0-second window ends with)
    u32 cnt;
    u32 pad;
};
struct GenStatus {
    u32 err;         // ERRBIT_*
    u32 overflow;    // a hash bucket holds more than GS_MAX hits: the host retries with smaller passes
    u32 n_new;       // cells the batch creates (k_gen_count)
    u32 n_inserted;  // cells k_gen_commit created
    u32 last_round;  // last round that ran (its parity picks the pass-flag buffer)
    u32 last_slot;   // ... and its word of changed[]
    u32 rounds_run;
    u32 hot_n;       // keys that qualified for the next hot set (the host adapts the threshold)
    u32 committed;   // k_gen_commit applied the pass (converged, no error, the new cells fit)
    u32 collide;     // ERRBIT_KEY_COLLISION: ~(the smallest hit index, relative to the pass, whose check word disagrees); 0 = none
    u32 pad2[2];
    // [slot]: that round changed the admitted set — as GEN_CHG_W words, each on a 128-byte line of its own, word =
    // (workgroup of k_gen_admit) % GEN_CHG_W.  ONE word that thousands of workgroups read or write costs ~3 ns apiece in the
    // L2 (k_gen_admit 6 -> 48 us when it was one; 64 words on two lines written with atomicOr by the 3 900 workgroups of a
    // round that changes nearly every request: 10 -> 48 us again, profiles/r06c); 16 lines take 1/16 of the writers each, in
    // parallel, as plain stores of 1 — and the launch that folded per-workgroup flags into one word (k_gen_admit_fold, ~5 us
    // per round) is gone.  Readers OR the words (gen_changed: one per lane).
    u32 changed[GEN_ROUNDS_MAX + 2][16 * 32];
};
constexpr u32 GEN_CHG_W = 16, GEN_CHG_STRIDE = 32;
// Did the round that wrote `slot` change anything?  Wave-wide (every lane of the wave must call it), uniform result.
__device__ __forceinline__ bool gen_changed(const GenStatus* g, u32 slot) {
    const u32 lane = threadIdx.x & 63u;
    return __any((int)(lane < GEN_CHG_W && g->changed[slot][lane * GEN_CHG_STRIDE] != 0u)) != 0;
}

struct GenArgs {
    Cell* table;
    u32 log2cap;
    u64 seed;
    const LimitDev* limits;
    u64 now;
    const Hit* hits;         // the pass's hits (caller's batch + hit0)
    const u32* hit_req;      // absolute request of every hit of the caller's batch (null: hit i is request i)
    const u32* b_req;        // the same, beside the partitioned records: b_req[p] belongs to b_hits[p] (k_bkt_scatter; null: gather)
    const u32* req_off;      // absolute CSR offsets (null: every hit its own request)
    const u64* req_delta;    // per-request u64 deltas of the caller's batch (null: the wire field)
    const u32* hit_check;    // hashed keys (rl_keyhash.h): the check word of every hit of the pass, else null
    PeerTables peers;        // engines that hold peer state (rl_merge_cells): the per-actor tables; has_peers says whether any exists
    u32 has_peers;
    int32_t* msg_status;     // hashed keys: per REQUEST of the caller's batch; k_gen_check_keys stores WIRE_ST_KEY_COLLISION for
                             // every request that carries a colliding hit (the caller takes them all out at once), else null
    u32 hit0, req0;          // the pass starts at this hit / request of the caller's batch
    u32 n_hits, n_req;       // size of the pass
    const BHit* b_hits;
    const uint2* ranges;
    u32 nb;
    const HotParam* hot_param;
    const unsigned short* chunk_tab;
    HotSet* hot_next;
    u32 hot_threshold;
    SHit* s_hits;
    SegInfo* seg_info;
    SegTot* seg_tot;
    SegTot* piece_sum;       // per hot chunk
    u32* req_stop;           // per request: its walk covers the hits [.., req_stop) (mark_reached; k_gen_final -> k_gen_reach)
    uint8_t* reached;        // per segment
    uint8_t* pass[2];
    uint8_t* admitted;       // per request, by the previous round (k_gen_admit)
    const uint8_t* admitted_hit;  // phased form (rl_gen_round_device): admission per HIT, decided by the host; else null
    const uint8_t* reached_hit;   // phased form (rl_gen_count_device): did the request's walk get to this hit; else null
    uint8_t* verdict;        // outputs, already offset to the pass
    int32_t* first_limited;
    int32_t* limited_limit;  // per request: the limit id of the first limited counter, -1 (null: not wanted)
    u64* remaining;
    u64* expires_in;
    GenStatus* gst;
    const Status* pst;       // the partition's status: a refused batch leaves the sorted arrays unwritten
    u32 load, update_mode, mark_reached;
    u32 load_deferred;       // load_counters: k_gen_round does not store remaining / expires_in, k_gen_load does once behind the rounds
    u32 pass_prefilled;      // the round's pass flags start out 1 (round 0: k_gen_piece_sum fills them; later rounds: the round's
                             // k_gen_admit, request by request) and k_gen_round only stores the FAILURES — a flag goes to its
                             // hit's index in request order, a random byte store per hit.  The phased form
                             // (rl_gen_round_device) sets it too: every call is a "round 0" whose k_gen_piece_sum first
                             // overwrites the caller's d_pass with 1s, then k_gen_round stores the failures — so d_pass is
                             // fully defined only THROUGH that prefill, and it must not alias d_admitted
    unsigned long long* trace;  // debugging (RL_GEN_TRACE=2): k_gen_sort's phase stamps, 8 words per workgroup
};

// The (absolute) request of the record at position p of the partitioned batch: beside the record when the partition carried
// it along (b_req), else through the record's index (a random read per hit).
template <class BH>
__device__ __forceinline__ u32 gen_req_of(const GenArgs& A, const BH& h, u32 p) {
    if (A.b_req) return A.b_req[p];
    return A.hit_req ? A.hit_req[A.hit0 + (h.idx_tag & 0xFFFFFFu)] : 0u;
}

__device__ __forceinline__ LimitDev gen_limit_row(const GenArgs& A, u32 limit) {
    const uint4 v = *reinterpret_cast<const uint4*>(&A.limits[limit & ~SIMPLE_FLAG]);
    LimitDev L;
    L.max_value = ((u64)v.y << 32) | v.x;
    L.window_us = ((u64)v.w << 32) | v.z;
    return L;
}

// The state of `key`'s cell before the batch (one probe chain, nothing is created).  (a0, b0) = the two halves
// of the key's HOME cell, loaded by the caller — early, beside whatever else it waits for.
__device__ __forceinline__ SegInfo gen_resolve_from(const GenArgs& A, u64 key, u32 hit_limit, u32 len, uint4 a0, uint4 b0) {
    SegInfo si{};
    si.len = len;
    const u32 mask = (1u << A.log2cap) - 1u;
    u32 slot = slot_of(key, A.seed, A.log2cap);
    si.slot = SLOT_INVALID;
    for (u32 step = 0; step <= mask; ++step) {
        const Cell* c = &A.table[slot];
        const uint4 a = step ? *reinterpret_cast<const uint4*>(c) : a0;
        const u64 tag = ((u64)a.y << 32) | a.x;
        if (tag == key) {
            const uint4 b = step ? reinterpret_cast<const uint4*>(c)[1] : b0;
            const u64 expiry = ((u64)b.y << 32) | b.x;
            si.slot = slot;
            si.limit = b.z;
            if (expiry <= A.now) {
                si.flags |= SF_EXPIRED0;
                // An engine that holds PEER state (rl_merge_cells): the first admitted hit restarts the window the way
                // CrCounterValue::inc_at does (cr_counter_value.rs:53-59) — our own value becomes the increment, the
                // peers' contributions to the window that just ended STAY (`others` is only cleared by a merge's reset,
                // :85-87,144-149) — so from the second hit on the cell reads increment + that stale part.  For an
                // expired cell value_at(now) is 0 by definition, so `s` is free to carry the stale part.
                if (A.has_peers) si.s = peers_window_sum(A.peers, A.seed, key, expiry, 0ull);
            } else {
                si.s = ((u64)a.w << 32) | a.z;  // value_at(now), atomic_expiring_value.rs:19-24
                si.ttl0 = expiry - A.now;       // ttl, atomic_expiring_value.rs:68-74
            }
            break;
        }
        if (tag == TAG_EMPTY) break;
        slot = (slot + 1) & mask;
    }
    if (si.slot == SLOT_INVALID) {
        si.limit = hit_limit;
        si.flags |= SF_NEW;
        if ((hit_limit & SIMPLE_FLAG) && !A.update_mode) {  // in_memory.rs:106-107 (k_bkt_hist checked already)
            atomicOr(&A.gst->err, ERRBIT_MISSING_SIMPLE);
            si.flags |= SF_BAD;
        }
    } else if (si.limit != hit_limit) {
        atomicOr(&A.gst->err, ERRBIT_KEY_LIMIT);
        si.flags |= SF_BAD;
    }
    const LimitDev L = gen_limit_row(A, si.limit);
    if (L.window_us == 0) {
        si.flags |= SF_ZEROWIN;  // every read sees an expired cell: value 0, ttl 0
        si.s = 0;
        si.ttl0 = 0;
    } else if (si.flags & SF_NEW) {
        si.ttl0 = L.window_us;  // AtomicExpiringValue::new(0, now + window), in_memory.rs:123-125
    }
    return si;
}
__device__ __forceinline__ SegInfo gen_resolve(const GenArgs& A, u64 key, u32 hit_limit, u32 len) {
    const uint4* c = reinterpret_cast<const uint4*>(&A.table[slot_of(key, A.seed, A.log2cap)]);
    return gen_resolve_from(A, key, hit_limit, len, c[0], c[1]);
}

// ---------------------------------------------------------------------------------------------
// k_gen_sort
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GS_BLOCK) void k_gen_sort(GenArgs A) {
    __shared__ u64 ekey[GS_E];
    __shared__ u32 eoff[GS_E];                       // hits of the cell, then its start offset in the bucket
    __shared__ unsigned short wcnt[GS_WAVES][GS_E];  // hits of the cell per wave; then where the wave's next hit of the cell goes
    __shared__ unsigned short etot[GS_E];   // hits of the cell (<= GS_LONG_MAX)
    __shared__ unsigned short elist[GS_E];  // the cells in use, in the order they were claimed
    __shared__ uint8_t efold[GS_E];
    __shared__ u32 s_w[GS_WAVES];
    __shared__ u32 s_full, s_nact;
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const u64 lt = (1ull << lane) - 1ull;
    if (A.pst->err) return;  // k_bkt_hist refused the batch
    if (blockIdx.x >= A.nb) {
        // ---- hot buckets: one key each, already in trace order: rewrite the records, chunk by chunk ---
        const u32 n_chunks = A.hot_param[HOT_MAX].chunk0;
        for (u32 c = blockIdx.x - A.nb; c < n_chunks; c += gridDim.x - A.nb) {
            const u32 hb = A.chunk_tab[c];
            const HotParam hp = A.hot_param[hb];
            const u32 first = hp.lo + (c - hp.chunk0) * HOT_CHUNK;
            constexpr int CU = HOT_CHUNK / GS_BLOCK;
            BHit hh[CU];
            u32 rq[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) {  // (unconditional loads from clamped positions: all in flight together)
                const u32 j = first + u * GS_BLOCK + tid;
                hh[u] = load_bhit(A.b_hits, j < hp.hi ? j : hp.hi - 1);
            }
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const u32 j = first + u * GS_BLOCK + tid;
                rq[u] = gen_req_of(A, hh[u], j < hp.hi ? j : hp.hi - 1);
            }
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const u32 j = first + u * GS_BLOCK + tid;
                if (j >= hp.hi) continue;
                const u32 idx = hh[u].idx_tag & 0xFFFFFFu;
                const u32 req = A.hit_req ? rq[u] - A.req0 : idx;
                *reinterpret_cast<uint4*>(A.s_hits + j) = make_uint4(hp.lo, req, idx, hh[u].delta);
                if ((hh[u].idx_tag >> 24) != limit_fold(hp.limit)) atomicOr(&A.gst->err, ERRBIT_KEY_LIMIT);
            }
            if (first == hp.lo && tid == 0) {
                A.seg_info[hp.lo] = gen_resolve(A, A.b_hits[hp.lo].key, hp.limit, hp.hi - hp.lo);
                A.reached[hp.lo] = 0;  // (k_gen_reach sets it; only segment heads are ever read)
            }
        }
        return;
    }
    const uint2 r = A.ranges[blockIdx.x];
    const u32 lo = r.x, L = r.y - r.x;
#define GS_STAMP(k) \
    if (A.trace && tid == 0) A.trace[(size_t)blockIdx.x * 8 + (k)] = wall_clock64();
    if (A.trace && tid == 0) A.trace[(size_t)blockIdx.x * 8 + 7] = L;
    GS_STAMP(0)
    if (L == 0) return;
    // wave w owns the contiguous (trace-ordered) positions [w * Lw, (w + 1) * Lw) of the bucket and walks them
    // in 64-hit steps, twice: count, then place.  A bucket of up to 512 hits — the usual one once the heavy keys
    // have buckets of their own — is read ONCE, two records per lane, and the requests of its records are gathered
    // while pass 1 runs; a longer one (any length up to GS_LONG_MAX) is re-read, records and requests requested
    // a step or two ahead.
    // (A bucket that is too long is only LOOKED at — its first GS_LONG_MAX hits, pass 1 alone: which keys are heavy — so that
    // the retry gives them buckets of their own.  Counting nothing there left the retry with the same bucket.)
    const bool too_long = L > (u32)GS_LONG_MAX;
    const u32 Lc = too_long ? (u32)GS_LONG_MAX : L;
    const u32 steps = (Lc + GS_BLOCK - 1) / GS_BLOCK;
    const u32 Lw = steps * 64;
    const u32 w_lo = w * Lw, w_hi = (w + 1) * Lw < Lc ? (w + 1) * Lw : Lc;
    const bool resident = steps <= 2;
    BHit hA{}, hB{};
    const bool okA = w_lo + lane < w_hi, okB = steps > 1 && w_lo + 64 + lane < w_hi;
    // (Loads that sit in a divergent `if` are waited for where the branch ends — the value has to be merged into
    // its register there — so every prefetch below is UNCONDITIONAL, from a clamped, always valid position.)
    if (resident) {
        hA = load_bhit(A.b_hits, lo + (w_lo + lane < L ? w_lo + lane : L - 1));
        hB = load_bhit(A.b_hits, lo + (w_lo + 64 + lane < L ? w_lo + 64 + lane : L - 1));
    }
    for (u32 e = tid; e < (u32)GS_E; e += GS_BLOCK) {
        ekey[e] = TAG_EMPTY;
        eoff[e] = 0;
#pragma unroll
        for (int ww = 0; ww < GS_WAVES; ++ww) wcnt[ww][e] = 0;
    }
    if (tid == 0) {
        s_full = 0;
        s_nact = 0;
    }
    __syncthreads();
    u32 rA = 0, rB = 0;
    if (resident && A.hit_req) {
        rA = gen_req_of(A, hA, lo + (w_lo + lane < L ? w_lo + lane : L - 1));
        rB = gen_req_of(A, hB, lo + (w_lo + 64 + lane < L ? w_lo + 64 + lane : L - 1));
    }
    // ---- pass 1: the bucket's cells (LDS hash), hits per (wave, cell) ---------------------------------
    auto count_step = [&](const BHit& h, bool ok) {
        u32 ent = 0;
        bool lost = false;
        if (ok) {
            u32 e = (u32)(fmix64(h.key ^ A.seed) >> 20) & (GS_E - 1);
            u32 step = 0;
            for (;; ++step) {
                if (step >= (u32)GS_E) {  // more distinct cells than the LDS hash holds
                    lost = true;
                    break;
                }
                u64 prev = ekey[e];
                if (prev == TAG_EMPTY) {
                    prev = atomicCAS(&ekey[e], TAG_EMPTY, h.key);
                    if (prev == TAG_EMPTY) {
                        efold[e] = (uint8_t)(h.idx_tag >> 24);  // the limit id every hit of the key must carry
                        elist[atomicAdd(&s_nact, 1u)] = (unsigned short)e;
                    }
                }
                if (prev == TAG_EMPTY || prev == h.key) break;
                e = (e + 1) & (GS_E - 1);
            }
            ent = e;
        }
        if (lost) s_full = 1;
        const u64 m = match_digit(ent, GS_E_LOG2, __ballot(ok && !lost));
        if (ok && !lost && (m & lt) == 0ull) wcnt[w][ent] = (unsigned short)(wcnt[w][ent] + (u32)__popcll(m));
    };
    if (resident) {
        count_step(hA, okA);
        if (steps > 1) count_step(hB, okB);
    } else {
        BHit nxt{};
        if (okA) nxt = load_bhit(A.b_hits, lo + w_lo + lane);
        for (u32 u = 0; u < steps; ++u) {
            const u32 p = w_lo + u * 64 + lane;
            const BHit h = nxt;
            if (u + 1 < steps && p + 64 < w_hi) nxt = load_bhit(A.b_hits, lo + p + 64);
            count_step(h, p < w_hi);
        }
    }
    __syncthreads();
    GS_STAMP(1)
    if (too_long || s_full) {
        // Too long (or too many cells) for the in-LDS sort: a traffic shift, the first batch.  Promote the
        // heavy keys seen so far: the host retries the pass with THIS set, which gives them buckets of their own.
        for (u32 e = tid; e < (u32)GS_E; e += GS_BLOCK) {
            u32 acc = 0;
#pragma unroll
            for (int ww = 0; ww < GS_WAVES; ++ww) acc += wcnt[ww][e];
            if (acc >= A.hot_threshold && A.hot_next && ekey[e] != TAG_EMPTY) hot_append(A.hot_next, ekey[e], acc, 1u);
        }
        if (tid == 0) atomicOr(&A.gst->overflow, 1u);
        return;
    }
    // per cell: the waves' exclusive offsets, the total; hot-set promotion like k_bkt_apply's commit
    for (u32 e = tid; e < (u32)GS_E; e += GS_BLOCK) {
        u32 acc = 0;
#pragma unroll
        for (int ww = 0; ww < GS_WAVES; ++ww) {
            const u32 c = wcnt[ww][e];
            wcnt[ww][e] = (unsigned short)acc;  // (acc < L <= GS_LONG_MAX)
            acc += c;
        }
        eoff[e] = acc;
        etot[e] = (unsigned short)acc;
        if (acc >= A.hot_threshold && A.hot_next) hot_append(A.hot_next, ekey[e], acc, 1u);
    }
    __syncthreads();
    {  // exclusive scan of eoff over the GS_E cells (8 per thread)
        constexpr int PER = GS_E / GS_BLOCK;
        u32 v[PER], sum = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            v[q] = eoff[tid * PER + q];
            sum += v[q];
        }
        u32 inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 o = __shfl_up(inc, off);
            if ((int)lane >= off) inc += o;
        }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        u32 base = inc - sum;
        for (u32 ww = 0; ww < w; ++ww) base += s_w[ww];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            eoff[tid * PER + q] = base;
            base += v[q];
        }
    }
    __syncthreads();
    GS_STAMP(2)
    // ---- the cells, first half: every distinct cell's home line is requested now and waited for after pass 2 ---
    const u32 n_act = s_nact;
    constexpr int RES = 2;  // cells per thread resolved with their loads in flight together (more: one by one below)
    u32 re[RES];
    uint4 ra[RES], rb[RES];
    bool rok[RES];
#pragma unroll
    for (int q = 0; q < RES; ++q) {
        const u32 k = tid + q * GS_BLOCK;
        rok[q] = k < n_act;
        re[q] = (u32)elist[rok[q] ? k : 0u];  // (n_act >= 1: the bucket is not empty)
        const uint4* c = reinterpret_cast<const uint4*>(&A.table[slot_of(ekey[re[q]], A.seed, A.log2cap)]);
        ra[q] = c[0];
        rb[q] = c[1];
    }
    // ---- pass 2: placement (the same walk: stable rank inside (wave, cell)) ----------------------------------
    auto place_step = [&](const BHit& h, u32 req_abs, bool ok) {
        u32 ent = 0;
        if (ok) {
            u32 e = (u32)(fmix64(h.key ^ A.seed) >> 20) & (GS_E - 1);
            while (ekey[e] != h.key) e = (e + 1) & (GS_E - 1);
            ent = e;
        }
        const u64 m = match_digit(ent, GS_E_LOG2, __ballot(ok));
        if (ok) {
            const u32 c = wcnt[w][ent];  // where this wave's next hit of the cell goes, inside the segment
            const u32 in_seg = c + (u32)__popcll(m & lt);
            if ((m & lt) == 0ull) wcnt[w][ent] = (unsigned short)(c + (u32)__popcll(m));
            const u32 seg = lo + eoff[ent];
            const u32 idx = h.idx_tag & 0xFFFFFFu;
            const u32 req = A.hit_req ? req_abs - A.req0 : idx;
            *reinterpret_cast<uint4*>(A.s_hits + seg + in_seg) = make_uint4(seg, req, idx, h.delta);
            if ((h.idx_tag >> 24) != (u32)efold[ent]) atomicOr(&A.gst->err, ERRBIT_KEY_LIMIT);
        }
    };
    if (resident) {
        place_step(hA, rA, okA);
        if (steps > 1) place_step(hB, rB, okB);
    } else {
        // two steps ahead: the records; one step ahead: their requests (a gather through the record's index)
        BHit h0{}, h1{}, h2{};
        u32 req0 = 0, req1 = 0;
        if (okA) {
            h0 = load_bhit(A.b_hits, lo + w_lo + lane);
            req0 = gen_req_of(A, h0, lo + w_lo + lane);
        }
        if (okB) h1 = load_bhit(A.b_hits, lo + w_lo + 64 + lane);
        for (u32 u = 0; u < steps; ++u) {
            const u32 p = w_lo + u * 64 + lane;
            if (u + 2 < steps && p + 128 < w_hi) h2 = load_bhit(A.b_hits, lo + p + 128);
            if (u + 1 < steps && p + 64 < w_hi) req1 = gen_req_of(A, h1, lo + p + 64);
            place_step(h0, req0, p < w_hi);
            h0 = h1;
            h1 = h2;
            req0 = req1;
        }
    }
    GS_STAMP(5)
    __syncthreads();  // (the records of this bucket are written: the first of every segment is read back below)
    GS_STAMP(3)
    // ---- the cells, second half: the limit id comes with the segment's first hit -------------------------------
    u32 ridx[RES];
#pragma unroll
    for (int q = 0; q < RES; ++q) ridx[q] = __builtin_nontemporal_load(&A.s_hits[lo + eoff[re[q]]].idx);
    u32 rlim[RES];
#pragma unroll
    for (int q = 0; q < RES; ++q) rlim[q] = A.hits[ridx[q]].limit;
#pragma unroll
    for (int q = 0; q < RES; ++q)
        if (rok[q]) {
            A.seg_info[lo + eoff[re[q]]] = gen_resolve_from(A, ekey[re[q]], rlim[q], etot[re[q]], ra[q], rb[q]);
            A.reached[lo + eoff[re[q]]] = 0;
        }
    for (u32 k = tid + RES * GS_BLOCK; k < n_act; k += GS_BLOCK) {  // (a bucket with more than 512 distinct cells)
        const u32 e = elist[k];
        const u32 seg = lo + eoff[e];
        const u32 idx = __builtin_nontemporal_load(&A.s_hits[seg].idx);
        A.seg_info[seg] = gen_resolve(A, ekey[e], A.hits[idx].limit, etot[e]);
        A.reached[seg] = 0;
    }
    if (A.trace) {
        __syncthreads();
        GS_STAMP(4)
    }
#undef GS_STAMP
}

// Is the request admitted by the previous round?  (round 0: everything is.)
__device__ __forceinline__ bool gen_admitted(const GenArgs& A, const uint8_t* __restrict__ pass_prev, u32 req) {
    return !pass_prev || A.admitted[req] != 0;
}
// The admission byte of a sorted record (req, idx): per request from k_gen_admit, or per hit from the host.
__device__ __forceinline__ uint8_t gen_adm_byte(const GenArgs& A, const uint8_t* __restrict__ pass_prev, u32 req, u32 idx) {
    if (A.admitted_hit) return A.admitted_hit[idx];
    return pass_prev ? A.admitted[req] : (uint8_t)1;
}

// ---------------------------------------------------------------------------------------------
// k_gen_admit: per request, the AND of its hits' pass flags of the previous round (coalesced: a request's
// flags are contiguous) — what k_gen_piece_sum / k_gen_round of this round read once per hit.  It is also where
// the fixpoint is detected: a round whose admitted set equals the one of the round before would reproduce that
// round's flags, so it does not have to run (changed[write_slot] stays 0 and everything enqueued behind returns
// at once).  Round 0 admitted every request.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gen_admit(GenArgs A, u32 round, u32 check_slot, u32 write_slot) {
    if (A.pst->err || A.gst->overflow || round == 0) return;
    if (check_slot && !gen_changed(A.gst, check_slot)) return;  // converged already
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        A.gst->last_slot = write_slot;
        A.gst->last_round = round - 1;  // the flags in force unless this round runs (the status block may be fresh:
                                        // the host clears it between groups of rounds)
    }
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    bool differs = false;
    if (r < A.n_req) {
        const uint8_t* pass_prev = A.pass[(round - 1) & 1u];
        const u32 b = A.req_off ? A.req_off[A.req0 + r] - A.hit0 : r;
        const u32 e = A.req_off ? A.req_off[A.req0 + r + 1] - A.hit0 : r + 1;
        uint8_t adm = 1;
        for (u32 q = b; q < e; ++q)
            if (!pass_prev[q]) {
                adm = 0;
                break;
            }
        const uint8_t before = round == 1 ? (uint8_t)1 : A.admitted[r];
        differs = adm != before;
        A.admitted[r] = adm;
        if (A.pass_prefilled) {  // this round's flags start out "passes" (a request's flags are contiguous)
            uint8_t* pass_next = A.pass[round & 1u];
            for (u32 q = b; q < e; ++q) pass_next[q] = 1;
        }
    }
    // Only a workgroup that HAS a difference says so, into one of GEN_CHG_W words (GenStatus::changed); the status block is
    // zero when the group of rounds starts, so a slot nobody wrote reads "nothing changed".
    const int any = __syncthreads_or(differs ? 1 : 0);
    if (threadIdx.x == 0 && any) A.gst->changed[write_slot][(blockIdx.x % GEN_CHG_W) * GEN_CHG_STRIDE] = 1u;
}

__device__ __forceinline__ u64 gen_delta(const GenArgs& A, const SHit& h) {
    return A.req_delta ? A.req_delta[A.req0 + h.req] : (u64)h.delta;
}

// Scan state of a run of hits that belong to ONE segment: what its admitted hits add.
struct Run {
    u64 sum;
    u64 last;
    u32 cnt;
};
__device__ __forceinline__ Run run_join(const Run& a, const Run& b) {  // a then b
    return Run{a.sum + b.sum, b.cnt ? b.last : a.last, a.cnt + b.cnt};
}

// ---------------------------------------------------------------------------------------------
// k_gen_piece_sum: the sorted batch is cut into PIECES of GS_MAX consecutive positions, whatever bucket or
// segment they fall into.  A segment that crosses a piece boundary needs, in the later piece, what its admitted
// hits in the earlier pieces add: piece_sum[k] = that sum for the segment that is open at the end of piece k
// (only computed when the segment really goes on in piece k + 1).  Most tails are a handful of hits; the
// pieces inside a heavy key's segment are summed whole.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GS_BLOCK) void k_gen_piece_sum(GenArgs A, u32 round, u32 check_slot) {
    __shared__ Run s_run[GS_WAVES];
    if (A.pst->err || A.gst->overflow) return;
    if (check_slot && !gen_changed(A.gst, check_slot)) return;  // converged: the round before changed nothing
    const uint8_t* pass_prev = (round == 0 || A.update_mode) ? nullptr : A.pass[(round - 1) & 1u];
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const u32 k = blockIdx.x;
    const u32 lo = k * (u32)GS_MAX, hi = lo + (u32)GS_MAX;
    if (round == 0 && A.pass_prefilled) {  // round 0's flags start out "passes" (coalesced; k_gen_round stores the failures)
        uint8_t* p0 = A.pass[0];
        for (u32 q = lo + tid; q < hi && q < A.n_hits; q += GS_BLOCK) p0[q] = 1;
    }
    if (hi >= A.n_hits) return;                     // the last piece has no successor
    const u32 t = A.s_hits[hi].seg;                 // the segment the next piece opens with ...
    if (t == hi) return;                            // ... starts there: nothing is carried over
    const u32 first = t > lo ? t : lo;              // its hits inside this piece: [first, hi)
    constexpr int PER = GS_MAX / GS_BLOCK;
    Run acc{0, 0, 0};
    if (lo + (tid + 1) * PER > first) {  // (a short tail: most threads have nothing to add, and wave-uniformly so)
        uint4 v[PER];
        uint8_t ad[PER];
        u64 dl[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) v[u] = *reinterpret_cast<const uint4*>(A.s_hits + lo + tid * PER + u);
#pragma unroll
        for (int u = 0; u < PER; ++u) ad[u] = gen_adm_byte(A, pass_prev, v[u].y, v[u].z);
#pragma unroll
        for (int u = 0; u < PER; ++u) dl[u] = A.req_delta ? A.req_delta[A.req0 + v[u].y] : (u64)v[u].w;
#pragma unroll
        for (int u = 0; u < PER; ++u)  // consecutive hits per thread: `last` is in order
            if (lo + tid * PER + u >= first && ad[u]) acc = run_join(acc, Run{dl[u], dl[u], 1u});
    }
    // in-order reduction over the lanes, then the waves
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        Run o;
        o.sum = __shfl_up(acc.sum, off);
        o.last = __shfl_up(acc.last, off);
        o.cnt = __shfl_up(acc.cnt, off);
        if ((int)lane >= off) acc = run_join(o, acc);
    }
    if (lane == 63) s_run[w] = acc;
    __syncthreads();
    if (tid == 0) {
        Run r = s_run[0];
        for (int ww = 1; ww < GS_WAVES; ++ww) r = run_join(r, s_run[ww]);
        A.piece_sum[k] = SegTot{r.sum, r.last, r.cnt, 0u};
    }
}

// ---------------------------------------------------------------------------------------------
// k_gen_round
// ---------------------------------------------------------------------------------------------
// One piece of the sorted batch: a whole hash bucket (<= GS_MAX hits, any number of segments) or one chunk
// of a hot segment (`carry` = what the earlier chunks' admitted hits add; carry_seg = that segment).
// Every thread owns 4 consecutive hits; runs of equal segment are scanned inside the thread, then across
// the threads with a segmented scan (wave shuffles + a fold over the 4 wave aggregates).
struct GenRoundLds {
    Run wrun[GS_WAVES];   // the run that is open at the end of each wave
    u32 wflag[GS_WAVES];  // a run starts inside the wave (the aggregate does not pass through)
    u32 wlast[GS_WAVES];  // segment of the wave's last hit
    u32 changed;
    u32 head;             // segment of the piece's first record
    Run carry;            // the piece carry, broadcast
    Run out_run;          // the run that is open at the end of the piece ...
    u32 out_seg;          // ... and its segment (the carry of a long bucket's next piece)
};

// LOAD_ONLY: the pass behind the rounds (k_gen_load) — the same scan, but what it stores is `remaining` / `expires_in` of
// every hit and nothing else (no pass flag, no segment total).
template <bool LOAD_ONLY>
__device__ __forceinline__ void gen_round_piece(const GenArgs& A, const uint8_t* __restrict__ pass_prev,
                                                uint8_t* __restrict__ pass_cur, u32 k, u32 lo, u32 n, GenRoundLds& S) {
    constexpr int PER = GS_MAX / GS_BLOCK;
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    SHit h[PER];
    bool ok[PER], adm[PER];
    u64 d[PER];
    Run pre[PER];  // what the admitted hits before this one, in its segment and inside this thread, add
    // Everything a hit needs from memory is requested up front, for all of the thread's hits, from clamped (always
    // valid) positions and with no divergent branch around the loads: a load inside an `if` is waited for where
    // the branch ends, which turns the thread's four hits into four round trips in a row.
    uint4 raw[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const u32 p = tid * PER + i;
        ok[i] = p < n;
        raw[i] = *reinterpret_cast<const uint4*>(A.s_hits + lo + (ok[i] ? p : n - 1));
    }
    const u32 pos0 = lo + tid * PER;
    const uint4 before = *reinterpret_cast<const uint4*>(A.s_hits + (pos0 ? pos0 - 1 : 0u));  // the record ahead of the thread's first
    uint8_t adm_raw[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) adm_raw[i] = gen_adm_byte(A, pass_prev, raw[i].y, raw[i].z);
    u64 d_raw[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) d_raw[i] = A.req_delta ? A.req_delta[A.req0 + raw[i].y] : (u64)raw[i].w;
    uint4 si_a[PER], si_b[PER];  // SegInfo of the hit's segment, as two 16-byte halves
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint4* sp = reinterpret_cast<const uint4*>(&A.seg_info[raw[i].x]);
        si_a[i] = sp[0];
        si_b[i] = sp[1];
    }
    // ---- the piece's carry ------------------------------------------------------------------------------------
    // The piece's first record says whether it opens inside a segment (round 5 read it by itself, ahead of everything: one
    // more round trip in front of every workgroup's chain; now thread 0's own first record is that record, and the pieces'
    // sums are requested while the gathers above are still in flight).  If it does: what the segment's admitted hits in the
    // pieces before add = the in-order fold of those pieces' sums (all of them end inside this segment).
    if (tid == 0) {
        S.head = raw[0].x;
        S.carry = Run{0, 0, 0};
    }
    __syncthreads();
    const u32 head = S.head;  // (uniform)
    u32 carry_seg = 0xFFFFFFFEu;
    if (head < lo) {
        carry_seg = head;
        if (tid < 64) {
            const u32 j0 = head / (u32)GS_MAX;
            const u32 n_before = k - j0;
            const u32 per = (n_before + 63) / 64;
            Run acc{0, 0, 0};
            for (u32 q = lane * per; q < (lane + 1) * per && q < n_before; ++q) {
                const SegTot t = A.piece_sum[j0 + q];
                acc = run_join(acc, Run{t.sum, t.last, t.cnt});
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                Run o;
                o.sum = __shfl_up(acc.sum, off);
                o.last = __shfl_up(acc.last, off);
                o.cnt = __shfl_up(acc.cnt, off);
                if ((int)lane >= off) acc = run_join(o, acc);
            }
            if (lane == 63) S.carry = acc;
        }
        __syncthreads();
    }
    const Run carry = S.carry;
    uint4 lim[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) lim[i] = *reinterpret_cast<const uint4*>(&A.limits[si_b[i].y & ~SIMPLE_FLAG]);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        h[i] = ok[i] ? SHit{raw[i].x, raw[i].y, raw[i].z, raw[i].w} : SHit{0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0};
        adm[i] = ok[i] && adm_raw[i] != 0;
        d[i] = ok[i] ? d_raw[i] : 0ull;
    }
    // ---- inside the thread -------------------------------------------------------------------------
    Run run{0, 0, 0};
    u32 last_seg = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        if (!ok[i]) continue;
        if (i > 0 && h[i].seg != h[i - 1].seg) run = Run{0, 0, 0};
        pre[i] = run;
        if (adm[i]) run = run_join(run, Run{d[i], d[i], 1u});
        last_seg = h[i].seg;
    }
    const u32 first_seg = h[0].seg;  // (0xFFFFFFFF for a thread without hits)
    // ---- across the threads: X(t) = the run open at the end of thread t.  Thread t passes the run of
    //      thread t-1 through iff all its hits belong to the segment thread t-1 ended with. -------------
    u32 prev_last = __shfl_up(last_seg, 1);
    __syncthreads();  // (S is reused by consecutive pieces)
    if (lane == 63) S.wlast[w] = last_seg;
    __syncthreads();
    if (lane == 0) prev_last = w ? S.wlast[w - 1] : carry_seg;
    const bool link = ok[0] && first_seg == prev_last;  // my first hit continues what came before me
    const bool through = !ok[0] || (link && first_seg == last_seg);
    Run X = ok[0] ? run : Run{0, 0, 0};
    u32 F = through ? 0u : 1u;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        Run o;
        o.sum = __shfl_up(X.sum, off);
        o.last = __shfl_up(X.last, off);
        o.cnt = __shfl_up(X.cnt, off);
        const u32 of = __shfl_up(F, off);
        if ((int)lane >= off && !F) {
            X = run_join(o, X);
            F = of;
        }
    }
    if (lane == 63) {
        S.wrun[w] = X;
        S.wflag[w] = F;
    }
    __syncthreads();
    Run C = carry;  // the run open at the end of the previous wave (wave 0: the piece carry)
    for (u32 ww = 0; ww < w; ++ww) C = S.wflag[ww] ? S.wrun[ww] : run_join(C, S.wrun[ww]);
    if (!F) X = run_join(C, X);
    Run cin;
    cin.sum = __shfl_up(X.sum, 1);
    cin.last = __shfl_up(X.last, 1);
    cin.cnt = __shfl_up(X.cnt, 1);
    if (lane == 0) cin = C;
    if (!link) cin = Run{0, 0, 0};
    if (n - 1 >= tid * PER && n - 1 < (tid + 1) * PER) {  // the thread that holds the piece's last hit
        S.out_run = X;
        S.out_seg = last_seg;
    }
    // ---- per hit -------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        if (!ok[i]) continue;
        const u32 pos = lo + tid * PER + i;
        Run pr = h[i].seg == first_seg ? run_join(cin, pre[i]) : pre[i];
        // exclusive BY REQUEST: hits of the same request on the same cell all read the value before any of
        // them is applied (in_memory.rs:105-139 reads, :146-153 updates afterwards)
        u64 dup_sum = 0;
        u32 dup_cnt = 0;
        {
            // the record right ahead is in registers; only a request that really repeats a cell walks back further
            const u32 prev_req = i > 0 ? h[i > 0 ? i - 1 : 0].req : before.y;
            if (adm[i] && pos > h[i].seg && prev_req == h[i].req) {
                for (u32 q = pos; q > h[i].seg;) {
                    --q;
                    const uint4 v = *reinterpret_cast<const uint4*>(A.s_hits + q);
                    if (v.y != h[i].req) break;
                    dup_sum += gen_delta(A, SHit{v.x, v.y, v.z, v.w});
                    dup_cnt += 1;
                }
            }
        }
        SegInfo si;
        si.s = ((u64)si_a[i].y << 32) | si_a[i].x;
        si.ttl0 = ((u64)si_a[i].w << 32) | si_a[i].z;
        si.slot = si_b[i].x;
        si.limit = si_b[i].y;
        si.flags = si_b[i].z;
        si.len = si_b[i].w;
        LimitDev Lm;
        Lm.max_value = ((u64)lim[i].y << 32) | lim[i].x;
        Lm.window_us = ((u64)lim[i].w << 32) | lim[i].z;
        const bool zw = (si.flags & SF_ZEROWIN) != 0;
        // (an expired cell reads 0 until an admitted hit of an EARLIER request has restarted it; from then on `s` — the
        // peers' stale part, 0 in an engine without peers — counts again: see k_gen_sort's resolve)
        const u64 base = (si.flags & SF_EXPIRED0) && (pr.cnt - dup_cnt) == 0u ? 0ull : si.s;
        const u64 v = zw ? 0ull : base + (pr.sum - dup_sum);
        const u64 sum = v + d[i];  // wraps like the reference's release build (in_memory.rs:88)
        const bool pass = A.update_mode ? true : sum <= Lm.max_value;
        if (!LOAD_ONLY && (!A.pass_prefilled || !pass)) pass_cur[h[i].idx] = pass ? 1 : 0;
        if (LOAD_ONLY || (A.load && !A.load_deferred)) {
            A.remaining[h[i].idx] = pass ? Lm.max_value - sum : 0ull;  // checked_sub().unwrap_or_default(), :88-89
            u64 ttl;
            if (zw) ttl = 0;
            else if (!(si.flags & SF_EXPIRED0)) ttl = si.ttl0;                  // alive, or created by this batch
            else ttl = (pr.cnt - dup_cnt) > 0 ? Lm.window_us : 0ull;           // an earlier admitted hit reopened the window
            A.expires_in[h[i].idx] = ttl;
        }
        // the segment's last hit publishes what the admitted hits add to the cell
        if (!LOAD_ONLY && pos + 1 == h[i].seg + si.len) {
            const Run tot = adm[i] ? run_join(pr, Run{d[i], d[i], 1u}) : pr;
            A.seg_tot[h[i].seg] = SegTot{tot.sum, tot.last, tot.cnt, 0u};
        }
    }
}

template <bool LOAD_ONLY>
__device__ __forceinline__ void gen_round_body(const GenArgs& A, const uint8_t* pass_prev, uint8_t* pass_cur, GenRoundLds& S) {
    // One workgroup per piece of GS_MAX positions (k_gen_piece_sum): every piece of every bucket at once — a
    // bucket that skew made long no longer sets the duration of the round.
    const u32 k = blockIdx.x;
    const u32 lo = k * (u32)GS_MAX;
    const u32 n = A.n_hits - lo < (u32)GS_MAX ? A.n_hits - lo : (u32)GS_MAX;
    gen_round_piece<LOAD_ONLY>(A, pass_prev, pass_cur, k, lo, n, S);
}

__global__ __launch_bounds__(GS_BLOCK) void k_gen_round(GenArgs A, u32 round, u32 check_slot, u32 write_slot) {
    __shared__ GenRoundLds S;
    if (A.pst->err || A.gst->overflow) return;
    if (check_slot && !gen_changed(A.gst, check_slot)) return;  // converged: the round before changed nothing
    const uint8_t* pass_prev = (round == 0 || A.update_mode) ? nullptr : A.pass[(round - 1) & 1u];
    uint8_t* pass_cur = A.pass[round & 1u];
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        A.gst->last_round = round;
        atomicAdd(&A.gst->rounds_run, 1u);
        if (check_slot == 0) {  // no k_gen_admit of this group before this round (round 0; a round whose k_gen_admit closed
                                // the group before): the next round always follows
            A.gst->last_slot = write_slot;
            A.gst->changed[write_slot][0] = 1u;
        }
    }
    gen_round_body<false>(A, pass_prev, pass_cur, S);
}

// load_counters, once per group of rounds instead of once per round: `remaining` / `expires_in` of every hit (as read BEFORE
// the update, in_memory.rs:114-116,134-136) are two more random 8-byte stores per hit — the stores that bound a round — and
// only the LAST round's values are ever read.  This kernel repeats the last round's scan (the admitted set it used is still
// in A.admitted — a k_gen_admit that found nothing changed rewrote the same values — and the pieces' carries in piece_sum)
// and stores just those two.  The phased form (rl_gen_round_device) keeps storing them per round: its caller reads them per round.
__global__ __launch_bounds__(GS_BLOCK) void k_gen_load(GenArgs A) {
    __shared__ GenRoundLds S;
    if (A.pst->err || A.gst->overflow) return;
    // (round 0 admitted every request; any later round read A.admitted — gen_adm_byte only asks whether pass_prev is null)
    const uint8_t* pass_prev = (A.gst->last_round == 0 || A.update_mode) ? nullptr : A.pass[0];
    gen_round_body<true>(A, pass_prev, nullptr, S);
}

// ---------------------------------------------------------------------------------------------
// k_gen_final: per request, from the last round's pass flags
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gen_final(GenArgs A) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && A.hot_next) A.gst->hot_n = A.hot_next->n;  // (the host adapts the threshold)
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    if (r >= A.n_req || A.pst->err || A.gst->overflow) return;
    const uint8_t* pass = A.pass[A.gst->last_round & 1u];
    const u32 b = A.req_off ? A.req_off[A.req0 + r] - A.hit0 : r;
    const u32 e = A.req_off ? A.req_off[A.req0 + r + 1] - A.hit0 : r + 1;
    int32_t first = -1;
    for (u32 q = b; q < e; ++q)
        if (!pass[q]) {
            first = (int32_t)q;
            break;
        }
    A.verdict[r] = first < 0 ? 0 : 1;
    if (A.first_limited) A.first_limited[r] = first < 0 ? -1 : (int32_t)(A.hit0 + (u32)first);  // index in the caller's batch
    // the limit whose name the reference reports (Authorization::Limited(name), in_memory.rs:91-93,97-99)
    if (A.limited_limit) A.limited_limit[r] = first < 0 ? -1 : (int32_t)(A.hits[first].limit & ~SIMPLE_FLAG);
    // !load_counters: the walk stops at its first limited counter (in_memory.rs:109-113,129-133); k_gen_reach marks
    // the cells the walks got to
    if (A.mark_reached) A.req_stop[r] = first < 0 ? e : (u32)first + 1;
}

// ---------------------------------------------------------------------------------------------
// k_gen_reach: reached[segment] = some request's walk got to this cell.  Only asked for cells the batch would
// CREATE (a cell a walk never reached is not created, in_memory.rs:109-113,129-133), so only their hits look
// their request up; from the sorted side, coalesced — the other way round (every hit stores its segment, every
// request marks the segments of its hits) costs two random 4-byte accesses per hit, which was what bounded
// k_gen_sort (3.1 M random stores: 85 us).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GS_BLOCK) void k_gen_reach(GenArgs A) {
    if (A.pst->err || A.gst->overflow || !A.mark_reached) return;
    constexpr int PER = GS_MAX / GS_BLOCK;
    const u32 base = blockIdx.x * (u32)GS_MAX + threadIdx.x * PER;
    uint4 v[PER];
    u32 fl[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) v[u] = *reinterpret_cast<const uint4*>(A.s_hits + (base + u < A.n_hits ? base + u : A.n_hits - 1));
#pragma unroll
    for (int u = 0; u < PER; ++u) fl[u] = A.seg_info[v[u].x].flags;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        if (base + u >= A.n_hits || !(fl[u] & SF_NEW)) continue;
        const bool got_here = A.reached_hit ? A.reached_hit[v[u].z] != 0 : v[u].z < A.req_stop[v[u].y];  // (idx < the walk's end)
        if (got_here && !A.reached[v[u].x]) A.reached[v[u].x] = 1;
    }
}

// Is position j the first hit of a segment?  (the sorted batch covers [0, n_hits) without gaps)
__device__ __forceinline__ bool gen_is_head(const GenArgs& A, u32 j) { return A.s_hits[j].seg == j; }

// ---------------------------------------------------------------------------------------------
// k_gen_count: cells the batch creates (new cells some request's walk reached)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gen_count(GenArgs A) {
    __shared__ u32 s_n;
    if (blockIdx.x == 0 && threadIdx.x == 0 && A.hot_next) A.gst->hot_n = A.hot_next->n;
    if (A.pst->err || A.gst->overflow) return;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    u32 mine = 0;
    for (u32 j = blockIdx.x * 256 + threadIdx.x; j < A.n_hits; j += gridDim.x * 256) {
        if (!gen_is_head(A, j)) continue;
        const SegInfo si = A.seg_info[j];
        if ((si.flags & SF_NEW) && !(si.flags & SF_BAD) && (!A.mark_reached || A.reached[j])) ++mine;
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(&s_n, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_n) atomicAdd(&A.gst->n_new, s_n);
}

// Hashed keys (include/rl_keyhash.h): every hit carries the 32-bit check word of its counter's identity.  Two hits of one
// segment (= one 64-bit key) with different words, or a hit whose word differs from the one stored in its key's cell, are
// two counters that share a key: the pass is REFUSED (ERRBIT_KEY_COLLISION, nothing is applied, `collide` names the hit)
// and the caller takes that request out — never a silent merge.  Launched between k_gen_sort and the commit.
// Every hit of a segment is compared with ONE reference — the word stored in the key's cell if the cell exists (and was
// created through hashed keys), else the word of the segment's first hit — so that one pass names EVERY hit that is not
// the counter the key belongs to: the caller takes all their requests out and runs the batch again once (one
// re-run per colliding message, capped at 64, turned 65 crafted messages into a failed batch of 262 144).
constexpr int32_t WIRE_ST_KEY_COLLISION = -103;  // msg_status[]: the request carries a counter whose 64-bit key is another counter's
__global__ __launch_bounds__(256) void k_gen_check_keys(GenArgs A) {
    if (A.pst->err || A.gst->overflow) return;
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j >= A.n_hits) return;
    const SHit h = A.s_hits[j];
    const u32 chk = A.hit_check[h.idx];
    const u32 slot = A.seg_info[h.seg].slot;
    u32 ref = slot == SLOT_INVALID ? 0u : A.table[slot].pad;  // 0: no cell yet, or one not created through hashed keys
    if (ref == 0u) ref = A.hit_check[A.s_hits[h.seg].idx];
    if (ref != chk) {
        atomicOr(&A.gst->err, ERRBIT_KEY_COLLISION);
        atomicMax(&A.gst->collide, ~h.idx);
        if (A.msg_status) A.msg_status[A.hit_req ? A.hit_req[A.hit0 + h.idx] : A.req0 + h.idx] = WIRE_ST_KEY_COLLISION;
    }
}


// ---------------------------------------------------------------------------------------------
// k_gen_commit: per cell
// ---------------------------------------------------------------------------------------------
// Applies the pass only if it is final and fits — decided HERE, on the device, from the status block, so that the
// host does not have to look at it between k_gen_count and this kernel: the fixpoint has converged (the last
// round changed nothing), no error, no overflow, and the table keeps 1/16 of its slots empty with the n_new cells
// the pass creates (`room` = cells that may still be created).  Otherwise nothing is written.
//
// `veto` (the key-sharded step, rl_gen_commit_gated_device): n_veto words `veto_stride` apart, one per rank of the job, each
// rank's own "not from me" (k_gen_veto) gathered on the device — the pass is applied only where EVERY rank's word is zero, so
// all ranks apply or none does without any of their hosts having looked in between.
__global__ __launch_bounds__(256) void k_gen_commit(GenArgs A, u32 room, const u32* __restrict__ veto, u32 n_veto, u32 veto_stride) {
    __shared__ u32 s_n;
    {
        const GenStatus* g = A.gst;
        if (A.pst->err || g->err || g->overflow || g->n_new > room) return;
        if (!A.update_mode && gen_changed(g, g->last_slot)) return;  // not converged yet
        u32 v = 0;
        for (u32 q = 0; q < n_veto; ++q) v |= veto[q * veto_stride];
        if (v) return;
    }
    if (threadIdx.x == 0) {
        s_n = 0;
        if (blockIdx.x == 0) A.gst->committed = 1;
    }
    __syncthreads();
    u32 created = 0;
    for (u32 j = blockIdx.x * 256 + threadIdx.x; j < A.n_hits; j += gridDim.x * 256) {
        if (!gen_is_head(A, j)) continue;
        const SegInfo si = A.seg_info[j];
        if (si.flags & SF_BAD) continue;
        // no request's walk got here: the cell is not even created (k_gen_reach marks NEW cells only; an existing
        // cell no walk reached has nothing admitted on it and is left alone below)
        if (A.mark_reached && (si.flags & SF_NEW) && !A.reached[j]) continue;
        const SegTot t = A.seg_tot[j];
        const LimitDev Lm = gen_limit_row(A, si.limit);
        u32 slot = si.slot;
        if (si.flags & SF_NEW) {
            // first touch creates the cell, admitted or not: AtomicExpiringValue::new(0, now + window)
            // (in_memory.rs:122-127; update_counter: :51-62)
            const u64 k = A.hits[A.s_hits[j].idx].key;
            const u32 mask = (1u << A.log2cap) - 1u;
            slot = slot_of(k, A.seed, A.log2cap);
            bool done = false;
            for (u32 step = 0; step <= mask; ++step) {
                const u64 old = atomicCAS(&A.table[slot].tag, TAG_EMPTY, k);
                if (old == TAG_EMPTY) {
                    done = true;
                    break;
                }
                slot = (slot + 1) & mask;
            }
            if (!done) {  // (cannot happen: the host checked the room for n_new cells)
                atomicOr(&A.gst->err, ERRBIT_TABLE_FULL);
                continue;
            }
            Cell* c = &A.table[slot];
            c->limit = si.limit;
            c->pad = A.hit_check ? A.hit_check[A.s_hits[j].idx] : 0u;  // the key's check word travels with the cell
            c->value = 0;
            c->expiry = A.now + Lm.window_us;
            ++created;
        }
        if (t.cnt == 0) continue;  // nothing admitted on this cell
        Cell* c = &A.table[slot];
        if (si.flags & SF_ZEROWIN) {  // expired at every update: (last delta, now + 0)
            c->value = t.last;
            c->expiry = A.now;
        } else if (si.flags & SF_EXPIRED0) {  // update_if_expired: the first admitted hit stores, the rest add (:36-42,87-99)
            // (+ s: the peers' part of the window that ended, which a local restart keeps — 0 without peers — and their
            // entries move on to the new window, so that merges, exports and sweeps go on finding them)
            if (A.has_peers) (void)peers_window_sum(A.peers, A.seed, c->tag, c->expiry, A.now + Lm.window_us);
            c->value = si.s + t.sum;
            c->expiry = A.now + Lm.window_us;
        } else {
            c->value = si.s + t.sum;  // fetch_add, wrapping
        }
    }
    for (int off = 32; off > 0; off >>= 1) created += __shfl_down(created, off);
    if ((threadIdx.x & 63u) == 0 && created) atomicAdd(&s_n, created);
    __syncthreads();
    if (threadIdx.x == 0 && s_n) atomicAdd(&A.gst->n_inserted, s_n);
}

// Where a pass that starts at hit `base_hit` ends: out[0] = the largest request r with req_off[r] <= base_hit +
// sub_max (the pass covers the requests before r), out[1] = req_off[r].
__global__ void k_gen_cuts(const u32* __restrict__ req_off, u32 n_req, u32 base_hit, u32 sub_max, u32 n_cuts,
                           u32* __restrict__ out) {
    if (blockIdx.x || threadIdx.x || !n_cuts) return;
    const u64 target = (u64)base_hit + sub_max;
    u32 a = 0, b = n_req;  // the largest r in [0, n_req] with req_off[r] <= target (req_off[0] = 0 qualifies)
    while (a < b) {
        const u32 m = (a + b + 1) >> 1;
        if ((u64)req_off[m] <= target) a = m;
        else b = m - 1;
    }
    out[0] = a;
    out[1] = req_off[a];
}

// ---------------------------------------------------------------------------------------------
// k_gen_tiny: the general form for a FEW requests (at most GT_MAX hits — the per-request calls of
// the trait, a request's 1..k counters, small micro-batches): one workgroup, one launch, no sort,
// no fixpoint, no host round trip.  The hits' cells are resolved in parallel (created like
// in_memory.rs:122-127), their state is copied to LDS, then ONE lane replays the requests in index
// order with the reference's own control flow (in_memory.rs:72-156: simple counters first, early
// return at the first limited counter unless load_counters, check everything before updating
// anything), and the touched cells are written back in parallel.  A cell this batch created that no
// request reached before stopping is dropped again (the reference would not have created it).
// ---------------------------------------------------------------------------------------------
constexpr u32 GT_MAX = 256;  // hits (one per thread)
constexpr u32 GT_MAX_REQ = 256;
constexpr u32 GT_ENT = 512;  // LDS cells
constexpr u32 GT_DIRTY = 1u, GT_CREATED = 2u, GT_REACHED = 4u;

// v_readlane / v_writelane with a wave-uniform lane index (device pass only: the host pass sees stubs)
__device__ __forceinline__ u32 gt_readlane(u32 v, u32 lane) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (u32)__builtin_amdgcn_readlane((int)v, (int)lane);
#else
    (void)lane;
    return v;
#endif
}
__device__ __forceinline__ u32 gt_writelane(u32 old, u32 x, u32 lane) {
    // (this compiler has no writelane builtin, and v_writelane_b32 with two SGPR operands needs M0: a compare and a
    // select over the wave do the same in two instructions)
    return (threadIdx.x & 63u) == lane ? x : old;
}

// The kernel's body (all 256 threads; ends with every thread past a barrier).  POST: the completion word goes to
// host_status as k_gen_tiny's protocol wants it; else {error bits, cells dropped, cells created} are left in
// out_counts[3] (workgroup memory) for the caller — k_gen_serve, which answers in its own format.  The output pointers
// may be global or workgroup memory.
template <bool POST>
__device__ __forceinline__ void gen_tiny_body(Cell* __restrict__ table, u32 log2cap, u64 seed, const Hit* __restrict__ hits,
                                              u32 n_hits, const u32* req_off, u32 n_req, const LimitDev* __restrict__ limits,
                                              u32 n_limits, u64 now, int load, uint8_t* verdict, int32_t* first_limited,
                                              u64* remaining, u64* expires_in, const u64* req_delta, Status* host_status,
                                              u32 done_seq, u32* out_counts) {
    __shared__ u64 s_key[GT_ENT], s_value[GT_ENT], s_expiry[GT_ENT];
    __shared__ u32 s_slot[GT_ENT], s_limit[GT_ENT], s_flags[GT_ENT];
    __shared__ u64 h_max[GT_MAX], h_win[GT_MAX];
    __shared__ u64 h_delta[GT_MAX];  // the REQUEST's delta (in_memory.rs:75 `delta: u64`), req_delta[] or the wire field
    __shared__ u32 h_lim[GT_MAX];
    __shared__ unsigned short h_ent[GT_MAX];
    __shared__ unsigned short s_owner[GT_ENT];  // the thread that claimed the LDS cell (the wave-resident replay below)
    __shared__ u32 s_req_off[GT_MAX_REQ + 1];  // (req_off may live in host-mapped memory: read it once, in parallel)
    __shared__ u32 s_err, s_created, s_dropped;
    __shared__ Status s_st;  // probe_from reports into a Status block
    Status* st = &s_st;
    const u32 tid = threadIdx.x;
    if (tid == 0) s_st.err = 0;
    for (u32 e = tid; e < GT_ENT; e += 256) {
        s_key[e] = TAG_EMPTY;
        s_flags[e] = 0;
    }
    if (tid == 0) s_err = s_created = s_dropped = 0;
    for (u32 r = tid; r <= n_req; r += 256) s_req_off[r] = req_off ? req_off[r] : r;
    __syncthreads();
    if (req_delta)  // one u64 delta per request, spread over its hits
        for (u32 r = tid; r < n_req; r += 256)
            for (u32 j = s_req_off[r]; j < s_req_off[r + 1] && j < GT_MAX; ++j) h_delta[j] = req_delta[r];
    __syncthreads();
    // ---- 1a: validate, find or create the cell, claim the key's LDS cell ---------------------------
    bool claimer = false;
    u32 my_slot = SLOT_INVALID, my_ent = 0;
    Hit h{};
    if (tid < n_hits) {
        h = load_hit(hits, tid);
        u32 err = 0;
        if ((h.limit & ~SIMPLE_FLAG) >= n_limits) err = ERRBIT_BAD_LIMIT;
        else if (h.key >= TAG_TOMB) err = ERRBIT_RESERVED_KEY;
        else {
            const LimitDev L = limits[h.limit & ~SIMPLE_FLAG];
            h_max[tid] = L.max_value;
            h_win[tid] = L.window_us;
            if (!req_delta) h_delta[tid] = h.delta;
            h_lim[tid] = h.limit;
            u32 created = 0;
            const u32 s0 = slot_of(h.key, seed, log2cap);
            my_slot = probe_from<PM_CHECK>(table, log2cap, s0, table[s0].tag, h.key, h.limit, limits, now, st, created);
            if (my_slot == SLOT_INVALID) {
                err = (h.limit & SIMPLE_FLAG) ? ERRBIT_MISSING_SIMPLE : ERRBIT_TABLE_FULL;
            } else {
                u32 e = (u32)(fmix64(h.key ^ seed) >> 20) & (GT_ENT - 1);
                for (;;) {
                    const u64 prev = atomicCAS(&s_key[e], TAG_EMPTY, h.key);
                    if (prev == TAG_EMPTY) {
                        claimer = true;
                        break;
                    }
                    if (prev == h.key) break;
                    e = (e + 1) & (GT_ENT - 1);
                }
                my_ent = e;
                h_ent[tid] = (unsigned short)e;
                if (created) {
                    atomicOr(&s_flags[e], GT_CREATED);
                    atomicAdd(&s_created, 1u);
                }
            }
        }
        if (err) atomicOr(&s_err, err);
    }
    __syncthreads();  // every created cell is complete (workgroup-scope fence) before anyone copies it
    // ---- 1b: the claimer copies the cell to LDS -----------------------------------------------------
    if (claimer) {
        const Cell* c = &table[my_slot];
        s_value[my_ent] = c->value;
        s_expiry[my_ent] = c->expiry;
        s_limit[my_ent] = c->limit;
        s_slot[my_ent] = my_slot;
        s_owner[my_ent] = (unsigned short)tid;
    }
    __syncthreads();
    if (tid < n_hits && my_slot != SLOT_INVALID && s_limit[my_ent] != h.limit) atomicOr(&s_err, ERRBIT_KEY_LIMIT);
    __syncthreads();
    const u32 err_all = s_err;
    // ---- 2: the requests are replayed in index order, in_memory.rs:72-156 ----------------------------------
    // Up to 64 hits and 64 requests — every per-request call, the usual micro-batch: WAVE-RESIDENT.  Hit j lives in lane
    // j's registers (limit, delta, window), a cell's state (value, expiry, flags) in the registers of the lane that
    // claimed it, and the walk is scalar code over v_readlane / v_writelane: a step costs a handful of cycles where the
    // one-lane loop below pays an LDS round trip per field (0.7 us per hit measured: 21 requests x 3 counters took 69 us).
    // Results are written to lane r (verdict, first_limited) / lane j (remaining, expires_in) and stored by all lanes at once.
    const bool wave_replay = n_hits <= 64u && n_req <= 64u;
    if (wave_replay && tid < 64u && !err_all) {
        const u32 lane = tid;
        const bool has = lane < n_hits && my_slot != SLOT_INVALID;
        u64 r_max = has ? h_max[lane] : 0ull, r_win = has ? h_win[lane] : 0ull, r_delta = has ? h_delta[lane] : 0ull;
        u32 r_lim = has ? h_lim[lane] : 0u;
        u32 r_own = has ? (u32)s_owner[my_ent] : 0u;                     // the lane that holds my cell's state
        u64 c_val = claimer ? s_value[my_ent] : 0ull, c_exp = claimer ? s_expiry[my_ent] : 0ull;
        u32 c_flg = 0;
        u64 o_rem = 0, o_exp = 0;  // lane j: what hit j loaded
        u32 o_res = 0;             // lane r: verdict | (first + 1) << 1
        // Requests that list their simple counters first (the mirror and the matcher do) need ONE walk in index order, not
        // two filtered ones: no hit of a request may be a simple counter right behind a qualified one of the same request.
        u32 my_req = 0;
        if (lane < n_hits) {
            u32 a = 0, bnd = n_req;  // the request r with s_req_off[r] <= lane < s_req_off[r + 1]
            while (bnd - a > 1) {
                const u32 m = (a + bnd) >> 1;
                if (s_req_off[m] <= lane) a = m;
                else bnd = m;
            }
            my_req = a;
        }
        const u32 prev_lim = (u32)__shfl_up((int)r_lim, 1), prev_req = (u32)__shfl_up((int)my_req, 1);
        const bool bad_pair = lane > 0 && lane < n_hits && prev_req == my_req && (prev_lim & SIMPLE_FLAG) == 0u &&
                              (r_lim & SIMPLE_FLAG) != 0u;
        const int n_pass = __ballot(bad_pair) == 0ull ? 1 : 2;
#define RL_RD32(v, l) gt_readlane((u32)(v), (u32)(l))
#define RL_RD64(v, l) (((u64)RL_RD32((u32)((v) >> 32), l) << 32) | (u64)RL_RD32((u32)(v), l))
#define RL_WR32(v, x, l) v = gt_writelane((v), (u32)(x), (u32)(l))
#define RL_WR64(v, x, l)                                                                                    \
    do {                                                                                                    \
        u32 lo_ = (u32)(v), hi_ = (u32)((v) >> 32);                                                         \
        RL_WR32(lo_, (u32)(x), l);                                                                          \
        RL_WR32(hi_, (u32)((u64)(x) >> 32), l);                                                             \
        v = ((u64)hi_ << 32) | lo_;                                                                         \
    } while (0)
        for (u32 r = 0; r < n_req; ++r) {
            const u32 b = s_req_off[r], e_ = s_req_off[r + 1];
            int32_t first = -1;
            bool stopped = false;
            for (int pass = 0; pass < n_pass && !stopped; ++pass) {  // simple counters (:105-118), then qualified (:121-139)
                for (u32 j = b; j < e_; ++j) {
                    if (n_pass == 2 && ((RL_RD32(r_lim, j) & SIMPLE_FLAG) == 0u) != (pass == 1)) continue;
                    const u32 c = RL_RD32(r_own, j);
                    RL_WR32(c_flg, RL_RD32(c_flg, c) | GT_REACHED, c);
                    const u64 exp = RL_RD64(c_exp, c), val = RL_RD64(c_val, c), mx = RL_RD64(r_max, j);
                    const u64 value = exp <= now ? 0ull : val;       // value_at(now)
                    const u64 sum = value + RL_RD64(r_delta, j);     // wraps like the release build
                    const bool within = sum <= mx;
                    if (load) {
                        RL_WR64(o_rem, within ? mx - sum : 0ull, j);          // checked_sub().unwrap_or_default(), :88-89
                        if (first < 0 && !within) first = (int32_t)j;          // :90-94
                        RL_WR64(o_exp, exp > now ? exp - now : 0ull, j);      // ttl, :114-116,134-136
                    } else if (!within) {  // :109-113, :129-133: return at once, nothing is updated
                        first = (int32_t)j;
                        stopped = true;
                        break;
                    }
                }
            }
            if (first < 0) {  // :146-153: update every counter, simple ones first
                for (int pass = 0; pass < n_pass; ++pass)
                    for (u32 j = b; j < e_; ++j) {
                        if (n_pass == 2 && ((RL_RD32(r_lim, j) & SIMPLE_FLAG) == 0u) != (pass == 1)) continue;
                        const u32 c = RL_RD32(r_own, j);
                        const u64 exp = RL_RD64(c_exp, c), d = RL_RD64(r_delta, j);
                        if (exp <= now) {  // atomic_expiring_value.rs:36-42,87-99
                            RL_WR64(c_exp, now + RL_RD64(r_win, j), c);
                            RL_WR64(c_val, d, c);
                        } else {
                            RL_WR64(c_val, RL_RD64(c_val, c) + d, c);
                        }
                        RL_WR32(c_flg, RL_RD32(c_flg, c) | GT_DIRTY, c);
                    }
            }
            RL_WR32(o_res, (first < 0 ? 0u : 1u) | ((u32)(first + 1) << 1), r);
        }
#undef RL_RD32
#undef RL_RD64
#undef RL_WR32
#undef RL_WR64
        if (lane < n_req) {
            verdict[lane] = (uint8_t)(o_res & 1u);
            if (first_limited) first_limited[lane] = (int32_t)(o_res >> 1) - 1;
        }
        if (load && lane < n_hits) {
            remaining[lane] = o_rem;
            expires_in[lane] = o_exp;
        }
        if (claimer) {  // the cells' state back to the LDS cells: phase 3 writes it out
            s_value[my_ent] = c_val;
            s_expiry[my_ent] = c_exp;
            atomicOr(&s_flags[my_ent], c_flg);
        }
    }
    if (!wave_replay && tid == 0 && !err_all) {
        for (u32 r = 0; r < n_req; ++r) {
            const u32 b = s_req_off[r], e_ = s_req_off[r + 1];
            int32_t first = -1;
            bool stopped = false;
            for (int pass = 0; pass < 2 && !stopped; ++pass) {  // simple counters (:105-118), then qualified (:121-139)
                for (u32 j = b; j < e_; ++j) {
                    if (((h_lim[j] & SIMPLE_FLAG) == 0u) != (pass == 1)) continue;
                    const u32 e = h_ent[j];
                    s_flags[e] |= GT_REACHED;
                    const u64 value = s_expiry[e] <= now ? 0ull : s_value[e];  // value_at(now)
                    const u64 sum = value + h_delta[j];                          // wraps like the release build
                    const bool within = sum <= h_max[j];
                    if (load) {
                        remaining[j] = within ? h_max[j] - sum : 0ull;  // checked_sub().unwrap_or_default(), :88-89
                        if (first < 0 && !within) first = (int32_t)j;   // :90-94
                        expires_in[j] = s_expiry[e] > now ? s_expiry[e] - now : 0ull;  // ttl, :114-116,134-136
                    } else if (!within) {  // :109-113, :129-133: return at once, nothing is updated
                        first = (int32_t)j;
                        stopped = true;
                        break;
                    }
                }
            }
            if (first < 0) {  // :146-153: update every counter, simple ones first
                for (int pass = 0; pass < 2; ++pass)
                    for (u32 j = b; j < e_; ++j) {
                        if (((h_lim[j] & SIMPLE_FLAG) == 0u) != (pass == 1)) continue;
                        const u32 e = h_ent[j];
                        if (s_expiry[e] <= now) {  // atomic_expiring_value.rs:36-42,87-99
                            s_expiry[e] = now + h_win[j];
                            s_value[e] = h_delta[j];
                        } else {
                            s_value[e] += h_delta[j];
                        }
                        s_flags[e] |= GT_DIRTY;
                    }
            }
            verdict[r] = first < 0 ? 0 : 1;
            if (first_limited) first_limited[r] = first;
        }
    }
    __syncthreads();
    // ---- 3: write back, drop what no request reached --------------------------------------------------
    if (!err_all) {
        for (u32 e = tid; e < GT_ENT; e += 256) {
            if (s_key[e] == TAG_EMPTY) continue;
            const u32 f = s_flags[e];
            Cell* c = &table[s_slot[e]];
            if (f & GT_DIRTY) {
                c->value = s_value[e];
                c->expiry = s_expiry[e];
            }
            if (!load && (f & GT_CREATED) && !(f & GT_REACHED)) {
                c->tag = TAG_TOMB;
                atomicAdd(&s_dropped, 1u);
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (POST) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // completion word written last, with the counts, as ONE 16-byte store (see apply_finish)
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(u32x4{s_err, s_dropped, s_created, done_seq},
                                        reinterpret_cast<u32x4*>(host_status));
        } else {
            out_counts[0] = s_err;
            out_counts[1] = s_dropped;
            out_counts[2] = s_created;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_gen_tiny(Cell* __restrict__ table, u32 log2cap, u64 seed,
                                                  const Hit* __restrict__ hits, u32 n_hits,
                                                  const u32* __restrict__ req_off, u32 n_req,
                                                  const LimitDev* __restrict__ limits, u32 n_limits, u64 now, int load,
                                                  uint8_t* __restrict__ verdict, int32_t* __restrict__ first_limited,
                                                  u64* __restrict__ remaining, u64* __restrict__ expires_in,
                                                  const u64* __restrict__ req_delta, Status* host_status, u32 done_seq) {
    gen_tiny_body<true>(table, log2cap, seed, hits, n_hits, req_off, n_req, limits, n_limits, now, load, verdict, first_limited,
                        remaining, expires_in, req_delta, host_status, done_seq, nullptr);
}

// ---------------------------------------------------------------------------------------------
// k_gen_serve: per-request calls — and micro-batches of up to SRV_MAX_HITS hits / requests — WITHOUT a launch per call.
// One request with a few counters (the trait's check_and_update called request by request: BASELINE.json configs[0]) is
// launch-latency-bound as a kernel of its own:
// ~10 us from the enqueue to the first wave, ~10 us for the stream to be seen idle again — 47 us per call for 10 us of
// work.  This kernel stays: one workgroup polls a MAILBOX in host-mapped memory, serves request after request with
// k_gen_tiny's body (same semantics, same code), and leaves by itself `linger` after the last one — so that nothing
// ever waits for it longer than that (a device-wide synchronise, the next kernel on the stream) — or at once when
// the host says so (any other engine call does, before it touches the device).
//   host -> device   the hits (the engine's host-mapped staging), now, the request's u64 delta; then cmd = {seq, hits,
//                    flags, 0}, seq LAST: a command is the one the server expects or it is not there yet
//   device -> host   done = {error bits | verdict << 8 | (first_limited + 1) << 16, cells dropped, cells created, seq},
//                    seq in a system-scope RELEASE store behind the rest (the kernel does not end behind a request:
//                    only write-through stores ever leave the L2); with load_counters two 16-byte slots per hit in front
//                    of it, each tagged with seq and its index
//   leaving          gone = {0, 0, 0, the seq it was waiting for}: the host then knows that command was NOT taken (it
//                    launches a new server for it) — the decision is one read of cmd, so "taken" and "gone" exclude
//                    each other.
// ---------------------------------------------------------------------------------------------
constexpr u32 SRV_MAX_HITS = 64;  // hits and requests of one command (the wave-resident replay's reach)
constexpr u32 SRV_LOAD = 1u, SRV_DELTA = 2u, SRV_QUIT = 4u;
struct ServeBox {
    u32 cmd[4];
    u64 now, delta;
    u32 pad0[8];
    u32 gone[4];
    u32 pad1[12];
    u32 slot[2 * SRV_MAX_HITS][4];  // {lo, hi, seq, index}: remaining of hit j at [2 j], expires_in at [2 j + 1]
    u32 rslot[SRV_MAX_HITS][2];     // commands of several requests: {verdict | (first_limited + 1) << 1, seq} of request r
};
static_assert(sizeof(ServeBox) == 128 + 40 * SRV_MAX_HITS, "ServeBox layout");

// What the lingering server tells the host travels as SELF-VALIDATING 16-byte units: value and tag {sequence number, index}
// leave in ONE store (system scope, written through), so the host can never see a tag whose value is not there yet — two
// 8-byte stores, even with the second a release, are two arrivals at the host (the withdrawn synchronise-free
// tiny path: acknowledged stores are not ordered arrivals), and a release per request is an L2 write-back per request.
__device__ __forceinline__ void srv_store16(void* dst, u32 a, u32 b, u32 c, u32 d) {
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v{a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ void srv_post(u64* dst, u64 first, u64 second) {
    srv_store16(dst, (u32)first, (u32)(first >> 32), (u32)second, (u32)(second >> 32));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(256) void k_gen_serve(Cell* __restrict__ table, u32 log2cap, u64 seed,
                                                   const LimitDev* __restrict__ limits, u32 n_limits,
                                                   const Hit* __restrict__ hits, const u32* __restrict__ req_off_many,
                                                   const u64* __restrict__ req_delta_many, ServeBox* box,
                                                   Status* host_status, u32 first_seq, u32 linger_ticks) {
    __shared__ u32 s_c[4];
    __shared__ u64 s_now, s_delta;
    __shared__ u32 s_off[2], s_counts[3];
    __shared__ uint8_t o_verdict[SRV_MAX_HITS];
    __shared__ int32_t o_first[SRV_MAX_HITS];
    __shared__ u64 o_rem[SRV_MAX_HITS], o_exp[SRV_MAX_HITS];
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    const u32 tid = threadIdx.x;
    u32 expect = first_seq;
    for (;;) {
        if (tid == 0) {
            const u64 t0 = wall_clock64();
            bool leave = false;
            for (;;) {
                if (__hip_atomic_load(&box->cmd[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == expect) break;
                if (wall_clock64() - t0 > (u64)linger_ticks) {
                    leave = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            u32 nh = 0, flags = 0, nr = 0;
            if (!leave) {
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                nh = __hip_atomic_load(&box->cmd[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                flags = __hip_atomic_load(&box->cmd[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                nr = __hip_atomic_load(&box->cmd[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                s_now = __hip_atomic_load(&box->now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                s_delta = __hip_atomic_load(&box->delta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((flags & SRV_QUIT) || nh == 0 || nh > SRV_MAX_HITS || nr == 0 || nr > SRV_MAX_HITS) leave = true;
            }
            s_c[0] = leave ? 1u : 0u;
            s_c[1] = nh;
            s_c[2] = flags;
            s_c[3] = nr;
            s_off[0] = 0;
            s_off[1] = nh;
        }
        __syncthreads();
        if (s_c[0]) {
            if (tid == 0) srv_post(reinterpret_cast<u64*>(box->gone), 0ull, (u64)expect << 32);
            return;
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);  // (the staging area was rewritten by the host: no line of it from the last request)
        const u32 n_hits = s_c[1], flags = s_c[2], n_req = s_c[3];
        const int load = (flags & SRV_LOAD) ? 1 : 0;
        // one request: its offsets and its delta came with the command; several: the CSR offsets / the deltas are in the staging
        const bool many = n_req > 1u;
        gen_tiny_body<false>(table, log2cap, seed, hits, n_hits, many ? req_off_many : s_off, n_req, limits, n_limits, s_now, load,
                             o_verdict, o_first, o_rem, o_exp,
                             (flags & SRV_DELTA) ? (many ? req_delta_many : &s_delta) : nullptr, nullptr, 0u, s_counts);
        if (load && tid < n_hits && !s_counts[0]) {
            // written THROUGH to the host (system scope) and acknowledged before the barrier: the kernel does not end
            // behind a request, so nothing else would ever push these lines out of the L2
            const u64 rem = o_rem[tid], exp = o_exp[tid];
            srv_store16(box->slot[2 * tid], (u32)rem, (u32)(rem >> 32), expect, 2u * tid);
            srv_store16(box->slot[2 * tid + 1], (u32)exp, (u32)(exp >> 32), expect, 2u * tid + 1u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (many && tid < n_req && !s_counts[0]) {
            __hip_atomic_store(reinterpret_cast<u64*>(box->rslot[tid]),
                               (u64)((u32)o_verdict[tid] | ((u32)(o_first[tid] + 1) << 1)) | ((u64)expect << 32), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (tid == 0) {
            const u32 w0 = s_counts[0] ? s_counts[0] : (((u32)o_verdict[0] << 8) | ((u32)(o_first[0] + 1) << 16));
            srv_post(reinterpret_cast<u64*>(host_status), (u64)w0 | ((u64)s_counts[1] << 32), (u64)s_counts[2] | ((u64)expect << 32));
        }
        expect = expect + 1u ? expect + 1u : 1u;
        __syncthreads();
    }
}

// This rank's word of the key-sharded step's all-or-nothing commit (k_gen_commit's `veto`): 1 = the cells the pass creates do
// not fit, 2 = an error bit of the pass or of its partition, 4 = hash buckets overflowed (the pass is to be begun again),
// 8 = `also` is non-zero (the caller's own reason: its last round still changed the admitted set).  gst null: a pass of no hits.
__global__ void k_gen_veto(const GenStatus* gst, const Status* pst, u32 room, const u32* also, u32* out) {
    if (threadIdx.x || blockIdx.x) return;
    u32 w = 0;
    if (gst) {
        if (pst->err | gst->err) w |= 2u;
        if (gst->overflow) w |= 4u;
        if (gst->n_new > room) w |= 1u;
    }
    if (also && *also) w |= 8u;
    *out = w;
}

// End of a pass: what the host decides on — error bits of the pass and of its partition, cells created, overflow /
// committed / "the last round still changed something", rounds run, size of the next hot set — as ONE 16-byte store
// into host-mapped memory, sequence number last.  Launched behind k_gen_commit: everything it reads is final.
__global__ __launch_bounds__(256) void k_gen_post(GenStatus* gst, const Status* __restrict__ pst, u32* host_word, u32 seq,
                                                  u32* __restrict__ scratch_words, u32 n_scratch_words) {
    __shared__ u32 s_clean;
    const u32 changed = gen_changed(gst, gst->last_slot) ? 1u : 0u;
    if (threadIdx.x == 0) {
        const u32 err = gst->err | pst->err;
        const u32 hot = gst->hot_n > 0xFFFEu ? 0xFFFFu : gst->hot_n;
        const u32 rr = gst->rounds_run > 0xFFu ? 0xFFu : gst->rounds_run;
        const u32 flags = (gst->overflow ? 1u : 0u) | (gst->committed ? 2u : 0u) | (changed << 2) | (rr << 8) | (hot << 16);
        // a pass that was applied leaves nothing the host still wants from the device: the status block and the rotating
        // scratch blocks are zeroed HERE for whatever batch comes next (two fill commands the host used to enqueue once it
        // had seen the word: 15-25 us between two calls); every other outcome keeps them for the host to look at
        s_clean = (!err && !gst->overflow && gst->committed) ? 1u : 0u;
        typedef u32 u32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(u32x4{err, gst->n_inserted, flags, seq}, reinterpret_cast<u32x4*>(host_word));
    }
    __syncthreads();
    if (!s_clean) return;
    for (u32 q = threadIdx.x; q < n_scratch_words; q += 256) scratch_words[q] = 0u;
    static_assert(sizeof(GenStatus) % 16 == 0, "GenStatus is zeroed 16 bytes at a time");
    uint4* g = reinterpret_cast<uint4*>(gst);
    for (u32 q = threadIdx.x; q < (u32)(sizeof(GenStatus) / 16); q += 256) g[q] = make_uint4(0u, 0u, 0u, 0u);
}

}  // namespace rl
