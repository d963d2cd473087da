// rl_bucket.hpp — the measured hot path: CounterStorage::check_and_update for a batch of
// single-counter requests as ONE read and ONE write per touched counter cell, no global atomics
// on the data path and no mid-batch host round trip.
//
// Sequential contract (reference limitador/src/storage/in_memory.rs:72-156, called once per
// request): hit i is decided against the cell state left by hits < i.  Hits on DIFFERENT cells
// never interact, so the batch is split by key hash into buckets; every hit of one key lands in
// the same bucket, a bucket is owned by exactly one workgroup, and inside the bucket the hits
// keep their trace order (the partition is stable).  The owner replays its bucket with the
// reference's arithmetic, but wavefront-parallel: per round of APPLY_R hits it aggregates the
// hits per key in an LDS hash (ds atomics), reads each NEW key's cell once from HBM, decides
//     run + sum(round) <= max    -> every hit of the key admitted, whatever the order
//     run + delta      >  max    -> this hit denied, whatever the order (run only grows)
//     otherwise (uniform delta)  -> admitted iff trace-order rank < (max - run) / delta
//     anything else              -> replayed hit by hit in trace order (wrapping u64, 0-s windows)
// (`run` = value_at(now) + what this batch has admitted so far, atomic_expiring_value.rs:19-24,
// 36-42), keeps the cell in LDS across rounds, and writes value (and expiry, when the window was
// reset: update_if_expired, atomic_expiring_value.rs:87-99) back once.
//
//   k_bkt_hist     per tile of the batch: hits per bucket (LDS histogram); validates the hits
//   k_bkt_scan     per bucket: exclusive scan of its tile counts; bucket totals
//   k_bkt_scatter  stable partition into bucket order (16-B records); one extra workgroup prepares
//                  the hot keys' buckets (HotParam)
//   k_hot_state    the hot keys' cells before the batch (one small workgroup, on the apply stream)
//   k_bkt_apply    (rl_apply.hpp) one workgroup per hash bucket: the replay described above; then the
//                  hot buckets, from positions
//
// Hot keys (a Zipf head, a simple limit every request of a namespace hits) would make one bucket
// — one workgroup — the critical path.  A key that absorbed HOT_PROMOTE hits in one batch gets a
// bucket of its own in the next: stable partition + one key per bucket means a hit's position in
// the bucket IS its trace-order rank on the key, so with a single delta value the verdict is
// rank < (max - value) / delta for any number of workgroups in parallel (apply_hot).
#pragma once
#include "rl_kernels.hpp"

namespace rl {

constexpr int BK_LOG2_MAX = 11;
constexpr int BK_MAX = 1 << BK_LOG2_MAX;
constexpr int HOT_MAX = 512;                  // keys that get a bucket of their own
constexpr int BKT_MAX = BK_MAX + HOT_MAX;     // hash buckets + hot-key buckets
constexpr int HOT_COLS = 3 * HOT_MAX;         // per tile and hot key: largest delta, largest ~delta, a limit id
constexpr int ROW_MAX = BKT_MAX + HOT_COLS;   // columns of one tile's row of the histogram matrix
constexpr int PT_BLOCK = 1024;                // 16 waves per workgroup
constexpr int PT_WAVES = PT_BLOCK / 64;
constexpr int PT_STEPS = 4;                   // 64-hit steps per wave (large batches)
constexpr int PT_TILE = PT_BLOCK * PT_STEPS;  // hits per workgroup (4096)
constexpr int PT_TILE_SMALL = PT_BLOCK;       // small batches: one step per wave, four times the workgroups
constexpr u32 PT_SMALL_MAX_TILES = 256;       // ... as long as that is at most one workgroup per CU
constexpr int HOT_CHUNK = 1024;              // hits per work item of a hot bucket
constexpr u32 HOT_PROMOTE = 160;              // hits in one batch that make (or keep) a key hot: the floor of the
                                              // threshold, which the host doubles while more keys qualify than fit
constexpr u32 HOT_LONG_BUCKET = 1024;         // a bucket this long is long BECAUSE of a key: promote it earlier
constexpr int HOT_HASH = 1024;                // LDS lookup table of the current hot set

// Keys that took a large share of the PREVIOUS batch ("hot": a Zipf head, a simple limit every
// request of a namespace hits).  Each gets a bucket to itself, so the stable partition leaves the
// key's hits contiguous and in trace order: position in the bucket == the hit's rank on the key.
// Any stale or arbitrary set is valid — the set only has to be the same in every kernel of one
// batch; it decides which code path a key takes, never a verdict.  k_bkt_apply builds the next
// batch's set from exact counts: hot keys that still got `hot_threshold` hits, plus every key of a hash
// bucket that absorbed as many (a traffic shift costs one batch of long buckets).  The host keeps
// the threshold at HOT_PROMOTE and doubles it while more keys qualify than there are hot buckets, so
// that under pressure the hottest keys are the ones that stay.
struct HotSet {
    u32 n;
    u32 pad;
    u64 key[HOT_MAX];
    // What the key looked like in the batch that picked it — PREDICTIONS for the batch that uses the set, each checked
    // where it is used (rl_part.hpp): `cnt` sizes the key's share of k_bkt_apply's hot work items (any split covers the
    // bucket); `d` is the delta and `limit` the limit id its hits are expected to carry (a bucket whose hits all do is
    // decided from positions alone); `flg & 1`: the key's window was full when the batch ended, so "limited" is the
    // default answer k_bkt_part writes for its hits and k_bkt_apply only stores the exceptions.
    u32 cnt[HOT_MAX];
    u32 d[HOT_MAX];
    u32 limit[HOT_MAX];
    u32 flg[HOT_MAX];
    // the same predictions as ONE 16-byte record per key — key | d + (HOT_FLG_DENY << 31) | limit — for the partition
    // role of k_bkt_step, which keeps only key fingerprints in LDS and checks a match with a single load
    uint4 rec[HOT_MAX];
};
constexpr u32 HOT_LIMIT_UNKNOWN = 0xFFFFFFFFu;  // (never a valid limit id: ids are range-checked against the limit table)
constexpr u32 HOT_FLG_DENY = 1u;
// Append a key to the set a batch picks (any thread of any workgroup).
__device__ __forceinline__ void hot_append(HotSet* hs, u64 key, u32 cnt, u32 d, u32 limit = HOT_LIMIT_UNKNOWN, u32 flg = 0u) {
    const u32 pos = atomicAdd(&hs->n, 1u);
    if (pos < (u32)HOT_MAX) {
        hs->key[pos] = key;
        hs->cnt[pos] = cnt;
        hs->d[pos] = d;
        hs->limit[pos] = limit;
        hs->flg[pos] = flg;
        // (a predicted delta that does not fit 31 bits is never matched: such a bucket is replayed anyway)
        hs->rec[pos] = make_uint4((u32)key, (u32)(key >> 32), (d & 0x7FFFFFFFu) | ((flg & HOT_FLG_DENY) << 31),
                                  d < 0x80000000u ? limit : HOT_LIMIT_UNKNOWN);
    }
}
// Per-batch scratch of the bucketed path.  Two of them alternate: the last workgroup of a batch's
// k_bkt_apply mirrors the status block to host-mapped memory (no copy command behind the batch) and
// zeroes the OTHER one for the next batch (no memset command in front of it).
struct alignas(16) BatchScratch {
    Status st;
    u32 ticket;  // ticket shards that are complete
    u32 pad[15];
    // Workgroups that have finished, counted in 8 shards (blockIdx & 7: one per XCD), each on a line
    // of its own: same-address device-scope atomics serialise at ~10-30 ns apiece, and one ticket word
    // taken by 2048 workgroups was 9 us of a 46 us kernel.
    struct Shard {
        u32 t;
        u32 pad[31];
    } shard[8];
};
// What k_bkt_apply needs to decide a hot key's bucket.  k_bkt_scatter fills the part that only depends on
// the batch (range, delta, limit id, chunks); k_hot_state — which runs on the apply stream, after the
// previous batch's k_bkt_apply — adds the state of the key's cell before the batch.
struct HotParam {
    u32 lo, hi;   // the bucket's range in the partitioned batch
    u32 fast;     // decided from positions (see apply_hot_chunk); else replayed by one workgroup   [k_hot_state]
    u32 limit;    // limit id | SIMPLE
    u64 s;        // value_at(now) before the batch                                                [k_hot_state]
    u64 room;     // hits the reference admits: the first `room` of the bucket                     [k_hot_state]
    u64 d;        // the bucket's delta (uniform buckets only)
    u32 chunk0;   // HOT_CHUNK-sized chunks of the uniform buckets before this one
    u32 slot;     // the key's cell, SLOT_INVALID if it has none yet                               [k_hot_state]
    u32 expired;  // the cell was expired: the first admitted hit resets the window                [k_hot_state]
    u32 uni;      // every hit of the bucket carries the same delta: the bucket owns chunks
    u64 key;      // the bucket's key (k_bkt_apply reads the cell itself when no k_hot_state runs: self_hot)
};
static_assert(sizeof(HotParam) == 64, "four dwordx4 per hot bucket");

// Record of the partitioned batch: the hit's key and delta, its index in the caller's batch (where
// the verdict goes) and an 8-bit fold of its limit id — one 16-byte store per hit.  The limit id
// itself is not carried: it is an attribute of the counter cell (the caller interns key -> limit),
// the fold only lets the engine notice a caller that sends one key with two limit ids (exact for
// ids below 128, 255/256 otherwise); a new cell reads the id from the caller's batch.
struct BHit {
    u64 key;
    u32 delta;
    u32 idx_tag;  // idx (24 bits: MAX_BATCH_HITS) | limit fold << 24
};
static_assert(sizeof(BHit) == 16, "one dwordx4 per record");
__host__ __device__ inline u32 limit_fold(u32 limit) {
    return (limit ^ (limit >> 8) ^ (limit >> 16) ^ (limit >> 24)) & 0xFFu;
}
__device__ __forceinline__ BHit load_bhit(const BHit* p, u32 j) {
    const uint4 v = *reinterpret_cast<const uint4*>(p + j);
    BHit h;
    h.key = ((u64)v.y << 32) | v.x;
    h.delta = v.z;
    h.idx_tag = v.w;
    return h;
}

__device__ __forceinline__ u32 bucket_of_hash(u64 hh, u32 bk_log2) {
    return bk_log2 ? (u32)(hh >> (64 - bk_log2)) : 0u;
}

// LDS lookup table of the hot set: build (all threads call it; needs a barrier afterwards).
__device__ __forceinline__ void hot_table_build(const HotSet* __restrict__ hot, u64 seed, u64* s_key,
                                                u32* s_idx) {
    // the count and the keys are fetched together (one HBM latency, not two)
    const u32 n_raw = hot->n;
    const u64 k = threadIdx.x < (u32)HOT_MAX ? hot->key[threadIdx.x] : TAG_EMPTY;
    for (u32 q = threadIdx.x; q < (u32)HOT_HASH; q += blockDim.x) {
        s_key[q] = TAG_EMPTY;
        s_idx[q] = 0xFFFFFFFFu;
    }
    __syncthreads();
    const u32 n = n_raw < (u32)HOT_MAX ? n_raw : (u32)HOT_MAX;
    if (threadIdx.x < n) {
        // A key listed twice still gets ONE slot, and the smaller of its indices: every workgroup
        // of every kernel of the batch must map a key to the same bucket.
        u32 q = (u32)(fmix64(k ^ seed) >> 8) & (HOT_HASH - 1);
        for (;;) {
            const u64 prev = atomicCAS(&s_key[q], TAG_EMPTY, k);
            if (prev == TAG_EMPTY || prev == k) break;
            q = (q + 1) & (HOT_HASH - 1);
        }
        atomicMin(&s_idx[q], threadIdx.x);
    }
}
// -1, or the key's index in the hot set.  `hh` = fmix64(key ^ seed).
__device__ __forceinline__ int hot_lookup(const u64* s_key, const u32* s_idx, u64 key, u64 hh) {
    u32 q = (u32)(hh >> 8) & (HOT_HASH - 1);
    for (;;) {
        const u64 k = s_key[q];
        if (k == key) return (int)s_idx[q];
        if (k == TAG_EMPTY) return -1;
        q = (q + 1) & (HOT_HASH - 1);
    }
}

// Lanes of the wave whose `d` equals mine (among `valid` lanes): one ballot per bucket-id bit.
// All BK_LOG2_MAX + 1 bits are walked whatever `nbits` says: a bin id has no bit above nbits (d < 2^nbits), so those
// ballots are empty and change nothing — and without a test per bit the twelve steps are straight-line code.  The
// MISMATCHES are accumulated, z |= ballot(bit) ^ -(my bit): two xors and (pairwise) one three-input or per bit and
// half, where "m &= bit ? bm : ~bm" compiled to ten vector instructions, two wait states and a scalar branch per bit.
__device__ __forceinline__ u64 match_digit(u32 d, u32 nbits, u64 valid) {
    (void)nbits;
    u32 zlo = 0, zhi = 0;
#pragma unroll
    for (u32 b = 0; b < (u32)BK_LOG2_MAX + 1u; ++b) {
        const u32 sb = (u32)__builtin_amdgcn_sbfe((int)d, b, 1u);  // -(bit b of d)
        const u64 bm = __ballot(sb != 0u);
        zlo |= (u32)bm ^ sb;
        zhi |= (u32)(bm >> 32) ^ sb;
    }
    return valid & ~(((u64)zhi << 32) | zlo);
}

// ---------------------------------------------------------------------------------------------
// k_bkt_hist: hist[tile * nrow + bucket] = hits of `tile` that belong to `bucket`
// (nbt = 2^bk_log2 hash buckets + HOT_MAX hot-key buckets; a row has nrow = nbt + HOT_COLS columns:
// behind the counts, per hot key, the largest delta, the largest ~delta and a limit id the tile saw —
// k_bkt_scan reduces those columns with max, so no global atomic is needed for them).
// Also the only place the batch is validated, so that k_bkt_apply can refuse to touch the table
// when the batch is malformed: limit id range, reserved keys, and (in_memory.rs:106-107) a
// simple counter must already have its cell.
// ---------------------------------------------------------------------------------------------
template <int STEPS>
__global__ __launch_bounds__(PT_BLOCK) void k_bkt_hist(const Cell* __restrict__ table, u32 log2cap,
                                                       u64 seed, const Hit* __restrict__ hits, u32 n,
                                                       const LimitDev* __restrict__ limits,
                                                       u32 n_limits, u32 bk_log2, u32 ntiles,
                                                       u32* __restrict__ hist, BatchScratch* bs,
                                                       const HotSet* __restrict__ hot, u32 check_simple,
                                                       u64* htrace, uint8_t* __restrict__ verdict_fill,
                                                       int32_t* __restrict__ first_fill) {
    __shared__ u32 s_hist[BKT_MAX];
    __shared__ u64 s_hot_key[HOT_HASH];
    __shared__ u32 s_hot_idx[HOT_HASH];
    __shared__ u32 s_dmax[HOT_MAX], s_ndmin[HOT_MAX], s_hlimit[HOT_MAX];
    const u32 tid = threadIdx.x;
    Status* st = &bs->st;
#define RL_HSTAMP(k)                                                                           \
    do {                                                                                       \
        if (htrace && threadIdx.x == 0) htrace[(size_t)blockIdx.x * 8 + (k)] = wall_clock64(); \
    } while (0)
    // the same tile -> XCD assignment as k_bkt_scatter (see there): the tile this workgroup streams now
    // is re-read from the same XCD's L2 by the scatter pass
    u32 tile;
    {
        const u32 x = blockIdx.x & 7u, j = blockIdx.x >> 3, per = ntiles >> 3, rem = ntiles & 7u;
        tile = x * per + (x < rem ? x : rem) + j;
    }
    RL_HSTAMP(0);
    const u32 nb = 1u << bk_log2;
    const u32 nbt = nb + HOT_MAX;
    for (u32 b = tid; b < nbt; b += PT_BLOCK) s_hist[b] = 0;
    if (tid < HOT_MAX) {
        s_dmax[tid] = 0;
        s_ndmin[tid] = 0;
    }
    const u32 base = tile * (PT_BLOCK * STEPS);
    Hit h[STEPS];
#pragma unroll
    for (int r = 0; r < STEPS; ++r) {
        const u32 i = base + r * PT_BLOCK + tid;
        if (i < n) h[r] = load_hit(hits, i);
    }
    // "admitted" is the default answer, written here as coalesced stores while the batch streams by; k_bkt_apply then
    // scatters only the denials (a scattered 1-byte store per hit was 5.8 us of its 39: scripts/exp/v2.sh)
    if (verdict_fill) {
#pragma unroll
        for (int r = 0; r < STEPS; ++r) {
            const u32 i = base + r * PT_BLOCK + tid;
            if (i < n) {
                verdict_fill[i] = 0;
                if (first_fill) first_fill[i] = -1;
            }
        }
    }
    RL_HSTAMP(1);
    hot_table_build(hot, seed, s_hot_key, s_hot_idx);
    __syncthreads();
    RL_HSTAMP(2);
    u32 err = 0;
#pragma unroll
    for (int r = 0; r < STEPS; ++r) {
        const u32 i = base + r * PT_BLOCK + tid;
        if (i < n) {
            if ((h[r].limit & ~SIMPLE_FLAG) >= n_limits) err |= ERRBIT_BAD_LIMIT;
            else if (h[r].key >= TAG_TOMB) err |= ERRBIT_RESERVED_KEY;
            else if ((h[r].limit & SIMPLE_FLAG) && check_simple) {  // (update_counter creates simple cells: in_memory.rs:60-62)
                u32 dummy = 0;
                u32 slot = slot_of(h[r].key, seed, log2cap);
                slot = probe_from<PM_LOOKUP>(const_cast<Cell*>(table), log2cap, slot, table[slot].tag, h[r].key,
                                             h[r].limit, limits, 0ull, st, dummy);
                if (slot == SLOT_INVALID) err |= ERRBIT_MISSING_SIMPLE;
            }
            const u64 hh = fmix64(h[r].key ^ seed);
            const int hidx = hot_lookup(s_hot_key, s_hot_idx, h[r].key, hh);
            if (hidx < 0) {
                atomicAdd(&s_hist[bucket_of_hash(hh, bk_log2)], 1u);
            } else {
                // Only a count is needed here (the stable ranks are k_bkt_scatter's job), so the lanes of
                // a wave that carry the same hot key simply queue up on its LDS counter — a few cycles
                // per lane, far cheaper than finding each other with ballots.
                atomicAdd(&s_hist[nb + hidx], 1u);
                s_hlimit[hidx] = h[r].limit;  // any of the key's hits: k_bkt_scatter checks them against the cell
                // extrema of the key's deltas: read first, so that only a new extreme is an atomic
                const u32 d = h[r].delta;
                if (d > s_dmax[hidx]) atomicMax(&s_dmax[hidx], d);
                if (~d > s_ndmin[hidx]) atomicMax(&s_ndmin[hidx], ~d);
            }
        }
    }
    RL_HSTAMP(3);
    if (err) atomicOr(&st->err, err);
    RL_HSTAMP(4);
    __syncthreads();
    RL_HSTAMP(5);
    u32* row = hist + (size_t)tile * (nbt + HOT_COLS);
    for (u32 b = tid; b < nbt; b += PT_BLOCK) row[b] = s_hist[b];
    if (tid < HOT_MAX) {
        const bool any = s_hist[nb + tid] != 0;
        row[nbt + tid] = any ? s_dmax[tid] : 0u;
        row[nbt + HOT_MAX + tid] = any ? s_ndmin[tid] : 0u;
        row[nbt + 2 * HOT_MAX + tid] = any ? s_hlimit[tid] : 0u;
    }
    RL_HSTAMP(6);
}

// ---------------------------------------------------------------------------------------------
// k_bkt_scan: per bucket (column of hist), exclusive scan over the tiles in place;
// total[bucket] = its hit count.  A workgroup owns 32 columns; its 32 thread groups split the
// tiles, sum their part, exchange the part sums through LDS and rewrite their part.
// ---------------------------------------------------------------------------------------------
constexpr u32 SC_REG = 8;  // tiles per thread group that k_bkt_scan keeps in registers (batches of up to 256 tiles = 1 M hits)
__global__ __launch_bounds__(1024, 8) void k_bkt_scan(u32* __restrict__ hist, u32 ntiles, u32 nbt,
                                                      u32* __restrict__ total) {
    __shared__ u32 s_part[32][32];
    const u32 nrow = nbt + HOT_COLS;
    const u32 cl = threadIdx.x & 31u;
    const u32 c = blockIdx.x * 32 + cl;
    const bool is_max = c >= nbt;  // the hot-key attribute columns: total[c] = max over the tiles
    const u32 g = threadIdx.x >> 5;
    const u32 per = (ntiles + 31) / 32;
    const u32 t_lo = g * per < ntiles ? g * per : ntiles;
    const u32 t_hi = t_lo + per < ntiles ? t_lo + per : ntiles;
    // Up to SC_REG tiles per group (the usual batch): the group's counts are read ONCE, all loads in flight together,
    // and stay in registers for the rewrite — one trip to memory instead of two (the kernel is pure latency, and
    // beside k_bkt_apply a trip costs three times what it costs alone).  (64 VGPRs: see launch_apply.)
    const bool cached = per <= SC_REG;  // (block-uniform)
    u32 v8[SC_REG];
    u32 sum = 0;
    if (c < nrow) {
        if (cached) {
#pragma unroll
            for (u32 q = 0; q < SC_REG; ++q) {
                const u32 t = t_lo + q;
                const u32 tc = t < t_hi ? t : ntiles - 1;  // (unconditional load of a valid row; masked below)
                const u32 v = hist[(size_t)tc * nrow + c];
                v8[q] = t < t_hi ? v : 0u;
            }
#pragma unroll
            for (u32 q = 0; q < SC_REG; ++q) sum = is_max ? (v8[q] > sum ? v8[q] : sum) : sum + v8[q];
        } else {
#pragma unroll 8
            for (u32 t = t_lo; t < t_hi; ++t) {
                const u32 v = hist[(size_t)t * nrow + c];
                sum = is_max ? (v > sum ? v : sum) : sum + v;
            }
        }
    }
    s_part[g][cl] = sum;
    __syncthreads();
    if (c >= nrow) return;
    u32 run = 0, all = 0;
    for (u32 gg = 0; gg < 32; ++gg) {
        const u32 x = s_part[gg][cl];
        if (gg < g) run += x;
        all = is_max ? (x > all ? x : all) : all + x;
    }
    if (!is_max) {
        if (cached) {
#pragma unroll
            for (u32 q = 0; q < SC_REG; ++q) {
                const u32 t = t_lo + q;
                if (t < t_hi) hist[(size_t)t * nrow + c] = run;
                run += v8[q];
            }
        } else {
#pragma unroll 8
            for (u32 t = t_lo; t < t_hi; ++t) {
                const u32 v = hist[(size_t)t * nrow + c];
                hist[(size_t)t * nrow + c] = run;
                run += v;
            }
        }
    }
    if (g == 0) total[c] = all;
}

// Exclusive prefix of `v` over the 1024 threads of the workgroup (thread order); `total` = sum.
__device__ __forceinline__ u32 block_excl_scan_1024(u32 v, u32* s_w, u32& total) {
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    u32 inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 o = __shfl_up(inc, off);
        if ((int)lane >= off) inc += o;
    }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    u32 woff = 0, tot = 0;
#pragma unroll
    for (u32 ww = 0; ww < (u32)PT_WAVES; ++ww) {
        const u32 x = s_w[ww];
        if (ww < w) woff += x;
        tot += x;
    }
    __syncthreads();
    total = tot;
    return woff + inc - v;
}

// ---------------------------------------------------------------------------------------------
// k_bkt_scatter: stable partition.  Every workgroup first rebuilds the bucket starts (exclusive
// scan of the bucket totals: cheaper than another launch); workgroup 0 also publishes the
// hash-bucket ranges in processing order (large buckets first) and the hot-bucket ranges.
// Wave w of a workgroup owns the contiguous hits [tile + w*256, tile + (w+1)*256) and walks them
// in 64-hit steps in trace order; the rank of a hit inside (wave, bucket) comes from a
// wave-private LDS counter plus its position among the lanes of the step that share the bucket.
// ---------------------------------------------------------------------------------------------
template <int STEPS>
__global__ __launch_bounds__(PT_BLOCK) void k_bkt_scatter(const Hit* __restrict__ hits, u32 n, u64 seed,
                                                          u32 bk_log2, const u32* __restrict__ hist,
                                                          const u32* __restrict__ total,
                                                          const HotSet* __restrict__ hot,
                                                          BHit* __restrict__ b_hits,
                                                          uint2* __restrict__ ranges, const Status* st,
                                                          u32 ntiles, HotParam* __restrict__ hot_param,
                                                          HotSet* __restrict__ hot_next, u32 hot_threshold,
                                                          unsigned short* __restrict__ chunk_tab, u32 all_chunks,
                                                          u64* htrace, const u32* __restrict__ hit_req,
                                                          u32* __restrict__ b_req) {
    // hit_req / b_req (the general resolver; may be null): the request of every hit travels WITH its record —
    // b_req[p] = hit_req[i] for the record that lands at p.  Read here it is a coalesced read (i runs with the lanes) and
    // a 4-byte store beside the record's; gathered later through the record's index (k_gen_sort did that) it is a random
    // 4-byte read per hit — 3.1 M of them were more than half of that kernel.
    // wave-private counters, [PT_WAVES][nbt] — dynamic, sized by the launch for the batch's bucket count
    // (PT_WAVES * nbt * 2 bytes: 80 KB with 2048 hash buckets, 48 KB with 1024), so that with fewer buckets the
    // workgroup fits a CU beside k_bkt_apply's workgroups
    extern __shared__ __align__(16) unsigned short s_cnt[];
    __shared__ u32 s_base[BKT_MAX];
    __shared__ u32 s_w[PT_WAVES];
    __shared__ u64 s_hot_key[HOT_HASH];
    __shared__ u32 s_hot_idx[HOT_HASH];
    const u32 tid = threadIdx.x;
    // One extra workgroup prepares the hot keys' buckets for k_bkt_apply.
    if (blockIdx.x == ntiles) {
        if (htrace && tid == 0) htrace[(size_t)2040 * 8] = wall_clock64();
        // ---- the hot keys of this batch: where their buckets are, the state of their cells before
        //      the batch, and how the bucket will be decided (apply_hot); next batch's hot set -------
        const u32 nb_ = 1u << bk_log2, nbt_ = nb_ + HOT_MAX;
        const u32 b0 = 3 * tid;
        u32 c3[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) c3[q] = b0 + q < nbt_ ? total[b0 + q] : 0u;
        u32 all;
        const u32 ex = block_excl_scan_1024(c3[0] + c3[1] + c3[2], s_w, all);
        u32* s_lo = s_base;  // s_lo[h] = start of hot bucket h
        {
            u32 lo3[3] = {ex, ex + c3[0], ex + c3[0] + c3[1]};
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (b0 + q >= nb_ && b0 + q < nbt_) s_lo[b0 + q - nb_] = lo3[q];
        }
        if (tid == 0) hot_next->n = 0;  // k_bkt_apply appends the keys it promotes
        __syncthreads();
        const bool refuse = st->err != 0;  // k_bkt_hist rejected the batch: empty ranges, nothing applied
        const u32 nh = hot->n < (u32)HOT_MAX ? hot->n : (u32)HOT_MAX;
        u32* s_nchunk = s_base + HOT_MAX;
        {
            // ranges[]: the hash buckets in the order k_bkt_apply's workgroups take them (workgroup w:
            // entries w, w + G, ...), longest first in 64 size classes, so that every workgroup draws the
            // same mix of long and short buckets (a bucket's time is its number of 512-hit rounds).
            u32* s_cls = s_base + 2 * HOT_MAX;
            u32* s_cls_lo = s_cls + 64;
            if (tid < 64) s_cls[tid] = 0;
            __syncthreads();
            u32 cls[3], r[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                cls[q] = 63u - (c3[q] / 32u < 63u ? c3[q] / 32u : 63u);
                r[q] = b0 + q < nb_ ? atomicAdd(&s_cls[cls[q]], 1u) : 0u;
            }
            __syncthreads();
            if (tid < 64) {
                u32 inc = s_cls[tid];
                const u32 own = inc;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const u32 o = __shfl_up(inc, off);
                    if ((int)tid >= off) inc += o;
                }
                s_cls_lo[tid] = inc - own;
            }
            __syncthreads();
            const u32 lo3[3] = {ex, ex + c3[0], ex + c3[0] + c3[1]};
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (b0 + q < nb_)
                    ranges[s_cls_lo[cls[q]] + r[q]] = refuse ? make_uint2(0, 0) : make_uint2(lo3[q], lo3[q] + c3[q]);
        }
        if (tid == HOT_MAX) hot_param[HOT_MAX] = HotParam{};
        if (tid < HOT_MAX) {
            HotParam hp{};
            hp.slot = SLOT_INVALID;
            const u32 cnt = refuse ? 0u : total[nb_ + tid];
            hp.lo = s_lo[tid];
            hp.hi = hp.lo + cnt;
            if (tid < nh && cnt) {
                hp.key = hot->key[tid];
                hp.limit = total[nbt_ + 2 * HOT_MAX + tid];
                const u32 dmax = total[nbt_ + tid], dmin = ~total[nbt_ + HOT_MAX + tid];
                hp.d = dmax;
                hp.uni = dmin == dmax ? 1u : 0u;
                if (cnt >= hot_threshold) hot_append(hot_next, hot->key[tid], cnt, dmax, hp.limit);
            }
            // (the general resolver walks EVERY hot bucket in chunks: all_chunks)
            s_nchunk[tid] = (hp.uni || all_chunks) ? (cnt + HOT_CHUNK - 1) / HOT_CHUNK : 0u;
            hot_param[tid] = hp;
        }
        __syncthreads();
        u32* s_c0 = s_base + 2 * HOT_MAX + 256;  // chunk0[] again, for the chunk table below
        if (tid <= HOT_MAX) {  // chunk0[h] = chunks of the fast buckets before h; entry HOT_MAX = all
            u32 acc = 0;
            for (u32 q = 0; q < tid; ++q) acc += s_nchunk[q];
            hot_param[tid].chunk0 = acc;
            s_c0[tid] = acc;
        }
        __syncthreads();
        // chunk_tab[c] = the fast bucket that owns chunk c: the LAST h with chunk0[h] <= c (a bucket
        // without chunks shares its successor's chunk0, so it is never the last one) — k_bkt_apply2
        for (u32 c = tid; c < s_c0[HOT_MAX]; c += PT_BLOCK) {
            u32 a = 0, b = HOT_MAX;  // invariant: chunk0[a] <= c < chunk0[b]
            while (b - a > 1) {
                const u32 m = (a + b) >> 1;
                if (s_c0[m] <= c) a = m;
                else b = m;
            }
            chunk_tab[c] = (unsigned short)a;
        }
        if (htrace && tid == 0) htrace[(size_t)2040 * 8 + 1] = wall_clock64();
        return;
    }
    // Workgroups go round-robin over the 8 XCDs, each with its own L2.  Consecutive tiles write
    // neighbouring records of every bucket's range, so each XCD takes a CONTIGUOUS run of tiles: the
    // 16-byte records of neighbouring tiles then meet in one L2 and leave it as full lines
    // (interleaved tiles: 30 MB written to HBM for 16 MB of records).
    u32 tile;
    {
        const u32 x = blockIdx.x & 7u, j = blockIdx.x >> 3, per = ntiles >> 3, rem = ntiles & 7u;
        tile = x * per + (x < rem ? x : rem) + j;
    }
    RL_HSTAMP(0);
    const u32 lane = tid & 63u, w = tid >> 6;
    const u32 nb = 1u << bk_log2;
    const u32 nbt = nb + HOT_MAX;
    const u32 wbase = tile * (PT_BLOCK * STEPS) + w * (64 * STEPS);
    uint4 raw[STEPS];
    u32 rq[STEPS];
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
        const u32 i = wbase + u * 64 + lane;
        if (i < n) raw[u] = *reinterpret_cast<const uint4*>(hits + i);
        rq[u] = (b_req && i < n) ? hit_req[i] : 0u;
    }
    hot_table_build(hot, seed, s_hot_key, s_hot_idx);
    {
        // bucket order in the partitioned arrays: hash buckets 0..nb-1, then the hot buckets
        const u32 b0 = 3 * tid;
        u32 c[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) c[q] = b0 + q < nbt ? total[b0 + q] : 0u;
        u32 all;
        const u32 ex = block_excl_scan_1024(c[0] + c[1] + c[2], s_w, all);
        u32 lo[3] = {ex, ex + c[0], ex + c[0] + c[1]};
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (b0 + q < nbt) s_base[b0 + q] = lo[q] + hist[(size_t)tile * (nbt + HOT_COLS) + b0 + q];
    }
    RL_HSTAMP(1);
    for (u32 b = tid; b < nbt; b += PT_BLOCK) {
#pragma unroll
        for (int ww = 0; ww < PT_WAVES; ++ww) s_cnt[ww * nbt + b] = 0;
    }
    __syncthreads();
    RL_HSTAMP(2);
    unsigned short rank[STEPS];
    unsigned short dig[STEPS];
    const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
        const u32 i = wbase + u * 64 + lane;
        const bool ok = i < n;
        const u64 valid = __ballot(ok);
        const u64 key = ((u64)raw[u].y << 32) | raw[u].x;
        u32 d = 0;
        if (ok) {
            const u64 hh = fmix64(key ^ seed);
            const int hi = hot_lookup(s_hot_key, s_hot_idx, key, hh);
            d = hi >= 0 ? nb + (u32)hi : bucket_of_hash(hh, bk_log2);
        }
        const u64 m = match_digit(d, (bk_log2 > 9u ? bk_log2 : 9u) + 1u, valid);
        u32 r = 0;
        if (ok) {
            const u32 c = s_cnt[w * nbt + d];
            r = c + (u32)__popcll(m & lt);
            if ((m & lt) == 0ull) s_cnt[w * nbt + d] = (unsigned short)(c + (u32)__popcll(m));
        }
        rank[u] = (unsigned short)r;
        dig[u] = (unsigned short)d;
    }
    RL_HSTAMP(3);
    __syncthreads();
    RL_HSTAMP(4);
    // wave-private counts -> exclusive offsets of the waves inside (tile, bucket)
    for (u32 b = tid; b < nbt; b += PT_BLOCK) {
        u32 acc = 0;
#pragma unroll
        for (int ww = 0; ww < PT_WAVES; ++ww) {
            const u32 c = s_cnt[ww * nbt + b];
            s_cnt[ww * nbt + b] = (unsigned short)acc;
            acc += c;
        }
    }
    __syncthreads();
    RL_HSTAMP(5);
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
        const u32 i = wbase + u * 64 + lane;
        if (i < n) {
            const u32 d = dig[u];
            const u32 dst = s_base[d] + s_cnt[w * nbt + d] + rank[u];
            *reinterpret_cast<uint4*>(b_hits + dst) =
                make_uint4(raw[u].x, raw[u].y, raw[u].w, i | (limit_fold(raw[u].z) << 24));
            if (b_req) b_req[dst] = rq[u];
        }
    }
    RL_HSTAMP(6);
}

// ---------------------------------------------------------------------------------------------
// k_hot_state: the state of every hot key's cell BEFORE this batch, and from it how the key's bucket
// is decided (see apply_hot_chunk in rl_apply.hpp).  One workgroup, one thread per hot key.  It runs
// on the apply stream between the previous batch's k_bkt_apply and this batch's: the partition of
// this batch (k_bkt_hist / scan / scatter) may overlap the previous batch's k_bkt_apply, the cell
// reads may not.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(HOT_MAX) void k_hot_state(const Cell* __restrict__ table, u32 log2cap, u64 seed,
                                                       const LimitDev* __restrict__ limits, u64 now,
                                                       const HotSet* __restrict__ hot, HotParam* __restrict__ hot_param,
                                                       Status* st) {
    const u32 tid = threadIdx.x;
    const u32 nh = hot->n < (u32)HOT_MAX ? hot->n : (u32)HOT_MAX;
    if (tid >= nh) return;
    HotParam hp = hot_param[tid];
    if (hp.hi == hp.lo) return;
    const u64 k = hot->key[tid];
    u32 dummy = 0;
    u32 slot = slot_of(k, seed, log2cap);
    slot = probe_from<PM_LOOKUP>(const_cast<Cell*>(table), log2cap, slot, table[slot].tag, k, 0u, limits, 0ull, st, dummy);
    hp.slot = slot;
    bool limit_ok = true;
    if (slot != SLOT_INVALID) {
        const Cell* c = &table[slot];
        const u64 expiry = c->expiry;
        hp.expired = expiry <= now ? 1u : 0u;
        hp.s = expiry <= now ? 0ull : c->value;  // value_at(now), atomic_expiring_value.rs:19-24
        limit_ok = c->limit == hp.limit;
    }
    const LimitDev L = limits[hp.limit & ~SIMPLE_FLAG];
    if (hp.uni && L.window_us != 0 && hp.s < (1ull << 62) && limit_ok && (slot != SLOT_INVALID || !(hp.limit & SIMPLE_FLAG))) {
        hp.fast = 1;
        hp.room = hp.s > L.max_value ? 0ull : (hp.d ? (L.max_value - hp.s) / hp.d : ~0ull);
    }
    hot_param[tid] = hp;
}

// ---------------------------------------------------------------------------------------------
// shared by the bucket replay kernels (rl_apply.hpp)
// ---------------------------------------------------------------------------------------------
constexpr int AP_BLOCK = 256;
constexpr u32 EF_EXPIRED = 1u;   // the cell was expired when first read: the first admitted hit resets the window
constexpr u32 EF_DIRTY = 2u;     // at least one hit admitted: the cell must be written back
constexpr u32 EF_SLOW = 4u;      // this round: replay the entry's hits one by one
constexpr u32 EF_BAD = 8u;       // a hit carried a limit id that is not the cell's / no cell
constexpr u32 EF_COUNT_SHIFT = 8;  // bits 8..31: hits this entry has absorbed (hot entries survive a rebuild)
constexpr u32 EF_HOT_MIN = 16;
constexpr u32 ENT_NONE = 0xFFFFu;
constexpr u32 TINY_MAX = 1024;   // batches up to this many hits are one bucket: k_bkt_tiny

// End of a batch's last kernel, one thread per workgroup: the last workgroup out hands the status
// block to the host and zeroes the scratch of a later batch (`bs_zero`; the scratches rotate).  No agent-scope fence (a release would write the
// XCD's whole L2 back, once per workgroup): everything the last workgroup reads was written with
// device-scope atomics, and this workgroup's own contribution has RETURNED before its ticket is taken.
__device__ __forceinline__ void apply_finish(u32 n_created, BatchScratch* bs, BatchScratch* bs_zero, Status* host_status,
                                             u32 done_seq, u32 G, const u32* hot_n_ptr, u32 hot_n) {
    u32 dep = 0;
    if (n_created) dep = atomicAdd(&bs->st.n_inserted, n_created);
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(dep) : "memory");
    const u32 shard = blockIdx.x & 7u;
    const u32 n_in_shard = (G + 7u - shard) >> 3;  // workgroups b < G with (b & 7) == shard
    const bool last = atomicAdd(&bs->shard[shard].t, 1u) == n_in_shard - 1 &&
                      atomicAdd(&bs->ticket, 1u) == (G < 8u ? G : 8u) - 1;
    if (last) {
        Status out{};
        out.err = __hip_atomic_load(&bs->st.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        out.n_inserted = __hip_atomic_load(&bs->st.n_inserted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        out.pad[0] = __hip_atomic_load(&bs->st.pad[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        out.pad[1] = __hip_atomic_load(&bs->st.pad[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // keys that qualified for the next hot set (read here: every other workgroup's promotions are done)
        out.pad[2] = hot_n_ptr ? __hip_atomic_load(hot_n_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : hot_n;
        // The host does not wait on an event (a marker in the queue costs ~5 us of idle device per
        // batch): it polls the first 16 bytes of the status block, written LAST and as ONE store, so
        // `n_removed == done_seq` means the whole block of this batch is there.
        out.n_removed = done_seq;
        typedef u32 u32x4 __attribute__((ext_vector_type(4)));
        u32x4* hp = reinterpret_cast<u32x4*>(host_status);
        const u32* o = reinterpret_cast<const u32*>(&out);
        for (int q = 1; q < 4; ++q) hp[q] = u32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_nontemporal_store(u32x4{o[0], o[1], o[2], o[3]}, hp);
        // the other scratch, zeroed for the next batch: explicit 16-byte stores (the plain assignment of
        // a 1 KB struct compiles to a byte loop: 15 us on one lane)
        u32x4* z = reinterpret_cast<u32x4*>(bs_zero);
#pragma unroll 8
        for (u32 q = 0; q < (u32)(sizeof(BatchScratch) / 16); ++q) z[q] = u32x4{0, 0, 0, 0};
    }
}

}  // namespace rl
