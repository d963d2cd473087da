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
//   k_bkt_apply    persistent workgroups, one hash bucket at a time: the replay described above,
//                  with the next bucket's inputs in flight; then the hot buckets, from positions
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
};
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
// What k_bkt_apply needs to decide a hot key's bucket, prepared once per batch by k_bkt_scatter.
struct HotParam {
    u32 lo, hi;   // the bucket's range in the partitioned batch
    u32 fast;     // decided from positions (see apply_hot); else replayed by one workgroup
    u32 limit;    // limit id | SIMPLE
    u64 s;        // value_at(now) before the batch
    u64 room;     // hits the reference admits: the first `room` of the bucket
    u64 d;        // the bucket's delta (fast only)
    u32 chunk0;   // HOT_CHUNK-sized chunks of the fast buckets before this one
    u32 slot;     // the key's cell, SLOT_INVALID if it has none yet
    u32 expired;  // the cell was expired: the first admitted hit resets the window
    u32 pad;
};

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
__device__ __forceinline__ u64 match_digit(u32 d, u32 nbits, u64 valid) {
    u64 m = valid;
#pragma unroll
    for (u32 b = 0; b < (u32)BK_LOG2_MAX + 1u; ++b) {
        if (b < nbits) {
            const bool bit = (d >> b) & 1u;
            const u64 bm = __ballot(bit);
            m &= bit ? bm : ~bm;
        }
    }
    return m;
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
                                                       const HotSet* __restrict__ hot, u64* htrace) {
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
            else if (h[r].limit & SIMPLE_FLAG) {
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
__global__ __launch_bounds__(1024) void k_bkt_scan(u32* __restrict__ hist, u32 ntiles, u32 nbt,
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
    u32 sum = 0;
    if (c < nrow) {
#pragma unroll 8
        for (u32 t = t_lo; t < t_hi; ++t) {
            const u32 v = hist[(size_t)t * nrow + c];
            sum = is_max ? (v > sum ? v : sum) : sum + v;
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
#pragma unroll 8
        for (u32 t = t_lo; t < t_hi; ++t) {
            const u32 v = hist[(size_t)t * nrow + c];
            hist[(size_t)t * nrow + c] = run;
            run += v;
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
                                                          uint2* __restrict__ ranges, Status* st,
                                                          const Cell* __restrict__ table, u32 log2cap,
                                                          const LimitDev* __restrict__ limits, u64 now,
                                                          u32 ntiles, HotParam* __restrict__ hot_param,
                                                          HotSet* __restrict__ hot_next,
                                                          const BatchScratch* __restrict__ bs, u32 hot_threshold,
                                                          unsigned short* __restrict__ chunk_tab, u64* htrace) {
    __shared__ __align__(16) unsigned short s_cnt[PT_WAVES][BKT_MAX];
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
                const u64 k = hot->key[tid];
                u32 dummy = 0;
                u32 slot = slot_of(k, seed, log2cap);
                slot = probe_from<PM_LOOKUP>(const_cast<Cell*>(table), log2cap, slot, table[slot].tag, k, 0u, limits,
                                             0ull, st, dummy);
                hp.slot = slot;
                hp.limit = total[nbt_ + 2 * HOT_MAX + tid];
                bool limit_ok = true;
                if (slot != SLOT_INVALID) {
                    const Cell* c = &table[slot];
                    const u64 expiry = c->expiry;
                    hp.expired = expiry <= now ? 1u : 0u;
                    hp.s = expiry <= now ? 0ull : c->value;
                    limit_ok = c->limit == hp.limit;
                }
                const u32 dmax = total[nbt_ + tid], dmin = ~total[nbt_ + HOT_MAX + tid];
                const LimitDev L = limits[hp.limit & ~SIMPLE_FLAG];
                hp.d = dmax;
                if (dmin == dmax && L.window_us != 0 && hp.s < (1ull << 62) && limit_ok &&
                    (slot != SLOT_INVALID || !(hp.limit & SIMPLE_FLAG))) {
                    hp.fast = 1;
                    hp.room = hp.s > L.max_value ? 0ull : (hp.d ? (L.max_value - hp.s) / hp.d : ~0ull);
                }
                if (cnt >= hot_threshold) {
                    const u32 pos = atomicAdd(&hot_next->n, 1u);
                    if (pos < (u32)HOT_MAX) hot_next->key[pos] = k;
                }
            }
            s_nchunk[tid] = hp.fast ? (cnt + HOT_CHUNK - 1) / HOT_CHUNK : 0u;
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
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
        const u32 i = wbase + u * 64 + lane;
        if (i < n) raw[u] = *reinterpret_cast<const uint4*>(hits + i);
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
        for (int ww = 0; ww < PT_WAVES; ++ww) s_cnt[ww][b] = 0;
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
            const u32 c = s_cnt[w][d];
            r = c + (u32)__popcll(m & lt);
            if ((m & lt) == 0ull) s_cnt[w][d] = (unsigned short)(c + (u32)__popcll(m));
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
            const u32 c = s_cnt[ww][b];
            s_cnt[ww][b] = (unsigned short)acc;
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
            const u32 dst = s_base[d] + s_cnt[w][d] + rank[u];
            *reinterpret_cast<uint4*>(b_hits + dst) =
                make_uint4(raw[u].x, raw[u].y, raw[u].w, i | (limit_fold(raw[u].z) << 24));
        }
    }
    RL_HSTAMP(6);
}

// ---------------------------------------------------------------------------------------------
// k_bkt_apply
// ---------------------------------------------------------------------------------------------
constexpr int AP_BLOCK = 256;
constexpr int AP_HPT = 2;                     // consecutive batch items per thread
constexpr int AP_R = AP_BLOCK * AP_HPT;       // hits per decide/commit round
constexpr int AP_WS = 4;                      // strips of AP_BLOCK hits per quick-path window
constexpr int AP_W = AP_BLOCK * AP_WS;        // hits per window
constexpr int AP_Q = 2048;                    // ring of deferred hits (< AP_R left over + one window)
constexpr int ENT_LOG2 = 10;
constexpr int ENT_N = 1 << ENT_LOG2;          // LDS cells per workgroup
constexpr int ENT_KEEP = ENT_N * 3 / 4 - AP_R;  // rebuild the LDS cells before a round if more are live
constexpr u32 EF_EXPIRED = 1u;   // the cell was expired when first read: the first admitted hit resets the window
constexpr u32 EF_DIRTY = 2u;     // at least one hit admitted: the cell must be written back
constexpr u32 EF_SLOW = 4u;      // this round: replay the entry's hits one by one
constexpr u32 EF_BAD = 8u;       // a hit carried a limit id that is not the cell's / no cell
constexpr u32 EF_COUNT_SHIFT = 8;  // bits 8..31: hits this entry has absorbed (hot entries survive a rebuild)
constexpr u32 EF_HOT_MIN = 16;
constexpr u32 ENT_NONE = 0xFFFFu;
constexpr int LIM_LDS = 512;   // limit-table rows the bucketed path supports (all of them live in LDS)

struct ApplyLds {
    u64 key[ENT_N];
    u64 run[ENT_N];    // value the next hit reads; for 0-second windows: the last admitted delta
    u64 rsum[ENT_N];   // this round: sum of deltas
    u64 cnt4[ENT_N];   // this round: hits per wave (4 x u16)
    u32 slot[ENT_N];
    u32 limit[ENT_N];  // the CELL's limit attribute
    u32 dmax[ENT_N];   // this round: largest delta
    u32 flags[ENT_N];
    u32 queue[AP_Q];   // deferred hits (positions in the bucketed arrays), trace order
    u32 h_delta[AP_R];
    unsigned short h_ent[AP_R];
    uint8_t h_verdict[AP_R];
    u32 wcnt[AP_WS][4];
    LimitDev lim[LIM_LDS];  // the limit table
    struct HotRange {
        u32 lo, hi, chunk0, fast;
    } hot[HOT_MAX + 1];  // where the hot buckets are (the full HotParam rows stay in global memory)
    u32 n_ent;
    u32 bucket_len;
    u32 promote_ok;  // 0 while a hot bucket is replayed: its key is kept or dropped by count, not promoted
    u32 any_slow;
    u32 n_created;
    u32 n_keep;
};

struct ApplyArgs {
    Cell* table;
    u32 log2cap;
    u64 seed;
    const BHit* b_hits;
    const Hit* hits;  // the caller's batch: only read for the limit id of a key that has no cell yet
    const LimitDev* limits;
    u32 n_limits;
    u64 now;
    uint8_t* verdict;
    int32_t* first_limited;
    Status* st;
    HotSet* hot_next;  // next batch's hot keys (appended to)
    const HotParam* hot_param;  // this batch's hot-bucket table
    u32 hot_threshold;          // hits in this batch that make a key hot for the next one
    u32 vmask;   // debug (RL_DEBUG_VMASK): AND-mask on the verdict index, 0xFFFFFFFF normally
    u64* trace;  // debug (RL_APPLY_TRACE=1): per-workgroup phase timestamps, 16 per workgroup; else null
};
#define RL_STAMP(k)                                                                       \
    do {                                                                                  \
        if (A.trace && threadIdx.x == 0) A.trace[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); \
    } while (0)

// Row of the limit table, always from LDS: a global load here would also wait for the prefetched
// inputs of the next bucket (the memory counter retires in order).  The host only takes this path
// when the whole table fits (n_limits <= LIM_LDS); ids were range-checked by k_bkt_hist.
__device__ __forceinline__ LimitDev limit_row(const ApplyLds& S, const ApplyArgs& A, u32 limit) {
    return S.lim[(limit & ~SIMPLE_FLAG) & (LIM_LDS - 1)];
}

// Write the dirty LDS cells back; optionally rebuild the LDS hash keeping only the hot entries.
__device__ __forceinline__ void apply_commit(ApplyLds& S, const ApplyArgs& A, bool rebuild) {
    constexpr int PER = ENT_N / AP_BLOCK;
    u64 k_key[PER], k_run[PER];
    u32 k_slot[PER], k_limit[PER], k_flags[PER];
    bool keep[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const u32 e = threadIdx.x + q * AP_BLOCK;
        keep[q] = false;
        const u64 key = S.key[e];
        if (key == TAG_EMPTY) continue;
        u32 f = S.flags[e];
        // promote: the key absorbed HOT_PROMOTE hits, or it is what made this bucket long (hits denied
        // on the spot by the window pass are not counted, so a saturated key shows fewer than it got)
        if (!rebuild && !(f & EF_BAD) && S.promote_ok &&
            ((f >> EF_COUNT_SHIFT) >= A.hot_threshold ||
             ((f >> EF_COUNT_SHIFT) >= A.hot_threshold / 4 && S.bucket_len >= HOT_LONG_BUCKET))) {
            const u32 pos = atomicAdd(&A.hot_next->n, 1u);
            if (pos < (u32)HOT_MAX) A.hot_next->key[pos] = key;
        }
        if (f & EF_DIRTY) {
            Cell* c = &A.table[S.slot[e]];
            c->value = S.run[e];
            if (f & EF_EXPIRED) c->expiry = A.now + limit_row(S, A, S.limit[e]).window_us;
            // the window is open again — except a 0-second one, which is expired at every read
            if (limit_row(S, A, S.limit[e]).window_us != 0) f &= ~EF_EXPIRED;
            f &= ~EF_DIRTY;
        }
        if (rebuild && (f >> EF_COUNT_SHIFT) >= EF_HOT_MIN && !(f & EF_BAD)) {
            keep[q] = true;
            k_key[q] = key;
            k_run[q] = S.run[e];
            k_slot[q] = S.slot[e];
            k_limit[q] = S.limit[e];
            k_flags[q] = f;
        }
    }
    if (!rebuild) {  // end of the bucket: leave the table empty
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const u32 e = threadIdx.x + q * AP_BLOCK;
            S.key[e] = TAG_EMPTY;
            S.flags[e] = 0;
        }
        return;
    }
    if (threadIdx.x == 0) S.n_keep = 0;
    __syncthreads();
    u32 nk = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) nk += keep[q] ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) nk += __shfl_down(nk, off);
    if ((threadIdx.x & 63u) == 0 && nk) atomicAdd(&S.n_keep, nk);
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const u32 e = threadIdx.x + q * AP_BLOCK;
        S.key[e] = TAG_EMPTY;
        S.flags[e] = 0;
    }
    __syncthreads();
    const bool reinsert = S.n_keep <= (u32)ENT_KEEP;
    if (reinsert) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            if (!keep[q]) continue;
            u32 e = (u32)(fmix64(k_key[q] ^ A.seed) >> 20) & (ENT_N - 1);
            while (atomicCAS(&S.key[e], TAG_EMPTY, k_key[q]) != TAG_EMPTY) e = (e + 1) & (ENT_N - 1);
            S.run[e] = k_run[q];
            S.slot[e] = k_slot[q];
            S.limit[e] = k_limit[q];
            S.flags[e] = k_flags[q];
        }
    }
    if (threadIdx.x == 0) S.n_ent = reinsert ? S.n_keep : 0u;
    __syncthreads();
}

// Inputs of one decide/commit round, fetched ahead of the round itself: the hits, and for every hit
// its home cell (tag + value, expiry, limit: one 64-byte line) — only the claimer of a new LDS cell
// consumes the latter.
struct RoundIn {
    BHit h[AP_HPT];
    u32 hslot[AP_HPT], climit[AP_HPT];
    u64 ctag[AP_HPT], cvalue[AP_HPT], cexpiry[AP_HPT];
};
// position of item p is first + p (from_queue == false) or S.queue[(first + p) % AP_Q]
__device__ __forceinline__ void round_load_hits(const ApplyLds& S, const ApplyArgs& A, bool from_queue,
                                                u32 first, u32 n_items, RoundIn& in) {
#pragma unroll
    for (int u = 0; u < AP_HPT; ++u) {
        const u32 p = threadIdx.x * AP_HPT + u;
        if (p < n_items) {
            const u32 j = from_queue ? S.queue[(first + p) & (AP_Q - 1)] : first + p;
            in.h[u] = load_bhit(A.b_hits, j);
        }
    }
}
__device__ __forceinline__ void round_load_lines(const ApplyArgs& A, u32 n_items, RoundIn& in) {
#pragma unroll
    for (int u = 0; u < AP_HPT; ++u) {
        const u32 p = threadIdx.x * AP_HPT + u;
        if (p >= n_items) continue;
        in.hslot[u] = slot_of(in.h[u].key, A.seed, A.log2cap);
        const Cell* c = &A.table[in.hslot[u]];
        const uint4 a = *reinterpret_cast<const uint4*>(c);
        const uint4 b = reinterpret_cast<const uint4*>(c)[1];  // expiry, limit
        in.ctag[u] = ((u64)a.y << 32) | a.x;
        in.cvalue[u] = ((u64)a.w << 32) | a.z;
        in.cexpiry[u] = ((u64)b.y << 32) | b.x;
        in.climit[u] = b.z;
    }
}

// One decide/commit round over up to AP_R hits (inputs already requested).  `mid` runs once the
// round's own table reads are done (after phase B): the place to request the NEXT bucket's inputs,
// since the memory counter retires in order and a later wait would also wait for them.
template <class Mid>
__device__ __forceinline__ void apply_round_core(ApplyLds& S, const ApplyArgs& A, u32 n_items, RoundIn& in,
                                                 Mid mid) {
    const u32 tid = threadIdx.x;
    const u32 lane = tid & 63u, w = tid >> 6;
    const u64 lt = (1ull << lane) - 1ull;
    BHit(&h)[AP_HPT] = in.h;
    u32(&hslot)[AP_HPT] = in.hslot;
    u32(&climit)[AP_HPT] = in.climit;
    u64(&ctag)[AP_HPT] = in.ctag;
    u64(&cvalue)[AP_HPT] = in.cvalue;
    u64(&cexpiry)[AP_HPT] = in.cexpiry;
    u32 idx[AP_HPT], ent[AP_HPT];
    bool ok[AP_HPT], creator[AP_HPT], leader[AP_HPT];
#pragma unroll
    for (int u = 0; u < AP_HPT; ++u) {
        const u32 p = tid * AP_HPT + u;
        ok[u] = p < n_items;
        creator[u] = leader[u] = false;
        ent[u] = 0;
        idx[u] = ok[u] ? (h[u].idx_tag & 0xFFFFFFu) : 0u;
    }
    RL_STAMP(2);
    // ---- A: find or claim the key's LDS cell, add this hit to the round's aggregates ------------
    // Staged over the thread's hits: the same LDS operation is issued for every hit before any result
    // is consumed, so the round trips of independent hits overlap instead of queueing behind each other.
    u32 n_new = 0;
    u64 first_key[AP_HPT];
#pragma unroll
    for (int u = 0; u < AP_HPT; ++u) {
        ent[u] = ok[u] ? (u32)(fmix64(h[u].key ^ A.seed) >> 20) & (ENT_N - 1) : 0u;
        first_key[u] = ok[u] ? S.key[ent[u]] : 0ull;
    }
#pragma unroll
    for (int u = 0; u < AP_HPT; ++u) {
        if (!ok[u]) continue;
        u32 e = ent[u];
        u64 prev = first_key[u];
        for (;;) {
            // a plain read first: the lanes that repeat a key already in LDS do not queue up on a CAS
            if (prev == TAG_EMPTY) prev = atomicCAS(&S.key[e], TAG_EMPTY, h[u].key);
            if (prev == TAG_EMPTY) {
                creator[u] = true;
                ++n_new;
                break;
            }
            if (prev == h[u].key) break;
            e = (e + 1) & (ENT_N - 1);
            prev = S.key[e];
        }
        ent[u] = e;
    }
    u32 seen_dmax[AP_HPT];
#pragma unroll
    for (int u = 0; u < AP_HPT; ++u) {
        const u32 p = tid * AP_HPT + u;
        if (!ok[u]) {
            S.h_ent[p] = (unsigned short)ENT_NONE;
            seen_dmax[u] = 0;
            continue;
        }
        atomicAdd(&S.rsum[ent[u]], (u64)h[u].delta);
        seen_dmax[u] = S.dmax[ent[u]];
        S.h_ent[p] = (unsigned short)ent[u];
        S.h_delta[p] = h[u].delta;
    }
    u64 before[AP_HPT];
#pragma unroll
    for (int u = 0; u < AP_HPT; ++u) {
        before[u] = 1;
        if (!ok[u]) continue;
        if (h[u].delta > seen_dmax[u]) atomicMax(&S.dmax[ent[u]], h[u].delta);  // only a new maximum is an atomic
        before[u] = atomicAdd(&S.cnt4[ent[u]], 1ull << (16 * w));
    }
#pragma unroll
    for (int u = 0; u < AP_HPT; ++u) leader[u] = ok[u] && before[u] == 0ull;
    for (int off = 32; off > 0; off >>= 1) n_new += __shfl_down(n_new, off);
    if (lane == 0 && n_new) atomicAdd(&S.n_ent, n_new);
    RL_STAMP(3);
    // ---- B: the claimer of a new LDS cell resolves the counter cell -------------------------------
    u32 created = 0;
#pragma unroll
    for (int u = 0; u < AP_HPT; ++u) {
        if (!creator[u]) continue;
        const u32 e = ent[u];
        u32 slot = hslot[u];
        u64 value = cvalue[u], expiry = cexpiry[u];
        u32 cl = climit[u];
        if (ctag[u] != h[u].key) {
            // Not at home: probe on, fetching whole cells so that a match needs no further read; the
            // limit id (caller's batch) is only read when the cell has to be created
            // (in_memory.rs:122-127).
            const u32 mask = (1u << A.log2cap) - 1u;
            u64 tag = ctag[u];  // the tag at `slot`; not ours
            bool done = false, missing_simple = false;
            for (u32 step = 0; step <= mask; ++step) {
                if (tag == TAG_EMPTY) {
                    const u32 hl = A.hits[idx[u]].limit;
                    if (hl & SIMPLE_FLAG) {  // in_memory.rs:106-107: a simple counter must pre-exist
                        missing_simple = true;
                        break;
                    }
                    const u64 old = atomicCAS(&A.table[slot].tag, TAG_EMPTY, h[u].key);
                    if (old == TAG_EMPTY || old == h[u].key) {
                        Cell* c = &A.table[slot];
                        if (old == TAG_EMPTY) {  // AtomicExpiringValue::new(0, now + window), in_memory.rs:123-125
                            c->value = 0;
                            c->expiry = A.now + A.limits[hl & ~SIMPLE_FLAG].window_us;
                            c->limit = hl;
                            ++created;
                        }
                        value = c->value;
                        expiry = c->expiry;
                        cl = c->limit;
                        done = true;
                        break;
                    }
                    // somebody else's key landed here first: keep probing
                }
                slot = (slot + 1) & mask;
                const Cell* c = &A.table[slot];
                const uint4 a = *reinterpret_cast<const uint4*>(c);
                const uint4 b = reinterpret_cast<const uint4*>(c)[1];
                const u64 ex = ((u64)b.y << 32) | b.x;
                const u32 li = b.z;
                tag = ((u64)a.y << 32) | a.x;
                if (tag == h[u].key) {
                    value = ((u64)a.w << 32) | a.z;
                    expiry = ex;
                    cl = li;
                    done = true;
                    break;
                }
            }
            if (!done) {
                atomicOr(&A.st->err, missing_simple ? ERRBIT_MISSING_SIMPLE : ERRBIT_TABLE_FULL);
                slot = SLOT_INVALID;
                value = 0;
                expiry = 0;
            }
        }
        const bool expired = expiry <= A.now;
        S.run[e] = expired ? 0ull : value;  // value_at(now), atomic_expiring_value.rs:19-24
        S.slot[e] = slot;
        S.limit[e] = cl;
        S.flags[e] = (expired ? EF_EXPIRED : 0u) | (slot == SLOT_INVALID ? EF_BAD : 0u);
    }
    if (created) atomicAdd(&S.n_created, created);
    mid();
    RL_STAMP(4);
    __syncthreads();
    RL_STAMP(5);
    // ---- C: verdicts -------------------------------------------------------------------------------
    uint8_t v[AP_HPT];
    bool need_rank[AP_HPT], slow[AP_HPT];
    u64 room[AP_HPT];
#pragma unroll
    for (int u = 0; u < AP_HPT; ++u) {
        v[u] = 1;
        need_rank[u] = slow[u] = false;
        room[u] = 0;
        if (!ok[u]) continue;
        const u32 e = ent[u];
        if (limit_fold(S.limit[e]) != (h[u].idx_tag >> 24)) {
            atomicOr(&S.flags[e], EF_BAD);
            atomicOr(&A.st->err, ERRBIT_KEY_LIMIT);
            continue;
        }
        const LimitDev Lu = limit_row(S, A, S.limit[e]);
        const u64 run = S.run[e], sum = S.rsum[e], c4 = S.cnt4[e];
        const u64 d = h[u].delta;
        const u64 cnt = (c4 & 0xFFFFull) + ((c4 >> 16) & 0xFFFFull) + ((c4 >> 32) & 0xFFFFull) + (c4 >> 48);
        u64 tot;
        const bool ovf = __builtin_add_overflow(run, sum, &tot);
        if (Lu.window_us == 0 || ovf) {
            slow[u] = true;  // every read sees an expired cell / the sum wraps: replay
        } else if (tot <= Lu.max_value) {
            v[u] = 0;
        } else if (run + d > Lu.max_value) {
            v[u] = 1;
        } else if (sum == cnt * (u64)S.dmax[e]) {  // all deltas of the round equal (and > 0 here)
            need_rank[u] = true;
            room[u] = (Lu.max_value - run) / d;
        } else {
            slow[u] = true;
        }
        if (slow[u]) {
            atomicOr(&S.flags[e], EF_SLOW);
            S.any_slow = 1;
        }
    }
    // trace-order rank among the round's hits on the same key: hits of earlier waves, then
    // earlier lanes of this wave, then earlier hits of this lane.
    for (;;) {
        bool have = false;
        u32 my_e = 0;
#pragma unroll
        for (int u = AP_HPT - 1; u >= 0; --u)
            if (need_rank[u]) {
                have = true;
                my_e = ent[u];
            }
        const u64 pend = __ballot(have);
        if (!pend) break;
        const u32 e0 = __shfl(my_e, __ffsll((long long)pend) - 1);
        u32 before_lane = 0;
#pragma unroll
        for (int u = 0; u < AP_HPT; ++u) before_lane += (u32)__popcll(__ballot(ok[u] && ent[u] == e0) & lt);
        const u64 c4 = S.cnt4[e0];
        u64 pre = 0;
        for (u32 ww = 0; ww < w; ++ww) pre += (c4 >> (16 * ww)) & 0xFFFFull;
        u32 mine = 0;
#pragma unroll
        for (int u = 0; u < AP_HPT; ++u) {
            if (need_rank[u] && ent[u] == e0) {
                v[u] = (pre + before_lane + mine) < room[u] ? 0 : 1;
                need_rank[u] = false;
            }
            if (ok[u] && ent[u] == e0) ++mine;
        }
    }
    __syncthreads();
    RL_STAMP(6);
    // ---- slow entries: one lane replays the round in trace order, reference arithmetic -----------
    if (S.any_slow) {
        if (tid == 0) {
            for (u32 p = 0; p < n_items; ++p) {
                const u32 e = S.h_ent[p];
                const u32 f = S.flags[e];
                if (!(f & EF_SLOW) || (f & EF_BAD)) continue;
                const LimitDev Le = limit_row(S, A, S.limit[e]);
                const u64 d = S.h_delta[p];
                const u64 cur = Le.window_us == 0 ? 0ull : S.run[e];
                const u64 sum = cur + d;  // wraps like the reference's release build (in_memory.rs:88)
                const bool adm = sum <= Le.max_value;
                if (adm) {
                    S.run[e] = Le.window_us == 0 ? d : sum;
                    S.flags[e] = f | EF_DIRTY | (Le.window_us == 0 ? EF_EXPIRED : 0u);
                }
                S.h_verdict[p] = adm ? 0 : 1;
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < AP_HPT; ++u)
            if (slow[u]) v[u] = S.h_verdict[tid * AP_HPT + u];
    }
    // ---- D: the round's first arriver of each key folds the round into `run` --------------------
#pragma unroll
    for (int u = 0; u < AP_HPT; ++u) {
        if (!ok[u]) continue;
        const u32 i = idx[u];
        A.verdict[i & A.vmask] = v[u];
        if (A.first_limited) A.first_limited[i] = v[u] ? (int32_t)i : -1;
        if (!leader[u]) continue;
        const u32 e = ent[u];
        u32 f = S.flags[e];
        const u64 c4 = S.cnt4[e];
        const u64 cnt = (c4 & 0xFFFFull) + ((c4 >> 16) & 0xFFFFull) + ((c4 >> 32) & 0xFFFFull) + (c4 >> 48);
        if (!(f & (EF_SLOW | EF_BAD))) {
            const LimitDev Le = limit_row(S, A, S.limit[e]);
            const u64 run = S.run[e], sum = S.rsum[e];
            const u64 dm = S.dmax[e];
            if (run + sum <= Le.max_value) {  // no overflow here: overflowing rounds are slow
                S.run[e] = run + sum;
                f |= EF_DIRTY;
            } else if (sum == cnt * dm && run + dm <= Le.max_value) {
                const u64 rm = (Le.max_value - run) / dm;
                const u64 n_adm = cnt < rm ? cnt : rm;
                if (n_adm) {
                    S.run[e] = run + n_adm * dm;
                    f |= EF_DIRTY;
                }
            }
        }
        const u32 seen = (f >> EF_COUNT_SHIFT) + (u32)cnt;
        f = (f & 0xFFu & ~EF_SLOW) | ((seen > 0xFFFFFFu ? 0xFFFFFFu : seen) << EF_COUNT_SHIFT);
        S.flags[e] = f;
        S.rsum[e] = 0;
        S.cnt4[e] = 0;
        S.dmax[e] = 0;
    }
    if (tid == 0) S.any_slow = 0;
    __syncthreads();
    RL_STAMP(7);
}

// A round that fetches its own inputs (everything but the first round of a bucket).
__device__ __forceinline__ void apply_round(ApplyLds& S, const ApplyArgs& A, bool from_queue, u32 first,
                                            u32 n_items) {
    if (S.n_ent > (u32)ENT_KEEP) {  // block-uniform (read after the previous round's barrier)
        __syncthreads();
        apply_commit(S, A, true);
    }
    RoundIn in;
    round_load_hits(S, A, from_queue, first, n_items, in);
    round_load_lines(A, n_items, in);
    apply_round_core(S, A, n_items, in, [] {});
}

// A whole bucket [lo, hi) of the partitioned batch, in trace order, by one workgroup.
// `in0` holds the (already requested) inputs of its first min(AP_R, hi - lo) hits.
template <class Mid>
__device__ __forceinline__ void apply_bucket(ApplyLds& S, const ApplyArgs& A, u32 lo, u32 hi, RoundIn& in0,
                                             Mid mid) {
    const u32 tid = threadIdx.x;
    const u32 lane = tid & 63u, w = tid >> 6;
    const u64 lt = (1ull << lane) - 1ull;
    // (the LDS cells are empty here: cleared once per launch and again by every bucket's final commit)
    if (tid == 0) {
        S.n_ent = 0;
        S.any_slow = 0;
        S.bucket_len = hi - lo;
    }
    __syncthreads();
    RL_STAMP(1);

    // The first AP_R hits of the bucket go straight through a decide/commit round.  In a bucket made
    // long by a frequent key this also brings that key's cell into LDS.
    apply_round_core(S, A, (hi - lo) < (u32)AP_R ? (hi - lo) : (u32)AP_R, in0, mid);

    // The rest of a long bucket streams through in windows: a hit whose key is already in LDS and
    // saturated (run + delta > max, and run can only grow) is denied on the spot — valid in any
    // order; everything else is deferred, in trace order, to a ring of positions that is drained in
    // decide/commit rounds.
    u32 q_head = 0, q_tail = 0;
    // two window buffers, used alternately: the next window's hits are requested before the current
    // one is examined (no copies of registers with a load in flight)
    auto window_load = [&](u32 wbase, BHit(&wh)[AP_WS]) {
#pragma unroll
        for (int u = 0; u < AP_WS; ++u) {
            const u32 j = wbase + u * AP_BLOCK + tid;
            if (j < hi) wh[u] = load_bhit(A.b_hits, j);
        }
    };
    auto window = [&](u32 wbase, BHit(&wh)[AP_WS], BHit(&wnext)[AP_WS]) {
        if (wbase + AP_W < hi) window_load(wbase + AP_W, wnext);
        bool unres[AP_WS];
        u32 pos[AP_WS];
#pragma unroll
        for (int u = 0; u < AP_WS; ++u) {
            const u32 j = wbase + u * AP_BLOCK + tid;
            pos[u] = j;
            unres[u] = false;
            if (j >= hi) continue;
            const BHit h = wh[u];
            const u32 i = h.idx_tag & 0xFFFFFFu;
            unres[u] = true;
            u32 e = (u32)(fmix64(h.key ^ A.seed) >> 20) & (ENT_N - 1);
            for (;;) {
                const u64 k = S.key[e];
                if (k == h.key) {
                    const u64 run = S.run[e];
                    const u32 f = S.flags[e];
                    const u32 cl = S.limit[e];
                    const LimitDev L = limit_row(S, A, cl);
                    if (limit_fold(cl) == (h.idx_tag >> 24) && !(f & EF_BAD) && L.window_us != 0 && run < (1ull << 63) &&
                        run + (u64)h.delta > L.max_value) {
                        A.verdict[i & A.vmask] = 1;
                        if (A.first_limited) A.first_limited[i] = (int32_t)i;
                        unres[u] = false;
                    }
                    break;
                }
                if (k == TAG_EMPTY) break;
                e = (e + 1) & (ENT_N - 1);
            }
        }
        // stable compaction of the unresolved hits into the ring: order (strip, wave, lane)
        u64 bal[AP_WS];
#pragma unroll
        for (int u = 0; u < AP_WS; ++u) {
            bal[u] = __ballot(unres[u]);
            if (lane == 0) S.wcnt[u][w] = (u32)__popcll(bal[u]);
        }
        __syncthreads();
        u32 total = 0;
#pragma unroll
        for (int u = 0; u < AP_WS; ++u) {
            u32 before = 0, strip_total = 0;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                const u32 c = S.wcnt[u][ww];
                if (ww < (int)w) before += c;
                strip_total += c;
            }
            if (unres[u]) S.queue[(q_tail + total + before + (u32)__popcll(bal[u] & lt)) & (AP_Q - 1)] = pos[u];
            total += strip_total;
        }
        q_tail += total;
        __syncthreads();
        while (q_tail - q_head >= (u32)AP_R) {
            apply_round(S, A, true, q_head, AP_R);
            q_head += AP_R;
        }
    };
    if (lo + AP_R < hi) {
        BHit wa[AP_WS], wb[AP_WS];
        window_load(lo + AP_R, wa);
        for (u32 wbase = lo + AP_R;;) {
            window(wbase, wa, wb);
            wbase += AP_W;
            if (wbase >= hi) break;
            window(wbase, wb, wa);
            wbase += AP_W;
            if (wbase >= hi) break;
        }
    }
    if (q_tail != q_head) apply_round(S, A, true, q_head, q_tail - q_head);
    RL_STAMP(8);
    apply_commit(S, A, false);  // also leaves every LDS cell empty for the next bucket
    RL_STAMP(9);
    __syncthreads();
}

// A hot key's bucket holds one key, in trace order: a hit's position in the bucket is its rank on
// the key.  With one delta value d > 0 for the whole bucket the reference admits exactly the first
// (max - s) / d hits (s = value_at(now): in_memory.rs:259-264 with atomic_expiring_value.rs:36-42
// applied after each admission), so every hit is decided from its position alone, by any number
// of workgroups.  Buckets that do not fit that form (mixed deltas, 0-second window, values near
// 2^64, a stale hot set that made two keys share...) are replayed by one worker through
// apply_bucket, which is exact for everything.

__device__ __forceinline__ void apply_hot(ApplyLds& S, const ApplyArgs& A, u32 worker, u32 n_workers) {
    const u32 tid = threadIdx.x;
    RL_STAMP(11);
    RL_STAMP(12);
    if (A.trace && worker == 0)
        for (u32 q = tid; q < (u32)HOT_MAX; q += AP_BLOCK)
            if (S.hot[q].hi != S.hot[q].lo) atomicAdd(&A.st->pad[S.hot[q].fast ? 0 : 1], S.hot[q].hi - S.hot[q].lo);
    // ---- fast buckets: chunks of HOT_CHUNK hits, grid-strided over the workers --------------------
    const u32 n_chunks = S.hot[HOT_MAX].chunk0;
    for (u32 c = worker; c < n_chunks; c += n_workers) {
        // the bucket that owns chunk c: the LAST h with chunk0[h] <= c (a bucket without chunks shares
        // its successor's chunk0, so it is never the last one)
        u32 a = 0, b = HOT_MAX;  // invariant: chunk0[a] <= c < chunk0[b]
        while (b - a > 1) {
            const u32 m = (a + b) >> 1;
            if (S.hot[m].chunk0 <= c) a = m;
            else b = m;
        }
        const HotParam hp = A.hot_param[a];  // one dependent read per chunk of HOT_CHUNK hits
        const u32 lo = hp.lo, hi = hp.hi;
        const u32 first = lo + (c - hp.chunk0) * HOT_CHUNK;
        const u64 room = hp.room;
        const u32 limit = hp.limit;
#pragma unroll
        for (int u = 0; u < HOT_CHUNK / AP_BLOCK; ++u) {
            const u32 j = first + u * AP_BLOCK + tid;
            if (j >= hi) continue;
            const BHit h = load_bhit(A.b_hits, j);
            const u32 i = h.idx_tag & 0xFFFFFFu;
            uint8_t v = (u64)(j - lo) < room ? 0 : 1;
            if ((h.idx_tag >> 24) != limit_fold(limit)) {  // one key, two limit ids: caller contract violation
                atomicOr(&A.st->err, ERRBIT_KEY_LIMIT);
                v = 1;
            }
            A.verdict[i & A.vmask] = v;
            if (A.first_limited) A.first_limited[i] = v ? (int32_t)i : -1;
        }
        if (first == lo && tid == 0) {
            // the bucket's first chunk also applies AtomicExpiringValue::update for the admitted hits
            const LimitDev L = limit_row(S, A, limit);
            const u64 cnt = hi - lo;
            const u64 n_adm = cnt < room ? cnt : room;
            u32 slot = hp.slot;
            bool expired = hp.expired != 0;
            if (slot == SLOT_INVALID) {  // first touch creates the cell (in_memory.rs:122-127), verdict or not
                u32 created = 0;
                const u64 key = A.b_hits[lo].key;
                slot = slot_of(key, A.seed, A.log2cap);
                slot = probe_from<PM_CHECK>(A.table, A.log2cap, slot, A.table[slot].tag, key, limit, A.limits,
                                            A.now, A.st, created);
                if (created) atomicAdd(&A.st->n_inserted, created);
                expired = false;
            }
            if (n_adm && slot != SLOT_INVALID) {
                Cell* cell = &A.table[slot];
                cell->value = hp.s + n_adm * hp.d;
                if (expired) cell->expiry = A.now + L.window_us;
            }
        }
    }
    RL_STAMP(13);
    // ---- everything else: replayed by one worker per bucket ---------------------------------------
    if (tid == 0) S.promote_ok = 0;  // (apply_bucket starts with a barrier)
    for (u32 hk = worker; hk < (u32)HOT_MAX; hk += n_workers) {
        const ApplyLds::HotRange hp = S.hot[hk];
        if (hp.fast || hp.hi == hp.lo) continue;
        RoundIn in0;
        const u32 n0 = (hp.hi - hp.lo) < (u32)AP_R ? (hp.hi - hp.lo) : (u32)AP_R;
        round_load_hits(S, A, false, hp.lo, n0, in0);
        round_load_lines(A, n0, in0);
        apply_bucket(S, A, hp.lo, hp.hi, in0, [] {});
    }
}

// End of a batch's last kernel, one thread per workgroup: the last workgroup out hands the status
// block to the host and resets the other scratch.  No agent-scope fence (a release would write the
// XCD's whole L2 back, once per workgroup): everything the last workgroup reads was written with
// device-scope atomics, and this workgroup's own contribution has RETURNED before its ticket is taken.
__device__ __forceinline__ void apply_finish(u32 n_created, BatchScratch* bs, BatchScratch* bs_next, Status* host_status,
                                             u32 done_seq, u32 G, const u32* hot_n_ptr, u32 hot_n, u32 tmode = 0) {
    u32 dep = 0;
    if (n_created) dep = atomicAdd(&bs->st.n_inserted, n_created);
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(dep) : "memory");
    bool last;
    if (tmode == 1) {  // (experiment) one ticket word
        last = atomicAdd(&bs->ticket, 1u) == G - 1;
    } else if (tmode == 2) {  // (experiment) shards by consecutive runs of workgroups
        const u32 per = (G + 7u) >> 3, shard = blockIdx.x / per;
        const u32 n_in_shard = (shard + 1) * per <= G ? per : G - shard * per;
        const u32 n_shards = (G + per - 1) / per;
        last = atomicAdd(&bs->shard[shard].t, 1u) == n_in_shard - 1 && atomicAdd(&bs->ticket, 1u) == n_shards - 1;
    } else {
        const u32 shard = blockIdx.x & 7u;
        const u32 n_in_shard = (G + 7u - shard) >> 3;  // workgroups b < G with (b & 7) == shard
        last = atomicAdd(&bs->shard[shard].t, 1u) == n_in_shard - 1 &&
               atomicAdd(&bs->ticket, 1u) == (G < 8u ? G : 8u) - 1;
    }
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
        u32x4* z = reinterpret_cast<u32x4*>(bs_next);
#pragma unroll 8
        for (u32 q = 0; q < (u32)(sizeof(BatchScratch) / 16); ++q) z[q] = u32x4{0, 0, 0, 0};
    }
}

// Persistent workgroups: workgroup g replays hash buckets g, g + G, g + 2G, ... (ranges[] is in
// processing order, large buckets first) and then takes its share of the hot-bucket chunks.  The
// inputs of the next bucket (its first AP_R hits, then their home cells) are requested while the
// current bucket is being decided, so the HBM latency of one bucket hides behind the LDS work of
// the previous one.
constexpr int AP_MAX_PER_WG = 16;  // buckets per workgroup: the host launches G >= nb / 16 workgroups

__global__ __launch_bounds__(AP_BLOCK) void k_bkt_apply(
    Cell* __restrict__ table, u32 log2cap, u64 seed, const BHit* __restrict__ b_hits,
    const Hit* __restrict__ hits, const uint2* __restrict__ ranges, u32 nb,
    const HotParam* __restrict__ hot_param,
    const LimitDev* __restrict__ limits, u32 n_limits, u64 now, uint8_t* __restrict__ verdict,
    int32_t* __restrict__ first_limited, BatchScratch* bs, BatchScratch* bs_next, Status* host_status, u32 done_seq,
    HotSet* hot_next, u32 hot_threshold, u32 vmask, u64* trace) {
    __shared__ ApplyLds S;
    __shared__ uint2 s_ranges[AP_MAX_PER_WG + 4];
    const u32 tid = threadIdx.x, G = gridDim.x;
    if (trace && tid == 0) trace[(size_t)blockIdx.x * 16 + 0] = wall_clock64();
    ApplyArgs A{table, log2cap, seed, b_hits, hits,  limits, n_limits, now,
                verdict, first_limited, &bs->st, hot_next, hot_param, hot_threshold, vmask, trace};
    auto first_n = [](uint2 r) { return (r.y - r.x) < (u32)AP_R ? (r.y - r.x) : (u32)AP_R; };
    // Three input buffers used round-robin (bucket t decides from b[t % 3] while the lines of bucket
    // t+1 land in b[(t+1) % 3] and the hits of bucket t+2 in b[(t+2) % 3]); the loop is unrolled by
    // three so that no register that is the target of a load in flight ever has to be copied.
    RoundIn b0, b1, b2;
    {
        // the first two buckets' hits are requested before anything else: they fly during the LDS set-up
        const uint2 r0 = blockIdx.x < nb ? ranges[blockIdx.x] : make_uint2(0, 0);
        const uint2 r1 = 2 * G - 1 - blockIdx.x < nb ? ranges[2 * G - 1 - blockIdx.x] : make_uint2(0, 0);
        round_load_hits(S, A, false, r0.x, first_n(r0), b0);
        round_load_hits(S, A, false, r1.x, first_n(r1), b1);
    }
    if (tid < AP_MAX_PER_WG + 4) {
        // row `tid` of ranges[] (longest buckets first), rows taken alternately left-to-right and
        // right-to-left: the workgroup that drew the longest bucket of one row draws the shortest of the next
        const u32 k = tid * G + ((tid & 1u) ? G - 1 - blockIdx.x : blockIdx.x);
        s_ranges[tid] = (tid < AP_MAX_PER_WG && k < nb) ? ranges[k] : make_uint2(0, 0);
    }
    for (u32 q = tid; q < (u32)LIM_LDS && q < n_limits; q += AP_BLOCK) S.lim[q] = limits[q];
    for (u32 e = tid; e < ENT_N; e += AP_BLOCK) {
        S.key[e] = TAG_EMPTY;
        S.rsum[e] = 0;
        S.cnt4[e] = 0;
        S.dmax[e] = 0;
        S.flags[e] = 0;
    }
    for (u32 q = tid; q <= (u32)HOT_MAX; q += AP_BLOCK) {
        const HotParam hp = hot_param[q];
        S.hot[q] = ApplyLds::HotRange{hp.lo, hp.hi, hp.chunk0, hp.fast};
    }
    if (tid == 0) {
        S.n_created = 0;
        S.promote_ok = 1;
    }
    __syncthreads();
    round_load_lines(A, first_n(s_ranges[0]), b0);
    auto step = [&](u32 t, RoundIn& cur, RoundIn& nxt, RoundIn& nn) -> bool {
        if (t >= (u32)AP_MAX_PER_WG || t * G >= nb) return false;
        const uint2 r_cur = s_ranges[t], r_nxt = s_ranges[t + 1], r_nn = s_ranges[t + 2];
        auto mid = [&] {
            round_load_lines(A, first_n(r_nxt), nxt);
            round_load_hits(S, A, false, r_nn.x, first_n(r_nn), nn);
        };
        if (r_cur.x != r_cur.y) apply_bucket(S, A, r_cur.x, r_cur.y, cur, mid);
        else mid();
        if (trace && tid == 0 && r_cur.x != r_cur.y) {
            trace[(size_t)blockIdx.x * 16 + 10] = r_cur.y - r_cur.x;
            trace[(size_t)blockIdx.x * 16 + 15] += ((u64)1 << 32) + (r_cur.y - r_cur.x);
        }
        return true;
    };
    for (u32 t = 0;; t += 3) {
        if (!step(t, b0, b1, b2)) break;
        if (!step(t + 1, b1, b2, b0)) break;
        if (!step(t + 2, b2, b0, b1)) break;
    }
    // k_bkt_hist refused the batch: the hot ranges are empty too, nothing is applied
    // (workers in reverse: the low workgroups drew the longest hash buckets — some of them a bucket
    // beyond one round — and the hot chunks are dealt from worker 0 up, so the extra chunk of an uneven
    // deal goes to the workgroups that finish their buckets first)
    apply_hot(S, A, G - 1 - blockIdx.x, G);
    RL_STAMP(14);
    __syncthreads();
    if (tid == 0) apply_finish(S.n_created, bs, bs_next, host_status, done_seq, G, &hot_next->n, 0u);
}

// ---------------------------------------------------------------------------------------------
// k_bkt_tiny: a batch of at most TINY_MAX hits IS one bucket — it is in trace order already — so one
// workgroup validates it, rewrites it as BHit records and replays it with the bucket code: one
// launch instead of four (a 1..1024-hit call: ~30 us -> ~15 us).  The hot set is left as it is.
// ---------------------------------------------------------------------------------------------
constexpr u32 TINY_MAX = 1024;

__global__ __launch_bounds__(AP_BLOCK) void k_bkt_tiny(
    Cell* __restrict__ table, u32 log2cap, u64 seed, const Hit* __restrict__ hits, u32 n, BHit* __restrict__ b_hits,
    const LimitDev* __restrict__ limits, u32 n_limits, u64 now, uint8_t* __restrict__ verdict,
    int32_t* __restrict__ first_limited, BatchScratch* bs, BatchScratch* bs_next, Status* host_status, u32 done_seq,
    u32 hot_n_report, u32 vmask) {
    __shared__ ApplyLds S;
    __shared__ u32 s_err;
    const u32 tid = threadIdx.x;
    if (tid == 0) s_err = 0;
    for (u32 q = tid; q < (u32)LIM_LDS && q < n_limits; q += AP_BLOCK) S.lim[q] = limits[q];
    for (u32 e = tid; e < ENT_N; e += AP_BLOCK) {
        S.key[e] = TAG_EMPTY;
        S.rsum[e] = 0;
        S.cnt4[e] = 0;
        S.dmax[e] = 0;
        S.flags[e] = 0;
    }
    if (tid == 0) {
        S.n_created = 0;
        S.promote_ok = 0;  // no promotion from here: the hot set belongs to the partitioned path
    }
    __syncthreads();
    // the checks of k_bkt_hist (nothing is applied to a malformed batch) + the record format of k_bkt_scatter
    u32 err = 0;
    for (u32 i = tid; i < n; i += AP_BLOCK) {
        const Hit h = load_hit(hits, i);
        if ((h.limit & ~SIMPLE_FLAG) >= n_limits) err |= ERRBIT_BAD_LIMIT;
        else if (h.key >= TAG_TOMB) err |= ERRBIT_RESERVED_KEY;
        else if (h.limit & SIMPLE_FLAG) {  // in_memory.rs:106-107: a simple counter must pre-exist
            u32 dummy = 0;
            u32 slot = slot_of(h.key, seed, log2cap);
            slot = probe_from<PM_LOOKUP>(table, log2cap, slot, table[slot].tag, h.key, h.limit, limits, 0ull, &bs->st,
                                         dummy);
            if (slot == SLOT_INVALID) err |= ERRBIT_MISSING_SIMPLE;
        }
        *reinterpret_cast<uint4*>(b_hits + i) =
            make_uint4((u32)h.key, (u32)(h.key >> 32), h.delta, i | (limit_fold(h.limit) << 24));
    }
    if (err) atomicOr(&s_err, err);
    __syncthreads();  // (also orders the b_hits stores before this workgroup's loads of them)
    if (s_err) {
        if (tid == 0) atomicOr(&bs->st.err, s_err);
    } else if (n) {
        ApplyArgs A{table, log2cap, seed, b_hits, hits, limits, n_limits, now, verdict, first_limited, &bs->st,
                    nullptr, nullptr, 0xFFFFFFFFu, vmask, nullptr};
        RoundIn in0;
        const u32 n0 = n < (u32)AP_R ? n : (u32)AP_R;
        round_load_hits(S, A, false, 0, n0, in0);
        round_load_lines(A, n0, in0);
        apply_bucket(S, A, 0, n, in0, [] {});
    }
    // the verdicts may go straight to host-mapped memory (rl_check_and_update_batch): every wave's stores
    // have been acknowledged before the completion word is written
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) apply_finish(S.n_created, bs, bs_next, host_status, done_seq, 1u, nullptr, hot_n_report);
}

}  // namespace rl
