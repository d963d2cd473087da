// rl_part.hpp — the single-pass stable partition of the measured hot path, and the "bucket view" through which
// k_bkt_apply reads it.
//
// Why it exists.  The replay of CounterStorage::check_and_update (reference limitador/src/storage/in_memory.rs:72-156,
// one call per request, in trace order) needs every hit of one key in ONE workgroup and in trace order; nothing else.
// The first design produced that with a GLOBAL stable partition — k_bkt_hist (read the batch), k_bkt_scan (a
// latency-bound launch over the tiles x buckets matrix), k_bkt_scatter (read the batch again, write it in 16-64-byte
// runs): three dependent launches, the batch read twice, 35 us alone and 59 us beside k_bkt_apply — the critical path
// of the step.  A global order is more than the replay needs:
//
//   k_bkt_part   every 4096-hit tile of the batch is sorted by bucket INSIDE ITS OWN 64 KB of the record buffer
//                (stable: wave-private LDS counters + ballot ranks, as before) and publishes, per bucket, where its
//                run starts and how long it is: runs[bucket][tile].  No tile needs anything from another tile: no
//                scan, no look-back, no second read — the batch is read once (16 B/hit) and written once (16 B/hit,
//                every line complete inside one workgroup's window), and the kernel is one wave of independent
//                workgroups.  It also validates the batch and writes the default verdict, as k_bkt_hist did.
//   BucketView   the owner of bucket b reads row b of `runs` (one coalesced 1 KB read), scans it in LDS, and from
//                then on maps "position p of my bucket, in trace order" to the record's address: tile t with
//                pre[t] <= p < pre[t+1], record start[t] + (p - pre[t]) of that tile.  Tiles are in trace order and
//                runs are stable, so positions ARE trace order.  The reads become 64-byte gathers instead of one
//                contiguous range — the same number of memory requests (a 1 M-hit batch has ~4 hits per (tile, bucket)).
//
// Hot keys (rl_bucket.hpp) keep their buckets of their own: bins nb .. nb+HOT_MAX-1 of every tile.  What used to need
// the scanned totals is now decided without them: the split of a hot bucket into work items comes from the count
// the key had when it was PICKED (HotSet::cnt — any split covers the bucket: item k of nk walks the 1024-position
// chunks k, k+nk, ...), and "every hit carries the same delta" is checked by every tile against the delta the set
// PREDICTS (one flag bit per run), so a workgroup knows from the row it reads anyway whether positions decide the bucket.
#pragma once
#include "rl_bucket.hpp"

namespace rl {

constexpr int TT_SMALL = 256;    // tiles a bucket view holds: the usual instantiation (batches up to 256 tiles)
constexpr int TT_LARGE = 1024;   // ... and the one for the largest batches (MAX_BATCH_HITS / 16384 tiles)
constexpr u32 PT_STEPS_MAX = 16; // 64-hit steps per wave in the largest tile (16384 hits)
constexpr u32 HOT_NK_MAX = 64;   // most work items one hot bucket is split into (beyond: the items stride)
constexpr int HOT_SLOTS = 2048;  // slots of the compact hot-set table (16-bit entries over the set's keys; load <= 1/4: a lookup looks
                                 // at 1.3 slots on average)

// runs[bin * TT + tile]: where the tile's hits of `bin` start inside the tile's window, how many there are, and (hot
// bins) whether one of them carries another delta than the hot set predicts.
constexpr u32 RUN_FLAG = 1u << 30;
__host__ __device__ inline u32 run_pack(u32 start, u32 cnt, bool flag) { return start | (cnt << 15) | (flag ? RUN_FLAG : 0u); }
__host__ __device__ inline u32 run_start(u32 v) { return v & 0x7FFFu; }
__host__ __device__ inline u32 run_count(u32 v) { return (v >> 15) & 0x7FFFu; }

// k_bkt_apply's hot work items, written by k_bkt_part's extra workgroup from the hot set alone: item k of the nk items
// of hot bucket hb.  Everything an item needs to start is in its own 32 bytes (one dependent load, not two).
struct HotItem {
    u64 key;
    u32 hb_k;    // hot bucket | k << 16
    u32 nk;      // items of this bucket (>= 1 for a key of the set)
    u32 d;       // the delta the set predicts for the key's hits
    u32 limit;   // the limit id the set predicts (HOT_LIMIT_UNKNOWN: none — the bucket is replayed)
    u32 flg;     // HOT_FLG_DENY: k_bkt_part wrote "limited" as the default answer of the key's hits
    u32 pad;
};
static_assert(sizeof(HotItem) == 32, "two dwordx4 per item");
struct HotItems {
    u32 n;  // items in it[]
    u32 pad[7];
    HotItem it[HOT_MAX * HOT_NK_MAX];
};

// ---------------------------------------------------------------------------------------------
// BucketView: position in the bucket (trace order) -> index of the record in the partitioned batch
// ---------------------------------------------------------------------------------------------
template <int TT>
struct BucketView {
    static constexpr int tiles = TT;
    u32 pre[TT + 1];           // pre[t] = hits of the bucket in tiles < t; pre[TT] = all of them
    unsigned short start[TT];  // the run's first record inside tile t's window
    u32 w[4];                  // scan scratch (one per wave of a 256-thread workgroup)
    u32 tile_shift;            // log2(hits per tile)
    u32 base;                  // added to every index (a contiguous range: one "tile" at `base`)
    u32 flags;                 // 1: some run of a hot bucket carries a delta the set did not predict
};

// A contiguous range [lo, lo + n) of the record buffer as a view (k_bkt_tiny; n < 2^32).  Thread 0 only writes; the
// caller places the barrier.
template <int TT>
__device__ __forceinline__ void view_single(BucketView<TT>& V, u32 lo, u32 n) {
    for (u32 t = threadIdx.x; t <= (u32)TT; t += blockDim.x) V.pre[t] = t ? n : 0u;
    for (u32 t = threadIdx.x; t < (u32)TT; t += blockDim.x) V.start[t] = 0;
    if (threadIdx.x == 0) {
        V.tile_shift = 0;
        V.base = lo;
        V.flags = 0;
    }
}

// Row `row` of the runs matrix -> view, in two steps so that the row's latency can run under something else:
// view_load requests the thread's share of the row, view_scan (all 256 threads; ends past a barrier) builds the view.
// The caller must have a barrier between the last use of the previous view and view_scan.
template <int TT>
struct ViewRow {
    u32 v[TT / AP_BLOCK];
};
template <int TT>
__device__ __forceinline__ ViewRow<TT> view_load(const u32* __restrict__ row, u32 ntiles) {
    constexpr int PER = TT / AP_BLOCK;
    static_assert(PER == 1 || PER == 4, "one run or one dwordx4 of runs per thread");
    const u32 tid = threadIdx.x;
    ViewRow<TT> r;
    if (PER == 1) {
        r.v[0] = row[tid < ntiles ? tid : 0u];  // (unconditional: a load inside a branch is waited for where the branch ends)
    } else {
        const uint4 q = *reinterpret_cast<const uint4*>(row + (tid * 4 < ntiles ? tid * 4 : 0u));
        r.v[0] = q.x;
        r.v[PER > 1 ? 1 : 0] = q.y;
        r.v[PER > 2 ? 2 : 0] = q.z;
        r.v[PER > 3 ? 3 : 0] = q.w;
    }
    return r;
}
template <int TT>
__device__ __forceinline__ void view_scan(BucketView<TT>& V, const ViewRow<TT>& r, u32 ntiles, u32 tile_shift) {
    constexpr int PER = TT / AP_BLOCK;
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    if (tid == 0) {
        V.tile_shift = tile_shift;
        V.base = 0;
        V.flags = 0;
    }
    u32 c[PER], sum = 0;
    bool mis = false;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const u32 t = tid * PER + q;
        c[q] = t < ntiles ? run_count(r.v[q]) : 0u;
        sum += c[q];
        mis = mis || (c[q] && (r.v[q] & RUN_FLAG));
    }
    u32 inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 o = __shfl_up(inc, off);
        if ((int)lane >= off) inc += o;
    }
    if (lane == 63) V.w[w] = inc;
    __syncthreads();
    u32 ex = inc - sum;
#pragma unroll
    for (u32 ww = 0; ww < (u32)(AP_BLOCK / 64); ++ww)
        if (ww < w) ex += V.w[ww];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const u32 t = tid * PER + q;
        V.pre[t] = ex;
        V.start[t] = (unsigned short)run_start(r.v[q]);
        ex += c[q];
    }
    if (tid == AP_BLOCK - 1) V.pre[TT] = ex;
    if (mis) atomicOr(&V.flags, 1u);
    __syncthreads();
}
template <int TT>
__device__ __forceinline__ void view_build(BucketView<TT>& V, const u32* __restrict__ row, u32 ntiles, u32 tile_shift) {
    const ViewRow<TT> r = view_load<TT>(row, ntiles);
    view_scan(V, r, ntiles, tile_shift);
}

template <int TT>
__device__ __forceinline__ u32 view_total(const BucketView<TT>& V) {
    return V.pre[TT];
}

// Index (into the partitioned batch) of the bucket's p-th hit in trace order; p < view_total.
template <int TT>
__device__ __forceinline__ u32 view_src(const BucketView<TT>& V, u32 p) {
    u32 t = 0;
#pragma unroll
    for (u32 s = TT / 2; s >= 1; s >>= 1)
        if (V.pre[t + s] <= p) t += s;  // the largest t with pre[t] <= p: its run holds position p
    return V.base + (t << V.tile_shift) + (u32)V.start[t] + (p - V.pre[t]);
}

// Hash bucket a workgroup of k_bkt_apply takes: workgroups go round-robin over the 8 XCDs, and neighbouring buckets'
// runs are neighbours in every tile's window, so each XCD is given a CONTIGUOUS range of buckets — the 64-byte
// segments two neighbouring runs share are then fetched into one L2, once.
__device__ __forceinline__ u32 bucket_of_workgroup(u32 wg, u32 nb) {
    if (nb < 8u) return wg;
    return (wg & 7u) * (nb >> 3) + (wg >> 3);
}

// ---------------------------------------------------------------------------------------------
// k_bkt_part
// ---------------------------------------------------------------------------------------------
template <int STEPS>
__global__ __launch_bounds__(PT_BLOCK) void k_bkt_part(const Cell* __restrict__ table, u32 log2cap, u64 seed,
                                                       const Hit* __restrict__ hits, u32 n,
                                                       const LimitDev* __restrict__ limits, u32 n_limits, u32 bk_log2,
                                                       u32 ntiles, u32 run_tt, BatchScratch* bs,
                                                       const HotSet* __restrict__ hot, u32 check_simple,
                                                       uint8_t* __restrict__ verdict_fill, int32_t* __restrict__ first_fill,
                                                       BHit* __restrict__ b_hits, u32* __restrict__ runs,
                                                       HotItems* __restrict__ items, HotSet* __restrict__ hot_next) {
    // wave-private counters, [PT_WAVES][nbt] — dynamic, sized by the launch for the batch's bucket count
    extern __shared__ __align__(16) unsigned short s_cnt[];
    __shared__ u32 s_base[BKT_MAX];
    __shared__ u32 s_w[PT_WAVES];
    __shared__ u64 s_hot_key[HOT_HASH];
    __shared__ u32 s_hot_idx[HOT_HASH];
    __shared__ u32 s_hot_d[HOT_MAX], s_hot_limit[HOT_MAX], s_hot_flg[HOT_MAX];
    __shared__ u32 s_mis[HOT_MAX];
    const u32 tid = threadIdx.x;
    if (blockIdx.x == ntiles) {
        // ---- one extra workgroup: k_bkt_apply's hot work items, from the hot set alone (nothing here waits for a tile) ----
        const u32 nh = hot->n < (u32)HOT_MAX ? hot->n : (u32)HOT_MAX;
        u32 nk = 0;
        if (tid < nh) {
            const u32 want = (hot->cnt[tid] + HOT_CHUNK - 1) / HOT_CHUNK;
            nk = want < 1u ? 1u : (want > HOT_NK_MAX ? HOT_NK_MAX : want);
        }
        u32 all;
        const u32 c0 = block_excl_scan_1024(nk, s_w, all);
        u32* s_c0 = s_base;
        if (tid < (u32)HOT_MAX) s_c0[tid] = c0;
        if (tid == (u32)HOT_MAX) {
            s_c0[HOT_MAX] = all;
            items->n = all;
            hot_next->n = 0;  // k_bkt_apply appends the keys it promotes or keeps
        }
        __syncthreads();
        // item c belongs to the LAST bucket h with chunk0[h] <= c (a bucket without items shares its successor's chunk0,
        // so it is never the last one)
        for (u32 c = tid; c < all; c += PT_BLOCK) {
            u32 a = 0, b = HOT_MAX;  // invariant: chunk0[a] <= c < chunk0[b]
            while (b - a > 1) {
                const u32 m = (a + b) >> 1;
                if (s_c0[m] <= c) a = m;
                else b = m;
            }
            HotItem it{};
            it.key = hot->key[a];
            it.hb_k = a | ((c - s_c0[a]) << 16);
            it.nk = s_c0[a + 1] - s_c0[a];
            it.d = hot->d[a];
            it.limit = hot->limit[a];
            it.flg = hot->flg[a];
            items->it[c] = it;
        }
        return;
    }
    // Workgroups go round-robin over the 8 XCDs, each with its own L2; every XCD takes a CONTIGUOUS run of tiles, so
    // the 4-byte entries neighbouring tiles write into one row of `runs` meet in one L2 and leave it as whole lines.
    u32 tile;
    {
        const u32 x = blockIdx.x & 7u, j = blockIdx.x >> 3, per = ntiles >> 3, rem = ntiles & 7u;
        tile = x * per + (x < rem ? x : rem) + j;
    }
    Status* st = &bs->st;
    const u32 lane = tid & 63u, w = tid >> 6;
    const u32 nb = 1u << bk_log2;
    const u32 nbt = nb + HOT_MAX;
    constexpr u32 TILE = PT_BLOCK * STEPS;
    const u32 tbase = tile * TILE;
    const u32 wbase = tbase + w * (64 * STEPS);
    uint4 raw[STEPS];
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
        const u32 i = wbase + u * 64 + lane;
        if (i < n) raw[u] = *reinterpret_cast<const uint4*>(hits + i);
    }
    {
        const u32 hd = tid < (u32)HOT_MAX ? hot->d[tid] : 0u;
        const u32 hl = tid < (u32)HOT_MAX ? hot->limit[tid] : 0u;
        const u32 hf = tid < (u32)HOT_MAX ? hot->flg[tid] : 0u;
        hot_table_build(hot, seed, s_hot_key, s_hot_idx);
        if (tid < (u32)HOT_MAX) {
            s_hot_d[tid] = hd;
            s_hot_limit[tid] = hl;
            s_hot_flg[tid] = hf;
            s_mis[tid] = 0;
        }
    }
    for (u32 b = tid; b < nbt; b += PT_BLOCK) {
#pragma unroll
        for (int ww = 0; ww < PT_WAVES; ++ww) s_cnt[ww * nbt + b] = 0;
    }
    __syncthreads();
    // ---- every hit: validated, its bin, its rank among the wave's hits of that bin (trace order) ----------------
    unsigned short rank[STEPS];
    unsigned short dig[STEPS];
    const u64 lt = (1ull << lane) - 1ull;
    u32 err = 0;
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
        const u32 i = wbase + u * 64 + lane;
        const bool ok = i < n;
        const u64 valid = __ballot(ok);
        const u64 key = ((u64)raw[u].y << 32) | raw[u].x;
        const u32 limit = raw[u].z, delta = raw[u].w;
        u32 d = 0;
        if (ok) {
            // the only place the batch is validated (k_bkt_apply refuses to touch the table when `err` is set): limit
            // id range, reserved keys, and (in_memory.rs:106-107) a simple counter must already have its cell
            if ((limit & ~SIMPLE_FLAG) >= n_limits) err |= ERRBIT_BAD_LIMIT;
            else if (key >= TAG_TOMB) err |= ERRBIT_RESERVED_KEY;
            else if ((limit & SIMPLE_FLAG) && check_simple) {
                u32 dummy = 0;
                u32 slot = slot_of(key, seed, log2cap);
                slot = probe_from<PM_LOOKUP>(const_cast<Cell*>(table), log2cap, slot, table[slot].tag, key, limit, limits, 0ull,
                                             st, dummy);
                if (slot == SLOT_INVALID) err |= ERRBIT_MISSING_SIMPLE;
            }
            const u64 hh = fmix64(key ^ seed);
            const int hi = hot_lookup(s_hot_key, s_hot_idx, key, hh);
            bool deny = false;
            if (hi >= 0) {
                d = nb + (u32)hi;
                // a hit that is not what the set predicts: the bucket is replayed hit by hit (plain store: every writer writes 1)
                if (delta != s_hot_d[hi] || limit != s_hot_limit[hi]) s_mis[hi] = 1;
                deny = (s_hot_flg[hi] & HOT_FLG_DENY) != 0u;
            } else {
                d = bucket_of_hash(hh, bk_log2);
            }
            // The default answer, written here as coalesced stores while the batch streams by: "admitted" — or, for the
            // hits of a hot key whose window was full when it was picked, "limited".  k_bkt_apply then scatters only the
            // answers that differ (a saturated hot key: none).
            if (verdict_fill) {
                verdict_fill[i] = deny ? 1 : 0;
                if (first_fill) first_fill[i] = deny ? (int32_t)i : -1;
            }
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
    if (err) atomicOr(&st->err, err);
    __syncthreads();
    // ---- per bin: the waves' exclusive offsets, the tile's count; exclusive scan over the bins = the runs' starts ----
    {
        const u32 b0 = 3 * tid;  // (nbt <= 2560 < 3 * 1024)
        u32 c[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const u32 b = b0 + q;
            u32 acc = 0;
            if (b < nbt) {
#pragma unroll
                for (int ww = 0; ww < PT_WAVES; ++ww) {
                    const u32 x = s_cnt[ww * nbt + b];
                    s_cnt[ww * nbt + b] = (unsigned short)acc;
                    acc += x;
                }
            }
            c[q] = acc;
        }
        u32 all;
        u32 ex = block_excl_scan_1024(c[0] + c[1] + c[2], s_w, all);
        u32* row = runs + tile;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const u32 b = b0 + q;
            if (b < nbt) {
                s_base[b] = ex;
                row[(size_t)b * run_tt] = run_pack(ex, c[q], b >= nb && s_mis[b - nb] != 0u);
            }
            ex += c[q];
        }
    }
    __syncthreads();
    // ---- the records, into the tile's own window: bins in order, every bin in trace order -----------------------
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
        const u32 i = wbase + u * 64 + lane;
        if (i < n) {
            const u32 d = dig[u];
            const u32 dst = tbase + s_base[d] + s_cnt[w * nbt + d] + rank[u];
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 recv = {raw[u].x, raw[u].y, raw[u].w, i | (limit_fold(raw[u].z) << 24)};
            *reinterpret_cast<u32x4*>(b_hits + dst) = recv;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_bkt_part_c: the same kernel in a COMPACT shape — 512 threads (8 waves, 8 x 64 hits per wave for a 4096-hit tile) and
// ~41 KB of LDS instead of 1024 threads and ~78 KB: small enough to be resident on a CU BESIDE the five workgroups of
// the replay (5 x 21.5 KB), where k_bkt_part's workgroups had to wait for two of them to leave (the replay's in-pipeline
// stretch of the partition, 24 -> 40 us, is mostly that wait).  What shrank: wave-private counters for 8 waves, the hot
// set as a 16-bit slot table over its keys (part_role's scheme), 16-bit bin starts.
// ---------------------------------------------------------------------------------------------
constexpr int PC_BLOCK = 512;
constexpr int PC_WAVES = PC_BLOCK / 64;

__device__ __forceinline__ u32 block_excl_scan_512(u32 v, u32* s_w, u32& total) {
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
    for (u32 ww = 0; ww < (u32)PC_WAVES; ++ww) {
        const u32 x = s_w[ww];
        if (ww < w) woff += x;
        tot += x;
    }
    __syncthreads();
    total = tot;
    return woff + inc - v;
}

template <int STEPS>
__global__ __launch_bounds__(PC_BLOCK) void k_bkt_part_c(const Cell* __restrict__ table, u32 log2cap, u64 seed,
                                                         const Hit* __restrict__ hits, u32 n,
                                                         const LimitDev* __restrict__ limits, u32 n_limits, u32 bk_log2,
                                                         u32 ntiles, u32 run_tt, BatchScratch* bs,
                                                         const HotSet* __restrict__ hot, u32 check_simple,
                                                         uint8_t* __restrict__ verdict_fill, int32_t* __restrict__ first_fill,
                                                         BHit* __restrict__ b_hits, u32* __restrict__ runs,
                                                         HotItems* __restrict__ items, HotSet* __restrict__ hot_next) {
    extern __shared__ __align__(16) unsigned short s_cnt[];  // [PC_WAVES][nbt]
    __shared__ unsigned short s_base[BKT_MAX];
    __shared__ u64 s_hkey[HOT_MAX];
    __shared__ unsigned short s_hslot[HOT_SLOTS];
    __shared__ u32 s_hot_d[HOT_MAX], s_hot_limit[HOT_MAX];
    __shared__ u32 s_mis[HOT_MAX / 32];
    __shared__ u32 s_w[PC_WAVES];
    const u32 tid = threadIdx.x;
    if (blockIdx.x == ntiles) {
        // ---- one extra workgroup: the replay's hot work items, from the hot set alone ----------------------------------
        const u32 nh = hot->n < (u32)HOT_MAX ? hot->n : (u32)HOT_MAX;
        u32 nk = 0;
        if (tid < nh) {
            const u32 want = (hot->cnt[tid] + HOT_CHUNK - 1) / HOT_CHUNK;
            nk = want < 1u ? 1u : (want > HOT_NK_MAX ? HOT_NK_MAX : want);
        }
        u32 all;
        const u32 c0 = block_excl_scan_512(nk, s_w, all);
        u32* s_c0 = reinterpret_cast<u32*>(s_hkey);  // chunk0[0 .. HOT_MAX]
        s_c0[tid] = c0;
        if (tid == 0) {
            s_c0[HOT_MAX] = all;
            items->n = all;
            hot_next->n = 0;  // the replay appends the keys it promotes or keeps
        }
        __syncthreads();
        for (u32 c = tid; c < all; c += PC_BLOCK) {
            u32 a = 0, b = HOT_MAX;  // invariant: chunk0[a] <= c < chunk0[b]
            while (b - a > 1) {
                const u32 m = (a + b) >> 1;
                if (s_c0[m] <= c) a = m;
                else b = m;
            }
            HotItem it{};
            it.key = hot->key[a];
            it.hb_k = a | ((c - s_c0[a]) << 16);
            it.nk = s_c0[a + 1] - s_c0[a];
            it.d = hot->d[a];
            it.limit = hot->limit[a];
            it.flg = hot->flg[a];
            items->it[c] = it;
        }
        return;
    }
    u32 tile;
    {
        const u32 x = blockIdx.x & 7u, j = blockIdx.x >> 3, per = ntiles >> 3, rem = ntiles & 7u;
        tile = x * per + (x < rem ? x : rem) + j;
    }
    Status* st = &bs->st;
    const u32 lane = tid & 63u, w = tid >> 6;
    const u32 nb = 1u << bk_log2;
    const u32 nbt = nb + HOT_MAX;
    constexpr u32 TILE = PC_BLOCK * STEPS;
    const u32 tbase = tile * TILE;
    const u32 wbase = tbase + w * (64 * STEPS);
    uint4 raw[STEPS];
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
        const u32 i = wbase + u * 64 + lane;
        raw[u] = *reinterpret_cast<const uint4*>(hits + (i < n ? i : n - 1));
    }
    // ---- LDS: empty tables; the hot set (one slot per key: a key listed twice keeps the smaller index) -------------
    const u32 nh = hot->n < (u32)HOT_MAX ? hot->n : (u32)HOT_MAX;
    const u64 hk = hot->key[tid < nh ? tid : 0u];
    const u32 hd = hot->d[tid < nh ? tid : 0u], hl = hot->limit[tid < nh ? tid : 0u], hf = hot->flg[tid < nh ? tid : 0u];
    for (u32 b = tid; b < (u32)(HOT_SLOTS / 2); b += PC_BLOCK) reinterpret_cast<u32*>(s_hslot)[b] = 0xFFFFFFFFu;
    for (u32 b = tid; b < (PC_WAVES * nbt + 1) / 2; b += PC_BLOCK) reinterpret_cast<u32*>(s_cnt)[b] = 0u;
    if (tid < (u32)(HOT_MAX / 32)) s_mis[tid] = 0u;
    s_hkey[tid] = tid < nh ? hk : TAG_EMPTY;
    s_hot_d[tid] = hd;
    s_hot_limit[tid] = hl;
    __syncthreads();
    if (tid < nh) {
        const u32 mine = tid | ((hf & HOT_FLG_DENY) ? 0x8000u : 0u);
        u32 s = (u32)(fmix64(hk ^ seed) >> 8) & (HOT_SLOTS - 1);
        for (;;) {
            u32* wp = reinterpret_cast<u32*>(s_hslot) + (s >> 1);
            const u32 sh = (s & 1u) * 16u;
            u32 old = *wp;
            bool done = false;
            for (;;) {
                const u32 x = (old >> sh) & 0xFFFFu;
                const bool same = x != 0xFFFFu && s_hkey[x & 0x7FFFu] == hk;
                if (x != 0xFFFFu && !same) break;       // another key's slot: probe on
                if (same && (x & 0x7FFFu) <= tid) {     // the same key with a smaller index is there already
                    done = true;
                    break;
                }
                const u32 prev = atomicCAS(wp, old, (old & ~(0xFFFFu << sh)) | (mine << sh));
                if (prev == old) {
                    done = true;
                    break;
                }
                old = prev;
            }
            if (done) break;
            s = (s + 1) & (HOT_SLOTS - 1);
        }
    }
    __syncthreads();
    // ---- every hit: validated, its bin, its rank among the wave's hits of that bin (trace order) ----------------
    unsigned short rank[STEPS];
    unsigned short dig[STEPS];  // bin | (default answer "limited") << 15
    const u64 lt = (1ull << lane) - 1ull;
    u32 err = 0;
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
        const u32 i = wbase + u * 64 + lane;
        const bool ok = i < n;
        const u64 valid = __ballot(ok);
        const u64 key = ((u64)raw[u].y << 32) | raw[u].x;
        const u32 limit = raw[u].z, delta = raw[u].w;
        u32 d = 0;
        bool deny = false;
        if (ok) {
            if ((limit & ~SIMPLE_FLAG) >= n_limits) err |= ERRBIT_BAD_LIMIT;
            else if (key >= TAG_TOMB) err |= ERRBIT_RESERVED_KEY;
            else if ((limit & SIMPLE_FLAG) && check_simple) {
                u32 dummy = 0;
                u32 slot = slot_of(key, seed, log2cap);
                slot = probe_from<PM_LOOKUP>(const_cast<Cell*>(table), log2cap, slot, table[slot].tag, key, limit, limits, 0ull,
                                             st, dummy);
                if (slot == SLOT_INVALID) err |= ERRBIT_MISSING_SIMPLE;
            }
            const u64 hh = fmix64(key ^ seed);
            int hi = -1;
            u32 q = (u32)(hh >> 8) & (HOT_SLOTS - 1);
            for (;;) {
                const u32 x = s_hslot[q];
                if (x == 0xFFFFu) break;
                if (s_hkey[x & 0x7FFFu] == key) {
                    hi = (int)(x & 0x7FFFu);
                    deny = (x >> 15) != 0u;
                    break;
                }
                q = (q + 1) & (HOT_SLOTS - 1);
            }
            if (hi >= 0) {
                d = nb + (u32)hi;
                // a hit that is not what the set predicts: the bucket is replayed hit by hit
                if (delta != s_hot_d[hi] || limit != s_hot_limit[hi]) atomicOr(&s_mis[hi >> 5], 1u << (hi & 31));
            } else {
                d = bucket_of_hash(hh, bk_log2);
            }
            if (verdict_fill) {
                verdict_fill[i] = deny ? 1 : 0;
                if (first_fill) first_fill[i] = deny ? (int32_t)i : -1;
            }
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
    if (err) atomicOr(&st->err, err);
    __syncthreads();
    // ---- per bin: the waves' exclusive offsets, the tile's count; exclusive scan over the bins = the runs' starts ----
    {
        constexpr int PER = 5;  // (nbt <= 2560 = 5 * 512)
        const u32 b0 = PER * tid;
        u32 c[PER], sum = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const u32 b = b0 + q;
            u32 acc = 0;
            if (b < nbt) {
#pragma unroll
                for (int ww = 0; ww < PC_WAVES; ++ww) {
                    const u32 x = s_cnt[ww * nbt + b];
                    s_cnt[ww * nbt + b] = (unsigned short)acc;
                    acc += x;
                }
            }
            c[q] = acc;
            sum += acc;
        }
        u32 all;
        u32 ex = block_excl_scan_512(sum, s_w, all);
        u32* row = runs + tile;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const u32 b = b0 + q;
            if (b < nbt) {
                s_base[b] = (unsigned short)ex;
                const bool mis = b >= nb && ((s_mis[(b - nb) >> 5] >> ((b - nb) & 31)) & 1u);
                row[(size_t)b * run_tt] = run_pack(ex, c[q], mis);
            }
            ex += c[q];
        }
    }
    __syncthreads();
    // ---- the records, into the tile's own window: bins in order, every bin in trace order -----------------------
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
        const u32 i = wbase + u * 64 + lane;
        if (i < n) {
            const u32 d = dig[u];
            const u32 dst = tbase + (u32)s_base[d] + s_cnt[w * nbt + d] + rank[u];
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 recv = {raw[u].x, raw[u].y, raw[u].w, i | (limit_fold(raw[u].z) << 24)};
            *reinterpret_cast<u32x4*>(b_hits + dst) = recv;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same partition as a ROLE of 256-thread workgroups, for k_bkt_step (rl_apply.hpp): the partition of batch j + 1
// runs INSIDE the launch that replays batch j — one stream, one launch per step, no event between the two.  (As a
// kernel of its own on a second stream, k_bkt_part cost the step 15 us: the replay slowed down beside its 16-wave
// workgroups and the apply stream sat idle ~12 us between two launches; with no partition at all a step was 36.5 us.)
// To live beside the replay's workgroups a tile's workgroup gets the replay's budget: 4 waves, 64 VGPRs, ~21 KB of LDS —
//   * the tile is walked TWICE (count, then place): a wave owns a quarter of the tile and walks it in 64-hit steps, so
//     the records are never held in registers; the second walk re-reads them from the L2;
//   * the hot set's LDS table holds 32-bit fingerprints + indices; a match is confirmed with one 16-byte load of the
//     set's packed record (HotSet::rec: key | predicted delta + default answer | predicted limit);
//   * wave-private counters [4][bins] and the bins' starts are 16-bit.
// Batches with more than PART_ROLE_BINS bins (2048 hash buckets: more than 2 M hits) take k_bkt_part.
// ---------------------------------------------------------------------------------------------
constexpr int PART_ROLE_BINS = 1024 + HOT_MAX;
constexpr int PR_WAVES = AP_BLOCK / 64;
struct PartParams {
    const Cell* table;
    u32 log2cap;
    u64 seed;
    const Hit* hits;
    u32 n;
    const LimitDev* limits;
    u32 n_limits, bk_log2, ntiles, run_tt, tile_shift, check_simple;
    BatchScratch* bs;
    const HotSet* hot;
    uint8_t* verdict_fill;
    int32_t* first_fill;
    BHit* b_hits;
    u32* runs;
    HotItems* items;
    HotSet* hot_next;
};
struct PartLds {
    unsigned short cnt[PR_WAVES][PART_ROLE_BINS];  // wave-private counts, then every (wave, bin)'s first record in the tile's window
    u64 hkey[HOT_MAX];                             // the hot set's keys, by index
    unsigned short hslot[HOT_SLOTS];               // hash slot -> index in the hot set | (default answer "limited") << 15; 0xFFFF = empty
    u32 mis[HOT_MAX / 32];                         // hot buckets that saw a hit the set does not predict
    u32 w[PR_WAVES];
};
static_assert(sizeof(PartLds) <= 163840 / 7, "seven workgroups of k_bkt_step per CU");

// -1, or the key's index in the hot set; `deny`: the default answer of its hits is "limited".
__device__ __forceinline__ int part_hot_lookup(const PartLds& L, u64 key, u64 hh, bool& deny) {
    u32 q = (u32)(hh >> 8) & (HOT_SLOTS - 1);
    for (;;) {
        const u32 x = L.hslot[q];
        if (x == 0xFFFFu) return -1;
        if (L.hkey[x & 0x7FFFu] == key) {
            deny = (x >> 15) != 0u;
            return (int)(x & 0x7FFFu);
        }
        q = (q + 1) & (HOT_SLOTS - 1);
    }
}

// cnt[w][d] += v (16-bit counters, two to an LDS word: the sum never carries, a tile has at most 16384 hits)
__device__ __forceinline__ void cnt_add(PartLds& L, u32 w, u32 d, u32 v) {
    u32* wp = reinterpret_cast<u32*>(&L.cnt[w][d & ~1u]);
    __hip_atomic_fetch_add(wp, v << ((d & 1u) * 16u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ u32 cnt_add_rtn(PartLds& L, u32 w, u32 d, u32 v) {
    u32* wp = reinterpret_cast<u32*>(&L.cnt[w][d & ~1u]);
    const u32 sh = (d & 1u) * 16u;
    return (__hip_atomic_fetch_add(wp, v << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> sh) & 0xFFFFu;
}

// One tile of the batch.  All 256 threads of the workgroup; `L` is the workgroup's LDS, reinterpreted.
__device__ __forceinline__ void part_role(PartLds& L, const PartParams& Q, u32 tile, u64* trace) {
    constexpr int GS = 4;   // walk 1: steps whose records are in flight together
    constexpr int GS2 = 2;  // walk 2 (more state per step)
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const u32 nb = 1u << Q.bk_log2, nbt = nb + HOT_MAX;
    const u32 tile_hits = 1u << Q.tile_shift;
    const u32 ngroups = tile_hits >> (8 + 2);  // 64-hit steps per wave, in groups of GS
    const u32 tbase = tile << Q.tile_shift;
    const u32 wbase = tbase + w * (tile_hits >> 2);
    const Hit* __restrict__ hits = Q.hits;
    const u32 n = Q.n;
    Status* st = &Q.bs->st;
#define RL_PSTAMP(k)                                                                    \
    do {                                                                                \
        if (trace && tid == 0) trace[(size_t)blockIdx.x * 8 + (k)] = wall_clock64();    \
    } while (0)
    RL_PSTAMP(0);
    // the first records are requested before anything else
    uint4 buf[GS];
#pragma unroll
    for (int k = 0; k < GS; ++k) {
        const u32 i = wbase + k * 64 + lane;
        buf[k] = *reinterpret_cast<const uint4*>(hits + (i < n ? i : n - 1));
    }
    // ---- LDS: empty tables; the hot set --------------------------------------------------------------------------
    for (u32 b = tid; b < (u32)(HOT_SLOTS / 2); b += AP_BLOCK) reinterpret_cast<u32*>(L.hslot)[b] = 0xFFFFFFFFu;
    for (u32 b = tid; b < (u32)(PR_WAVES * PART_ROLE_BINS / 2); b += AP_BLOCK) reinterpret_cast<u32*>(&L.cnt[0][0])[b] = 0u;
    if (tid < (u32)(HOT_MAX / 32)) L.mis[tid] = 0u;
    const u32 nh = Q.hot->n < (u32)HOT_MAX ? Q.hot->n : (u32)HOT_MAX;
    u64 hk[HOT_MAX / AP_BLOCK];
    u32 hf[HOT_MAX / AP_BLOCK];
#pragma unroll
    for (int q = 0; q < HOT_MAX / AP_BLOCK; ++q) {
        const u32 h = tid + q * AP_BLOCK;
        hk[q] = Q.hot->key[h < nh ? h : 0u];
        hf[q] = Q.hot->flg[h < nh ? h : 0u];
        L.hkey[h] = h < nh ? hk[q] : TAG_EMPTY;
    }
    RL_PSTAMP(1);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < HOT_MAX / AP_BLOCK; ++q) {
        const u32 h = tid + q * AP_BLOCK;
        if (h >= nh) continue;
        // A key listed twice gets ONE slot, and the smaller of its indices — every workgroup must map a key to the same
        // bucket: a slot whose index names the same key is taken over with the smaller index (rare: a 16-bit CAS loop).
        const u32 mine = h | ((hf[q] & HOT_FLG_DENY) ? 0x8000u : 0u);
        u32 s = (u32)(fmix64(hk[q] ^ Q.seed) >> 8) & (HOT_SLOTS - 1);
        for (;;) {
            u32* wp = reinterpret_cast<u32*>(L.hslot) + (s >> 1);
            const u32 sh = (s & 1u) * 16u;
            u32 old = *wp;
            bool done = false;
            for (;;) {
                const u32 x = (old >> sh) & 0xFFFFu;
                const bool same = x != 0xFFFFu && L.hkey[x & 0x7FFFu] == hk[q];
                if (x != 0xFFFFu && !same) break;             // another key's slot: probe on
                if (same && (x & 0x7FFFu) <= h) {              // the same key with a smaller index is there already
                    done = true;
                    break;
                }
                const u32 prev = atomicCAS(wp, old, (old & ~(0xFFFFu << sh)) | (mine << sh));
                if (prev == old) {
                    done = true;
                    break;
                }
                old = prev;
            }
            if (done) break;
            s = (s + 1) & (HOT_SLOTS - 1);
        }
    }
    __syncthreads();
    RL_PSTAMP(2);
    const u64 lt = (1ull << lane) - 1ull;
    const u32 nbits = (Q.bk_log2 > 9u ? Q.bk_log2 : 9u) + 1u;
    // ---- walk 1: validate, count per (wave, bin) — LDS only ----------------------------------------------------------
    u32 err = 0;
    for (u32 g = 0; g < ngroups; ++g) {
        uint4 cur[GS];
#pragma unroll
        for (int k = 0; k < GS; ++k) cur[k] = buf[k];
        if (g + 1 < ngroups) {
#pragma unroll
            for (int k = 0; k < GS; ++k) {
                const u32 i = wbase + ((g + 1) * GS + k) * 64 + lane;
                buf[k] = *reinterpret_cast<const uint4*>(hits + (i < n ? i : n - 1));
            }
        }
#pragma unroll
        for (int k = 0; k < GS; ++k) {
            const u32 i = wbase + (g * GS + k) * 64 + lane;
            const bool ok = i < n;
            const u64 valid = __ballot(ok);
            const u64 key = ((u64)cur[k].y << 32) | cur[k].x;
            const u32 limit = cur[k].z;
            u32 d = 0;
            if (ok) {
                // the only place the batch is validated (the replay refuses to touch the table when `err` is set): limit
                // id range, reserved keys, and (in_memory.rs:106-107) a simple counter must already have its cell
                if ((limit & ~SIMPLE_FLAG) >= Q.n_limits) err |= ERRBIT_BAD_LIMIT;
                else if (key >= TAG_TOMB) err |= ERRBIT_RESERVED_KEY;
                else if ((limit & SIMPLE_FLAG) && Q.check_simple) {
                    u32 dummy = 0;
                    u32 slot = slot_of(key, Q.seed, Q.log2cap);
                    slot = probe_from<PM_LOOKUP>(const_cast<Cell*>(Q.table), Q.log2cap, slot, Q.table[slot].tag, key, limit,
                                                 Q.limits, 0ull, st, dummy);
                    if (slot == SLOT_INVALID) err |= ERRBIT_MISSING_SIMPLE;
                }
                const u64 hh = fmix64(key ^ Q.seed);
                bool deny;
                const int hi = part_hot_lookup(L, key, hh, deny);
                d = hi >= 0 ? nb + (u32)hi : bucket_of_hash(hh, Q.bk_log2);
            }
            const u64 m = match_digit(d, nbits, valid);
            // one LDS atomic per distinct bin of the step, nothing waits for it: a read-modify-write here would chain
            // every step of the walk behind the LDS latency of the one before it (22 us per walk instead of 6)
            if (ok && (m & lt) == 0ull) cnt_add(L, w, d, (u32)__popcll(m));
        }
    }
    if (err) atomicOr(&st->err, err);
    RL_PSTAMP(3);
    // the second walk starts where the first one did
#pragma unroll
    for (int k = 0; k < GS; ++k) {
        const u32 i = wbase + k * 64 + lane;
        buf[k] = *reinterpret_cast<const uint4*>(hits + (i < n ? i : n - 1));
    }
    __syncthreads();
    // ---- per bin: the tile's count; exclusive scan over the bins = the runs' starts; cnt[w][b] <- where wave w's
    //      hits of bin b start inside the tile's window ---------------------------------------------------------------
    constexpr int PER = PART_ROLE_BINS / AP_BLOCK;  // 6 consecutive bins per thread
    {
        const u32 b0 = PER * tid;
        u32 tot[PER], sum = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const u32 b = b0 + q;
            tot[q] = 0;
            if (b < nbt) {
#pragma unroll
                for (int ww = 0; ww < PR_WAVES; ++ww) tot[q] += (u32)L.cnt[ww][b];
            }
            sum += tot[q];
        }
        u32 inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 o = __shfl_up(inc, off);
            if ((int)lane >= off) inc += o;
        }
        if (lane == 63) L.w[w] = inc;
        __syncthreads();
        u32 ex = inc - sum;
#pragma unroll
        for (u32 ww = 0; ww < (u32)PR_WAVES; ++ww)
            if (ww < w) ex += L.w[ww];
        u32* row = Q.runs + tile;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const u32 b = b0 + q;
            if (b < nbt) {
                row[(size_t)b * Q.run_tt] = run_pack(ex, tot[q], false);  // (a hot bin may get its flag after walk 2)
                u32 acc = ex;
#pragma unroll
                for (int ww = 0; ww < PR_WAVES; ++ww) {
                    const u32 x = L.cnt[ww][b];
                    L.cnt[ww][b] = (unsigned short)acc;
                    acc += x;
                }
            }
            ex += tot[q];
        }
    }
    __syncthreads();
    RL_PSTAMP(4);
    // ---- walk 2: the records, into the tile's own window (bins in order, every bin in trace order), and the default
    //      answer of every request — "admitted", or "limited" for the hits of a hot key whose window was full when the
    //      set was picked — as coalesced stores; the replay then scatters only the answers that differ.  A hot key's
    //      hits are also checked against what the set predicts (its packed record: one load per hot hit, consumed at the
    //      end of the group) -------------------------------------------------------------------------------------------
    for (u32 g = 0; g < ngroups * (GS / GS2); ++g) {
        uint4 cur[GS2];
#pragma unroll
        for (int k = 0; k < GS2; ++k) cur[k] = buf[k];
        if (g + 1 < ngroups * (GS / GS2)) {
#pragma unroll
            for (int k = 0; k < GS2; ++k) {
                const u32 i = wbase + ((g + 1) * GS2 + k) * 64 + lane;
                buf[k] = *reinterpret_cast<const uint4*>(hits + (i < n ? i : n - 1));
            }
        }
        int hi[GS2];
        uint2 pred[GS2];
        u32 old[GS2], before[GS2];
        bool dn[GS2];
        // the group's steps first take their places (one LDS atomic per distinct bin of a step, issued back to back: the
        // LDS serves a wave's operations in order, so step k + 1 sees step k's) ...
#pragma unroll
        for (int k = 0; k < GS2; ++k) {
            const u32 i = wbase + (g * GS2 + k) * 64 + lane;
            const bool ok = i < n;
            const u64 valid = __ballot(ok);
            const u64 key = ((u64)cur[k].y << 32) | cur[k].x;
            u32 d = 0;
            bool deny = false;
            hi[k] = -1;
            if (ok) {
                const u64 hh = fmix64(key ^ Q.seed);
                hi[k] = part_hot_lookup(L, key, hh, deny);
                d = hi[k] >= 0 ? nb + (u32)hi[k] : bucket_of_hash(hh, Q.bk_log2);
            }
            // (unconditional: a load inside a branch is waited for where the branch ends)
            const uint4 r = Q.hot->rec[hi[k] >= 0 ? hi[k] : 0];
            pred[k] = make_uint2(r.z & 0x7FFFFFFFu, r.w);
            const u64 m = match_digit(d, nbits, valid);
            const bool leader = ok && (m & lt) == 0ull;
            const u32 got = leader ? cnt_add_rtn(L, w, d, (u32)__popcll(m)) : 0u;
            // every lane of the bin reads its leader's answer (the lowest lane of the match)
            old[k] = __shfl(got, m ? __ffsll((long long)m) - 1 : 0);
            before[k] = (u32)__popcll(m & lt);
            dn[k] = deny;
        }
        // ... then the records go out
#pragma unroll
        for (int k = 0; k < GS2; ++k) {
            const u32 i = wbase + (g * GS2 + k) * 64 + lane;
            if (i < n) {
                const u32 dst = tbase + old[k] + before[k];
                *reinterpret_cast<uint4*>(Q.b_hits + dst) =
                    make_uint4(cur[k].x, cur[k].y, cur[k].w, i | (limit_fold(cur[k].z) << 24));
                if (Q.verdict_fill) {
                    Q.verdict_fill[i] = dn[k] ? 1 : 0;
                    if (Q.first_fill) Q.first_fill[i] = dn[k] ? (int32_t)i : -1;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < GS2; ++k)  // a hit that is not what the set predicts: the bucket is replayed hit by hit
            if (hi[k] >= 0 && (cur[k].w != pred[k].x || cur[k].z != pred[k].y)) atomicOr(&L.mis[hi[k] >> 5], 1u << (hi[k] & 31));
    }
    RL_PSTAMP(5);
    __syncthreads();
    // ---- hot bins that saw a hit the set does not predict: flag the tile's run (rare) -------------------------------
    for (u32 h = tid; h < (u32)HOT_MAX; h += AP_BLOCK) {
        if ((L.mis[h >> 5] >> (h & 31)) & 1u) {
            // (the word was stored above by another thread of this workgroup, barriers in between: an L2 atomic, not a
            // read-modify-write through this thread's L1)
            atomicOr(Q.runs + tile + (size_t)(nb + h) * Q.run_tt, RUN_FLAG);
        }
    }
    RL_PSTAMP(6);
#undef RL_PSTAMP
}

// The hot work items of the batch being partitioned, from the hot set alone (one workgroup; see k_bkt_part's extra one).
__device__ __forceinline__ void plan_role(PartLds& L, const PartParams& Q) {
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const HotSet* __restrict__ hot = Q.hot;
    const u32 nh = hot->n < (u32)HOT_MAX ? hot->n : (u32)HOT_MAX;
    constexpr int PER = HOT_MAX / AP_BLOCK;
    u32 nk[PER], sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const u32 h = tid * PER + q;
        nk[q] = 0;
        if (h < nh) {
            const u32 want = (hot->cnt[h] + HOT_CHUNK - 1) / HOT_CHUNK;
            nk[q] = want < 1u ? 1u : (want > HOT_NK_MAX ? HOT_NK_MAX : want);
        }
        sum += nk[q];
    }
    u32 inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 o = __shfl_up(inc, off);
        if ((int)lane >= off) inc += o;
    }
    if (lane == 63) L.w[w] = inc;
    __syncthreads();
    u32 ex = inc - sum, all = 0;
#pragma unroll
    for (u32 ww = 0; ww < (u32)PR_WAVES; ++ww) {
        if (ww < w) ex += L.w[ww];
        all += L.w[ww];
    }
    u32* s_c0 = reinterpret_cast<u32*>(L.hkey);  // chunk0[0 .. HOT_MAX]
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        s_c0[tid * PER + q] = ex;
        ex += nk[q];
    }
    if (tid == AP_BLOCK - 1) {
        s_c0[HOT_MAX] = all;
        Q.items->n = all;
        Q.hot_next->n = 0;  // the replay appends the keys it promotes or keeps
    }
    __syncthreads();
    for (u32 c = tid; c < all; c += AP_BLOCK) {
        u32 a = 0, b = HOT_MAX;  // invariant: chunk0[a] <= c < chunk0[b]
        while (b - a > 1) {
            const u32 m = (a + b) >> 1;
            if (s_c0[m] <= c) a = m;
            else b = m;
        }
        HotItem it{};
        it.key = hot->key[a];
        it.hb_k = a | ((c - s_c0[a]) << 16);
        it.nk = s_c0[a + 1] - s_c0[a];
        it.d = hot->d[a];
        it.limit = hot->limit[a];
        it.flg = hot->flg[a];
        Q.items->it[c] = it;
    }
}

}  // namespace rl
