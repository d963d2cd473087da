// rl_apply.hpp — k_bkt_step: the bucket replay of the single-counter hot path (rl_part.hpp: the single-pass partition
// that feeds it, k_bkt_part / _c / _l, and the bucket view through which a workgroup finds its hits; rl_bucket.hpp: the
// shared record / hot-set types).  The algorithm: reference limitador/src/storage/in_memory.rs:72-156 applied hit by hit
// in trace order, one read and one write per touched counter cell.
//
// Machine mapping (as measured on MI355X, rounds 3-5):
//   * ONE workgroup per hash bucket (1024 for a 1 M-hit batch) + 256 workgroups that walk the hot keys' work items, not
//     persistent: a bucket's dependent chain (runs row -> view -> records -> home cells -> LDS aggregation -> verdicts
//     -> write-back: 6.5 us + 5.3 us per 256-hit round) is hidden by OTHER workgroups — 80 VGPRs (no scratch) and
//     21.5 KB of DYNAMIC LDS keep five of them resident per CU, so all 1280 of a launch are resident at once and the
//     launch's span is its slowest workgroup (four beside a partition workgroup of the next batch: 82-96 VGPRs x two
//     waves per SIMD).
//   * the limit table and the hot work items are read from global memory (L2-resident, a few hundred bytes): no LDS
//     copies, no row limit on the limit table (16-bit limit ids in LDS; engines with more than 32768 rows take the
//     32-bit instantiation).
//   * one 64-bit LDS atomic per hit carries the round's delta sum AND the per-wave hit counts; deltas >= 2^23 cannot
//     share the 32-bit sum field and send their key through the sequential replay, which is exact for everything.
//   * default verdicts are written by the partition as coalesced stores; the replay only stores the answers that differ.
//   * completion: the last workgroup out (tickets sharded per XCD: one ticket word taken by 2048 workgroups was 9 us of
//     serialised atomics) stores the batch's sequence number + status as ONE 16-byte store into host-mapped memory.
//   * k_bkt_step can also carry the partition of the NEXT batch as a role of the same launch (RL_FUSE=1: one stream, one
//     launch per step — the default for engines of <= 256 k hits per batch; rl_part.hpp part_role).
//
//   k_bkt_step       the batch's hash buckets + hot work items (+ the partition role)
//   k_bkt_tiny       a batch of <= TINY_MAX hits is one bucket: validate + replay in ONE launch
//   k_bkt_count_new  dry run: how many cells would the batch create (all-or-nothing under TABLE_FULL)
#pragma once
#include "rl_part.hpp"

namespace rl {

// (the replay's scattered stores written THROUGH — sc1 on the cells' 8-byte write-backs, on the one-byte verdicts too: the
// boundary of two dependent kernels is 1.1-2.0 us behind write-through stores against 3.4-4.7 behind 16 MB left dirty,
// profiles/r04h_kernel_gap.txt — built as compile-time variants in round 5 and never faster in the pipeline:
// scripts/exp/patches/step_write_through.patch)
__device__ __forceinline__ void store_cell_value(u64* p, u64 v) { *p = v; }
__device__ __forceinline__ void store_verdict(uint8_t* p, u32 v) { *p = (uint8_t)v; }

constexpr u32 AP2_BIG_DELTA = 1u << 23;  // 512 hits x (2^23 - 1) < 2^32: the round's sum fits 32 bits

// NARROW: the cell's limit attribute in 16 bits (id < 2^15 | simple flag).  With it the workgroup's LDS is 19.5 KB and
// EIGHT workgroups fit a CU (160 KB), i.e. all 2048 buckets of a full-size batch are resident at once — with 22 KB
// (seven per CU) the last 256 buckets were a second generation that started when the first workgroups ended and ran at
// one workgroup per CU: the SQ counters of that kernel show 17 resident waves per CU on average of 28 possible
// (scripts/exp/v1.sh).  Engines with more than 32768 limit rows take the wide form.
template <bool NARROW>
struct LimitWord {
    typedef u32 type;
};
template <>
struct LimitWord<true> {
    typedef unsigned short type;
};
template <bool NARROW>
__device__ __forceinline__ typename LimitWord<NARROW>::type pack_limit(u32 l) {
    if (NARROW) return (typename LimitWord<NARROW>::type)((l & 0x7FFFu) | ((l & SIMPLE_FLAG) ? 0x8000u : 0u));
    return (typename LimitWord<NARROW>::type)l;
}
template <bool NARROW>
__device__ __forceinline__ u32 unpack_limit(typename LimitWord<NARROW>::type x) {
    if (NARROW) return ((u32)x & 0x7FFFu) | (((u32)x & 0x8000u) ? SIMPLE_FLAG : 0u);
    return (u32)x;
}

template <int HPT, int ENT_LOG2, bool NARROW = false, int TT = TT_SMALL>
struct Apply2Lds {
    static constexpr int E = 1 << ENT_LOG2;
    static constexpr int R = AP_BLOCK * HPT;          // hits per decide/commit round
    static constexpr int KEEP = E * 3 / 4 - R;        // rebuild the LDS cells before a round if more are live
    static constexpr bool narrow = NARROW;
    static_assert(KEEP > 0, "the LDS hash must hold one round at load <= 3/4");
    u64 key[E];
    u64 run[E];    // value the next hit reads; for 0-second windows: the last admitted delta
    u64 agg[E];    // this round: hits per wave (4 x 8 bits, bits 32..63) | sum of deltas < 2^23 (bits 0..31)
    u32 slot[E];
    u32 dmax[E];   // this round: largest delta
    u32 flags[E];  // EF_* | hits absorbed << EF_COUNT_SHIFT
    typename LimitWord<NARROW>::type limit[E];  // the CELL's limit attribute (pack_limit)
    unsigned short h_ent[R];  // the round's hits -> LDS cell (for the sequential replay of slow cells)
    BucketView<TT> V;         // where the bucket's hits are (rl_part.hpp)
    u32 n_ent;
    u32 bucket_len;
    u32 promote_ok;
    u32 any_slow;
    u32 n_created;
    u32 n_keep;
};

struct Apply2Args {
    Cell* table;
    u32 log2cap;
    u64 seed;
    const BHit* b_hits;
    const Hit* hits;  // the caller's batch: only read for the limit id of a key that has no cell yet
    const LimitDev* limits;
    u64 now;
    uint8_t* verdict;
    int32_t* first_limited;
    Status* st;
    HotSet* hot_next;
    const HotItems* items;            // the hot buckets' work items (k_bkt_part)
    const u32* runs;                  // [bin][run_tt] the tiles' runs (k_bkt_part)
    u32 run_tt, ntiles, tile_shift, nb;
    u32 hot_threshold;
    u32 hot_long;     // a hash bucket of at least this many hits is long BECAUSE of a key (twice the batch's mean bucket)
    u32 sparse_out;   // verdict[] / first_limited[] already say "admitted" (k_bkt_hist): only denials are stored
    u32* hot_arrive;  // [HOT_MAX] work items of a hot bucket that have read the key's cell; zero between kernels
};

__device__ __forceinline__ LimitDev limit_row2(const Apply2Args& A, u32 limit) {
    // 16 bytes, L1/L2-resident; ids were range-checked by k_bkt_hist
    const uint4 v = *reinterpret_cast<const uint4*>(&A.limits[limit & ~SIMPLE_FLAG]);
    LimitDev L;
    L.max_value = ((u64)v.y << 32) | v.x;
    L.window_us = ((u64)v.w << 32) | v.z;
    return L;
}

__device__ __forceinline__ u32 agg_count(u64 agg) {
    const u32 c4 = (u32)(agg >> 32);
    return (c4 & 0xFFu) + ((c4 >> 8) & 0xFFu) + ((c4 >> 16) & 0xFFu) + (c4 >> 24);
}

// Write the dirty LDS cells back; optionally rebuild the LDS hash keeping only the hot entries.
// Ends with every thread past a barrier when `rebuild`.
template <class LDS>
__device__ __forceinline__ void apply2_commit(LDS& S, const Apply2Args& A, bool rebuild) {
    constexpr int E = LDS::E;
    constexpr int PER = E / AP_BLOCK;
    constexpr int KEEP = LDS::KEEP;
    constexpr bool NARROW = LDS::narrow;
    u64 k_key[PER], k_run[PER];
    u32 k_slot[PER], k_flags[PER];
    typename LimitWord<NARROW>::type k_limit[PER];
    bool keep[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const u32 e = threadIdx.x + q * AP_BLOCK;
        keep[q] = false;
        const u64 key = S.key[e];
        if (key == TAG_EMPTY) continue;
        u32 f = S.flags[e];
        // promote: the key absorbed hot_threshold hits, or it is what made this bucket long
        if (!rebuild && !(f & EF_BAD) && S.promote_ok &&
            ((f >> EF_COUNT_SHIFT) >= A.hot_threshold ||
             ((f >> EF_COUNT_SHIFT) >= A.hot_threshold / 4 && S.bucket_len >= A.hot_long))) {
            // (predicted delta 1: a key whose hits carry another one is replayed in its first hot batch, and the last
            // work item of that bucket then hands the delta it saw to the next set)
            hot_append(A.hot_next, key, f >> EF_COUNT_SHIFT, 1u, unpack_limit<NARROW>(S.limit[e]));
        }
        if (f & EF_DIRTY) {
            Cell* c = &A.table[S.slot[e]];
            store_cell_value(&c->value, S.run[e]);
            if (f & EF_EXPIRED) {  // (the limit row is only read when the window is reset: no dependent load in front of
                                   // the usual write-back)
                const LimitDev L = limit_row2(A, unpack_limit<NARROW>(S.limit[e]));
                c->expiry = A.now + L.window_us;  // update_if_expired, atomic_expiring_value.rs:87-99
                // the window is open again — except a 0-second one, which is expired at every read
                if (L.window_us != 0) f &= ~EF_EXPIRED;
            }
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
    if (!rebuild) return;  // end of the bucket: the workgroup is done with its LDS cells
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
    const bool reinsert = S.n_keep <= (u32)KEEP;
    if (reinsert) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            if (!keep[q]) continue;
            u32 e = (u32)(fmix64(k_key[q] ^ A.seed) >> 20) & (E - 1);
            while (atomicCAS(&S.key[e], TAG_EMPTY, k_key[q]) != TAG_EMPTY) e = (e + 1) & (E - 1);
            S.run[e] = k_run[q];
            S.slot[e] = k_slot[q];
            S.limit[e] = k_limit[q];
            S.flags[e] = k_flags[q];
        }
    }
    if (threadIdx.x == 0) S.n_ent = reinsert ? S.n_keep : 0u;
    __syncthreads();
}

// One decide/commit round over the hits [first, first + n_items) of the partitioned batch
// (n_items <= R), in trace order.  Ends with every thread past a barrier.
template <class LDS>
__device__ __forceinline__ void apply2_round(LDS& S, const Apply2Args& A, u32 first, u32 n_items, u32 n_next,
                                             uint4 (&hnext)[LDS::R / AP_BLOCK]) {
    constexpr int E = LDS::E;
    constexpr int HPT = LDS::R / AP_BLOCK;
    constexpr bool NARROW = LDS::narrow;
    const u32 tid = threadIdx.x;
    const u32 lane = tid & 63u, w = tid >> 6;
    const u64 lt = (1ull << lane) - 1ull;
    BHit h[HPT];
    u32 hslot[HPT], climit[HPT];
    u64 ctag[HPT], cvalue[HPT], cexpiry[HPT];
    u32 idx[HPT], ent[HPT];
    bool ok[HPT], creator[HPT], leader[HPT];
    // ---- the round's inputs: the hits, then every hit's home cell (one 32-byte read) --------------
    // Both loads are UNCONDITIONAL (a lane past the end of the round repeats the round's last hit): a load inside
    // a divergent `if` is waited for where the branch ends, and the home cells are not needed before phase B —
    // phase A's LDS work runs under their latency.
    // The round's records were requested a round ago (`hnext`, by apply2_bucket for the first round); the NEXT round's
    // are requested now — after the cell reads, so that finding them in the bucket view (a binary search in LDS) also
    // runs under the cells' latency — and with the rounds of a bucket only the cell reads are a dependent trip to HBM
    // per round.  (The next round's cells are not read ahead: a rebuild between the rounds writes cells back.)
    uint4 ca[HPT], cb[HPT];
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
        const u32 p = tid * HPT + u;
        ok[u] = p < n_items;
        creator[u] = leader[u] = false;
        ent[u] = 0;
        h[u].key = ((u64)hnext[u].y << 32) | hnext[u].x;
        h[u].delta = hnext[u].z;
        h[u].idx_tag = hnext[u].w;
    }
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
        idx[u] = ok[u] ? (h[u].idx_tag & 0xFFFFFFu) : 0u;
        hslot[u] = slot_of(h[u].key, A.seed, A.log2cap);
        const Cell* c = &A.table[hslot[u]];
        ca[u] = *reinterpret_cast<const uint4*>(c);
        cb[u] = reinterpret_cast<const uint4*>(c)[1];  // expiry, limit
    }
    if (n_next) {  // (block-uniform)
#pragma unroll
        for (int u = 0; u < HPT; ++u) {
            const u32 p = tid * HPT + u;
            hnext[u] = *reinterpret_cast<const uint4*>(A.b_hits + view_src(S.V, first + n_items + (p < n_next ? p : n_next - 1)));
        }
    }
    // ---- A: find or claim the key's LDS cell, add this hit to the round's aggregates ------------
    u32 n_new = 0;
    u64 first_key[HPT];
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
        ent[u] = ok[u] ? (u32)(fmix64(h[u].key ^ A.seed) >> 20) & (E - 1) : 0u;
        first_key[u] = ok[u] ? S.key[ent[u]] : 0ull;
    }
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
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
            e = (e + 1) & (E - 1);
            prev = S.key[e];
        }
        ent[u] = e;
    }
    u64 before[HPT];
    u32 seen_dmax[HPT];
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
        const u32 p = tid * HPT + u;
        before[u] = ~0ull;
        seen_dmax[u] = 0;
        if (!ok[u]) {
            S.h_ent[p] = (unsigned short)ENT_NONE;
            continue;
        }
        const u32 d = h[u].delta;
        // ONE atomic: this wave's hit count (8 bits per wave) and the delta (32-bit sum field)
        before[u] = atomicAdd(&S.agg[ent[u]], (1ull << (32 + 8 * w)) | (u64)(d < AP2_BIG_DELTA ? d : 0u));
        seen_dmax[u] = S.dmax[ent[u]];
        S.h_ent[p] = (unsigned short)ent[u];
    }
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
        if (!ok[u]) continue;
        if (h[u].delta > seen_dmax[u]) atomicMax(&S.dmax[ent[u]], h[u].delta);  // only a new maximum is an atomic
        leader[u] = (before[u] >> 32) == 0ull;  // the round's first arriver on this key
    }
    for (int off = 32; off > 0; off >>= 1) n_new += __shfl_down(n_new, off);
    if (lane == 0 && n_new) atomicAdd(&S.n_ent, n_new);
    // ---- B: the claimer of a new LDS cell resolves the counter cell -------------------------------
    u32 created = 0;
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
        if (!creator[u]) continue;
        const u32 e = ent[u];
        u32 slot = hslot[u];
        ctag[u] = ((u64)ca[u].y << 32) | ca[u].x;
        cvalue[u] = ((u64)ca[u].w << 32) | ca[u].z;
        cexpiry[u] = ((u64)cb[u].y << 32) | cb[u].x;
        climit[u] = cb[u].z;
        u64 value = cvalue[u], expiry = cexpiry[u];
        u32 cl = climit[u];
        if (ctag[u] != h[u].key) {
            // Not at home: probe on, fetching whole cells so that a match needs no further read; the
            // limit id (caller's batch) is only read when the cell has to be created (in_memory.rs:122-127).
            const u32 mask = (1u << A.log2cap) - 1u;
            u64 tag = ctag[u];
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
                            c->expiry = A.now + limit_row2(A, hl).window_us;
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
                tag = ((u64)a.y << 32) | a.x;
                if (tag == h[u].key) {
                    value = ((u64)a.w << 32) | a.z;
                    expiry = ((u64)b.y << 32) | b.x;
                    cl = b.z;
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
        S.limit[e] = pack_limit<NARROW>(cl);
        S.flags[e] = (expired ? EF_EXPIRED : 0u) | (slot == SLOT_INVALID ? EF_BAD : 0u);
    }
    if (created) atomicAdd(&S.n_created, created);
    __syncthreads();
    // ---- C: verdicts -------------------------------------------------------------------------------
    uint8_t v[HPT];
    bool need_rank[HPT], slow[HPT];
    u64 room[HPT];
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
        v[u] = 1;
        need_rank[u] = slow[u] = false;
        room[u] = 0;
        if (!ok[u]) continue;
        const u32 e = ent[u];
        const u32 el = unpack_limit<NARROW>(S.limit[e]);
        if (limit_fold(el) != (h[u].idx_tag >> 24)) {
            atomicOr(&S.flags[e], EF_BAD);
            atomicOr(&A.st->err, ERRBIT_KEY_LIMIT);
            continue;
        }
        const LimitDev Lu = limit_row2(A, el);
        const u64 run = S.run[e], agg = S.agg[e];
        const u64 sum = agg & 0xFFFFFFFFull;
        const u64 cnt = agg_count(agg);
        const u32 dm = S.dmax[e];
        const u64 d = h[u].delta;
        u64 tot;
        const bool ovf = __builtin_add_overflow(run, sum, &tot);
        if (Lu.window_us == 0 || ovf || dm >= AP2_BIG_DELTA) {
            slow[u] = true;  // every read sees an expired cell / the sum wraps or left the sum field: replay
        } else if (tot <= Lu.max_value) {
            v[u] = 0;
        } else if (run + d > Lu.max_value) {
            v[u] = 1;
        } else if (sum == cnt * (u64)dm) {  // all deltas of the round equal (and > 0 here)
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
        for (int u = HPT - 1; u >= 0; --u)
            if (need_rank[u]) {
                have = true;
                my_e = ent[u];
            }
        const u64 pend = __ballot(have);
        if (!pend) break;
        const u32 e0 = __shfl(my_e, __ffsll((long long)pend) - 1);
        u32 before_lane = 0;
#pragma unroll
        for (int u = 0; u < HPT; ++u) before_lane += (u32)__popcll(__ballot(ok[u] && ent[u] == e0) & lt);
        const u32 c4 = (u32)(S.agg[e0] >> 32);
        u32 pre = 0;
        for (u32 ww = 0; ww < w; ++ww) pre += (c4 >> (8 * ww)) & 0xFFu;
        u32 mine = 0;
#pragma unroll
        for (int u = 0; u < HPT; ++u) {
            if (need_rank[u] && ent[u] == e0) {
                v[u] = (u64)(pre + before_lane + mine) < room[u] ? 0 : 1;
                need_rank[u] = false;
            }
            if (ok[u] && ent[u] == e0) ++mine;
        }
    }
    __syncthreads();
    // ---- slow entries: one lane replays the round in trace order, reference arithmetic -----------
    // (rare: the lane reads each hit's record again and stores its verdict itself — no LDS copies of the round's
    // deltas and verdicts, which is what lets eight workgroups share a CU)
    if (S.any_slow) {
        if (tid == 0) {
            for (u32 p = 0; p < n_items; ++p) {
                const u32 e = S.h_ent[p];
                const u32 f = S.flags[e];
                if (!(f & EF_SLOW) || (f & EF_BAD)) continue;
                const LimitDev Le = limit_row2(A, unpack_limit<NARROW>(S.limit[e]));
                const BHit hp = load_bhit(A.b_hits, view_src(S.V, first + p));
                const u64 d = hp.delta;
                const u64 cur = Le.window_us == 0 ? 0ull : S.run[e];
                const u64 sum = cur + d;  // wraps like the reference's release build (in_memory.rs:88)
                const bool adm = sum <= Le.max_value;
                if (adm) {
                    S.run[e] = Le.window_us == 0 ? d : sum;
                    S.flags[e] = f | EF_DIRTY | (Le.window_us == 0 ? EF_EXPIRED : 0u);
                }
                const u32 i = hp.idx_tag & 0xFFFFFFu;
                if (!adm || !A.sparse_out) {
                    A.verdict[i] = adm ? 0 : 1;
                    if (A.first_limited) A.first_limited[i] = adm ? -1 : (int32_t)i;
                }
            }
        }
        __syncthreads();
    }
    // ---- D: the round's first arriver of each key folds the round into `run` --------------------
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
        if (!ok[u]) continue;
        const u32 i = idx[u];
        if (!slow[u] && (v[u] || !A.sparse_out)) {  // (a slow hit's verdict was stored by the replay above)
            store_verdict(&A.verdict[i], v[u]);
            if (A.first_limited) A.first_limited[i] = v[u] ? (int32_t)i : -1;
        }
        if (!leader[u]) continue;
        const u32 e = ent[u];
        u32 f = S.flags[e];
        const u64 agg = S.agg[e];
        const u64 cnt = agg_count(agg);
        if (!(f & (EF_SLOW | EF_BAD))) {
            const LimitDev Le = limit_row2(A, unpack_limit<NARROW>(S.limit[e]));
            const u64 run = S.run[e], sum = agg & 0xFFFFFFFFull;
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
        S.agg[e] = 0;
        S.dmax[e] = 0;
    }
    if (tid == 0) S.any_slow = 0;
    __syncthreads();
}

// Empty LDS cells (all threads; the caller places the barrier).
template <class LDS>
__device__ __forceinline__ void apply2_clear(LDS& S) {
    constexpr int E = LDS::E;
    for (u32 e = threadIdx.x; e < (u32)E; e += AP_BLOCK) {
        S.key[e] = TAG_EMPTY;
        S.agg[e] = 0;
        S.dmax[e] = 0;
        S.flags[e] = 0;
    }
}

// Positions [lo, hi) of the bucket S.V describes, in trace order, by one workgroup.  The LDS cells
// must be empty on entry; they are NOT cleared on exit (callers that replay a second bucket clear
// them again).
template <class LDS>
__device__ __forceinline__ void apply2_bucket(LDS& S, const Apply2Args& A, u32 lo, u32 hi) {
    constexpr int R = LDS::R;
    constexpr int KEEP = LDS::KEEP;
    if (threadIdx.x == 0) {
        S.n_ent = 0;
        S.any_slow = 0;
        S.bucket_len = hi - lo;
    }
    __syncthreads();
    uint4 hnext[R / AP_BLOCK];
    {
        const u32 n0 = (hi - lo) < (u32)R ? (hi - lo) : (u32)R;
#pragma unroll
        for (int u = 0; u < R / AP_BLOCK; ++u) {
            const u32 p = threadIdx.x * (R / AP_BLOCK) + u;
            hnext[u] = *reinterpret_cast<const uint4*>(A.b_hits + view_src(S.V, lo + (p < n0 ? p : n0 - 1)));
        }
    }
    for (u32 first = lo; first < hi; first += R) {
        if (first != lo && S.n_ent > (u32)KEEP) {  // block-uniform (read after the previous round's barrier)
            __syncthreads();
            apply2_commit(S, A, true);
        }
        const u32 n_items = (hi - first) < (u32)R ? (hi - first) : (u32)R;
        const u32 left = hi - first - n_items;
        apply2_round(S, A, first, n_items, left < (u32)R ? left : (u32)R, hnext);
    }
    apply2_commit(S, A, false);
}

// Hot buckets (one key each; bins nb.. of the partition): stable partition + one key per bucket means a hit's position
// in the bucket IS its trace-order rank on the key, so with one delta value d for the whole bucket the reference admits
// exactly the first (max - value_at(now)) / d positions (in_memory.rs:85-102 applied hit by hit) and any number of
// workgroups can decide their share of the positions in parallel.  Work item k of the bucket's nk (HotItem, from
// k_bkt_part) takes the 1024-position chunks k, k + nk, ...: whatever the bucket's real length, the items cover it.
// k_bkt_part wrote a DEFAULT answer for the bucket's hits — "limited" if the key's window was full when the set was
// picked, else "admitted" — so only the positions whose answer differs are visited at all: for a key that stays
// saturated (or stays far from its limit) an item reads the key's cell and its row of runs, and nothing else.
// Every item's workgroup reads the key's cell itself.  The cell must not change before every item of the bucket has read
// it, so the workgroups count themselves in (hot_arrive) once their waves' reads have returned and the LAST one in applies
// AtomicExpiringValue::update for the bucket's admitted hits — or, when the bucket cannot be decided from positions (a
// tile saw a hit the set did not predict — another delta, another limit id —, a 0-second window, a value near 2^64, a
// missing simple cell: every item comes to that same conclusion from the same unchanged cell and the same runs), replays
// the bucket with the general bucket code.  Nobody waits for anybody: no spinning, no ordering between workgroups beyond
// the counter.  The last one in also keeps the key in the next hot set if it still earns it.
template <class LDS>
__device__ __forceinline__ void apply2_hot_item(LDS& S, const Apply2Args& A, u32 c) {
    constexpr int PER = HOT_CHUNK / AP_BLOCK;
    constexpr int TTV = decltype(S.V)::tiles;
    const u32 tid = threadIdx.x;
    const uint4 ia = *reinterpret_cast<const uint4*>(&A.items->it[c]);
    const uint4 ib = reinterpret_cast<const uint4*>(&A.items->it[c])[1];
    const u64 key = ((u64)ia.y << 32) | ia.x;
    const u32 hb = ia.z & 0xFFFFu, k = ia.z >> 16, nk = ia.w, d_pred = ib.x, limit = ib.y;
    const bool dv = (ib.z & HOT_FLG_DENY) != 0u;
    const bool limit_known = limit != HOT_LIMIT_UNKNOWN;
    // requested together: the bucket's row of runs, the limit's row, the key's cell
    const ViewRow<TTV> vrow = view_load<TTV>(A.runs + (size_t)(A.nb + hb) * A.run_tt, A.ntiles);
    const LimitDev L = limit_row2(A, limit_known ? limit : 0u);
    // ---- the key's cell as it is before this batch (every lane reads the same addresses) --------------------
    const u32 mask = (1u << A.log2cap) - 1u;
    u32 slot = slot_of(key, A.seed, A.log2cap);
    u64 value = 0, expiry = 0;
    u32 cl = 0;
    bool found = false;
    for (u32 step = 0; step <= mask; ++step) {
        const Cell* cp = &A.table[slot];
        const uint4 a = *reinterpret_cast<const uint4*>(cp);
        const uint4 b = reinterpret_cast<const uint4*>(cp)[1];
        const u64 tag = ((u64)a.y << 32) | a.x;
        if (tag == key) {
            value = ((u64)a.w << 32) | a.z;
            expiry = ((u64)b.y << 32) | b.x;
            cl = b.z;
            found = true;
            break;
        }
        if (tag == TAG_EMPTY) break;
        slot = (slot + 1) & mask;
    }
    __syncthreads();  // (the previous user of S.V and S.n_keep is done)
    view_scan(S.V, vrow, A.ntiles, A.tile_shift);
    // ---- count this workgroup in: every wave's read of the cell has returned (they are past the barriers above).
    //      The answer is only needed at the end: it travels under the position loop.
    u32 arrived = 0;
    if (tid == 0) arrived = atomicAdd(&A.hot_arrive[hb], 1u);
    const u32 total = view_total(S.V);
    const bool mis = S.V.flags != 0u;
    const bool expired = found && expiry <= A.now;
    const u64 s = (found && !expired) ? value : 0ull;  // value_at(now), atomic_expiring_value.rs:19-24
    const bool fast = total && !mis && limit_known && (!found || cl == limit) && L.window_us != 0 && s < (1ull << 62) &&
                      (found || !(limit & SIMPLE_FLAG));
    const u64 room = s > L.max_value ? 0ull : (d_pred ? (L.max_value - s) / d_pred : ~0ull);
    const u32 n_adm = room < (u64)total ? (u32)room : total;  // the reference admits the first n_adm positions
    if (fast) {
        // positions whose answer is not the default one
        const u32 dlo = dv ? 0u : n_adm, dhi = dv ? n_adm : total;
        for (u32 cb = k * HOT_CHUNK; cb < dhi; cb += nk * HOT_CHUNK) {
            if (cb + HOT_CHUNK <= dlo) continue;
            u32 h_tag[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const u32 j = cb + u * AP_BLOCK + tid;
                h_tag[u] = A.b_hits[view_src(S.V, j < total ? j : total - 1)].idx_tag;
            }
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const u32 j = cb + u * AP_BLOCK + tid;
                if (j < dlo || j >= dhi) continue;
                const u32 i = h_tag[u] & 0xFFFFFFu;
                store_verdict(&A.verdict[i], dv ? 0u : 1u);
                if (A.first_limited) A.first_limited[i] = dv ? -1 : (int32_t)i;
            }
        }
    }
    if (tid == 0) S.n_keep = arrived;
    __syncthreads();
    const bool last = S.n_keep + 1u == nk;  // (block-uniform)
    __syncthreads();  // S.n_keep is free again
    if (!last) return;
    if (tid == 0) {
        atomicExch(&A.hot_arrive[hb], 0u);  // for the next kernel
        // still hot?  Then with what this batch showed: the delta and the limit id its hits carry, and whether its
        // window is full now (the next default answer).
        if (total >= A.hot_threshold) {
            u32 d_next = d_pred, l_next = found ? cl : limit;
            if (mis || !limit_known) {
                const BHit h0 = load_bhit(A.b_hits, view_src(S.V, 0));
                d_next = h0.delta;
                if (!found) l_next = A.hits[h0.idx_tag & 0xFFFFFFu].limit;
            }
            hot_append(A.hot_next, key, total, d_next, l_next, (fast && room <= (u64)total) ? HOT_FLG_DENY : 0u);
        }
    }
    if (!total) return;
    if (!fast) {
        // the reference's arithmetic hit by hit, by this workgroup; every answer is stored if the default was "limited"
        Apply2Args B = A;
        if (dv) B.sparse_out = 0;
        apply2_clear(S);
        if (tid == 0) S.promote_ok = 0;  // kept or dropped by count (above), not promoted
        __syncthreads();
        apply2_bucket(S, B, 0, total);
        __syncthreads();
        return;
    }
    if (tid == 0) {
        // AtomicExpiringValue::update for the admitted hits
        u32 wslot = found ? slot : SLOT_INVALID;
        bool reset = expired;
        if (!found) {  // first touch creates the cell (in_memory.rs:122-127), verdict or not
            u32 created = 0;
            wslot = slot_of(key, A.seed, A.log2cap);
            wslot = probe_from<PM_CHECK>(A.table, A.log2cap, wslot, A.table[wslot].tag, key, limit, A.limits, A.now, A.st,
                                         created);
            if (created) atomicAdd(&A.st->n_inserted, created);
            reset = false;
        }
        if (n_adm && wslot != SLOT_INVALID) {
            Cell* cell = &A.table[wslot];
            store_cell_value(&cell->value, s + (u64)n_adm * d_pred);
            if (reset) cell->expiry = A.now + L.window_us;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_bkt_count_new: how many cells would this (already partitioned) batch create?  Same walk as
// k_bkt_apply up to the point where a new key's probe chain ends at an empty slot — counted, not
// claimed; nothing is written to the table, no verdict is produced.  The host runs it only when the
// cheap bound "every hit could be a new key" does not fit the table, to decide BEFORE anything is
// applied whether the batch fits (all-or-nothing under RL_ERR_TABLE_FULL).  Exact per bucket; a key
// that is dropped by a rebuild of the LDS cells of a long bucket and met again is counted twice
// (an over-estimate, never an under-estimate).
// ---------------------------------------------------------------------------------------------
template <int ENT_LOG2, int TT>
__global__ __launch_bounds__(AP_BLOCK) void k_bkt_count_new(const Cell* __restrict__ table, u32 log2cap, u64 seed,
                                                            const BHit* __restrict__ b_hits,
                                                            const u32* __restrict__ runs, u32 run_tt, u32 ntiles,
                                                            u32 tile_shift, u32 nb, u32* __restrict__ n_new_out) {
    constexpr int E = 1 << ENT_LOG2;
    constexpr u32 KEEP = E * 3 / 4 - AP_BLOCK;
    __shared__ u64 s_key[E];
    __shared__ BucketView<TT> V;
    __shared__ u32 s_n_ent, s_new;
    const u32 tid = threadIdx.x, G = gridDim.x;
    if (tid == 0) s_new = 0;
    u32 my_new = 0;
    for (u32 bin = blockIdx.x; bin < nb + (u32)HOT_MAX; bin += G) {
        __syncthreads();
        view_build(V, runs + (size_t)bin * run_tt, ntiles, tile_shift);
        const u32 hi = view_total(V);
        if (!hi) continue;
        for (u32 e = tid; e < (u32)E; e += AP_BLOCK) s_key[e] = TAG_EMPTY;
        if (tid == 0) s_n_ent = 0;
        __syncthreads();
        for (u32 first = 0; first < hi; first += AP_BLOCK) {
            if (s_n_ent > KEEP) {  // block-uniform (read after a barrier)
                __syncthreads();
                for (u32 e = tid; e < (u32)E; e += AP_BLOCK) s_key[e] = TAG_EMPTY;
                if (tid == 0) s_n_ent = 0;
                __syncthreads();
            }
            const u32 j = first + tid;
            bool creator = false;
            u64 key = 0;
            if (j < hi) {
                key = load_bhit(b_hits, view_src(V, j)).key;
                u32 e = (u32)(fmix64(key ^ seed) >> 20) & (E - 1);
                for (;;) {
                    u64 prev = s_key[e];
                    if (prev == TAG_EMPTY) prev = atomicCAS(&s_key[e], TAG_EMPTY, key);
                    if (prev == TAG_EMPTY) {
                        creator = true;
                        break;
                    }
                    if (prev == key) break;
                    e = (e + 1) & (E - 1);
                }
            }
            if (creator) {
                atomicAdd(&s_n_ent, 1u);
                const u32 mask = (1u << log2cap) - 1u;
                u32 slot = slot_of(key, seed, log2cap);
                for (u32 step = 0; step <= mask; ++step) {
                    const u64 tag = table[slot].tag;
                    if (tag == key) break;
                    if (tag == TAG_EMPTY) {
                        ++my_new;
                        break;
                    }
                    slot = (slot + 1) & mask;
                }
            }
            __syncthreads();
        }
    }
    for (int off = 32; off > 0; off >>= 1) my_new += __shfl_down(my_new, off);
    if ((tid & 63u) == 0 && my_new) atomicAdd(&s_new, my_new);
    __syncthreads();
    if (tid == 0 && s_new) atomicAdd(n_new_out, s_new);
}

// k_bkt_apply's body (the __global__ entry below only adds the launch bounds).
struct ApplyParams {
    Cell* table;
    u32 log2cap;
    u64 seed;
    const BHit* b_hits;
    const Hit* hits;
    const u32* runs;
    u32 run_tt, ntiles, tile_shift, nb;
    const HotItems* items;
    const LimitDev* limits;
    u64 now;
    uint8_t* verdict;
    int32_t* first_limited;
    BatchScratch* bs;
    BatchScratch* bs_zero;
    Status* host_status;
    u32 done_seq;
    HotSet* hot_next;
    u32 hot_threshold;
    u32* hot_arrive;
    u32 sparse_out;
    u32 hot_long;
    u64* trace;  // RL_APPLY_TRACE: [workgroup][8] wall-clock stamps (100 MHz) of the phases below, else null
};

// G: workgroups of the launch that replay (the first G of the grid).
template <int HPT, int ENT_LOG2, bool NARROW, int TT>
__device__ __forceinline__ void bkt_apply_body(const ApplyParams& P, const u32 G) {
    // The workgroup's LDS is DYNAMIC (the launch passes sizeof(Apply2Lds)): the compiler derives the register budget
    // from the occupancy the static LDS allows, and with 21 KB of it (seven workgroups per CU) it hands the kernel 88
    // VGPRs whatever waves_per_eu asks for.  The budget that matters here is 64 (see RL_DEF_APPLY).
    extern __shared__ __align__(16) unsigned char s_dyn[];
    Apply2Lds<HPT, ENT_LOG2, NARROW, TT>& S = *reinterpret_cast<Apply2Lds<HPT, ENT_LOG2, NARROW, TT>*>(s_dyn);
    const u32 tid = threadIdx.x;
    Apply2Args A{P.table, P.log2cap, P.seed, P.b_hits, P.hits, P.limits, P.now, P.verdict, P.first_limited, &P.bs->st,
                 P.hot_next, P.items, P.runs, P.run_tt, P.ntiles, P.tile_shift, P.nb, P.hot_threshold,
                 P.hot_long, P.sparse_out, P.hot_arrive};
#define RL_ASTAMP(k)                                                                            \
    do {                                                                                        \
        if (P.trace && tid == 0) P.trace[(size_t)blockIdx.x * 8 + (k)] = wall_clock64();        \
    } while (0)
    RL_ASTAMP(0);
    // k_bkt_part refused the batch (a malformed hit): nothing is applied, the status block says why
    const bool refused = __hip_atomic_load(&P.bs->st.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    if (!refused) {
        // Workgroups [0, nb) take one hash bucket each; the rest walk the hot buckets' work items.  (Appended to the
        // bucket workgroups, an item was 7 us on top of the longest buckets: the kernel's span is its slowest workgroup.)
        const bool own = blockIdx.x < P.nb;
        if (own) {
            // the row of the workgroup's own bucket is requested first; the LDS cells are cleared under its latency
            const u32 bin = bucket_of_workgroup(blockIdx.x, P.nb);
            const ViewRow<TT> vrow = view_load<TT>(P.runs + (size_t)bin * P.run_tt, P.ntiles);
            apply2_clear(S);
            if (tid == 0) {
                S.n_created = 0;
                S.promote_ok = 1;
            }
            view_scan(S.V, vrow, P.ntiles, P.tile_shift);  // (its barriers also cover the clear)
            RL_ASTAMP(1);
            const u32 total = view_total(S.V);
            if (total) apply2_bucket(S, A, 0, total);
            if (P.trace && tid == 0) P.trace[(size_t)blockIdx.x * 8 + 6] = total;
            RL_ASTAMP(2);
        } else {
            u32 n_items = P.items->n;
            if (n_items > (u32)(HOT_MAX * HOT_NK_MAX)) n_items = HOT_MAX * HOT_NK_MAX;
            if (tid == 0) {
                S.n_created = 0;
                S.promote_ok = 0;
            }
            RL_ASTAMP(1);
            RL_ASTAMP(2);
            u32 done = 0;
            for (u32 c = blockIdx.x - P.nb; c < n_items; c += G - P.nb, ++done) apply2_hot_item(S, A, c);
            if (P.trace && tid == 0) P.trace[(size_t)blockIdx.x * 8 + 7] = done;
        }
        RL_ASTAMP(3);
    }
    // every wave's stores have been acknowledged before the workgroup's ticket is taken
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    RL_ASTAMP(4);
    if (tid == 0) apply_finish(refused ? 0u : S.n_created, P.bs, P.bs_zero, P.host_status, P.done_seq, G, &P.hot_next->n, 0u);
    RL_ASTAMP(5);
#undef RL_ASTAMP
}

// ---------------------------------------------------------------------------------------------
// k_bkt_step: ONE launch per step of the pipeline — the replay of batch j (workgroups [0, n_apply_wgs): one per hash
// bucket, then the ones that walk the hot work items) and, beside it, the partition of batch j + 1 (one workgroup per
// tile + one for its hot work items: part_role / plan_role, rl_part.hpp).  The two halves share nothing but the device:
// the partition writes the buffers the NEXT launch replays.  Either half may be absent (n_*_wgs = 0): the first batch
// of a burst is only partitioned, the last one only replayed.
// The register budget is pinned (waves_per_eu), not left to the occupancy the compiler derives from the LDS size: 80
// VGPRs and ~21 KB of LDS per workgroup keep every workgroup of a 1 M-hit replay (1024 buckets + 256) resident at once.
// ---------------------------------------------------------------------------------------------
struct StepParams {
    ApplyParams A;
    PartParams Q;
    u32 n_apply_wgs;
    u32 n_part_wgs;  // tiles + 1
};
#define RL_DEF_STEP(NAME, HPT, ENT_LOG2, MAX_VGPR, NARROW, TT)                                                             \
    typedef Apply2Lds<HPT, ENT_LOG2, NARROW, TT> NAME##_lds;                                                               \
    __global__ __launch_bounds__(AP_BLOCK) __attribute__((amdgpu_waves_per_eu(512 / MAX_VGPR, 512 / MAX_VGPR))) void NAME( \
        const StepParams S) {                                                                                              \
        if (blockIdx.x < S.n_apply_wgs) {                                                                                  \
            bkt_apply_body<HPT, ENT_LOG2, NARROW, TT>(S.A, S.n_apply_wgs);                                                 \
        } else {                                                                                                           \
            extern __shared__ __align__(16) unsigned char s_dyn[];                                                         \
            PartLds& L = *reinterpret_cast<PartLds*>(s_dyn);                                                               \
            const u32 j = blockIdx.x - S.n_apply_wgs, ntiles = S.Q.ntiles;                                                 \
            if (j >= ntiles) {                                                                                             \
                plan_role(L, S.Q);                                                                                         \
            } else {                                                                                                       \
                /* every XCD takes a CONTIGUOUS run of tiles (n_apply_wgs is a multiple of 8: j & 7 is the XCD) */         \
                const u32 x = j & 7u, per = ntiles >> 3, rem = ntiles & 7u;                                                \
                part_role(L, S.Q, x * per + (x < rem ? x : rem) + (j >> 3), S.A.trace);                                               \
            }                                                                                                              \
        }                                                                                                                  \
    }
RL_DEF_STEP(k_bkt_step, 1, 9, 80, true, TT_SMALL)         // the usual one: limit ids in 16 bits, views of up to 256 tiles
RL_DEF_STEP(k_bkt_step_wide, 1, 9, 80, false, TT_SMALL)   // engines with more than 32768 limit rows
RL_DEF_STEP(k_bkt_step_large, 1, 9, 96, false, TT_LARGE)  // batches of more than 256 tiles
RL_DEF_STEP(k_bkt_step_v64, 1, 9, 64, true, TT_SMALL)     // (RL_APPLY2_CFG: other register budgets)
RL_DEF_STEP(k_bkt_step_v96, 1, 9, 96, true, TT_SMALL)
#undef RL_DEF_STEP
// ---------------------------------------------------------------------------------------------
// k_bkt_tiny: a batch of at most TINY_MAX hits IS one bucket — it is in trace order already — so one
// workgroup validates it (the checks of k_bkt_part), rewrites it as BHit records and replays it with
// the bucket code: one launch instead of two.  The hot set is left as it is.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AP_BLOCK) void k_bkt_tiny(
    Cell* __restrict__ table, u32 log2cap, u64 seed, const Hit* __restrict__ hits, u32 n, BHit* __restrict__ b_hits,
    const LimitDev* __restrict__ limits, u32 n_limits, u64 now, uint8_t* __restrict__ verdict,
    int32_t* __restrict__ first_limited, BatchScratch* bs, BatchScratch* bs_zero, Status* host_status, u32 done_seq,
    u32 hot_n_report) {
    __shared__ Apply2Lds<1, 9, false, TT_SMALL> S;
    __shared__ u32 s_err;
    const u32 tid = threadIdx.x;
    if (tid == 0) {
        s_err = 0;
        S.n_created = 0;
        S.promote_ok = 0;  // no promotion from here: the hot set belongs to the partitioned path
    }
    apply2_clear(S);
    view_single(S.V, 0u, n);
    __syncthreads();
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
        Apply2Args A{table, log2cap, seed, b_hits, hits, limits, now, verdict, first_limited, &bs->st,
                     nullptr, nullptr, nullptr, 0u, 1u, 0u, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, nullptr};
        apply2_bucket(S, A, 0, n);
    }
    // the verdicts may go straight to host-mapped memory (rl_check_and_update_batch): every wave's stores
    // have been acknowledged before the completion word is written
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) apply_finish(S.n_created, bs, bs_zero, host_status, done_seq, 1u, nullptr, hot_n_report);
}

}  // namespace rl
