// rl_kernels.hpp — hand-written gfx950 kernels of the counter engine.
//
// Batch pipeline for CounterStorage::check_and_update (reference
// limitador/src/storage/in_memory.rs:72-156), single-counter requests, one `now` per batch:
//
//   k_probe   every hit finds (or creates: in_memory.rs:122-127) its cell, hits of one
//             workgroup tile on the same cell are combined in an LDS hash, and one pair of
//             global atomics per (tile, cell) adds the tile's delta sum / hit count to the
//             cell's scratch words.  The lane whose count atomic returned 0 is the cell's
//             commit leader for this batch.
//   k_decide  every hit reads its cell: s = value_at(now) (atomic_expiring_value.rs:19-24),
//             total = sum of all deltas of the batch on that cell.
//               s + total <= max        -> every hit on the cell is admitted, whatever the order
//               s + delta  > max        -> this hit is denied, whatever the order (the running
//                                          value only grows inside a batch)
//               otherwise               -> the verdict depends on trace order: the hit is
//                                          appended to the ordered list and the cell is marked.
//   (ordered resolver, only when the list is non-empty: rl_ordered.hpp)
//   k_commit  the leader of each touched cell applies AtomicExpiringValue::update
//             (atomic_expiring_value.rs:36-42) once for the admitted sum and zeroes the scratch.
//
// All arithmetic is u64; `value + delta` wraps like the reference's release build.
#pragma once
#include "rl_cell.hpp"

namespace rl {

struct Hit {  // == rl_hit
    u64 key;
    u32 limit;
    u32 delta;
};
static_assert(sizeof(Hit) == 16, "rl_hit is 16 bytes");

constexpr int PROBE_BLOCK = 256;
constexpr int PROBE_HPT = 4;  // hits per thread
constexpr int PROBE_TILE = PROBE_BLOCK * PROBE_HPT;
constexpr int AGG_N = 2048;  // LDS aggregation table entries (2x tile)
constexpr u32 AGG_EMPTY = 0xFFFFFFFFu;
constexpr u32 PROBE_MAX_BLOCKS = 1024;  // workgroups of k_probe (tiles are grid-strided over them)

__device__ __forceinline__ Hit load_hit(const Hit* hits, u32 i) {
    const uint4 v = *reinterpret_cast<const uint4*>(hits + i);
    Hit h;
    h.key = ((u64)v.y << 32) | v.x;
    h.limit = v.z;
    h.delta = v.w;
    return h;
}

// Probe modes
constexpr int PM_LOOKUP = 0;         // never insert
constexpr int PM_CHECK = 1;          // insert qualified, simple must pre-exist (in_memory.rs:106-107)
constexpr int PM_UPDATE = 2;         // insert both (update_counter, in_memory.rs:51-62)

// Linear probing over 64-byte cells.  Returns the slot or SLOT_INVALID.
// `first_tag` is the tag already loaded from the start slot.
template <int MODE, bool FRESH = false>
__device__ __forceinline__ u32 probe_from(Cell* __restrict__ table, u32 log2cap, u32 slot,
                                          u64 first_tag, u64 key, u32 limit,
                                          const LimitDev* __restrict__ limits, u64 now,
                                          Status* st, u32& n_created) {
    const u32 mask = (1u << log2cap) - 1u;
    u64 tag = first_tag;
    for (u32 step = 0; step <= mask; ++step) {
        if (tag == key) return slot;
        if (tag == TAG_EMPTY) {
            if (MODE == PM_LOOKUP) return SLOT_INVALID;
            if (MODE == PM_CHECK && (limit & SIMPLE_FLAG)) {
                atomicOr(&st->err, ERRBIT_MISSING_SIMPLE);
                return SLOT_INVALID;
            }
            const u64 old = atomicCAS(&table[slot].tag, TAG_EMPTY, key);
            if (old == TAG_EMPTY) {
                // Creator: AtomicExpiringValue::new(0, now + window), in_memory.rs:123-125.
                // Scratch words of an EMPTY cell are already zero.
                Cell* c = &table[slot];
                c->value = 0;
                c->expiry = now + limits[limit & ~SIMPLE_FLAG].window_us;
                c->limit = limit;
                if (FRESH) c->pad = 1u;  // created by this batch, not yet confirmed (rl_general.hpp)
                n_created++;
                return slot;
            }
            if (old == key) return slot;
            // somebody else's key landed here first: keep probing
        }
        slot = (slot + 1) & mask;
        tag = table[slot].tag;
    }
    atomicOr(&st->err, ERRBIT_TABLE_FULL);
    return SLOT_INVALID;
}

// ---------------------------------------------------------------------------------------------
// k_probe
// ---------------------------------------------------------------------------------------------
template <int MODE, bool FRESH = false, bool WITH_SUM = true>
__global__ __launch_bounds__(PROBE_BLOCK) void k_probe(Cell* __restrict__ table, u32 log2cap,
                                                        u64 seed, const Hit* __restrict__ hits,
                                                        u32 n, const LimitDev* __restrict__ limits,
                                                        u32 n_limits, u64 now, u64 delta_limit,
                                                        u32* __restrict__ hit_slot, Status* st) {
    __shared__ u32 a_slot[AGG_N];
    __shared__ u32 a_cnt[AGG_N];
    __shared__ u64 a_sum[AGG_N];
    __shared__ u32 a_lead[AGG_N];
    __shared__ u32 s_created;

    const u32 tid = threadIdx.x;
    if (tid == 0) s_created = 0;
    // Tiles are grid-strided over a bounded number of workgroups: the count of created cells goes
    // through one same-address global atomic per workgroup (they serialise at ~30 ns each).
    const u32 n_tiles = (n + PROBE_TILE - 1) / PROBE_TILE;
    for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    for (u32 e = tid; e < AGG_N; e += PROBE_BLOCK) {
        a_slot[e] = AGG_EMPTY;
        a_cnt[e] = 0;
        a_sum[e] = 0;
        a_lead[e] = 0;
    }
    __syncthreads();

    const u32 base = tile * PROBE_TILE;
    Hit h[PROBE_HPT];
    u32 slot[PROBE_HPT];
    u64 tag0[PROBE_HPT];
    bool ok[PROBE_HPT];
    // Issue the independent loads of all PROBE_HPT hits before resolving any of them.
#pragma unroll
    for (int u = 0; u < PROBE_HPT; ++u) {
        const u32 i = base + u * PROBE_BLOCK + tid;
        ok[u] = i < n;
        if (ok[u]) h[u] = load_hit(hits, i);
    }
#pragma unroll
    for (int u = 0; u < PROBE_HPT; ++u) {
        if (ok[u]) {
            if ((h[u].limit & ~SIMPLE_FLAG) >= n_limits) {
                atomicOr(&st->err, ERRBIT_BAD_LIMIT);
                ok[u] = false;
            } else if (h[u].key >= TAG_TOMB) {
                atomicOr(&st->err, ERRBIT_RESERVED_KEY);
                ok[u] = false;
            }
        }
        if (ok[u]) {
            slot[u] = slot_of(h[u].key, seed, log2cap);
            tag0[u] = table[slot[u]].tag;
        }
    }
    u32 created = 0;
    u32 my_e[PROBE_HPT];
    bool claimed[PROBE_HPT];
#pragma unroll
    for (int u = 0; u < PROBE_HPT; ++u) {
        claimed[u] = false;
        my_e[u] = 0;
        if (!ok[u]) {
            slot[u] = SLOT_INVALID;
            continue;
        }
        slot[u] = probe_from<MODE, FRESH>(table, log2cap, slot[u], tag0[u], h[u].key, h[u].limit,
                                          limits, now, st, created);
        if (slot[u] == SLOT_INVALID) continue;
        // LDS aggregation keyed by the cell slot.
        u32 e = (slot[u] * 0x9E3779B1u) >> (32 - 11);  // AGG_N == 2^11
        for (;;) {
            const u32 prev = atomicCAS(&a_slot[e], AGG_EMPTY, slot[u]);
            if (prev == AGG_EMPTY) {
                claimed[u] = true;
                break;
            }
            if (prev == slot[u]) break;
            e = (e + 1) & (AGG_N - 1);
        }
        atomicAdd(&a_cnt[e], 1u);
        if (WITH_SUM) {
            if ((u64)h[u].delta >= delta_limit) atomicOr(&st->err, ERRBIT_BIG_DELTA);
            atomicAdd(&a_sum[e], (u64)h[u].delta);
        }
        my_e[u] = e;
    }
    if (created) atomicAdd(&s_created, created);
    __syncthreads();
    // One global atomic per (tile, cell): count (top 24 bits) and delta sum (low 40 bits).
    for (u32 e = tid; e < AGG_N; e += PROBE_BLOCK) {
        const u32 s = a_slot[e];
        if (s != AGG_EMPTY) {
            const u64 add = ((u64)a_cnt[e] << PEND_SHIFT) | (WITH_SUM ? (a_sum[e] & PEND_SUM_MASK) : 0ull);
            const u64 old = atomicAdd(&table[s].pend, add);
            a_lead[e] = ((old >> PEND_SHIFT) == 0ull);
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PROBE_HPT; ++u) {
        const u32 i = base + u * PROBE_BLOCK + tid;
        if (i < n) {
            u32 v = slot[u];
            if (v != SLOT_INVALID && claimed[u] && a_lead[my_e[u]]) v |= LEADER_BIT;
            hit_slot[i] = v;
        }
    }
    __syncthreads();  // the LDS aggregation table is re-initialised for the next tile
    }
    if (tid == 0 && s_created) atomicAdd(&st->n_inserted, s_created);
}

// ---------------------------------------------------------------------------------------------
// k_decide
// ---------------------------------------------------------------------------------------------
constexpr int DECIDE_BLOCK = 256;

__global__ __launch_bounds__(DECIDE_BLOCK) void k_decide(
    Cell* __restrict__ table, const Hit* __restrict__ hits, u32 n,
    const LimitDev* __restrict__ limits, u64 now, const u32* __restrict__ hit_slot,
    uint8_t* __restrict__ verdict, int32_t* __restrict__ first_limited, u32* __restrict__ ord_list,
    Status* st) {
    __shared__ u32 s_cnt;
    __shared__ u32 s_base;
    const u32 tid = threadIdx.x;
    const u32 i = blockIdx.x * DECIDE_BLOCK + tid;
    if (tid == 0) s_cnt = 0;
    __syncthreads();

    bool ordered = false;
    if (i < n) {
        const u32 hs = hit_slot[i];
        const u32 slot = hs & SLOT_MASK;
        uint8_t v = 1;
        if (slot != SLOT_INVALID) {
            const Hit h = load_hit(hits, i);
            Cell* c = &table[slot];
            const u64 value = c->value;
            const u64 expiry = c->expiry;
            const u64 total = c->pend & PEND_SUM_MASK;
            const u32 climit = c->limit;
            if (climit != h.limit) atomicOr(&st->err, ERRBIT_KEY_LIMIT);
            const LimitDev L = limits[h.limit & ~SIMPLE_FLAG];
            const u64 d = h.delta;
            if (L.window_us == 0) {
                // 0-second window: the cell is expired at every read, every hit stands alone
                // and the cell ends as (last admitted delta, now).
                const bool okv = d <= L.max_value;
                if (okv) atomicMax(&c->aux, ((u64)(i + 1) << 32) | d);
                v = okv ? 0 : 1;
            } else {
                const u64 s = (expiry <= now) ? 0ull : value;  // value_at, :19-24
                u64 st_sum;
                const bool ovf = __builtin_add_overflow(s, total, &st_sum);
                if (!ovf && st_sum <= L.max_value) {
                    v = 0;
                } else if (!ovf && s + d > L.max_value) {
                    v = 1;
                } else {
                    ordered = true;
                    v = 2;  // placeholder; the resolver overwrites it
                    c->amb = AMB_PENDING;
                }
            }
        }
        verdict[i] = v;
        if (first_limited) first_limited[i] = (v == 1) ? (int32_t)i : -1;
    }
    // Block-aggregated append to the ordered list.
    const u64 ball = __ballot(ordered);
    const u32 lane = __lane_id();
    u32 wave_base = 0;
    if (ball) {
        if (lane == 0) wave_base = atomicAdd(&s_cnt, (u32)__popcll(ball));
        wave_base = __shfl(wave_base, 0);
    }
    __syncthreads();
    if (tid == 0 && s_cnt) s_base = atomicAdd(&st->n_ord, s_cnt);
    __syncthreads();
    if (ordered) {
        const u32 pos = s_base + wave_base + (u32)__popcll(ball & ((1ull << lane) - 1ull));
        ord_list[pos] = i;
    }
}

// ---------------------------------------------------------------------------------------------
// k_commit
// ---------------------------------------------------------------------------------------------
// Grid-stride over the hits with a bounded grid: the count of dropped cells goes through one
// same-address global atomic per WORKGROUP, and those serialise at ~30 ns each on one address —
// a workgroup per 256 hits would be 12 k of them for a 3 M-hit batch (0.5 ms, measured).
constexpr u32 COMMIT_MAX_BLOCKS = 1024;

__global__ __launch_bounds__(256) void k_commit(Cell* __restrict__ table,
                                                const Hit* __restrict__ hits, u32 n,
                                                const LimitDev* __restrict__ limits, u64 now,
                                                const u32* __restrict__ hit_slot,
                                                int drop_unreached, Status* st) {
    __shared__ u32 s_dropped;
    if (threadIdx.x == 0) s_dropped = 0;
    __syncthreads();
    u32 my_dropped = 0;
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const u32 hs = hit_slot[i];
        if (!(hs & LEADER_BIT)) continue;
        Cell* c = &table[hs & SLOT_MASK];
        const LimitDev L = limits[c->limit & ~SIMPLE_FLAG];
        const u64 expiry = c->expiry;
        // A cell created by k_probe that no request actually reached (its request stopped at an
        // earlier limited counter, in_memory.rs:109-113,129-133) must not exist.
        const bool dropped = drop_unreached && c->pad == 1u;
        if (dropped) {
            c->tag = TAG_TOMB;
            ++my_dropped;
        } else {
            const bool expired = expiry <= now;
            if (L.window_us == 0) {
                const u64 a = c->aux;
                if (a) {  // update(): expired -> (delta, now + 0)
                    c->value = a & 0xFFFFFFFFull;
                    c->expiry = now;
                }
            } else {
                const u32 amb = c->amb;
                if (amb == AMB_NONE) {
                    const u64 s = expired ? 0ull : c->value;
                    u64 sum;
                    if (!__builtin_add_overflow(s, c->pend & PEND_SUM_MASK, &sum) && sum <= L.max_value) {
                        c->value = sum;
                        if (expired) c->expiry = now + L.window_us;  // update_if_expired, :87-99
                    }
                } else if (amb == AMB_ADMIT) {
                    c->value = c->aux;
                    if (expired) c->expiry = now + L.window_us;
                }
            }
        }
        cell_clear_scratch(c);  // (also resets `pad`)
    }
    for (int off = 32; off > 0; off >>= 1) my_dropped += __shfl_down(my_dropped, off);
    if (__lane_id() == 0 && my_dropped) atomicAdd(&s_dropped, my_dropped);
    __syncthreads();
    if (threadIdx.x == 0 && s_dropped) atomicAdd(&st->n_removed, s_dropped);
}

// Leaves the table's scratch clean after a failed batch (nothing is applied).
__global__ __launch_bounds__(256) void k_abort(Cell* __restrict__ table, u32 n,
                                               const u32* __restrict__ hit_slot) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32 slot = hit_slot[i] & SLOT_MASK;
    if (slot == SLOT_INVALID) return;
    cell_clear_scratch(&table[slot]);
}

// ---------------------------------------------------------------------------------------------
// update_counter (in_memory.rs:47-69): k_probe<PM_UPDATE>, optional k_update_aux, k_update_commit
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_update_aux(Cell* __restrict__ table,
                                                    const Hit* __restrict__ hits, u32 n,
                                                    const LimitDev* __restrict__ limits,
                                                    const u32* __restrict__ hit_slot) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32 slot = hit_slot[i] & SLOT_MASK;
    if (slot == SLOT_INVALID) return;
    const Hit h = load_hit(hits, i);
    if (limits[h.limit & ~SIMPLE_FLAG].window_us == 0)
        atomicMax(&table[slot].aux, ((u64)(i + 1) << 32) | (u64)h.delta);
}

__global__ __launch_bounds__(256) void k_update_commit(Cell* __restrict__ table, u32 n,
                                                       const LimitDev* __restrict__ limits, u64 now,
                                                       const u32* __restrict__ hit_slot) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32 hs = hit_slot[i];
    if (!(hs & LEADER_BIT)) return;
    Cell* c = &table[hs & SLOT_MASK];
    const LimitDev L = limits[c->limit & ~SIMPLE_FLAG];
    if (L.window_us == 0) {
        c->value = c->aux & 0xFFFFFFFFull;
        c->expiry = now;
    } else if (c->expiry <= now) {
        c->value = c->pend & PEND_SUM_MASK;  // first update stores, the rest fetch_add (:36-42)
        c->expiry = now + L.window_us;
    } else {
        c->value += c->pend & PEND_SUM_MASK;  // wraps like fetch_add
    }
    cell_clear_scratch(c);
}

// Exact fallback for update_counter batches whose deltas could carry out of the packed 40-bit sum:
// one lane replays AtomicExpiringValue::update hit by hit (slots already resolved by k_probe).
__global__ void k_update_serial(Cell* __restrict__ table, const Hit* __restrict__ hits, u32 n,
                                const LimitDev* __restrict__ limits, u64 now,
                                const u32* __restrict__ hit_slot) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (u32 i = 0; i < n; ++i) {
        const u32 slot = hit_slot[i] & SLOT_MASK;
        if (slot == SLOT_INVALID) continue;
        Cell* c = &table[slot];
        const u64 w = limits[c->limit & ~SIMPLE_FLAG].window_us;
        const u64 d = hits[i].delta;
        if (c->expiry <= now) {
            c->expiry = now + w;
            c->value = d;
        } else {
            c->value += d;
        }
        c->pend = 0;
        c->aux = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// is_within_limits (in_memory.rs:20-35): read-only
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_within(Cell* __restrict__ table, u32 log2cap, u64 seed,
                                                const Hit* __restrict__ hits, u32 n,
                                                const LimitDev* __restrict__ limits, u32 n_limits,
                                                u64 now, uint8_t* __restrict__ within, Status* st) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Hit h = load_hit(hits, i);
    if ((h.limit & ~SIMPLE_FLAG) >= n_limits) {
        atomicOr(&st->err, ERRBIT_BAD_LIMIT);
        within[i] = 0;
        return;
    }
    u32 dummy = 0;
    u32 slot = slot_of(h.key, seed, log2cap);
    slot = probe_from<PM_LOOKUP>(table, log2cap, slot, table[slot].tag, h.key, h.limit, limits, now,
                                 st, dummy);
    u64 value = 0;
    if (slot != SLOT_INVALID) {
        const Cell* c = &table[slot];
        value = (c->expiry <= now) ? 0ull : c->value;
    }
    const u64 M = limits[h.limit & ~SIMPLE_FLAG].max_value;
    within[i] = (M >= (u64)(value + (u64)h.delta)) ? 1 : 0;  // :34, wrapping add
}

// ---------------------------------------------------------------------------------------------
// Table maintenance: init, bulk insert, streaming scans, compaction
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_table_init(Cell* __restrict__ table, u64 cap) {
    // One 16-byte store per lane, 4 lanes per cell: fully coalesced streaming fill.
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    uint4* p = reinterpret_cast<uint4*>(table);
    const u64 total = cap * 4;
    for (u64 q = gid; q < total; q += stride) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if ((q & 3) == 0) {
            v.x = 0xFFFFFFFFu;
            v.y = 0xFFFFFFFFu;
        }
        p[q] = v;
    }
}

struct CellRow {  // == rl_cell_row
    u64 key;
    u32 limit;
    u32 reserved;
    u64 value;
    u64 expiry;
};

// Insert rows; overwrite=1 replaces (value, expiry, limit) of an existing key (snapshot load),
// overwrite=0 keeps an existing cell (add_counter's entry().or_default(), in_memory.rs:41).
__global__ __launch_bounds__(256) void k_insert_rows(Cell* __restrict__ table, u32 log2cap,
                                                     u64 seed, const CellRow* __restrict__ rows,
                                                     u64 n, int overwrite, Status* st) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const CellRow r = rows[i];
    if (r.key >= TAG_TOMB) {
        atomicOr(&st->err, ERRBIT_RESERVED_KEY);
        return;
    }
    const u32 mask = (1u << log2cap) - 1u;
    u32 slot = slot_of(r.key, seed, log2cap);
    for (u32 step = 0; step <= mask; ++step) {
        u64 tag = table[slot].tag;
        if (tag == TAG_EMPTY) {
            tag = atomicCAS(&table[slot].tag, TAG_EMPTY, r.key);
            if (tag == TAG_EMPTY) {
                table[slot].value = r.value;
                table[slot].expiry = r.expiry;
                table[slot].limit = r.limit;
                atomicAdd(&st->n_inserted, 1u);
                return;
            }
        }
        if (tag == r.key) {
            if (overwrite) {
                table[slot].value = r.value;
                table[slot].expiry = r.expiry;
                table[slot].limit = r.limit;
            }
            return;
        }
        slot = (slot + 1) & mask;
    }
    atomicOr(&st->err, ERRBIT_TABLE_FULL);
}

// Streaming scan over the whole table.  Each lane reads the first 32 bytes of one cell as two
// 16-byte loads (the whole 64-byte line is fetched once per cell: 64 B/slot of HBM traffic).
constexpr int SCAN_GET = 0;          // append cells of `arg_limit` with ttl(now) > 0   (get_counters)
constexpr int SCAN_DELETE_LIMIT = 1; // tombstone cells of `arg_limit`                 (delete_counters)
constexpr int SCAN_CLEAR_SIMPLE = 2; // tombstone simple cells                          (clear)
constexpr int SCAN_SWEEP = 3;        // tombstone QUALIFIED cells with expiry <= now    (sweep_expired)
constexpr int SCAN_DUMP = 4;         // append every live cell, raw                     (dump_cells)

template <int MODE>
__global__ __launch_bounds__(256) void k_scan(Cell* __restrict__ table, u64 cap, u32 arg_limit,
                                              u64 now, CellRow* __restrict__ out, u64 out_cap,
                                              Status* st, unsigned long long* out_total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u32 removed = 0;
    for (u64 s = gid; s < cap; s += stride) {
        const uint4* p = reinterpret_cast<const uint4*>(&table[s]);
        const uint4 a = p[0];  // tag, value
        const u64 tag = ((u64)a.y << 32) | a.x;
        if (tag >= TAG_TOMB) continue;
        const uint4 b = p[1];  // expiry, limit, cnt
        const u64 value = ((u64)a.w << 32) | a.z;
        const u64 expiry = ((u64)b.y << 32) | b.x;
        const u32 limit = b.z;
        bool emit = false, kill = false;
        if (MODE == SCAN_GET) emit = (limit == arg_limit) && (expiry > now);
        if (MODE == SCAN_DUMP) emit = true;
        if (MODE == SCAN_DELETE_LIMIT) kill = (limit == arg_limit);
        if (MODE == SCAN_CLEAR_SIMPLE) kill = (limit & SIMPLE_FLAG) != 0;
        if (MODE == SCAN_SWEEP) kill = !(limit & SIMPLE_FLAG) && (expiry <= now);
        if (emit) {
            const u64 pos = atomicAdd(out_total, 1ull);
            if (pos < out_cap) {
                CellRow r;
                r.key = tag;
                r.limit = limit;
                r.reserved = 0;
                if (MODE == SCAN_GET) {
                    r.value = value;  // not expired here, so value_at(now) == value
                    r.expiry = expiry - now;
                } else {
                    r.value = value;
                    r.expiry = expiry;
                }
                out[pos] = r;
            }
        }
        if (kill) {
            table[s].tag = TAG_TOMB;
            removed++;
        }
    }
    if (MODE == SCAN_DELETE_LIMIT || MODE == SCAN_CLEAR_SIMPLE || MODE == SCAN_SWEEP) {
        // wave-aggregated count
        for (int off = 32; off > 0; off >>= 1) removed += __shfl_down(removed, off);
        if (__lane_id() == 0 && removed) atomicAdd(&st->n_removed, removed);
    }
}

// Compaction: re-insert every live cell of `src` into the (initialised, empty) `dst`.
__global__ __launch_bounds__(256) void k_rehash(const Cell* __restrict__ src, u64 src_cap,
                                                Cell* __restrict__ dst, u32 dst_log2cap, u64 seed,
                                                Status* st) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u32 mask = (1u << dst_log2cap) - 1u;
    u32 moved = 0;
    for (u64 s = gid; s < src_cap; s += stride) {
        const uint4* p = reinterpret_cast<const uint4*>(&src[s]);
        const uint4 a = p[0];
        const u64 tag = ((u64)a.y << 32) | a.x;
        if (tag >= TAG_TOMB) continue;
        const uint4 b = p[1];  // expiry, limit, cnt
        u32 slot = slot_of(tag, seed, dst_log2cap);
        for (u32 step = 0; step <= mask; ++step) {
            const u64 old = atomicCAS(&dst[slot].tag, TAG_EMPTY, tag);
            if (old == TAG_EMPTY) {
                dst[slot].value = ((u64)a.w << 32) | a.z;
                dst[slot].expiry = ((u64)b.y << 32) | b.x;
                dst[slot].limit = b.z;
                ++moved;
                break;
            }
            slot = (slot + 1) & mask;
        }
    }
    // one atomic per workgroup (same-address atomics serialise at ~30 ns each)
    __shared__ u32 s_moved;
    if (threadIdx.x == 0) s_moved = 0;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) moved += __shfl_down(moved, off);
    if ((threadIdx.x & 63u) == 0 && moved) atomicAdd(&s_moved, moved);
    __syncthreads();
    if (threadIdx.x == 0 && s_moved) atomicAdd(&st->n_inserted, s_moved);
}

}  // namespace rl
