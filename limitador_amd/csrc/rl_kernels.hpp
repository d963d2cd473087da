// rl_kernels.hpp — the kernels of the counter engine that are not part of a batch pipeline: the shared probe
// loop, is_within_limits (read-only), and table maintenance (init, bulk insert, streaming scans, rehash).
// The batch pipelines live in rl_bucket.hpp / rl_apply.hpp (single-counter requests) and rl_general.hpp
// (multi-counter requests, load_counters, update_counter).  All arithmetic is u64; `value + delta` wraps
// like the reference's release build.
#pragma once
#include "rl_cell.hpp"

namespace rl {

struct Hit {  // == rl_hit
    u64 key;
    u32 limit;
    u32 delta;
};
static_assert(sizeof(Hit) == 16, "rl_hit is 16 bytes");

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

// Linear probing over 32-byte cells.  Returns the slot or SLOT_INVALID.
// `first_tag` is the tag already loaded from the start slot.
template <int MODE>
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
                Cell* c = &table[slot];
                c->value = 0;
                c->expiry = now + limits[limit & ~SIMPLE_FLAG].window_us;
                c->limit = limit;
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
// is_within_limits (in_memory.rs:20-35): read-only
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_within(Cell* __restrict__ table, u32 log2cap, u64 seed,
                                                const Hit* __restrict__ hits, u32 n,
                                                const u64* __restrict__ delta64,
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
    const u64 d = delta64 ? delta64[i] : (u64)h.delta;  // the trait's `delta: u64` (in_memory.rs:20)
    within[i] = (M >= (u64)(value + d)) ? 1 : 0;  // :34, wrapping add
}

// RateLimiter::is_rate_limited (lib.rs:362-409) for a batch of REQUESTS whose counters the matcher derived
// (hits[req_off[r] .. req_off[r + 1]), in counters_that_apply's order): find_first_limited_counter walks a request's
// counters with is_within_limits above and stops at the first one that does not fit — that counter's limit is the one
// whose name the reference reports.  Read-only: nothing is created, nothing is counted.  One lane per request (a request
// derives at most one counter per limit of its namespace: a handful).  `force_delta` != 0 replaces every hit's delta:
// the Kuadrant CheckRateLimit method checks with 1 whatever hits_addend says (envoy_rls/kuadrant_service.rs:62-64).
// Hashed keys (hit_check != null, include/rl_keyhash.h): a cell whose check word or limit id is another counter's is not
// this counter's cell — the request is marked WIRE_ST_KEY_COLLISION (-103) in msg_status and answered by the caller's
// exact path, like the counting entry points do; with exact keys a limit id that differs is RL_ERR_KEY_LIMIT.
__global__ __launch_bounds__(256) void k_req_within(const Cell* __restrict__ table, u32 log2cap, u64 seed,
                                                    const Hit* __restrict__ hits, const u32* __restrict__ req_off, u32 n_req,
                                                    const u32* __restrict__ hit_check, int32_t* __restrict__ msg_status,
                                                    const LimitDev* __restrict__ limits, u32 n_limits, u64 now,
                                                    u32 force_delta, uint8_t* __restrict__ verdict,
                                                    int32_t* __restrict__ limited_limit, Status* st) {
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_req) return;
    const u32 q0 = req_off[r], q1 = req_off[r + 1];
    const u32 mask = (1u << log2cap) - 1u;
    uint8_t limited = 0;
    int32_t which = -1;
    for (u32 q = q0; q < q1; ++q) {
        const Hit h = load_hit(hits, q);
        const u32 lid = h.limit & ~SIMPLE_FLAG;
        if (lid >= n_limits) {
            atomicOr(&st->err, ERRBIT_BAD_LIMIT);
            break;
        }
        u64 value = 0;
        u32 slot = slot_of(h.key, seed, log2cap);
        for (u32 step = 0; step <= mask; ++step, slot = (slot + 1) & mask) {
            const uint4 a = *reinterpret_cast<const uint4*>(&table[slot]);
            const u64 tag = ((u64)a.y << 32) | a.x;
            if (tag == TAG_EMPTY) break;  // unwrap_or_default(): 0
            if (tag != h.key) continue;
            const uint4 b = reinterpret_cast<const uint4*>(&table[slot])[1];
            const u64 expiry = ((u64)b.y << 32) | b.x;
            const bool other = (b.z & ~SIMPLE_FLAG) != lid || (hit_check && b.w != 0u && b.w != hit_check[q]);
            if (other) {
                if (hit_check && msg_status) msg_status[r] = -103;
                else atomicOr(&st->err, ERRBIT_KEY_LIMIT);
                q = q1;  // (the request is not answered here)
                break;
            }
            value = expiry <= now ? 0ull : (((u64)a.w << 32) | a.z);  // value_at(now), atomic_expiring_value.rs:19-24
            break;
        }
        if (q >= q1) break;
        const u64 d = force_delta ? (u64)force_delta : (u64)h.delta;
        if (!(limits[lid].max_value >= (u64)(value + d))) {  // in_memory.rs:34, wrapping add
            limited = 1;
            which = (int32_t)lid;
            break;
        }
    }
    verdict[r] = limited;
    if (limited_limit) limited_limit[r] = which;
}

// ---------------------------------------------------------------------------------------------
// Table maintenance: init, bulk insert, streaming scans, compaction
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_table_init(Cell* __restrict__ table, u64 cap) {
    // One 16-byte store per lane, 2 lanes per cell: fully coalesced streaming fill.
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    uint4* p = reinterpret_cast<uint4*>(table);
    const u64 total = cap * 2;
    for (u64 q = gid; q < total; q += stride) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if ((q & 1) == 0) {
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
                table[slot].pad = r.reserved;  // (the check word of a hashed key, 0 otherwise)
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

// Streaming scan over the whole table.  Each lane reads one 32-byte cell as two 16-byte loads: a wave
// reads 2 KB of consecutive memory per step (32 B/slot of HBM traffic).
constexpr int SCAN_GET = 0;          // append cells of `arg_limit` with ttl(now) > 0   (get_counters)
constexpr int SCAN_DELETE_LIMIT = 1; // tombstone cells of `arg_limit`                 (delete_counters)
constexpr int SCAN_CLEAR_SIMPLE = 2; // tombstone simple cells                          (clear)
constexpr int SCAN_SWEEP = 3;        // tombstone QUALIFIED cells with expiry <= now    (sweep_expired)
constexpr int SCAN_DUMP = 4;         // append every live cell, raw                     (dump_cells)
constexpr int SCAN_PEER = 5;         // a PEER table (rl_merge_cells): tombstone entries of windows that are over
                                     // (expiry <= now), of `arg_limit` (delete_counters; ~0u: none) or — arg_limit ==
                                     // 0xFFFFFFFE — of every simple limit (clear)

template <int MODE>
__global__ __launch_bounds__(256) void k_scan(Cell* __restrict__ table, u64 cap, u32 arg_limit,
                                              u64 now, CellRow* __restrict__ out, u64 out_cap,
                                              Status* st, unsigned long long* out_total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u32 lane = threadIdx.x & 63u;
    u32 removed = 0;
    u64 counted = 0;  // count-only calls (out_cap == 0): one atomic per workgroup at the end
    // (every lane of a wave runs the same number of iterations: cap and stride are multiples of 64)
    for (u64 s = gid; s < cap; s += stride) {
        const uint4* p = reinterpret_cast<const uint4*>(&table[s]);
        const uint4 a = p[0];  // tag, value
        const uint4 b = p[1];  // expiry, limit
        const u64 tag = ((u64)a.y << 32) | a.x;
        const bool live = tag < TAG_TOMB;
        const u64 value = ((u64)a.w << 32) | a.z;
        const u64 expiry = ((u64)b.y << 32) | b.x;
        const u32 limit = b.z;
        bool emit = false, kill = false;
        if (MODE == SCAN_GET) emit = live && (limit == arg_limit) && (expiry > now);
        if (MODE == SCAN_DUMP) emit = live;
        if (MODE == SCAN_DELETE_LIMIT) kill = live && (limit == arg_limit);
        if (MODE == SCAN_CLEAR_SIMPLE) kill = live && (limit & SIMPLE_FLAG) != 0;
        if (MODE == SCAN_SWEEP) kill = live && !(limit & SIMPLE_FLAG) && (expiry <= now);
        if (MODE == SCAN_SWEEP) emit = kill && out_cap != 0;  // (rl_sweep_expired_rows: the swept cells are reported)
        if (MODE == SCAN_PEER)
            kill = live && (expiry <= now || limit == arg_limit || (arg_limit == 0xFFFFFFFEu && (limit & SIMPLE_FLAG) != 0));
        if (MODE == SCAN_GET || MODE == SCAN_DUMP || (MODE == SCAN_SWEEP && out_cap != 0)) {
            const u64 bal = __ballot(emit);
            if (out_cap == 0) {
                if (lane == 0) counted += (u64)__popcll(bal);
            } else if (bal) {
                // one atomic per wave step reserves the rows of its matching lanes (a same-address atomic per
                // matching CELL serialises: 10 M rows took 6.3 ms)
                u64 base = 0;
                if (lane == 0) base = atomicAdd(out_total, (unsigned long long)__popcll(bal));
                base = __shfl(base, 0);
                if (emit) {
                    const u64 pos = base + (u64)__popcll(bal & ((1ull << lane) - 1ull));
                    if (pos < out_cap) {
                        CellRow r;
                        r.key = tag;
                        r.limit = limit;
                        r.reserved = MODE == SCAN_DUMP ? b.w : 0u;  // (a dump carries the check word of a hashed key: rl_keyhash.h)
                        r.value = value;  // SCAN_GET: not expired here, so value_at(now) == value
                        r.expiry = MODE == SCAN_GET ? expiry - now : expiry;
                        out[pos] = r;
                    }
                }
            }
        }
        if (kill) {
            table[s].tag = TAG_TOMB;
            removed++;
        }
    }
    if (MODE == SCAN_GET || MODE == SCAN_DUMP) {
        if (out_cap == 0) {
            __shared__ unsigned long long s_cnt;
            if (threadIdx.x == 0) s_cnt = 0;
            __syncthreads();
            if (lane == 0 && counted) atomicAdd(&s_cnt, (unsigned long long)counted);
            __syncthreads();
            if (threadIdx.x == 0 && s_cnt) atomicAdd(out_total, s_cnt);
        }
    } else {
        // one atomic per workgroup
        __shared__ u32 s_rem;
        if (threadIdx.x == 0) s_rem = 0;
        __syncthreads();
        for (int off = 32; off > 0; off >>= 1) removed += __shfl_down(removed, off);
        if (lane == 0 && removed) atomicAdd(&s_rem, removed);
        __syncthreads();
        if (threadIdx.x == 0 && s_rem) atomicAdd(&st->n_removed, s_rem);
    }
}

// ---------------------------------------------------------------------------------------------
// Compaction IN PLACE (same capacity, same seed): drop the tombstones and close the gaps they leave.
//
// With linear probing and tombstones (never EMPTY again), every slot between a key's home and its cell is non-empty,
// so a key's home lies in the same CLUSTER (maximal run of non-empty slots) as its cell and clusters never interact:
// a cluster is REBUILT inside its own slots.  One lane walks it left to right and re-inserts every live cell the way
// an insert would — into the first free slot at or after its home, where "free" = a slot of the cluster before the
// walk's position that holds nothing final (a tombstone, or a cell that has moved on) or the cell's own slot.
// Linear-probing insertion of the cluster's cells into its emptied range, in slot order: every cell lands at or
// before the slot it came from (the cells placed inside [home, d) come from slots inside [home, d), so one of
// [home, d] is free), nothing unread is overwritten, and what is left free at the end becomes EMPTY.  (Packing the
// cells with pos = max(home, previous pos + 1) in SLOT order is wrong: a later cell with an earlier home then sits
// behind a gap that was emptied — that formula needs the cells sorted by home.)
// No second table, no atomics, no failure mode (the rebuild into a fresh table, k_rehash, is a 1 GB fill + 10 M random
// compare-and-swaps + 1 GB of hipMalloc / hipFree for the bench's table: 1.7 ms).
//
// The table is read ONCE, as coalesced 2 KB loads: a wave owns a SEGMENT — the slots between two EMPTY slots of the
// unmodified table, `bounds[w]` = the first EMPTY slot at or after w * 4096 (k_compact_bounds: a launch of its own,
// ~1.4 cells read per segment, so that nobody decides a boundary from a table that is being rewritten) — and no
// cluster crosses a segment's end.  It walks the segment in steps of 64 slots: every lane loads its slot's cell
// (+ 16 slots of look-ahead) into a per-wave LDS window, the next step's loads are in flight meanwhile; a lane whose
// slot STARTS a cluster (non-empty, the slot before it EMPTY) rebuilds that cluster, reading from the window (from
// the table where the cluster runs past it).  What a cluster's walk has touched lies before its end, `carry_end`:
// later steps ignore what they loaded up to there (it may have been rewritten) and take slot carry_end as EMPTY.
// The free slots of a cluster's first 64 slots are a bit mask in a register; a longer cluster continues with the
// table itself as the state (free = EMPTY; the lane reads its own stores).
// ---------------------------------------------------------------------------------------------
constexpr u32 CSEG_LOG2 = 12;   // slots per segment (one wave)
constexpr u32 CWIN_EXTRA = 16;  // look-ahead slots of the LDS window

__global__ __launch_bounds__(256) void k_compact_bounds(const Cell* __restrict__ table, u64 cap, u64* __restrict__ bounds,
                                                        u32 n_seg) {
    const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_seg) return;
    u64 s = (u64)w << CSEG_LOG2;  // (absolute: may run past the segment, past the table's end for the last ones)
    for (u64 k = 0; k < cap && table[s & (cap - 1)].tag != TAG_EMPTY; ++k) ++s;
    bounds[w] = s;
}

// What a slot holds, as the walks need it: EMPTY, a tombstone, or a live cell's distance from its home — computed by
// all 64 lanes at once (the hash of the key is most of a walk's arithmetic, and a walk runs on the few lanes whose slot
// starts a cluster).
constexpr u32 CCODE_EMPTY = 0xFFFFFFFFu, CCODE_TOMB = 0xFFFFFFFEu;
__device__ __forceinline__ u32 compact_code(u64 tag, u64 slot, u64 seed, u32 log2cap, u32 mask) {
    return tag == TAG_EMPTY ? CCODE_EMPTY : tag == TAG_TOMB ? CCODE_TOMB : (((u32)slot - slot_of(tag, seed, log2cap)) & mask);
}

__global__ __launch_bounds__(256) void k_compact_seg(Cell* table, u32 log2cap, u64 seed, const u64* __restrict__ bounds,
                                                     u32 n_seg, Status* st, u32 count) {
    __shared__ uint4 s_win[4][(64 + CWIN_EXTRA) * 2];
    __shared__ u32 s_code[4][64 + CWIN_EXTRA];
    const u64 cap = 1ull << log2cap;
    const u32 mask = (u32)(cap - 1);
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const u32 gw = blockIdx.x * 4u + wv;
    const uint4 e0 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u), z0 = make_uint4(0u, 0u, 0u, 0u);
    auto cell = [&](u64 slot) { return reinterpret_cast<uint4*>(&table[(u32)slot & mask]); };
    u32 live = 0;
    if (gw < n_seg) {
        const u64 lo = bounds[gw], hi = gw + 1 < n_seg ? bounds[gw + 1] : bounds[0] + cap;
        uint4* win = s_win[wv];
        u32* code = s_code[wv];
        u64 carry_end = lo;   // everything up to here is final (slot carry_end itself is an EMPTY slot of the unmodified table)
        bool prev_ne = false;  // the slot before this step's first one was non-empty when it was loaded
        uint4 a0 = z0, a1 = z0, x0 = z0, x1 = z0;
        if (lo + 1 < hi) {
            const uint4* p = cell(lo + 1 + lane);
            a0 = p[0];
            a1 = p[1];
            if (lane < CWIN_EXTRA) {
                const uint4* px = cell(lo + 1 + 64 + lane);
                x0 = px[0];
                x1 = px[1];
            }
        }
        for (u64 base = lo + 1; base < hi; base += 64) {
            __builtin_amdgcn_wave_barrier();  // (the walks of the step before have read the window)
            const u64 slot = base + lane;
            const u64 own_tag = ((u64)a0.y << 32) | a0.x;
            win[2 * lane] = a0;
            win[2 * lane + 1] = a1;
            code[lane] = compact_code(own_tag, slot, seed, log2cap, mask);
            if (lane < CWIN_EXTRA) {
                win[2 * (64 + lane)] = x0;
                win[2 * (64 + lane) + 1] = x1;
                code[64 + lane] = compact_code(((u64)x0.y << 32) | x0.x, slot + 64, seed, log2cap, mask);
            }
            if (base + 64 < hi) {  // the next step's cells, in flight while this step's clusters are rebuilt
                const uint4* p = cell(base + 64 + lane);
                a0 = p[0];
                a1 = p[1];
                if (lane < CWIN_EXTRA) {
                    const uint4* px = cell(base + 128 + lane);
                    x0 = px[0];
                    x1 = px[1];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool ne = own_tag != TAG_EMPTY;
            const u64 ne_mask = __ballot(ne);
            const u64 tomb_mask = __ballot(own_tag == TAG_TOMB);
            bool before_ne = lane ? ((ne_mask >> (lane - 1)) & 1ull) != 0 : prev_ne;
            if (slot - 1 == carry_end) before_ne = false;
            const bool start = ne && !before_ne && slot > carry_end && slot < hi;
            u64 end = 0;
            bool walk = start;
            if (start) {
                // a cluster that ends inside these 64 slots and holds no tombstone stays as it is
                const u64 rest = ~ne_mask >> lane;
                if (rest) {
                    const u32 len = (u32)__builtin_ctzll(rest);
                    if (!((tomb_mask >> lane) & ((1ull << len) - 1ull))) {
                        live += len;
                        end = slot + len;
                        walk = false;
                    }
                }
            }
            if (walk) {
                auto rd_code = [&](u64 sl) {
                    const u64 r = sl - base;
                    if (r < 64 + CWIN_EXTRA) return code[r];
                    const uint4 c0 = *cell(sl);
                    return compact_code(((u64)c0.y << 32) | c0.x, sl, seed, log2cap, mask);
                };
                auto rd = [&](u64 sl, uint4& c0, uint4& c1) {
                    const u64 r = sl - base;
                    if (r < 64 + CWIN_EXTRA) {
                        c0 = win[2 * r];
                        c1 = win[2 * r + 1];
                    } else {
                        const uint4* p = cell(sl);
                        c0 = p[0];
                        c1 = p[1];
                    }
                };
                const u64 a = slot;
                u64 freem = 0;  // bit q: slot a + q (q < 64) holds nothing final
                u32 d = 0;
                bool ended = false;
                for (; d < 64; ++d) {
                    const u32 cd = rd_code(a + d);
                    if (cd == CCODE_EMPTY) {
                        ended = true;
                        break;
                    }
                    if (cd == CCODE_TOMB) {
                        freem |= 1ull << d;
                        continue;
                    }
                    const u32 hd = cd > d ? 0u : d - cd;  // the home's distance from the cluster's start (cd > d cannot happen
                                                          // in a well-formed table: the home lies inside the cluster)
                    const u64 cand = (freem >> hd) << hd;
                    if (cand) {
                        const u32 q = (u32)__builtin_ctzll(cand);
                        uint4 c0, c1;
                        rd(a + d, c0, c1);
                        uint4* t = cell(a + q);
                        t[0] = c0;
                        t[1] = c1;
                        freem = (freem & ~(1ull << q)) | (1ull << d);
                    }
                    ++live;
                }
                for (u64 f = freem; f; f &= f - 1) {
                    uint4* z = cell(a + (u32)__builtin_ctzll(f));
                    z[0] = e0;
                    z[1] = z0;
                }
                if (!ended) {
                    // a cluster of more than 64 slots: the table is the state from here on
                    for (; d <= mask; ++d) {
                        const u32 cd = rd_code(a + d);
                        if (cd == CCODE_EMPTY) break;
                        uint4* p = cell(a + d);
                        if (cd == CCODE_TOMB) {
                            p[0] = e0;
                            p[1] = z0;
                            continue;
                        }
                        u32 q = cd > d ? 0u : d - cd;
                        for (; q < d; ++q) {
                            const uint4 t0 = *cell(a + q);
                            if ((((u64)t0.y << 32) | t0.x) == TAG_EMPTY) break;
                        }
                        if (q < d) {
                            uint4 c0, c1;
                            rd(a + d, c0, c1);
                            uint4* t = cell(a + q);
                            t[0] = c0;
                            t[1] = c1;
                            p[0] = e0;
                            p[1] = z0;
                        }
                        ++live;
                    }
                }
                end = a + d;
            }
            const u64 smask = __ballot(start);
            if (smask) {
                const int top = 63 - __builtin_clzll(smask);  // clusters are disjoint and in order: the last start ends last
                const u32 elo = __shfl((u32)end, top), ehi = __shfl((u32)(end >> 32), top);
                carry_end = ((u64)ehi << 32) | elo;
            }
            prev_ne = (ne_mask >> 63) & 1ull;
        }
    }
    if (!count) return;
    __shared__ u32 s_live;
    if (threadIdx.x == 0) s_live = 0;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) live += __shfl_down(live, off);
    if (lane == 0 && live) atomicAdd(&s_live, live);
    __syncthreads();
    if (threadIdx.x == 0 && s_live) atomicAdd(&st->n_inserted, s_live);
}

// Compaction: re-insert every live cell of `src` into the (initialised, empty) `dst`.
__global__ __launch_bounds__(256) void k_rehash(const Cell* __restrict__ src, u64 src_cap,
                                                Cell* __restrict__ dst, u32 dst_log2cap, u64 seed,
                                                Status* st, u32 count) {
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
        bool placed = false;
        for (u32 step = 0; step <= mask; ++step) {
            const u64 old = atomicCAS(&dst[slot].tag, TAG_EMPTY, tag);
            if (old == TAG_EMPTY) {
                dst[slot].value = ((u64)a.w << 32) | a.z;
                dst[slot].expiry = ((u64)b.y << 32) | b.x;
                dst[slot].limit = b.z;
                dst[slot].pad = b.w;
                ++moved;
                placed = true;
                break;
            }
            slot = (slot + 1) & mask;
        }
        if (!placed) atomicOr(&st->err, ERRBIT_TABLE_FULL);  // (a smaller table that cannot hold the live cells)
    }
    if (!count) return;  // (peer tables: only the error bit)
    // one atomic per workgroup (same-address atomics serialise at ~30 ns each)
    __shared__ u32 s_moved;
    if (threadIdx.x == 0) s_moved = 0;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) moved += __shfl_down(moved, off);
    if ((threadIdx.x & 63u) == 0 && moved) atomicAdd(&s_moved, moved);
    __syncthreads();
    if (threadIdx.x == 0 && s_moved) atomicAdd(&st->n_inserted, s_moved);
}


// ---------------------------------------------------------------------------------------------
// Cross-node merge: CrCounterValue::merge_at
// (limitador/src/storage/distributed/cr_counter_value.rs:81-113) for ONE incoming actor's values.
//
// The reference keeps, per counter, our own value and a map actor -> value of the others; a read is their sum
// (:38-47).  Here the cell's `value` is that SUM; what each remote actor is known to have contributed lives in a
// small table per actor (same cell layout: value = the actor's value, expiry = the window the knowledge belongs
// to — the cell's expiry when it was recorded; knowledge of another window counts as 0).  One row = what actor
// `A` reports for one counter (its (expiry, value): local_values(), :131-141).
//   expired rows are ignored                                     (:83  `if expiry > when`)
//   the earliest future expiry wins                              (:84  AtomicExpiryTime::merge_at, atomic_expiring_value.rs:113-130)
//   a cell that is expired at `when` restarts from the row       (:85-87 reset: value 0, others cleared, expiry = other's)
//   the actor's value only ever grows: max(known, incoming)      (:96-110); a row about OURSELVES raises our own
//                                                                 part to it (:91-95)
// ---------------------------------------------------------------------------------------------
constexpr int MERGE_MAX_ACTORS = 8;
struct PeerTables {
    Cell* t[MERGE_MAX_ACTORS];  // null: that actor has never been merged
    u32 log2cap;
};

__device__ __forceinline__ Cell* peer_find(Cell* table, u32 log2cap, u64 seed, u64 key, bool create) {
    const u32 mask = (1u << log2cap) - 1u;
    u32 slot = slot_of(key, seed, log2cap);
    for (u32 step = 0; step <= mask; ++step) {
        u64 tag = table[slot].tag;
        if (tag == TAG_EMPTY && create) tag = atomicCAS(&table[slot].tag, TAG_EMPTY, key) == TAG_EMPTY ? key : table[slot].tag;
        if (tag == key) return &table[slot];
        if (tag == TAG_EMPTY) return nullptr;
        slot = (slot + 1) & mask;
    }
    return nullptr;
}

// What the peers contributed to the window that ends at `window` (their entries carry the window they belong to): the
// `others` of CrCounterValue for that window.  `restamp` != 0: the entries move on to that window (a LOCAL restart keeps
// them: CrCounterValue::inc_at, cr_counter_value.rs:53-59, resets only our own value).
__device__ __forceinline__ u64 peers_window_sum(const PeerTables& peers, u64 seed, u64 key, u64 window, u64 restamp) {
    u64 sum = 0;
    for (int a = 0; a < MERGE_MAX_ACTORS; ++a)
        if (peers.t[a]) {
            Cell* p = peer_find(peers.t[a], peers.log2cap, seed, key, false);
            if (p && p->expiry == window) {
                sum += p->value;
                if (restamp) p->expiry = restamp;
            }
        }
    return sum;
}

__global__ __launch_bounds__(256) void k_merge_rows(Cell* __restrict__ table, u32 log2cap, u64 seed, PeerTables peers,
                                                    u32 actor, u32 self_actor, const CellRow* __restrict__ rows, u64 n,
                                                    u64 now, Status* st) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const CellRow r = rows[i];
    if (r.key >= TAG_TOMB) {
        atomicOr(&st->err, ERRBIT_RESERVED_KEY);
        return;
    }
    if (r.expiry <= now) return;  // an expired set is ignored
    // find or create the counter (a counter first heard of from a peer: From<(SystemTime, BTreeMap)>, :163-173)
    const u32 mask = (1u << log2cap) - 1u;
    u32 slot = slot_of(r.key, seed, log2cap);
    Cell* c = nullptr;
    bool created = false;
    for (u32 step = 0; step <= mask; ++step) {
        u64 tag = table[slot].tag;
        if (tag == TAG_EMPTY) {
            const u64 old = atomicCAS(&table[slot].tag, TAG_EMPTY, r.key);
            if (old == TAG_EMPTY) {
                created = true;
                tag = r.key;
            } else {
                tag = old;
            }
        }
        if (tag == r.key) {
            c = &table[slot];
            break;
        }
        slot = (slot + 1) & mask;
    }
    if (!c) {
        atomicOr(&st->err, ERRBIT_TABLE_FULL);
        return;
    }
    if (created) {
        c->value = 0;
        c->expiry = r.expiry;
        c->limit = r.limit;
        atomicAdd(&st->n_inserted, 1u);
    } else {
        const u64 cur = c->expiry;
        if (r.expiry < cur && r.expiry > now) {  // the earliest expiry that is still in the future
            c->expiry = r.expiry;
            for (int a = 0; a < MERGE_MAX_ACTORS; ++a)  // what we know of the peers belongs to this same window
                if (peers.t[a]) {
                    Cell* p = peer_find(peers.t[a], peers.log2cap, seed, r.key, false);
                    if (p && p->expiry == cur) p->expiry = r.expiry;
                }
        }
        if (c->expiry <= now) {  // expired here: restart from the incoming set (reset)
            c->value = 0;
            c->expiry = r.expiry;  // (knowledge recorded for the old window no longer matches: it counts as 0)
        }
    }
    const u64 window = c->expiry;
    if (actor == self_actor) {
        // our own value as another replica remembers it: only a LARGER one is news (we restarted and lost state)
        u64 others = 0;
        for (int a = 0; a < MERGE_MAX_ACTORS; ++a)
            if (peers.t[a]) {
                const Cell* p = peer_find(peers.t[a], peers.log2cap, seed, r.key, false);
                if (p && p->expiry == window) others += p->value;
            }
        const u64 ours = c->value - others;
        if (r.value > ours) c->value += r.value - ours;
        return;
    }
    Cell* p = peer_find(peers.t[actor], peers.log2cap, seed, r.key, true);
    if (!p) {
        atomicOr(&st->err, ERRBIT_TABLE_FULL);
        return;
    }
    const u64 known = p->expiry == window ? p->value : 0ull;
    if (r.value > known) {
        c->value += r.value - known;
        p->value = r.value;
        p->expiry = window;
        p->limit = r.limit;
    } else if (p->expiry != window) {  // first word of this window from that actor, nothing to add (value 0)
        p->value = known;
        p->expiry = window;
        p->limit = r.limit;
    }
}

// local_values() of every live, unexpired counter: (key, limit, OUR OWN part of the value, expiry) — what a node
// sends to its peers.  Streaming scan; with peers, one probe per peer table per cell.
__global__ __launch_bounds__(256) void k_export_local(const Cell* __restrict__ table, u64 cap, u64 seed, PeerTables peers,
                                                      u64 now, CellRow* __restrict__ out, u64 out_cap,
                                                      unsigned long long* out_total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u32 lane = threadIdx.x & 63u;
    for (u64 s = gid; s < cap; s += stride) {
        const uint4* q = reinterpret_cast<const uint4*>(&table[s]);
        const uint4 a = q[0], b = q[1];
        const u64 tag = ((u64)a.y << 32) | a.x;
        const u64 expiry = ((u64)b.y << 32) | b.x;
        const bool emit = tag < TAG_TOMB && expiry > now;
        u64 ours = ((u64)a.w << 32) | a.z;
        if (emit)
            for (int p = 0; p < MERGE_MAX_ACTORS; ++p)
                if (peers.t[p]) {
                    const Cell* e = peer_find(peers.t[p], peers.log2cap, seed, tag, false);
                    if (e && e->expiry == expiry) ours -= e->value;
                }
        const u64 bal = __ballot(emit);
        if (!bal) continue;
        u64 base = 0;
        if (lane == 0) base = atomicAdd(out_total, (unsigned long long)__popcll(bal));
        base = __shfl(base, 0);
        if (emit) {
            const u64 pos = base + (u64)__popcll(bal & ((1ull << lane) - 1ull));
            if (pos < out_cap) out[pos] = CellRow{tag, b.z, 0u, ours, expiry};
        }
    }
}

// End of a stream-ordered maintenance command (rl_sweep_expired_submit): hand its device-side status to the host-mapped
// block the caller polls, with the pipeline's completion protocol (apply_finish, rl_bucket.hpp) — the first 16 bytes
// {err, n_ord = cells removed, n_inserted = 0, n_removed = the command's sequence number} are ONE store, written last.
__global__ void k_post_status(const Status* __restrict__ d_st, Status* host_status, u32 done_seq) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    u32x4* hp = reinterpret_cast<u32x4*>(host_status);
    for (int q = 1; q < 4; ++q) hp[q] = u32x4{0, 0, 0, 0};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_nontemporal_store(u32x4{d_st->err, d_st->n_removed, 0u, done_seq}, hp);
}

}  // namespace rl
