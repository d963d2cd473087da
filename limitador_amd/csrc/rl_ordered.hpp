// rl_ordered.hpp — trace-order resolver for the hits whose verdict depends on the order of the
// requests inside the batch (the cell's window fills up part-way through the batch).
//
// The reference decides request i after fully applying requests < i
// (limitador/src/storage/in_memory.rs:72-156 is called once per request).  For the hits that
// k_decide could not settle order-free, the ordered list is sorted by (cell slot, hit index):
// each cell's hits become one contiguous segment in trace order, and the segment is replayed
// with the reference's own arithmetic:
//     admitted  <=>  running + delta <= max          (in_memory.rs:259-264, wrapping add)
//     running  +=  delta  on admission               (atomic_expiring_value.rs:36-42)
// Uniform-delta segments (the overwhelmingly common case: delta == 1) are decided in closed
// form from the rank inside the segment; anything else is walked sequentially by one lane.
#pragma once
#include "rl_kernels.hpp"

namespace rl {

// sort key = slot << 32 | hit index
__global__ __launch_bounds__(256) void k_ord_keys(const u32* __restrict__ ord_list, u32 n_ord,
                                                  const u32* __restrict__ hit_slot,
                                                  u64* __restrict__ keys) {
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_ord) return;
    const u32 idx = ord_list[j];
    keys[j] = ((u64)(hit_slot[idx] & SLOT_MASK) << 32) | idx;
}

// Segment heads publish their position in the cell.
__global__ __launch_bounds__(256) void k_ord_heads(Cell* __restrict__ table,
                                                   const u64* __restrict__ keys, u32 n_ord) {
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_ord) return;
    const u32 slot = (u32)(keys[j] >> 32);
    if (j == 0 || (u32)(keys[j - 1] >> 32) != slot) table[slot].seg = j;
}

// Flag segments that cannot use the closed form.
__global__ __launch_bounds__(256) void k_ord_uniform(Cell* __restrict__ table,
                                                     const u64* __restrict__ keys, u32 n_ord,
                                                     const Hit* __restrict__ hits, u64 now) {
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_ord) return;
    const u64 k = keys[j];
    const u32 slot = (u32)(k >> 32);
    Cell* c = &table[slot];
    const u32 seg = c->seg;
    const u32 d = hits[(u32)k].delta;
    const u32 d0 = hits[(u32)keys[seg]].delta;
    bool bad = d != d0;
    if (j == seg) {
        // s + total overflowed in k_decide: every hit of the cell is here and the walk must use
        // wrapping arithmetic.
        const u64 s = (c->expiry <= now) ? 0ull : c->value;
        u64 tmp;
        bad = bad || __builtin_add_overflow(s, c->pend & PEND_SUM_MASK, &tmp);
    }
    if (bad) c->nonuni = 1;
}

__global__ __launch_bounds__(256) void k_ord_resolve(Cell* __restrict__ table,
                                                     const u64* __restrict__ keys, u32 n_ord,
                                                     const Hit* __restrict__ hits,
                                                     const LimitDev* __restrict__ limits, u64 now,
                                                     uint8_t* __restrict__ verdict,
                                                     int32_t* __restrict__ first_limited) {
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_ord) return;
    const u64 k = keys[j];
    const u32 slot = (u32)(k >> 32);
    const u32 idx = (u32)k;
    Cell* c = &table[slot];
    const u32 seg = c->seg;
    const u64 M = limits[c->limit & ~SIMPLE_FLAG].max_value;
    const u64 s = (c->expiry <= now) ? 0ull : c->value;
    if (!c->nonuni) {
        const u64 d = hits[idx].delta;
        const u64 rank = j - seg;
        // every hit in a uniform segment fits on its own: s + d <= M, no overflow
        const bool adm = (d == 0) || (rank + 1 <= (M - s) / d);
        verdict[idx] = adm ? 0 : 1;
        if (first_limited) first_limited[idx] = adm ? -1 : (int32_t)idx;
        const bool last = (j + 1 == n_ord) || ((u32)(keys[j + 1] >> 32) != slot);
        if (last) {
            const u64 len = rank + 1;
            u64 n_adm = len;
            if (d != 0) {
                const u64 room = (M - s) / d;
                n_adm = len < room ? len : room;
            }
            c->aux = s + n_adm * d;
            c->amb = (n_adm > 0 || d == 0) ? AMB_ADMIT : AMB_DENY;
        }
    } else if (j == seg) {
        u64 r = s;
        bool any = false;
        for (u32 q = seg; q < n_ord; ++q) {
            const u64 kq = keys[q];
            if ((u32)(kq >> 32) != slot) break;
            const u32 iq = (u32)kq;
            const u64 sum = r + (u64)hits[iq].delta;  // wraps like the reference
            const bool adm = sum <= M;
            if (adm) {
                r = sum;
                any = true;
            }
            verdict[iq] = adm ? 0 : 1;
            if (first_limited) first_limited[iq] = adm ? -1 : (int32_t)iq;
        }
        c->aux = r;
        c->amb = any ? AMB_ADMIT : AMB_DENY;
    }
}

}  // namespace rl
