// rl_general.hpp — the general, exact form of check_and_update for a batch: multi-counter
// requests (all-or-nothing across the counters of a request, in_memory.rs:141-153) and/or
// load_counters (remaining / expires_in of every counter, in_memory.rs:87-95,114-116).
//
// Sequential definition: request i is decided against the table state left by requests < i.
// With A the set of admitted requests, hit h of request i on cell c reads
//     v_h = value_at(c, now) + SUM{ delta_j : j < i, j in A, j touches c }        (wrapping u64)
// and passes iff v_h + delta_i <= max_h; request i is admitted iff all its hits pass.  A is the
// unique fixpoint of that map (induction on i), and iterating from "everything admitted" fixes at
// least one more request of the trace prefix per round, so the loop below terminates with exactly
// the sequential answer (SURVEY.md §7 hard part 1).
//
// Data flow per round, over the hits sorted by (cell slot, hit index):
//   k_gen_contrib   c[j] = admitted[req(j)] ? (delta, 1) : (0, 0)
//   exclusive scan  G = scan(c)            (one global scan; a cell's prefix is G[j] - G[seg start],
//                                           exact in modular arithmetic)
//   k_gen_eval      per hit: v_h, pass, remaining, expires_in
//   k_gen_requests  per request: AND of its hits' pass flags, first failing hit, changed flag
// then k_gen_finish publishes each touched cell's final value for k_commit.
#pragma once
#include "rl_kernels.hpp"

namespace rl {

struct Contrib {
    u64 sum;
    u64 cnt;
};
struct ContribPlus {
    __host__ __device__ Contrib operator()(const Contrib& a, const Contrib& b) const {
        return Contrib{a.sum + b.sum, a.cnt + b.cnt};
    }
};

// hit index -> request index
__global__ __launch_bounds__(256) void k_gen_hit_req(const u32* __restrict__ req_off, u32 n_req,
                                                     u32* __restrict__ hit_req) {
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_req) return;
    const u32 b = req_off[r], e = req_off[r + 1];
    for (u32 q = b; q < e; ++q) hit_req[q] = r;
}

__global__ __launch_bounds__(256) void k_gen_keys(const u32* __restrict__ hit_slot, u32 n,
                                                  u64* __restrict__ keys) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    keys[i] = ((u64)(hit_slot[i] & SLOT_MASK) << 32) | i;
}

__global__ __launch_bounds__(256) void k_gen_fill_u8(uint8_t* __restrict__ p, u32 n, uint8_t v) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ __launch_bounds__(256) void k_gen_contrib(const u64* __restrict__ keys, u32 n,
                                                     const Hit* __restrict__ hits,
                                                     const u32* __restrict__ hit_req,
                                                     const uint8_t* __restrict__ admitted,
                                                     Contrib* __restrict__ c) {
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const u32 idx = (u32)keys[j];
    const u32 r = hit_req ? hit_req[idx] : idx;
    const bool a = admitted[r] != 0;
    c[j] = Contrib{a ? (u64)hits[idx].delta : 0ull, a ? 1ull : 0ull};
}

// Value a hit reads, pass flag, and the load_counters outputs.
__global__ __launch_bounds__(256) void k_gen_eval(
    const Cell* __restrict__ table, const u64* __restrict__ keys, u32 n, const Contrib* __restrict__ G,
    const Hit* __restrict__ hits, const u32* __restrict__ hit_req, const uint8_t* __restrict__ admitted,
    const LimitDev* __restrict__ limits, u64 now, uint8_t* __restrict__ pass_hit,
    u64* __restrict__ remaining, u64* __restrict__ expires_in) {
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const u64 k = keys[j];
    const u32 idx = (u32)k;
    const Cell* c = &table[(u32)(k >> 32)];
    const u32 seg = c->seg;
    const u32 r = hit_req ? hit_req[idx] : idx;
    const Contrib g = G[j], g0 = G[seg];
    u64 pre_sum = g.sum - g0.sum;
    u64 pre_cnt = g.cnt - g0.cnt;
    // exclusive BY REQUEST: hits of the same request on the same cell all read the value before
    // any of them is applied (in_memory.rs:105-139 reads, :146-153 updates afterwards)
    if (admitted[r]) {
        for (u32 q = j; q > seg;) {
            --q;
            const u32 iq = (u32)keys[q];
            if ((hit_req ? hit_req[iq] : iq) != r) break;
            pre_sum -= (u64)hits[iq].delta;
            pre_cnt -= 1;
        }
    }
    const LimitDev L = limits[c->limit & ~SIMPLE_FLAG];
    const u64 expiry = c->expiry;
    const bool expired0 = expiry <= now;
    const u64 s = expired0 ? 0ull : c->value;
    const u64 v = (L.window_us == 0) ? 0ull : s + pre_sum;
    const u64 sum = v + (u64)hits[idx].delta;  // wraps like the reference
    const bool pass = sum <= L.max_value;
    pass_hit[idx] = pass ? 1 : 0;
    if (remaining) remaining[idx] = pass ? L.max_value - sum : 0ull;  // checked_sub().unwrap_or_default()
    if (expires_in) {
        u64 ttl;
        if (L.window_us == 0) ttl = 0;
        else if (!expired0) ttl = expiry - now;
        else ttl = pre_cnt > 0 ? L.window_us : 0ull;  // an earlier admitted hit reopened the window
        expires_in[idx] = ttl;
    }
}

__global__ __launch_bounds__(256) void k_gen_requests(const u32* __restrict__ req_off, u32 n_req,
                                                      const uint8_t* __restrict__ pass_hit,
                                                      uint8_t* __restrict__ admitted,
                                                      uint8_t* __restrict__ verdict,
                                                      int32_t* __restrict__ first_limited,
                                                      u32* __restrict__ changed) {
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_req) return;
    const u32 b = req_off ? req_off[r] : r;
    const u32 e = req_off ? req_off[r + 1] : r + 1;
    int32_t first = -1;
    for (u32 q = b; q < e; ++q)
        if (!pass_hit[q]) {
            first = (int32_t)q;
            break;
        }
    const uint8_t adm = first < 0 ? 1 : 0;
    if (admitted[r] != adm) {
        admitted[r] = adm;
        *changed = 1u;
    }
    verdict[r] = adm ? 0 : 1;
    if (first_limited) first_limited[r] = first;
}

// After convergence: final value of every touched cell (for k_commit) — written by the last
// element of each segment; 0-second windows record their last admitted delta.
__global__ __launch_bounds__(256) void k_gen_finish(Cell* __restrict__ table,
                                                    const u64* __restrict__ keys, u32 n,
                                                    const Contrib* __restrict__ G,
                                                    const Hit* __restrict__ hits,
                                                    const u32* __restrict__ hit_req,
                                                    const uint8_t* __restrict__ admitted,
                                                    const LimitDev* __restrict__ limits, u64 now) {
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const u64 k = keys[j];
    const u32 slot = (u32)(k >> 32);
    const u32 idx = (u32)k;
    Cell* c = &table[slot];
    const u32 r = hit_req ? hit_req[idx] : idx;
    const bool a = admitted[r] != 0;
    const u64 d = hits[idx].delta;
    const LimitDev L = limits[c->limit & ~SIMPLE_FLAG];
    if (L.window_us == 0) {
        if (a) atomicMax(&c->aux, ((u64)(j + 1) << 32) | d);
        return;
    }
    const bool last = (j + 1 == n) || ((u32)(keys[j + 1] >> 32) != slot);
    if (last) {
        const Contrib g = G[j], g0 = G[c->seg];
        const u64 tot_sum = g.sum - g0.sum + (a ? d : 0ull);
        const u64 tot_cnt = g.cnt - g0.cnt + (a ? 1ull : 0ull);
        const u64 s = (c->expiry <= now) ? 0ull : c->value;
        c->aux = s + tot_sum;
        c->amb = tot_cnt ? AMB_ADMIT : AMB_DENY;
    }
}

// !load_counters only: a request stops at its first limited counter (in_memory.rs:109-113,
// 129-133), so the qualified cells of its later counters are never created.  Cells created by
// k_probe carry pad = 1; every hit that the sequential walk reaches confirms its cell (pad = 2);
// k_commit drops the unconfirmed ones.
__global__ __launch_bounds__(256) void k_gen_reach(Cell* __restrict__ table,
                                                   const u32* __restrict__ req_off, u32 n_req,
                                                   const u32* __restrict__ hit_slot,
                                                   const int32_t* __restrict__ first_limited) {
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_req) return;
    const u32 b = req_off[r];
    u32 e = req_off[r + 1];
    const int32_t f = first_limited[r];
    if (f >= 0) e = (u32)f + 1;
    for (u32 q = b; q < e; ++q) {
        Cell* c = &table[hit_slot[q] & SLOT_MASK];
        if (c->pad == 1u) c->pad = 2u;
    }
}

}  // namespace rl
