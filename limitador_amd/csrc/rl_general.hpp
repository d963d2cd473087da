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

__global__ __launch_bounds__(256) void k_gen_tiny(Cell* __restrict__ table, u32 log2cap, u64 seed,
                                                  const Hit* __restrict__ hits, u32 n_hits,
                                                  const u32* __restrict__ req_off, u32 n_req,
                                                  const LimitDev* __restrict__ limits, u32 n_limits, u64 now, int load,
                                                  uint8_t* __restrict__ verdict, int32_t* __restrict__ first_limited,
                                                  u64* __restrict__ remaining, u64* __restrict__ expires_in,
                                                  Status* host_status, u32 done_seq) {
    __shared__ u64 s_key[GT_ENT], s_value[GT_ENT], s_expiry[GT_ENT];
    __shared__ u32 s_slot[GT_ENT], s_limit[GT_ENT], s_flags[GT_ENT];
    __shared__ u64 h_max[GT_MAX], h_win[GT_MAX];
    __shared__ u32 h_delta[GT_MAX], h_lim[GT_MAX];
    __shared__ unsigned short h_ent[GT_MAX];
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
            h_delta[tid] = h.delta;
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
    }
    __syncthreads();
    if (tid < n_hits && my_slot != SLOT_INVALID && s_limit[my_ent] != h.limit) atomicOr(&s_err, ERRBIT_KEY_LIMIT);
    __syncthreads();
    const u32 err_all = s_err;
    // ---- 2: one lane replays the requests, in_memory.rs:72-156 ----------------------------------------
    if (tid == 0 && !err_all) {
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
                    const u64 sum = value + (u64)h_delta[j];                     // wraps like the release build
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
                            s_value[e] = (u64)h_delta[j];
                        } else {
                            s_value[e] += (u64)h_delta[j];
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // completion word written last, with the counts, as ONE 16-byte store (see apply_finish)
        typedef u32 u32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(u32x4{s_err, s_dropped, s_created, done_seq},
                                    reinterpret_cast<u32x4*>(host_status));
    }
}

}  // namespace rl
