// rl_route.hpp — multi-GPU descriptor routing: stable partition of a batch by owner shard.
//
// Keys are hash-partitioned across the GPUs of a node (owner = owner_of(key)); each GPU owns a
// private table for its share of the key space.  Before the all-to-all every ingress GPU groups
// its hits by owner.  The partition is STABLE (original order inside each group), so that the
// concatenation "from rank 0 | from rank 1 | ..." an owner receives is exactly the global trace
// order restricted to its keys — the order the sequential reference semantics are defined on —
// without shipping per-hit sequence numbers.
//
// Three small kernels (count per tile, scan of the tile x owner table, scatter).
#pragma once
#include "rl_kernels.hpp"

namespace rl {

constexpr int ROUTE_BLOCK = 256;
constexpr int ROUTE_ROUNDS = 8;                         // 64-hit rounds per wave
constexpr int ROUTE_WAVE_TILE = 64 * ROUTE_ROUNDS;      // contiguous hits owned by one wave
constexpr int ROUTE_TILE = 4 * ROUTE_WAVE_TILE;         // hits per workgroup
constexpr int ROUTE_MAX_WORLD = 16;
constexpr int ROUTE_MAX_BLOCKS = 8192;

// cnt layout: [owner][block], owner-major, so one linear exclusive scan yields final offsets.
__global__ __launch_bounds__(ROUTE_BLOCK) void k_route_count(const Hit* __restrict__ hits, u32 n,
                                                              u64 seed, u32 world,
                                                              u32* __restrict__ cnt) {
    __shared__ u32 s_cnt[ROUTE_MAX_WORLD];
    const u32 tid = threadIdx.x;
    if (tid < ROUTE_MAX_WORLD) s_cnt[tid] = 0;
    __syncthreads();
    // (per wave: one ballot per owner and round, one LDS atomic per owner — an LDS atomic per HIT put all 2048 of a
    // workgroup's on ONE word at world 1: 15 us for the launch, gpurun_out/r13d)
    const u32 base = blockIdx.x * ROUTE_TILE;
    const u32 lane = tid & 63u;
    u32 own[ROUTE_TILE / ROUTE_BLOCK];
#pragma unroll
    for (int r = 0; r < ROUTE_TILE / ROUTE_BLOCK; ++r) {
        const u32 i = base + r * ROUTE_BLOCK + tid;
        own[r] = i < n ? owner_of(hits[i].key, seed, world) : 0xFFFFFFFFu;
    }
    for (u32 o = 0; o < world; ++o) {
        u32 c = 0;
#pragma unroll
        for (int r = 0; r < ROUTE_TILE / ROUTE_BLOCK; ++r) c += (u32)__popcll(__ballot(own[r] == o));
        if (lane == 0 && c) atomicAdd(&s_cnt[o], c);
    }
    __syncthreads();
    if (tid < world) cnt[tid * gridDim.x + blockIdx.x] = s_cnt[tid];
}

// Single workgroup: exclusive scan of cnt[world * nblk] in place; counts[o] = group sizes.
__global__ __launch_bounds__(256) void k_route_scan(u32* __restrict__ cnt, u32 nblk, u32 world,
                                                    u32* __restrict__ counts) {
    __shared__ u32 s_part[4];
    __shared__ u32 s_owner_tot[ROUTE_MAX_WORLD];
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u32 total = nblk * world;
    if (tid < ROUTE_MAX_WORLD) s_owner_tot[tid] = 0;
    __syncthreads();
    // 256 elements per step, one per thread (coalesced), the next step's element requested before this step's scan (one
    // thread walking the 256 partial sums in a row was most of this kernel's 5-11 us: gpurun_out/r13d)
    u32 carry = 0;
    u32 nxt = tid < total ? cnt[tid] : 0u;
    for (u32 base = 0; base < total; base += 256) {
        const u32 q = base + tid;
        const u32 v = nxt;
        nxt = q + 256 < total ? cnt[q + 256] : 0u;
        if (q < total && v) atomicAdd(&s_owner_tot[q / nblk], v);
        u32 inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 o2 = __shfl_up(inc, off);
            if ((int)lane >= off) inc += o2;
        }
        __syncthreads();  // (s_part of the step before has been read)
        if (lane == 63) s_part[wave] = inc;
        __syncthreads();
        u32 ex = carry + inc - v;
        for (u32 w2 = 0; w2 < wave; ++w2) ex += s_part[w2];
        if (q < total) cnt[q] = ex;
        carry += s_part[0] + s_part[1] + s_part[2] + s_part[3];
    }
    __syncthreads();
    if (tid < world) counts[tid] = s_owner_tot[tid];
}

__global__ __launch_bounds__(ROUTE_BLOCK) void k_route_scatter(const Hit* __restrict__ hits, u32 n,
                                                                u64 seed, u32 world,
                                                                const u32* __restrict__ offs,
                                                                Hit* __restrict__ out,
                                                                u32* __restrict__ perm) {
    __shared__ u32 s_wcnt[4][ROUTE_MAX_WORLD];  // per wave, per owner: count, then running offset
    const u32 tid = threadIdx.x;
    const u32 wave = tid >> 6;
    const u32 lane = tid & 63;
    if (tid < 4 * ROUTE_MAX_WORLD) (&s_wcnt[0][0])[tid] = 0;
    __syncthreads();
    const u32 wbase = blockIdx.x * ROUTE_TILE + wave * ROUTE_WAVE_TILE;
    Hit h[ROUTE_ROUNDS];
    u32 own[ROUTE_ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUTE_ROUNDS; ++r) {
        const u32 i = wbase + r * 64 + lane;
        own[r] = 0xFFFFFFFFu;
        if (i < n) {
            h[r] = load_hit(hits, i);
            own[r] = owner_of(h[r].key, seed, world);
        }
    }
    // pass 1: per-wave counts (ballots are wave-uniform: lane 0 publishes)
    for (u32 o = 0; o < world; ++o) {
        u32 c = 0;
#pragma unroll
        for (int r = 0; r < ROUTE_ROUNDS; ++r) c += (u32)__popcll(__ballot(own[r] == o));
        if (lane == 0) s_wcnt[wave][o] = c;
    }
    __syncthreads();
    // wave offsets: block base for the owner + counts of earlier waves
    if (tid < world) {
        u32 run = offs[tid * gridDim.x + blockIdx.x];
        for (int w = 0; w < 4; ++w) {
            const u32 c = s_wcnt[w][tid];
            s_wcnt[w][tid] = run;
            run += c;
        }
    }
    __syncthreads();
    // pass 2: stable scatter.  For each owner the running offset lives in a (wave-uniform)
    // register, rounds are visited in order, so positions inside a group keep the input order.
    u32 dest[ROUTE_ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUTE_ROUNDS; ++r) dest[r] = 0;
    for (u32 o = 0; o < world; ++o) {
        u32 run = s_wcnt[wave][o];
#pragma unroll
        for (int r = 0; r < ROUTE_ROUNDS; ++r) {
            const u64 b = __ballot(own[r] == o);
            if (own[r] == o) dest[r] = run + (u32)__popcll(b & ((1ull << lane) - 1ull));
            run += (u32)__popcll(b);
        }
    }
#pragma unroll
    for (int r = 0; r < ROUTE_ROUNDS; ++r) {
        if (own[r] != 0xFFFFFFFFu) {
            const u32 i = wbase + r * 64 + lane;
            uint4 v;
            v.x = (u32)h[r].key;
            v.y = (u32)(h[r].key >> 32);
            v.z = h[r].limit;
            v.w = h[r].delta;
            *reinterpret_cast<uint4*>(out + dest[r]) = v;
            perm[dest[r]] = i;
        }
    }
}

__global__ __launch_bounds__(256) void k_unpermute_u8(const uint8_t* __restrict__ src,
                                                      const u32* __restrict__ perm, u32 n,
                                                      uint8_t* __restrict__ dst) {
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) dst[perm[j]] = src[j];
}

__global__ __launch_bounds__(256) void k_unpermute_u64(const u64* __restrict__ src, const u32* __restrict__ perm, u32 n,
                                                       u64* __restrict__ dst) {
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) dst[perm[j]] = src[j];
}

// ---- the ingress side of the key-sharded MULTI-counter step (include/rl_sharded.h, rl_sharded_check_requests_device) ----
// A request's counters live on several owners; the all-or-nothing rule (in_memory.rs:141-153) is the AND of its hits'
// pass flags, taken where the request entered.  Hits travel in ROUTED order (stable partition by owner, perm[j] = the
// ingress index of routed hit j); everything per request happens in ingress order.

// request of every hit (ingress order), and the id that travels with routed hit j: base + its request
__global__ __launch_bounds__(256) void k_req_of_hit(const u32* __restrict__ req_off, u32 n_req, u32* __restrict__ req_of_hit) {
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_req) return;
    for (u32 q = req_off[r]; q < req_off[r + 1]; ++q) req_of_hit[q] = r;
}
__global__ __launch_bounds__(256) void k_req_id_sorted(const u32* __restrict__ req_of_hit, const u32* __restrict__ perm, u32 n,
                                                       u32 base, u32* __restrict__ req_id_sorted) {
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) req_id_sorted[j] = base + req_of_hit[perm[j]];
}

// One round, ingress side: a request is admitted iff every one of its hits passed on its owner (pass_home: the flags
// back in ingress order); first[r] = its first failing hit (the counter the reference reports: in_memory.rs:90-99,
// 141-143) or -1; *changed is raised if the admitted set differs from the previous round's (first round: from
// "everything admitted", which is what the owners assumed).
__global__ __launch_bounds__(256) void k_req_and(const uint8_t* __restrict__ pass_home, const u32* __restrict__ req_off, u32 n_req,
                                                 u32 first_round, uint8_t* __restrict__ adm, int32_t* __restrict__ first,
                                                 uint8_t* __restrict__ verdict, u32* __restrict__ changed) {
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    bool ch = false;
    if (r < n_req) {
        int32_t f = -1;
        const u32 e = req_off[r + 1];
        for (u32 q = req_off[r]; q < e; ++q)
            if (pass_home[q] == 0) {
                f = (int32_t)q;
                break;
            }
        const uint8_t a = f < 0 ? 1 : 0;
        ch = first_round ? a == 0 : adm[r] != a;
        adm[r] = a;
        first[r] = f;
        verdict[r] = a ? 0 : 1;
    }
    if (__syncthreads_or(ch ? 1 : 0) && threadIdx.x == 0) atomicOr(changed, 1u);
}

// per routed hit: a byte of its request (the admission for the owners' next round)
__global__ __launch_bounds__(256) void k_req_spread(const uint8_t* __restrict__ req_byte, const u32* __restrict__ req_of_hit,
                                                    const u32* __restrict__ perm, u32 n, uint8_t* __restrict__ out_sorted) {
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) out_sorted[j] = req_byte[req_of_hit[perm[j]]];
}

// per routed hit: did its request's walk get to it — it stops at the first limited counter unless the values are loaded
// (in_memory.rs:109-113,129-133)
__global__ __launch_bounds__(256) void k_req_reached(const int32_t* __restrict__ first, const u32* __restrict__ req_of_hit,
                                                     const u32* __restrict__ perm, u32 n, uint8_t* __restrict__ out_sorted) {
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const u32 pos = perm[j];
    const int32_t f = first[req_of_hit[pos]];
    out_sorted[j] = (f < 0 || pos <= (u32)f) ? 1 : 0;
}

// Up to four device-to-device copies in ONE launch (the router's own segments of an exchange: what a rank sends to
// itself).  hipMemcpyAsync costs the host 8-10 us per call whatever it moves; a kernel launch ~3.
constexpr u32 COPY_SEGS_MAX = 8;
struct CopySegs {
    void* dst[COPY_SEGS_MAX];
    const void* src[COPY_SEGS_MAX];
    u64 bytes[COPY_SEGS_MAX];
    u32 n;
};
__global__ __launch_bounds__(256) void k_copy_segs(const CopySegs S) {
    const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x, stride = (u64)gridDim.x * 256;
    for (u32 k = 0; k < S.n; ++k) {
        const u64 b = S.bytes[k];
        if ((((u64)S.dst[k] | (u64)S.src[k]) & 15ull) == 0ull) {  // 16-byte words, then the bytes behind the last whole one
            const uint4* s4 = static_cast<const uint4*>(S.src[k]);
            uint4* d4 = static_cast<uint4*>(S.dst[k]);
            for (u64 i = gid; i < (b >> 4); i += stride) d4[i] = s4[i];
            const uint8_t* s1 = static_cast<const uint8_t*>(S.src[k]);
            uint8_t* d1 = static_cast<uint8_t*>(S.dst[k]);
            for (u64 i = (b & ~15ull) + gid; i < b; i += stride) d1[i] = s1[i];
        } else {
            const uint8_t* s1 = static_cast<const uint8_t*>(S.src[k]);
            uint8_t* d1 = static_cast<uint8_t*>(S.dst[k]);
            for (u64 i = gid; i < b; i += stride) d1[i] = s1[i];
        }
    }
}

}  // namespace rl
