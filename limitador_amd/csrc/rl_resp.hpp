// rl_resp.hpp — the answer of the wire path built where the decisions are: the serialized
// envoy.service.ratelimit.v3.RateLimitResponse of every request of a batch (rls.proto:62-71,182: overall_code = 1,
// response_headers_to_add = 3 of HeaderValue { key = 1; value = 2 }) as ShouldRateLimit builds it
// (limitador-server/src/envoy_rls/server.rs:176-206), with the draft-03 headers of RateLimitHeaders::DraftVersion03:
//   X-RateLimit-Limit      `{max}, {max};w={secs}[;name="{name}"], ...` — the most restrictive counter's max, then one entry
//                          per counter of the request, the counters sorted by remaining (stable)
//   X-RateLimit-Remaining  remaining of the most restrictive counter
//   X-RateLimit-Reset      its expires_in, in whole seconds (Duration::as_secs)
// (CheckResult::response_header, limitador/src/lib.rs:235-275; the three headers sorted by key, envoy_rls/server.rs:44-57).
// Round 4 assembled these bytes on the host from four result arrays copied back per call (the derived counters, their
// remaining / expires_in, the request offsets: 26 MB for 262 144 messages) — 2.8 x the cost of the decisions.  Here one lane
// per request computes the length of its response (k_resp<false>), an exclusive scan gives the offsets, and the same code
// writes the bytes (k_resp<true>): what crosses PCIe is the responses themselves.  What a LIMIT contributes to
// X-RateLimit-Limit — `, {max};w={secs}[;name="{name}"]` — does not depend on the request: the host formats it once per limit
// (rl_resp_table_set), the kernel copies it.  gfx950 only.
#pragma once
#include "rl_kernels.hpp"
#include "rl_wire.hpp"

namespace rl {

struct RespArgs {
    const uint8_t* blob;       // the limits' fragments (allocated in multiples of 16 bytes)
    u32 blob_len;
    const WireStr* frag;       // [n_frag], indexed by limit id
    u32 n_frag;
    const LimitDev* limits;    // the engine's limit rows (max_value)
    u32 n_limits;
    const int32_t* status;     // per request: 0 = answered; anything else derives no response (-101, no domain, is the EMPTY
                               // message: Code::Unknown = 0 is the proto3 default, server.rs:105-115); null: all 0
    const uint8_t* verdict;    // per request
    u64 out_cap;               // WRITE: bytes `out` holds — a block whose responses end beyond it writes nothing (the caller
                               // sized `out` from a bound and checks the total afterwards)
    const u32* req_off;        // [n + 1] into hits / remaining / expires_in (null without headers)
    const Hit* hits;
    const u64* remaining;
    const u64* expires_in;
    u32 n;
    u32 with_headers;
};

__device__ __forceinline__ u32 resp_varint_len(u32 v) { return v < 0x80u ? 1u : v < 0x4000u ? 2u : v < 0x200000u ? 3u : v < 0x10000000u ? 4u : 5u; }
__device__ __forceinline__ u32 resp_dec_len(u64 v) {
    u32 n = 1;
    while (v >= 10ull) {
        v /= 10ull;
        ++n;
    }
    return n;
}

template <bool WRITE>
struct RespOut {
    uint8_t* p;
    u32 n;
    __device__ __forceinline__ void byte(uint8_t b) {
        if (WRITE) p[n] = b;
        ++n;
    }
    __device__ __forceinline__ void varint(u32 v) {
        while (v >= 0x80u) {
            byte((uint8_t)(v | 0x80u));
            v >>= 7;
        }
        byte((uint8_t)v);
    }
    __device__ __forceinline__ void bytes(const uint8_t* s, u32 len) {
        if (WRITE) {
            u32 k = 0;
            for (; k + 4u <= len; k += 4u) {  // (four loads in flight, then four stores: a byte at a time is one round trip each)
                const uint8_t b0 = s[k], b1 = s[k + 1], b2 = s[k + 2], b3 = s[k + 3];
                p[n + k] = b0;
                p[n + k + 1] = b1;
                p[n + k + 2] = b2;
                p[n + k + 3] = b3;
            }
            for (; k < len; ++k) p[n + k] = s[k];
        }
        n += len;
    }
    __device__ __forceinline__ void dec(u64 v) {
        const u32 len = resp_dec_len(v);
        if (WRITE)
            for (u32 k = len; k-- > 0;) {
                p[n + k] = (uint8_t)('0' + (u32)(v % 10ull));
                v /= 10ull;
            }
        n += len;
    }
    // HeaderValue { key, value } as one element of response_headers_to_add; the caller writes the vlen value bytes next
    __device__ __forceinline__ void header(const char* key, u32 klen, u32 vlen) {
        const u32 hv = 1u + resp_varint_len(klen) + klen + 1u + resp_varint_len(vlen) + vlen;
        byte(0x1A);  // field 3, length-delimited
        varint(hv);
        byte(0x0A);  // HeaderValue.key
        varint(klen);
        bytes(reinterpret_cast<const uint8_t*>(key), klen);
        byte(0x12);  // HeaderValue.value
        varint(vlen);
    }
};

__device__ __forceinline__ u32 resp_lid(const RespArgs& A, u32 q) { return load_hit(A.hits, q).limit & ~SIMPLE_FLAG; }
// `, {max};w={secs}[;name=...]` of a limit; one this table does not know contributes `, 0;w=0` (the host did the same)
template <bool WRITE>
__device__ __forceinline__ void resp_frag(const RespArgs& A, RespOut<WRITE>& o, u32 lid) {
    if (lid < A.n_frag) {
        const WireStr f = A.frag[lid];
        o.bytes(A.blob + f.off, f.len);
    } else {
        const char z[] = ", 0;w=0";
        o.bytes(reinterpret_cast<const uint8_t*>(z), 7u);
    }
}

// One lane per request.  WRITE = false: len[r] = the length of request r's response (len[n] = 0: the scan's total lands
// there).  WRITE = true: the bytes, at out + off[r] — through LDS: a lane writes its response byte by byte, and byte stores
// of 256 lanes to 256 different lines were 39 M memory transactions for 262 144 responses (405 us, 96 G/s: the chip's
// transaction rate, not its bandwidth).  The 256 responses of a workgroup are one contiguous range of the output, so the
// lanes build them in LDS — laid out with the range's misalignment, so that LDS word k is global word k — and the
// workgroup copies the range out as aligned 16-byte words, coalesced.  A range that does not fit (RESP_LDS bytes: names of
// hundreds of bytes on most counters) is written directly, as before.  The limits' fragments — most of a response's bytes,
// read by every lane at its own address — are copied into LDS first when they fit (RESP_BLOB_LDS): with LDS staging alone the
// kernel was still 359 us, a chain of one global byte load per byte per lane.
constexpr u32 RESP_LDS = 44u * 1024u;
constexpr u32 RESP_BLOB_LDS = 16u * 1024u;

// `out` may be host memory the device can write (the engine's pinned staging): the copy-out below is then the transfer
// itself — 16-byte stores, coalesced, what a copy kernel would issue — and the responses need no device buffer and no copy
// command.  block0: the launch covers the workgroups [block0, block0 + gridDim.x) (the engine launches the kernel in a few
// pieces with an event behind each, so that the host can hand on the first responses while the last ones are written).
template <bool WRITE>
__global__ __launch_bounds__(256) void k_resp(RespArgs A, u32* __restrict__ len, const u32* __restrict__ off, uint8_t* __restrict__ out,
                                              u32 block0, u32 n_blocks) {
    __shared__ __attribute__((aligned(16))) uint8_t s_buf[WRITE ? RESP_LDS : 16u];
    __shared__ __attribute__((aligned(16))) uint8_t s_blob[WRITE ? RESP_BLOB_LDS : 16u];
    // (WRITE: the launch's workgroups share the blocks [block0, block0 + n_blocks) round-robin — with host memory as `out`
    // the NUMBER of workgroups writing at once is the engine's choice, not the batch's size: see responses_locked)
    const uint8_t* const blob_global = A.blob;
    for (u32 block = block0 + blockIdx.x; block < block0 + n_blocks; block += gridDim.x) {
    if (WRITE && block != block0 + blockIdx.x) {
        __syncthreads();  // (s_buf is reused)
        A.blob = blob_global;
    }
    const u32 r = block * 256 + threadIdx.x;
    u32 g0 = 0, g_len = 0, mis = 0;
    bool staged = false;
    if (WRITE) {  // (block-uniform)
        const u32 r0 = block * 256u, r1 = r0 + 256u < A.n ? r0 + 256u : A.n;
        if (r0 >= A.n) return;  // (past the last request: so is every later block of this workgroup)
        g0 = off[r0];
        g_len = off[r1] - g0;
        if ((u64)g0 + g_len > A.out_cap) return;  // (so does every later block: offsets ascend)
        mis = (u32)((reinterpret_cast<unsigned long long>(out) + g0) & 15ull);
        staged = g_len + mis <= RESP_LDS;
        if (A.with_headers && A.blob_len <= RESP_BLOB_LDS) {
            if (block == block0 + blockIdx.x) {
                for (u32 k = threadIdx.x * 4u; k < A.blob_len; k += 1024u)  // (the blob is allocated in multiples of 16 bytes)
                    *reinterpret_cast<u32*>(s_blob + k) = *reinterpret_cast<const u32*>(blob_global + k);
                __syncthreads();
            }
            A.blob = s_blob;
        }
    }
    if (!WRITE && r == A.n) len[r] = 0;
    RespOut<WRITE> o{nullptr, 0u};
    if (WRITE && r < A.n) o.p = staged ? s_buf + mis + (off[r] - g0) : out + off[r];
    if (r < A.n && (!A.status || A.status[r] == 0)) {
        o.byte(0x08);  // overall_code
        o.byte(A.verdict[r] ? 2 : 1);  // OVER_LIMIT : OK
        const u32 q0 = A.with_headers ? A.req_off[r] : 0u, q1 = A.with_headers ? A.req_off[r + 1] : 0u;
        if (q1 > q0) {
            // the most restrictive counter: least remaining, the storage's order among equals (a stable sort's first)
            u32 f = q0;
            u64 rem_f = A.remaining[q0];
            u32 frag_sum = 0;
            for (u32 q = q0; q < q1; ++q) {
                const u64 x = A.remaining[q];
                if (x < rem_f) {
                    rem_f = x;
                    f = q;
                }
                const u32 l = resp_lid(A, q);
                frag_sum += l < A.n_frag ? A.frag[l].len : 7u;
            }
            const u32 lf = resp_lid(A, f);
            const u64 max_f = lf < A.n_limits && lf < A.n_frag ? A.limits[lf].max_value : 0ull;
            o.header("X-RateLimit-Limit", 17u, resp_dec_len(max_f) + frag_sum);
            o.dec(max_f);
            // the counters in (remaining, position) order: the successor of (prev_rem, prev_q) is the least pair above it
            // (a handful of counters per request: no array, no scratch)
            u64 prev_rem = 0;
            u32 prev_q = q0;
            for (u32 k = 0; k < q1 - q0; ++k) {
                u64 best_rem = ~0ull;
                u32 best_q = q1;
                for (u32 q = q0; q < q1; ++q) {
                    const u64 x = A.remaining[q];
                    const bool above = k == 0 || x > prev_rem || (x == prev_rem && q > prev_q);
                    if (above && (best_q == q1 || x < best_rem)) {
                        best_rem = x;
                        best_q = q;
                    }
                }
                resp_frag<WRITE>(A, o, resp_lid(A, best_q));
                prev_rem = best_rem;
                prev_q = best_q;
            }
            o.header("X-RateLimit-Remaining", 21u, resp_dec_len(rem_f));
            o.dec(rem_f);
            const u64 secs = A.expires_in[f] / 1000000ull;  // Duration::as_secs
            o.header("X-RateLimit-Reset", 17u, resp_dec_len(secs));
            o.dec(secs);
        }
    }
    if (!WRITE) {
        if (r < A.n) len[r] = o.n;
        return;
    }
    if (!staged) continue;
    __syncthreads();
    // LDS [mis, mis + g_len) -> out [g0, g0 + g_len): aligned 16-byte words in the middle, bytes at the two ragged ends
    uint8_t* const gbase = out + g0 - mis;  // 16-byte aligned
    const u32 end = mis + g_len, n_words = (end + 15u) >> 4;
    for (u32 w = threadIdx.x; w < n_words; w += 256u) {
        const u32 b = w << 4;
        if (b >= mis && b + 16u <= end) {
            *reinterpret_cast<uint4*>(gbase + b) = *reinterpret_cast<const uint4*>(s_buf + b);
        } else {
            for (u32 k = b < mis ? mis : b; k < b + 16u && k < end; ++k) gbase[k] = s_buf[k];
        }
    }
    }  // blocks of this workgroup
}

}  // namespace rl
