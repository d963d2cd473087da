// rl_wire.hpp — the wire path without host dictionaries: serialized RateLimitRequests in, counters out.
//
// What the dictionary path (rl_match.hpp + csrc/host/ingest.cpp) does on the host — walk the protobuf, intern every
// namespace / key / value string under a reader-writer lock, pack ids — happens HERE, one thread per message:
//     envoy_rls/server.rs:97-137      domain -> namespace, descriptors[i].entries -> the context, one map per descriptor (a
//                                     repeated key keeps its LAST value: HashMap::insert), hits_addend 0 -> 1
//     lib.rs:507-522, limit.rs:157-174, 133-148, counter.rs:19-31   counters_that_apply (the slot form of rl_match.hpp)
//     storage/keys.rs:220-248         the counter's key: a hash of its canonical key bytes (include/rl_keyhash.h)
// The host only concatenates the messages.  Strings are compared as BYTES (namespaces, descriptor keys, condition
// literals: a hash picks the candidate, the bytes decide), so matching is exact; only the counter's identity is a hash,
// and that one is checked where it is used (rl_keyhash.h: the 32-bit check word of the cell).
//
//   k_wire_count   decode + slot values + evaluate: per request the bit mask of the limits that apply, the hashes of the
//                  values its variables read, namespace, delta, status; per WORKGROUP the number of counters
//   (k_match_scan2 of rl_match.hpp: the workgroups' totals -> offsets, {total, error bits} to the host)
//   k_wire_fill    the rl_hit records + the check word of every hit, in the order the storage walks a request's counters
#pragma once
#include "../../include/rl_keyhash.h"
#include "rl_match.hpp"

namespace rl {

constexpr u32 WIRE_BLOB_MAX = 8192;  // bytes of all table strings (namespaces, slot keys, literals)
constexpr u32 WIRE_LIT_TAB = 1024;   // open-addressing table of the condition literals (at most WIRE_LIT_TAB / 2 of them)
constexpr u32 WIRE_NON_LITERAL = (1u << MATCH_VAL_BITS) - 1u;  // "a value no condition names": equal to no literal id
constexpr int32_t WIRE_ST_UNKNOWN_DOMAIN = -101;               // == RLI_UNKNOWN_DOMAIN
constexpr int32_t WIRE_ST_MALFORMED = -1;                      // == RL_ERR_INVALID

struct WireStr {  // == rl_wire_str
    u32 off, len;
};
struct WireLit {
    u64 h1;
    u32 off;
    unsigned short len, id;  // id 0xFFFF: empty
};
static_assert(sizeof(WireLit) == 16, "WireLit");

struct WireTables {
    const uint8_t* blob;
    u32 blob_len;
    const WireStr* ns;  // [n_ns], index = namespace id (0: the namespace without limits, "")
    u32 n_ns;
    WireStr slot_key[MATCH_SLOTS];
    u32 slot_desc[MATCH_SLOTS];  // the descriptor the slot's key is read from: slot = (descriptors[slot_desc], slot_key)
    u64 desc_mask;               // bit i: some slot reads descriptors[i] (bit 0 always) — the others are skipped by wire type,
                                 // not decoded, exactly like csrc/host/ingest.cpp's reader does
    const WireLit* lit;    // [WIRE_LIT_TAB]
    const u64* prefix;     // [limit id][2]: rl_kh_bytes of the limit's canonical prefix
    rl_hkey hkey;          // the ingest's secret: every hash of this path is SipHash-2-4-128 under it (include/rl_keyhash.h)
    u32 var_slot_mask;     // slots some limit reads as a variable (their values' hashes are kept for k_wire_fill)
};

struct DWire {
    const uint8_t* p;
    const uint8_t* end;
    __device__ __forceinline__ bool done() const { return p >= end; }
    __device__ __forceinline__ bool varint(u64& v) {
        u64 r = 0;
        for (int shift = 0; shift < 64 && p < end; shift += 7) {
            const uint8_t b = *p++;
            r |= (u64)(b & 0x7F) << shift;
            if (!(b & 0x80)) {
                v = r;
                return true;
            }
        }
        return false;
    }
    __device__ __forceinline__ bool bytes(DWire& sub) {  // length-delimited payload
        u64 n;
        if (!varint(n) || n > (u64)(end - p)) return false;
        sub.p = p;
        sub.end = p + n;
        p += n;
        return true;
    }
    __device__ __forceinline__ bool skip(u32 wire_type) {
        u64 v;
        DWire w;
        switch (wire_type) {
            case 0: return varint(v);
            case 1:
                if (end - p < 8) return false;
                p += 8;
                return true;
            case 2: return bytes(w);
            case 5:
                if (end - p < 4) return false;
                p += 4;
                return true;
            default: return false;  // groups are not used by these messages
        }
    }
};

// str::from_utf8 (what prost demands of a `string` field): no overlong forms, no surrogates, nothing above U+10FFFF.
__device__ __forceinline__ bool wire_utf8_ok(const uint8_t* p, const uint8_t* end) {
    while (p < end) {
        const uint8_t b = *p++;
        if (b < 0x80) continue;
        u32 need;
        uint8_t lo = 0x80, hi = 0xBF;
        if (b >= 0xC2 && b <= 0xDF) need = 1;
        else if (b >= 0xE0 && b <= 0xEF) {
            need = 2;
            if (b == 0xE0) lo = 0xA0;
            if (b == 0xED) hi = 0x9F;
        } else if (b >= 0xF0 && b <= 0xF4) {
            need = 3;
            if (b == 0xF0) lo = 0x90;
            if (b == 0xF4) hi = 0x8F;
        } else return false;
        if ((u32)(end - p) < need) return false;
        if (*p < lo || *p > hi) return false;
        ++p;
        for (u32 q = 1; q < need; ++q, ++p)
            if ((*p & 0xC0) != 0x80) return false;
    }
    return true;
}

struct WireLds {
    uint8_t blob[WIRE_BLOB_MAX];
    WireLit lit[WIRE_LIT_TAB];
    WireStr ns[MATCH_LDS_NS];
};

__device__ __forceinline__ bool wire_bytes_eq(const uint8_t* g, u32 len, const uint8_t* s) {
    for (u32 i = 0; i < len; ++i)
        if (g[i] != s[i]) return false;
    return true;
}

constexpr u32 WIRE_MSG_LDS = 20u * 1024u;  // bytes of messages a workgroup stages (256 messages of up to 80 bytes on average)

__global__ __launch_bounds__(256) void k_wire_count(const uint8_t* wire, const u32* __restrict__ msg_off, u32 n_req,
                                                    WireTables W, MatchTables T, u32* __restrict__ req_ns,
                                                    u32* __restrict__ req_delta, int32_t* __restrict__ status,
                                                    unsigned long long* __restrict__ mask, uint4* __restrict__ slot_h,
                                                    MatchScan* ms) {
    __shared__ MatchLdsTables S;
    __shared__ WireLds L;
    // The messages of the workgroup's 256 requests are one contiguous range of the batch's bytes.  A lane walks ITS message
    // a byte at a time — a chain of dependent byte loads, each a round trip to the L2 when it reads global memory (87 us for
    // 262 144 messages of ~60 bytes) — so the range is brought into LDS first, 16 bytes per lane and coalesced, when it fits.
    __shared__ __attribute__((aligned(16))) uint8_t s_msg[WIRE_MSG_LDS];
    const u32 tid = threadIdx.x;
    match_stage_tables(S, T);
    for (u32 q = tid; q < W.blob_len; q += 256) L.blob[q] = W.blob[q];
    for (u32 q = tid; q < WIRE_LIT_TAB; q += 256) L.lit[q] = W.lit[q];
    for (u32 q = tid; q < W.n_ns; q += 256) L.ns[q] = W.ns[q];
    const u32 r0 = blockIdx.x * 256, r1 = r0 + 256 < n_req ? r0 + 256 : n_req;
    const u32 g0 = msg_off[r0], g1 = msg_off[r1];
    const u32 a0 = g0 & ~15u;  // (the staging of the batch's bytes is 16-byte aligned and padded: whole 16-byte words)
    const bool staged = g1 - a0 <= WIRE_MSG_LDS;
    if (staged)
        for (u32 q = a0 + tid * 16u; q < g1; q += 256u * 16u)
            *reinterpret_cast<uint4*>(s_msg + (q - a0)) = *reinterpret_cast<const uint4*>(wire + q);
    __syncthreads();
    // Every offset below is relative to byte a0 of the batch, which is `wire` from here on: the start of the LDS copy, or
    // the byte itself in global memory.  (NOT `s_msg - a0`: arithmetic on an LDS pointer is 32-bit, a "negative" intermediate
    // wraps inside the LDS aperture and the flat address it turns into is nowhere — a memory fault that aborts the host.)
    wire = staged ? static_cast<const uint8_t*>(s_msg) : wire + a0;
    const u32 r = blockIdx.x * 256 + tid;
    u32 err = 0, k = 0;
    if (r < n_req) {
        const u32 m0 = msg_off[r] - a0, m1 = msg_off[r + 1] - a0;
        // ---- RateLimitRequest { domain = 1; repeated RateLimitDescriptor descriptors = 2; uint32 hits_addend = 3 } ----
        DWire w{wire + m0, wire + m1};
        u32 dom_off = 0, dom_len = 0, n_desc = 0;
        u64 hits_addend = 0;
#pragma unroll
        for (u32 sl = 0; sl < MATCH_SLOTS; ++sl) S.v[tid][sl] = MATCH_NO_VALUE;
        bool ok = true;
        while (ok && !w.done()) {
            u64 tag;
            if (!w.varint(tag)) {
                ok = false;
                break;
            }
            const u32 field = (u32)(tag >> 3), wt = (u32)(tag & 7);
            DWire sub;
            // (a KNOWN field with another wire type than its declared one is a decode error for prost, and so are strings
            // that are not UTF-8: such a message derives nothing — csrc/host/ingest.cpp's reader has the same rules)
            if (field == 1) {
                if (wt != 2 || !w.bytes(sub) || !wire_utf8_ok(sub.p, sub.end)) ok = false;
                else {
                    dom_off = (u32)(sub.p - wire);
                    dom_len = (u32)(sub.end - sub.p);
                }
            } else if (field == 2) {
                if (wt != 2 || !w.bytes(sub)) ok = false;
                else {
                    // ---- RateLimitDescriptor { repeated Entry entries = 1; ... }: descriptor number d_idx of the context (the
                    //      reference binds the whole list, one map per descriptor: envoy_rls/server.rs:121-137) ----
                    const u32 d_idx = n_desc++;
                    if (d_idx > 63u || !((W.desc_mask >> d_idx) & 1ull)) sub.p = sub.end;  // (no limit reads this descriptor)
                    while (ok && !sub.done()) {
                        u64 t2;
                        if (!sub.varint(t2)) {
                            ok = false;
                            break;
                        }
                        const u32 f2 = (u32)(t2 >> 3), w2 = (u32)(t2 & 7);
                        DWire ent;
                        if (f2 == 1) {
                            if (w2 != 2 || !sub.bytes(ent)) {
                                ok = false;
                                break;
                            }
                            // ---- Entry { key = 1; value = 2 } (a field given twice keeps its last occurrence) ----
                            u32 ko = 0, kl = 0, vo = 0, vl = 0;
                            while (ok && !ent.done()) {
                                u64 t3;
                                if (!ent.varint(t3)) {
                                    ok = false;
                                    break;
                                }
                                const u32 f3 = (u32)(t3 >> 3), w3 = (u32)(t3 & 7);
                                DWire str;
                                if (f3 == 1 || f3 == 2) {
                                    if (w3 != 2 || !ent.bytes(str) || !wire_utf8_ok(str.p, str.end)) ok = false;
                                    else if (f3 == 1) {
                                        ko = (u32)(str.p - wire);
                                        kl = (u32)(str.end - str.p);
                                    } else {
                                        vo = (u32)(str.p - wire);
                                        vl = (u32)(str.end - str.p);
                                    }
                                } else if (!ent.skip(w3)) {
                                    ok = false;
                                }
                            }
                            if (!ok) break;
                            // a key some limit reads -> its slot: the id of the condition literal the value equals (bytes
                            // compared), its hash for the counter's key; a repeated key keeps its LAST value (HashMap::insert)
                            u32 sl_hit = MATCH_SLOTS;
#pragma unroll
                            for (u32 sl = 0; sl < MATCH_SLOTS; ++sl)
                                if (sl < T.slots.n && W.slot_desc[sl] == d_idx && W.slot_key[sl].len == kl &&
                                    wire_bytes_eq(wire + ko, kl, L.blob + W.slot_key[sl].off))
                                    sl_hit = sl;
                            if (sl_hit < MATCH_SLOTS) {
                                const rl_h128 h = rl_kh_bytes(wire + vo, vl, W.hkey);
                                u32 v = WIRE_NON_LITERAL;
                                for (u32 q = (u32)h.h1 & (WIRE_LIT_TAB - 1u);; q = (q + 1u) & (WIRE_LIT_TAB - 1u)) {
                                    const WireLit e = L.lit[q];
                                    if (e.id == 0xFFFFu) break;
                                    if (e.h1 == h.h1 && e.len == vl && wire_bytes_eq(wire + vo, vl, L.blob + e.off)) {
                                        v = e.id;
                                        break;
                                    }
                                }
                                S.v[tid][sl_hit] = v;
                                if ((W.var_slot_mask >> sl_hit) & 1u)
                                    slot_h[(size_t)r * MATCH_SLOTS + sl_hit] = make_uint4((u32)h.h1, (u32)(h.h1 >> 32), (u32)h.h2, (u32)(h.h2 >> 32));
                            }
                        } else if (!sub.skip(w2)) {
                            ok = false;
                        }
                    }
                }
            } else if (field == 3) {
                if (wt != 0 || !w.varint(hits_addend)) ok = false;
            } else if (!w.skip(wt)) {
                ok = false;
            }
        }
        int32_t st = 0;
        if (!ok) st = WIRE_ST_MALFORMED;
        else if (dom_len == 0) st = WIRE_ST_UNKNOWN_DOMAIN;  // envoy_rls/server.rs:105-115
        u32 ns = 0;
        if (st == 0) {
            // a namespace no limit names has no counters (lib.rs:434-440): namespace 0
            for (u32 q = 1; q < W.n_ns; ++q)
                if (L.ns[q].len == dom_len && wire_bytes_eq(wire + dom_off, dom_len, L.blob + L.ns[q].off)) {
                    ns = q;
                    break;
                }
        } else {
#pragma unroll
            for (u32 sl = 0; sl < MATCH_SLOTS; ++sl) S.v[tid][sl] = MATCH_NO_VALUE;  // a refused message derives nothing
        }
        u32 delta = (u32)hits_addend;  // a uint32 on the wire; 0 means 1 (server.rs:131-137)
        if (delta == 0) delta = 1;
        req_ns[r] = ns;
        req_delta[r] = delta;
        status[r] = st;
        // ---- the limits of the namespace that apply (as k_match_count2) ----
        unsigned long long m = 0ull;
        if (ns >= T.n_ns) {
            err |= ERRBIT_BAD_LIMIT;
        } else {
            const u32 l0 = S.ns[ns], l1 = S.ns[ns + 1];
            for (u32 li = l0; li < l1; ++li) {
                const MatchLimitF Lm = S.l[li];
                const u32 nc = Lm.shape & 0xFFu, nv = (Lm.shape >> 8) & 0xFFu;
                bool app = true;
                for (u32 c = 0; c < nc; ++c) {
                    const MatchCondF cd = S.c[Lm.cond_off + c];
                    const u32 val = S.v[tid][cd.slot_op & 0xFFu];
                    // NoSuchKey -> false, whatever the operator (limit/cel.rs:321-338)
                    app = app && val != MATCH_NO_VALUE && ((val == cd.value) == ((cd.slot_op >> 8) == 0u));
                }
                for (u32 q = 0; q < nv; ++q)  // a variable the request does not carry: no counter (limit/cel.rs:176-191)
                    app = app && S.v[tid][(Lm.vslots >> (4u * q)) & 0xFu] != MATCH_NO_VALUE;
                if (app) {
                    m |= 1ull << (li - l0);
                    ++k;
                }
            }
        }
        mask[r] = m;
    }
    if (err) atomicOr(&ms->st.err, err);
    u32 total;
    (void)match_block_scan(k, S.w, total);
    if (tid == 0) ms->wg_tot[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void k_wire_fill(const u32* __restrict__ req_ns, const u32* __restrict__ req_delta, u32 n_req,
                                                   MatchTables T, const u64* __restrict__ prefix, rl_hkey hkey,
                                                   const unsigned long long* __restrict__ mask, const uint4* __restrict__ slot_h,
                                                   const MatchScan* ms, u32* __restrict__ req_off, Hit* __restrict__ hits,
                                                   u32* __restrict__ hit_check, u32* __restrict__ hit_req, u32 max_hits) {
    __shared__ MatchLdsTables S;
    const u32 tid = threadIdx.x;
    // enqueued before the host has seen the total (its round trip runs under this kernel)
    const bool fits = req_off[n_req] <= max_hits;
    match_stage_tables(S, T);
    __syncthreads();
    const u32 r = blockIdx.x * 256 + tid;
    const bool active = r < n_req;
    const unsigned long long m = active ? mask[r] : 0ull;
    u32 total;
    const u32 k = (u32)__popcll(m);
    const u32 out = ms->wg_tot[blockIdx.x] + match_block_scan(k, S.w, total);
    if (!active) return;
    req_off[r] = out;
    if (!m || !fits) return;
    const u32 l0 = S.ns[req_ns[r]];  // (a request of an unknown namespace has no mask bits)
    const u32 delta = req_delta[r];
    u32 kk = 0;
    for (int pass = 0; pass < 2; ++pass) {  // limits without variables first (in_memory.rs:105,121)
        unsigned long long mm = m;
        while (mm) {
            const u32 i = (u32)__builtin_ctzll(mm);
            mm &= mm - 1ull;
            const MatchLimitF Lm = S.l[l0 + i];
            const u32 nv = (Lm.shape >> 8) & 0xFFu;
            if ((nv != 0u) != (pass == 1)) continue;
            const u32 lid = Lm.limit & ~SIMPLE_FLAG;
            rl_h128 P, vals[MATCH_MAX_VARS_F];
            P.h1 = prefix[2 * lid];
            P.h2 = prefix[2 * lid + 1];
#pragma unroll
            for (u32 q = 0; q < MATCH_MAX_VARS_F; ++q) {
                if (q < nv) {
                    const uint4 h = slot_h[(size_t)r * MATCH_SLOTS + ((Lm.vslots >> (4u * q)) & 0xFu)];
                    vals[q].h1 = ((u64)h.y << 32) | h.x;
                    vals[q].h2 = ((u64)h.w << 32) | h.z;
                }
            }
            uint64_t key;
            uint32_t chk;
            rl_counter_key(P, vals, nv, hkey, &key, &chk);
            Hit h;
            h.key = key;
            h.limit = Lm.limit;
            h.delta = delta;
            hits[out + kk] = h;
            hit_check[out + kk] = chk;
            if (hit_req) hit_req[out + kk] = r;
            ++kk;
        }
    }
}

}  // namespace rl
