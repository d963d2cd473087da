// rl_match.hpp — limit matching and counter-key derivation on the device: the step UPSTREAM of
// CounterStorage::check_and_update in the reference,
//     RateLimiter::counters_that_apply   limitador/src/lib.rs:507-522
//     Limit::applies                      limitador/src/limit.rs:157-174
//     Limit::resolve_variables            limitador/src/limit.rs:133-148
//     Counter::new                        limitador/src/counter.rs:19-31
// for the predicate shapes rate-limit configurations are made of
//     descriptors[0]['key'] == 'value'     descriptors[0]['key'] != 'value'      (conditions)
//     descriptors[0]['key']                                                     (variables)
// (limitador-server/sandbox/limits.yaml, examples/limits.yaml, doc/how-it-works.md).  Anything
// else is CEL and stays on the host: the caller simply does not put such a limit in the match table.
//
// Strings never reach the device: the ingest side dictionary-encodes descriptor keys and values to
// dense ids (exact — two different strings never share an id), so a request is
//     namespace id, [(key id, value id)] entries of descriptors[0], delta
// and a compiled limit is a list of (key id, ==|!=, value id) conditions plus the key ids of its
// variables.  Semantics restated from the reference:
//   * a condition on a key the request does not carry is FALSE, for == and for != alike
//     (limit/cel.rs:321-338: NoSuchKey -> Ok(false));
//   * a limit whose variable is not in the request does not produce a counter
//     (limit/cel.rs:176-191: NoSuchKey -> Ok(None); counter.rs:22-24);
//   * every limit of the namespace that applies yields one counter (lib.rs:512-521) — here in
//     match-table order, simple counters (no variables) first, as the storage walks them
//     (in_memory.rs:105,121);
//   * the counter's identity is (limit, resolved variable values) (counter.rs:123-138): the key is
//     the PACKED ids — limit id + 1 in bits 52.., value ids in bits 0..25 and 26..51 — injective, so
//     exact; dictionaries beyond 2^26 values or limits with more than two variables keep the host
//     path.
//
//   k_match_count   requests x limits of their namespace -> counters per request
//   (exclusive scan of the counts: rocPRIM)
//   k_match_fill    the same evaluation again, writing the rl_hit records at the scanned offsets
#pragma once
#include "rl_kernels.hpp"

namespace rl {

constexpr u32 MATCH_MAX_VARS = 2;
constexpr u32 MATCH_VAL_BITS = 26;
constexpr u32 MATCH_NO_VALUE = 0xFFFFFFFFu;

struct MatchCond {  // == rl_match_cond
    u32 key;
    u32 op;  // 0: ==   1: !=
    u32 value;
};
struct MatchLimit {  // == rl_match_limit
    u32 limit;     // limit id | SIMPLE_FLAG (set iff n_vars == 0)
    u32 ns;        // namespace id; the table is sorted by ns
    u32 cond_off;  // first condition in the condition array
    u32 n_cond;
    u32 n_vars;
    u32 var_key[MATCH_MAX_VARS];  // descriptor keys of the variables, in variable-name order
    u32 pad;
};
static_assert(sizeof(MatchLimit) == 32, "rl_match_limit is 32 bytes");
static_assert(sizeof(MatchCond) == 12, "rl_match_cond is 12 bytes");

__host__ __device__ inline u64 match_key(u32 limit_id, u32 n_vars, u32 v0, u32 v1) {
    return ((u64)(limit_id + 1u) << (2 * MATCH_VAL_BITS)) | ((u64)(n_vars > 1 ? v1 : 0u) << MATCH_VAL_BITS) |
           (u64)(n_vars > 0 ? v0 : 0u);
}

constexpr u32 MATCH_REG_ENTRIES = 8;   // descriptor entries of a request kept in registers
constexpr u32 MATCH_LDS_LIMITS = 256;  // the match table is staged in LDS when it is this small ...
constexpr u32 MATCH_LDS_CONDS = 512;   // ... (a few KB: every request walks all limits of its namespace)
constexpr u32 MATCH_LDS_NS = 64;

// The request's descriptor entries: the first MATCH_REG_ENTRIES in registers, the rest re-read.
struct ReqEntries {
    u32 k[MATCH_REG_ENTRIES], v[MATCH_REG_ENTRIES];
    u32 n, b;
    const u32* ent_key;
    const u32* ent_val;
    // value id of descriptor key `key`, MATCH_NO_VALUE if absent.  The first entry wins, like the
    // first insertion into the reference's context map.
    __device__ __forceinline__ u32 value_of(u32 key) const {
#pragma unroll
        for (u32 q = 0; q < MATCH_REG_ENTRIES; ++q)
            if (q < n && k[q] == key) return v[q];
        for (u32 q = MATCH_REG_ENTRIES; q < n; ++q)
            if (ent_key[b + q] == key) return ent_val[b + q];
        return MATCH_NO_VALUE;
    }
};

// Does limit L apply to the request, and with which variable values?  (limit.rs:157-174, 133-148)
__device__ __forceinline__ bool limit_applies(const MatchLimit& L, const MatchCond* conds, const ReqEntries& R,
                                              u32 (&vars)[MATCH_MAX_VARS]) {
    for (u32 c = 0; c < L.n_cond; ++c) {
        const MatchCond cd = conds[L.cond_off + c];
        const u32 v = R.value_of(cd.key);
        if (v == MATCH_NO_VALUE) return false;  // NoSuchKey -> false, whatever the operator
        if ((v == cd.value) != (cd.op == 0u)) return false;
    }
    vars[0] = vars[1] = 0;
    for (u32 q = 0; q < L.n_vars; ++q) {
        const u32 v = R.value_of(L.var_key[q]);
        if (v == MATCH_NO_VALUE) return false;
        vars[q] = v;
    }
    return true;
}

// One thread per request.  FILL == false: count[r] = counters of request r.  FILL == true: write
// them at hit_off[r].., simple counters first (two passes over the namespace's limits).
template <bool FILL, bool IN_LDS>
__global__ __launch_bounds__(256) void k_match(const u32* __restrict__ req_ns, const u32* __restrict__ ent_off,
                                               const u32* __restrict__ ent_key, const u32* __restrict__ ent_val,
                                               const u32* __restrict__ req_delta, u32 n_req,
                                               const MatchLimit* __restrict__ limits, u32 n_limits,
                                               const u32* __restrict__ ns_off, u32 n_ns,
                                               const MatchCond* __restrict__ conds, u32 n_conds,
                                               u32* __restrict__ count, const u32* __restrict__ hit_off,
                                               Hit* __restrict__ hits, Status* st) {
    __shared__ MatchLimit s_limits[MATCH_LDS_LIMITS];
    __shared__ MatchCond s_conds[MATCH_LDS_CONDS];
    __shared__ u32 s_ns_off[MATCH_LDS_NS + 1];
    // IN_LDS (chosen by the host when the table is small): no pointer that could be either LDS or global
    // survives, so the loads stay ds_read / global_load instead of flat_load
    if (IN_LDS) {
        for (u32 q = threadIdx.x; q < n_limits; q += 256) s_limits[q] = limits[q];
        for (u32 q = threadIdx.x; q < n_conds; q += 256) s_conds[q] = conds[q];
        for (u32 q = threadIdx.x; q <= n_ns; q += 256) s_ns_off[q] = ns_off[q];
        __syncthreads();
    }
    const MatchLimit* T = IN_LDS ? (const MatchLimit*)s_limits : limits;
    const MatchCond* Cd = IN_LDS ? (const MatchCond*)s_conds : conds;
    const u32* NO = IN_LDS ? (const u32*)s_ns_off : ns_off;
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_req) return;
    const u32 ns = req_ns[r];
    ReqEntries R;
    R.b = ent_off[r];
    R.n = ent_off[r + 1] - R.b;
    R.ent_key = ent_key;
    R.ent_val = ent_val;
#pragma unroll
    for (u32 q = 0; q < MATCH_REG_ENTRIES; ++q) {
        R.k[q] = q < R.n ? ent_key[R.b + q] : MATCH_NO_VALUE;
        R.v[q] = q < R.n ? ent_val[R.b + q] : 0u;
    }
    u32 k = 0;
    if (ns < n_ns) {
        const u32 l0 = NO[ns], l1 = NO[ns + 1];
        const u32 delta = FILL ? req_delta[r] : 0u;
        const u32 out = FILL ? hit_off[r] : 0u;
        for (int pass = 0; pass < 2; ++pass) {  // pass 0: limits without variables, pass 1: with
            for (u32 li = l0; li < l1; ++li) {
                const MatchLimit L = T[li];
                if ((L.n_vars != 0u) != (pass == 1)) continue;
                u32 vars[MATCH_MAX_VARS];
                if (!limit_applies(L, Cd, R, vars)) continue;
                if (vars[0] >> MATCH_VAL_BITS || vars[1] >> MATCH_VAL_BITS) atomicOr(&st->err, ERRBIT_RESERVED_KEY);
                if (FILL) {
                    Hit h;
                    h.key = match_key(L.limit & ~SIMPLE_FLAG, L.n_vars, vars[0], vars[1]);
                    h.limit = L.limit;
                    h.delta = delta;
                    hits[out + k] = h;
                }
                ++k;
            }
        }
    } else {
        atomicOr(&st->err, ERRBIT_BAD_LIMIT);  // unknown namespace id
    }
    if (!FILL) count[r] = k;
}

// ---------------------------------------------------------------------------------------------
// The same evaluation for the tables limit files produce: at most MATCH_SLOTS distinct descriptor keys in all
// conditions and variables, at most 64 limits per namespace, the table in LDS.  The host renames the keys to
// SLOTS; a request's entries are reduced ONCE to "value of slot s" (static register indexing, no scratch), the
// conditions read slots, and the count pass leaves a bit mask of the limits that apply, so the fill pass
// evaluates nothing twice.  (The generic k_match above: 105 + 128 us per 1 M requests x 8 limits, its entry
// arrays in scratch memory.)
// ---------------------------------------------------------------------------------------------
constexpr u32 MATCH_SLOTS = 8;
struct MatchSlots {
    u32 key[MATCH_SLOTS];
    u32 n;
};
constexpr u32 MATCH_MAX_VARS_F = 8;  // variables of a limit in the slot form (the packed exact key still takes two: match_key)
struct MatchLimitF {
    u32 limit;     // limit id | SIMPLE_FLAG
    u32 cond_off;  // first condition in the MatchCondF array
    u32 shape;     // n_cond | n_vars << 8 | slot of variable 0 << 16 | slot of variable 1 << 24
    u32 vslots;    // slot of variable q in bits 4q .. 4q + 3 (all of them; the two in `shape` are what the packed-key kernels read)
};
struct MatchCondF {
    u32 slot_op;  // slot | op << 8
    u32 value;
};

template <bool FILL>
__global__ __launch_bounds__(256) void k_match_fast(const u32* __restrict__ req_ns, const u32* __restrict__ ent_off,
                                                    const u32* __restrict__ ent_key, const u32* __restrict__ ent_val,
                                                    const u32* __restrict__ req_delta, u32 n_req,
                                                    const MatchLimitF* __restrict__ limits, u32 n_limits,
                                                    const u32* __restrict__ ns_off, u32 n_ns,
                                                    const MatchCondF* __restrict__ conds, u32 n_conds, MatchSlots slots,
                                                    u32* __restrict__ count, unsigned long long* __restrict__ mask,
                                                    const u32* __restrict__ hit_off, Hit* __restrict__ hits, Status* st,
                                                    u32* __restrict__ hit_req, u32 max_hits) {
    // The fill pass is enqueued before the host has seen the total of the count pass (its round trip runs under this
    // kernel): a batch that expands to more counters than the staging buffers hold writes nothing — the host refuses it.
    if (FILL && hit_off[n_req] > max_hits) return;
    __shared__ MatchLimitF s_l[MATCH_LDS_LIMITS];
    __shared__ MatchCondF s_c[MATCH_LDS_CONDS];
    __shared__ u32 s_ns[MATCH_LDS_NS + 1];
    __shared__ u32 s_v[256][MATCH_SLOTS + 1];  // (+1: rows on different banks)
    const u32 tid = threadIdx.x;
    for (u32 q = tid; q < n_limits; q += 256) s_l[q] = limits[q];
    for (u32 q = tid; q < n_conds; q += 256) s_c[q] = conds[q];
    for (u32 q = tid; q <= n_ns; q += 256) s_ns[q] = ns_off[q];
    __syncthreads();
    const u32 r = blockIdx.x * 256 + tid;
    if (r >= n_req) return;
    const u32 ns = req_ns[r];
    const u32 b = ent_off[r], n = ent_off[r + 1] - b;
    // value of every slot: the FIRST entry with the slot's key wins, like the first insertion into the
    // reference's context map
    u32 v[MATCH_SLOTS];
#pragma unroll
    for (u32 sl = 0; sl < MATCH_SLOTS; ++sl) v[sl] = MATCH_NO_VALUE;
#pragma unroll
    for (u32 q = 0; q < MATCH_REG_ENTRIES; ++q) {
        if (q < n) {
            const u32 k = ent_key[b + q], val = ent_val[b + q];
#pragma unroll
            for (u32 sl = 0; sl < MATCH_SLOTS; ++sl)
                if (sl < slots.n && k == slots.key[sl] && v[sl] == MATCH_NO_VALUE) v[sl] = val;
        }
    }
    for (u32 q = MATCH_REG_ENTRIES; q < n; ++q) {
        const u32 k = ent_key[b + q], val = ent_val[b + q];
#pragma unroll
        for (u32 sl = 0; sl < MATCH_SLOTS; ++sl)
            if (sl < slots.n && k == slots.key[sl] && v[sl] == MATCH_NO_VALUE) v[sl] = val;
    }
#pragma unroll
    for (u32 sl = 0; sl < MATCH_SLOTS; ++sl) s_v[tid][sl] = v[sl];
    if (ns >= n_ns) {
        atomicOr(&st->err, ERRBIT_BAD_LIMIT);  // unknown namespace id
        if (!FILL) {
            count[r] = 0;
            mask[r] = 0ull;
        }
        return;
    }
    const u32 l0 = s_ns[ns], l1 = s_ns[ns + 1];
    if (!FILL) {
        unsigned long long m = 0ull;
        u32 k = 0;
        for (u32 li = l0; li < l1; ++li) {
            const MatchLimitF L = s_l[li];
            const u32 nc = L.shape & 0xFFu, nv = (L.shape >> 8) & 0xFFu;
            bool ok = true;
            for (u32 c = 0; c < nc; ++c) {
                const MatchCondF cd = s_c[L.cond_off + c];
                const u32 val = s_v[tid][cd.slot_op & 0xFFu];
                // NoSuchKey -> false, whatever the operator (limit/cel.rs:321-338)
                ok = ok && val != MATCH_NO_VALUE && ((val == cd.value) == ((cd.slot_op >> 8) == 0u));
            }
            if (nv > 0) ok = ok && s_v[tid][(L.shape >> 16) & 0xFFu] != MATCH_NO_VALUE;  // limit/cel.rs:176-191
            if (nv > 1) ok = ok && s_v[tid][(L.shape >> 24) & 0xFFu] != MATCH_NO_VALUE;
            if (ok) {
                // a value id that does not fit the packed key keeps the host path: said here, so that the fill pass
                // has nothing left to refuse
                if ((nv > 0 && s_v[tid][(L.shape >> 16) & 0xFFu] >> MATCH_VAL_BITS) ||
                    (nv > 1 && s_v[tid][(L.shape >> 24) & 0xFFu] >> MATCH_VAL_BITS))
                    atomicOr(&st->err, ERRBIT_RESERVED_KEY);
                m |= 1ull << (li - l0);
                ++k;
            }
        }
        count[r] = k;
        mask[r] = m;
    } else {
        const unsigned long long m = mask[r];
        const u32 delta = req_delta[r];
        const u32 out = hit_off[r];
        u32 k = 0;
        for (int pass = 0; pass < 2; ++pass) {  // limits without variables first (in_memory.rs:105,121)
            unsigned long long mm = m;
            while (mm) {
                const u32 i = (u32)__builtin_ctzll(mm);
                mm &= mm - 1ull;
                const MatchLimitF L = s_l[l0 + i];
                const u32 nv = (L.shape >> 8) & 0xFFu;
                if ((nv != 0u) != (pass == 1)) continue;
                const u32 v0 = nv > 0 ? s_v[tid][(L.shape >> 16) & 0xFFu] : 0u;
                const u32 v1 = nv > 1 ? s_v[tid][(L.shape >> 24) & 0xFFu] : 0u;
                if (v0 >> MATCH_VAL_BITS || v1 >> MATCH_VAL_BITS) atomicOr(&st->err, ERRBIT_RESERVED_KEY);
                Hit h;
                h.key = match_key(L.limit & ~SIMPLE_FLAG, nv, v0, v1);
                h.limit = L.limit;
                h.delta = delta;
                hits[out + k] = h;
                if (hit_req) hit_req[out + k] = r;  // (what k_gen_hit_req would derive from the offsets again)
                ++k;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The slot form without the library scan, the copy commands and the marker (count pass + rocPRIM scan + two copies + an
// event + fill pass were 125 us of a 0.54 ms call):
//   k_match_count2   evaluate; per request the bit mask of the limits that apply; per WORKGROUP the number of counters
//   k_match_scan2    one workgroup: exclusive scan of the workgroups' totals; {total, error bits, call number} handed to
//                    the host as ONE 16-byte store into host-mapped memory — the host polls that word while ...
//   k_match_fill2    ... the records are written: a workgroup's base + a scan of its own requests' mask popcounts gives
//                    every request its offset (the CSR offsets the resolver wants are written here too), in the order
//                    the storage walks a request's counters (limits without variables first, in_memory.rs:105,121).
// A batch that expands to more counters than the staging buffers hold writes nothing; the host sees the total and
// refuses it.
// (Tried first, measured, withdrawn: all of it in ONE kernel with ticketed workgroups and a decoupled look-back over
// their totals — 128 us against 34 + 4 + 38: four thousand workgroups polling each other's state words across the
// XCDs, whose L2s only meet in memory.)
// ---------------------------------------------------------------------------------------------
struct MatchScan {
    Status st;       // the matcher's own error word (zero between calls: k_match_scan2 clears it)
    u32 wg_tot[1];   // [workgroups]: counters of the workgroup's requests, then (in place) the counters before it
};

struct MatchTables {
    const MatchLimitF* limits;
    u32 n_limits;
    const u32* ns_off;
    u32 n_ns;
    const MatchCondF* conds;
    u32 n_conds;
    MatchSlots slots;
};

struct MatchLdsTables {
    MatchLimitF l[MATCH_LDS_LIMITS];
    MatchCondF c[MATCH_LDS_CONDS];
    u32 ns[MATCH_LDS_NS + 1];
    u32 v[256][MATCH_SLOTS + 1];  // (+1: rows on different banks)
    u32 w[4];
};

__device__ __forceinline__ void match_stage_tables(MatchLdsTables& S, const MatchTables& T) {
    const u32 tid = threadIdx.x;
    for (u32 q = tid; q < T.n_limits; q += 256) S.l[q] = T.limits[q];
    for (u32 q = tid; q < T.n_conds; q += 256) S.c[q] = T.conds[q];
    for (u32 q = tid; q <= T.n_ns; q += 256) S.ns[q] = T.ns_off[q];
}

// value of every slot of request r -> S.v[tid]: the FIRST entry with the slot's key wins, like the first insertion into
// the reference's context map
__device__ __forceinline__ void match_slot_values(MatchLdsTables& S, const MatchSlots& slots, const u32* __restrict__ ent_off,
                                                  const u32* __restrict__ ent_key, const u32* __restrict__ ent_val, u32 r) {
    const u32 tid = threadIdx.x;
    const u32 b = ent_off[r], n = ent_off[r + 1] - b;
    u32 v[MATCH_SLOTS];
#pragma unroll
    for (u32 sl = 0; sl < MATCH_SLOTS; ++sl) v[sl] = MATCH_NO_VALUE;
    // The entries' loads are the kernel's one chain (offsets -> entries -> LDS walk): four entries at a time come as ONE
    // 16-byte load per array when the request's first entry is 16-byte aligned (a batch of 4-entry requests always is) instead
    // of eight scalar loads issued one behind the other.
    u32 ek[MATCH_REG_ENTRIES], ev[MATCH_REG_ENTRIES];
    if ((((size_t)(ent_key + b) | (size_t)(ent_val + b)) & 15u) == 0u && n >= 4u) {
        const uint4 k4 = *reinterpret_cast<const uint4*>(ent_key + b), v4 = *reinterpret_cast<const uint4*>(ent_val + b);
        ek[0] = k4.x, ek[1] = k4.y, ek[2] = k4.z, ek[3] = k4.w;
        ev[0] = v4.x, ev[1] = v4.y, ev[2] = v4.z, ev[3] = v4.w;
        if (n >= 8u) {
            const uint4 k8 = *reinterpret_cast<const uint4*>(ent_key + b + 4), v8 = *reinterpret_cast<const uint4*>(ent_val + b + 4);
            ek[4] = k8.x, ek[5] = k8.y, ek[6] = k8.z, ek[7] = k8.w;
            ev[4] = v8.x, ev[5] = v8.y, ev[6] = v8.z, ev[7] = v8.w;
        } else {
#pragma unroll
            for (u32 q = 4; q < MATCH_REG_ENTRIES; ++q) {
                ek[q] = q < n ? ent_key[b + q] : 0u;
                ev[q] = q < n ? ent_val[b + q] : 0u;
            }
        }
    } else {
#pragma unroll
        for (u32 q = 0; q < MATCH_REG_ENTRIES; ++q) {
            ek[q] = q < n ? ent_key[b + q] : 0u;
            ev[q] = q < n ? ent_val[b + q] : 0u;
        }
    }
#pragma unroll
    for (u32 q = 0; q < MATCH_REG_ENTRIES; ++q) {
        if (q < n) {
#pragma unroll
            for (u32 sl = 0; sl < MATCH_SLOTS; ++sl)
                if (sl < slots.n && ek[q] == slots.key[sl] && v[sl] == MATCH_NO_VALUE) v[sl] = ev[q];
        }
    }
    for (u32 q = MATCH_REG_ENTRIES; q < n; ++q) {
        const u32 ek = ent_key[b + q], val = ent_val[b + q];
#pragma unroll
        for (u32 sl = 0; sl < MATCH_SLOTS; ++sl)
            if (sl < slots.n && ek == slots.key[sl] && v[sl] == MATCH_NO_VALUE) v[sl] = val;
    }
#pragma unroll
    for (u32 sl = 0; sl < MATCH_SLOTS; ++sl) S.v[tid][sl] = v[sl];
}

// exclusive offset of this thread's k among the workgroup's 256, and the workgroup's total (all threads; one barrier)
__device__ __forceinline__ u32 match_block_scan(u32 k, u32* s_w, u32& total) {
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    u32 inc = k;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 o = __shfl_up(inc, off);
        if ((int)lane >= off) inc += o;
    }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    u32 woff = 0, tot = 0;
#pragma unroll
    for (u32 ww = 0; ww < 4; ++ww) {
        const u32 x = s_w[ww];
        if (ww < w) woff += x;
        tot += x;
    }
    total = tot;
    return woff + inc - k;
}

__global__ __launch_bounds__(256) void k_match_count2(const u32* __restrict__ req_ns, const u32* __restrict__ ent_off,
                                                      const u32* __restrict__ ent_key, const u32* __restrict__ ent_val,
                                                      u32 n_req, MatchTables T, unsigned long long* __restrict__ mask,
                                                      MatchScan* ms) {
    __shared__ MatchLdsTables S;
    const u32 tid = threadIdx.x;
    match_stage_tables(S, T);
    __syncthreads();
    const u32 r = blockIdx.x * 256 + tid;
    u32 err = 0, k = 0;
    unsigned long long m = 0ull;
    if (r < n_req) {
        const u32 ns = req_ns[r];
        match_slot_values(S, T.slots, ent_off, ent_key, ent_val, r);
        if (ns >= T.n_ns) {
            err |= ERRBIT_BAD_LIMIT;  // unknown namespace id
        } else {
            const u32 l0 = S.ns[ns], l1 = S.ns[ns + 1];
            for (u32 li = l0; li < l1; ++li) {
                const MatchLimitF L = S.l[li];
                const u32 nc = L.shape & 0xFFu, nv = (L.shape >> 8) & 0xFFu;
                bool ok = true;
                for (u32 c = 0; c < nc; ++c) {
                    const MatchCondF cd = S.c[L.cond_off + c];
                    const u32 val = S.v[tid][cd.slot_op & 0xFFu];
                    // NoSuchKey -> false, whatever the operator (limit/cel.rs:321-338)
                    ok = ok && val != MATCH_NO_VALUE && ((val == cd.value) == ((cd.slot_op >> 8) == 0u));
                }
                if (nv > 0) ok = ok && S.v[tid][(L.shape >> 16) & 0xFFu] != MATCH_NO_VALUE;  // limit/cel.rs:176-191
                if (nv > 1) ok = ok && S.v[tid][(L.shape >> 24) & 0xFFu] != MATCH_NO_VALUE;
                if (ok) {
                    // a value id that does not fit the packed key keeps the host path: said here, the fill pass has
                    // nothing left to refuse
                    if ((nv > 0 && S.v[tid][(L.shape >> 16) & 0xFFu] >> MATCH_VAL_BITS) ||
                        (nv > 1 && S.v[tid][(L.shape >> 24) & 0xFFu] >> MATCH_VAL_BITS))
                        err |= ERRBIT_RESERVED_KEY;
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

__global__ __launch_bounds__(1024) void k_match_scan2(MatchScan* ms, u32 g, u32* __restrict__ req_off_end, u32* host_word,
                                                      u32 call) {
    __shared__ u32 s_w[16];
    __shared__ u32 s_carry;
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (u32 base = 0; base < g; base += 1024) {  // (block-uniform)
        const u32 i = base + tid;
        const u32 x = i < g ? ms->wg_tot[i] : 0u;
        u32 inc = x;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 o = __shfl_up(inc, off);
            if ((int)lane >= off) inc += o;
        }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        u32 woff = 0, tot = 0;
#pragma unroll
        for (u32 ww = 0; ww < 16; ++ww) {
            const u32 y = s_w[ww];
            if (ww < w) woff += y;
            tot += y;
        }
        const u32 carry = s_carry;
        if (i < g) ms->wg_tot[i] = carry + woff + inc - x;
        __syncthreads();
        if (tid == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (tid == 0) {
        const u32 total = s_carry, err = ms->st.err;
        *req_off_end = total;
        ms->st.err = 0;
        typedef u32 u32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(u32x4{total, err, 0u, call}, reinterpret_cast<u32x4*>(host_word));
    }
}

__global__ __launch_bounds__(256) void k_match_fill2(const u32* __restrict__ req_ns, const u32* __restrict__ ent_off,
                                                     const u32* __restrict__ ent_key, const u32* __restrict__ ent_val,
                                                     const u32* __restrict__ req_delta, u32 n_req, MatchTables T,
                                                     const unsigned long long* __restrict__ mask, const MatchScan* ms,
                                                     u32* __restrict__ req_off, Hit* __restrict__ hits,
                                                     u32* __restrict__ hit_req, u32 max_hits) {
    __shared__ MatchLdsTables S;
    const u32 tid = threadIdx.x;
    // enqueued before the host has seen the total (its round trip runs under this kernel)
    const bool fits = req_off[n_req] <= max_hits;
    match_stage_tables(S, T);
    __syncthreads();
    const u32 r = blockIdx.x * 256 + tid;
    const bool active = r < n_req;
    const unsigned long long m = active ? mask[r] : 0ull;
    if (active && m) match_slot_values(S, T.slots, ent_off, ent_key, ent_val, r);
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
            const MatchLimitF L = S.l[l0 + i];
            const u32 nv = (L.shape >> 8) & 0xFFu;
            if ((nv != 0u) != (pass == 1)) continue;
            const u32 v0 = nv > 0 ? S.v[tid][(L.shape >> 16) & 0xFFu] : 0u;
            const u32 v1 = nv > 1 ? S.v[tid][(L.shape >> 24) & 0xFFu] : 0u;
            Hit h;
            h.key = match_key(L.limit & ~SIMPLE_FLAG, nv, v0, v1);
            h.limit = L.limit;
            h.delta = delta;
            hits[out + kk] = h;
            if (hit_req) hit_req[out + kk] = r;
            ++kk;
        }
    }
}

// first_limited (index into hits) -> the limit id whose name the reference reports
// (Authorization::Limited(name), in_memory.rs:91-93,97-99), -1 when the request is not limited.
__global__ __launch_bounds__(256) void k_match_limited_limit(const int32_t* __restrict__ first_limited,
                                                             const Hit* __restrict__ hits, u32 n_req,
                                                             int32_t* __restrict__ out) {
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_req) return;
    const int32_t f = first_limited[r];
    out[r] = f < 0 ? -1 : (int32_t)(hits[f].limit & ~SIMPLE_FLAG);
}

// ---------------------------------------------------------------------------------------------
// Exclusive scan of the per-request counter counts for the GENERIC matcher (tables that do not take the slot form),
// without a library: k_xscan_sums (1024 counts per workgroup -> its total) -> k_xscan_tot (one workgroup: the totals
// in place -> what lies before each workgroup) -> k_xscan_apply (every workgroup scans its own 1024 counts on top of
// its base).  out[i] = sum of in[0 .. i), i < n.
// ---------------------------------------------------------------------------------------------
constexpr u32 XSCAN_PER_WG = 1024;

__global__ __launch_bounds__(256) void k_xscan_sums(const u32* __restrict__ in, u32 n, u32* __restrict__ wg_tot) {
    __shared__ u32 s_w[4];
    const u32 i0 = blockIdx.x * XSCAN_PER_WG + threadIdx.x * 4;
    u32 k = 0;
#pragma unroll
    for (u32 q = 0; q < 4; ++q)
        if (i0 + q < n) k += in[i0 + q];
    u32 total;
    (void)match_block_scan(k, s_w, total);
    if (threadIdx.x == 0) wg_tot[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void k_xscan_tot(u32* __restrict__ wg_tot, u32 g) {
    __shared__ u32 s_w[16];
    __shared__ u32 s_carry;
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (u32 base = 0; base < g; base += 1024) {  // (block-uniform)
        const u32 i = base + tid;
        const u32 x = i < g ? wg_tot[i] : 0u;
        u32 inc = x;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 o = __shfl_up(inc, off);
            if ((int)lane >= off) inc += o;
        }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        u32 woff = 0, tot = 0;
#pragma unroll
        for (u32 ww = 0; ww < 16; ++ww) {
            const u32 y = s_w[ww];
            if (ww < w) woff += y;
            tot += y;
        }
        const u32 carry = s_carry;
        if (i < g) wg_tot[i] = carry + woff + inc - x;
        __syncthreads();
        if (tid == 0) s_carry = carry + tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_xscan_apply(const u32* __restrict__ in, u32 n, const u32* __restrict__ wg_tot,
                                                     u32* __restrict__ out) {
    __shared__ u32 s_w[4];
    const u32 i0 = blockIdx.x * XSCAN_PER_WG + threadIdx.x * 4;
    u32 c[4], k = 0;
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
        c[q] = i0 + q < n ? in[i0 + q] : 0u;
        k += c[q];
    }
    u32 total;
    u32 ex = wg_tot[blockIdx.x] + match_block_scan(k, s_w, total);
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
        if (i0 + q < n) out[i0 + q] = ex;
        ex += c[q];
    }
}

}  // namespace rl
