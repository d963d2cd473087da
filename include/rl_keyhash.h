/*
 * rl_keyhash.h — counter keys derived from the CANONICAL KEY BYTES of a counter, the same function on the host and on
 * the device (row f1 of SURVEY.md §8: "hash the canonical key bytes on device").
 *
 * The reference names a counter by the byte string of key_for_counter_v2
 * (limitador/src/storage/keys.rs:220-248): version byte 1 + postcard of
 *     CounterKey { ns: &str, seconds: u64, conditions: Vec<String> (sorted), variables: Vec<(&str, &str)> (sorted by name) }
 * (postcard: LEB128 varints, a string = varint length + bytes, a Vec = varint length + elements).  The engine's table
 * wants a 64-bit key.  The key is a KEYED hash of exactly those bytes, computed hierarchically so that the part a LIMIT
 * fixes is hashed once, on the host, when the limit is compiled:
 *
 *     K       = a 128-bit secret of the ingest (rl_hkey; random per rli_create, or set by the host: rli_set_hash_key)
 *     prefix  = 0x01, str(ns), varint(seconds), varint(#conditions), str(condition_i)..., varint(#variables),
 *               str(variable_name_i)...                                  (the canonical bytes WITHOUT the values)
 *     P       = SipHash-2-4-128_K(prefix)                               (rl_kh_bytes)                    -- per limit
 *     V_i     = SipHash-2-4-128_K(bytes of value_i)                     (variables in name order)        -- per request
 *     S       = SipHash-2-4-128_K(P.h1, P.h2, V_1.h1, V_1.h2, ... as 8-byte little-endian words)         (rl_kh_words)
 *     key     = S.h1 (the two reserved tags 0xFF..FE / 0xFF..FF folded down by 2)
 *     check   = upper 32 bits of S.h2, never 0
 *
 * (prefix, values) <-> canonical bytes is a bijection (both parse uniquely), and S hashes a FIXED-WIDTH encoding of the
 * digests (their count is in SipHash's length byte), so two counters share (key, check) only if 128-bit digests collide
 * on the way or the 96 bits collide at the end.  The 64-bit key addresses the cell; the 32-bit check is stored beside it
 * (the cell's spare word) when the cell is created and compared on every later touch: a mismatch is REPORTED (the request
 * is answered RLI_HOST_ONLY and not applied), never merged.
 *
 * WHY KEYED.  Rounds 3-4 used MurmurHash3_x64_128 with seed 0: every step of it can be inverted, so a
 * caller who chooses descriptor values could BUILD a value whose counter shares alice's 64-bit key (and pick the check word
 * too: a silent merge) in microseconds, and a generic birthday search needs only ~2^32 hashes; a secret seed does not help
 * MurmurHash (its collisions are seed-independent).  SipHash-2-4 is a PRF: without K an attacker cannot compute, let alone
 * steer, any of P, V_i or S, so the odds below hold for ADVERSARIAL inputs too, as long as K stays secret.  Odds, for N live
 * counters: two of them share a key with probability ~N^2 / 2^65 (N = 10^7: 2.7e-6 over the table's life), in which case
 * the second one is refused; an UNDETECTED merge needs the check to collide as well: ~N^2 / 2^97 (N = 10^7: 6e-16).
 * What K costs the host: every ingest that feeds one table (several front-ends, a restart that reloads a snapshot) must be
 * given the same K — it names the cells — so a host that persists counters persists K with them.
 *
 * SipHash-2-4 (Aumasson, Bernstein 2012; 128-bit output as in the authors' reference: v1 ^= 0xee at the start, v2 ^= 0xee
 * before the first finalisation, v1 ^= 0xdd before the second) is restated here from its published description; pinned by
 * the paper's 64-bit test vector and the reference's first 128-bit vectors (tests/test_keyhash_cpu.py).  The device reads
 * strings byte by byte, so no alignment is assumed.
 */
#ifndef RL_KEYHASH_H
#define RL_KEYHASH_H
#include <stdint.h>

#if defined(__HIPCC__)
#define RL_KH_FN __host__ __device__ static inline
#else
#define RL_KH_FN static inline
#endif

typedef struct rl_h128 {
    uint64_t h1, h2;
} rl_h128;

/* the ingest's 128-bit secret */
typedef struct rl_hkey {
    uint64_t k0, k1;
} rl_hkey;

RL_KH_FN uint64_t rl_kh_rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

typedef struct rl_sip {
    uint64_t v0, v1, v2, v3;
} rl_sip;

RL_KH_FN rl_sip rl_sip_round(rl_sip s) {
    s.v0 += s.v1;
    s.v1 = rl_kh_rotl(s.v1, 13);
    s.v1 ^= s.v0;
    s.v0 = rl_kh_rotl(s.v0, 32);
    s.v2 += s.v3;
    s.v3 = rl_kh_rotl(s.v3, 16);
    s.v3 ^= s.v2;
    s.v0 += s.v3;
    s.v3 = rl_kh_rotl(s.v3, 21);
    s.v3 ^= s.v0;
    s.v2 += s.v1;
    s.v1 = rl_kh_rotl(s.v1, 17);
    s.v1 ^= s.v2;
    s.v2 = rl_kh_rotl(s.v2, 32);
    return s;
}

RL_KH_FN rl_sip rl_sip_init(rl_hkey k) {
    rl_sip s;
    s.v0 = k.k0 ^ 0x736f6d6570736575ull;
    s.v1 = k.k1 ^ 0x646f72616e646f6dull ^ 0xeeull; /* (128-bit output) */
    s.v2 = k.k0 ^ 0x6c7967656e657261ull;
    s.v3 = k.k1 ^ 0x7465646279746573ull;
    return s;
}

/* one 8-byte word (little endian) into the state: c = 2 compression rounds */
RL_KH_FN rl_sip rl_sip_word(rl_sip s, uint64_t m) {
    s.v3 ^= m;
    s = rl_sip_round(s);
    s = rl_sip_round(s);
    s.v0 ^= m;
    return s;
}

/* the last word (the message's length in its top byte) and the finalisation: d = 4 rounds per output word */
RL_KH_FN rl_h128 rl_sip_finish(rl_sip s, uint64_t last_word) {
    rl_h128 out;
    s = rl_sip_word(s, last_word);
    s.v2 ^= 0xeeull;
    s = rl_sip_round(s);
    s = rl_sip_round(s);
    s = rl_sip_round(s);
    s = rl_sip_round(s);
    out.h1 = s.v0 ^ s.v1 ^ s.v2 ^ s.v3;
    s.v1 ^= 0xddull;
    s = rl_sip_round(s);
    s = rl_sip_round(s);
    s = rl_sip_round(s);
    s = rl_sip_round(s);
    out.h2 = s.v0 ^ s.v1 ^ s.v2 ^ s.v3;
    return out;
}

RL_KH_FN uint64_t rl_kh_le64(const uint8_t *p, uint32_t n) { /* the first n (<= 8) bytes, little endian */
    uint64_t v = 0;
    for (uint32_t i = 0; i < n; ++i) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

/* SipHash-2-4-128_k(p[0..len)) */
RL_KH_FN rl_h128 rl_kh_bytes(const uint8_t *p, uint32_t len, rl_hkey k) {
    rl_sip s = rl_sip_init(k);
    uint32_t i = 0;
    for (; i + 8 <= len; i += 8) s = rl_sip_word(s, rl_kh_le64(p + i, 8));
    return rl_sip_finish(s, ((uint64_t)(len & 0xFFu) << 56) | rl_kh_le64(p + i, len - i));
}

/* SipHash-2-4-128_k of n 8-byte words (the message is their little-endian bytes: 8 n bytes, no tail) */
RL_KH_FN rl_h128 rl_kh_words(const uint64_t *w, uint32_t n, rl_hkey k) {
    rl_sip s = rl_sip_init(k);
    for (uint32_t i = 0; i < n; ++i) s = rl_sip_word(s, w[i]);
    return rl_sip_finish(s, (uint64_t)((8u * n) & 0xFFu) << 56);
}

/* (key, check) of the counter of a limit with prefix hash `prefix` and the hashes of its variables' values, in
 * variable-name order (n_vals = 0: a limit without variables; n_vals <= 8). */
RL_KH_FN void rl_counter_key(rl_h128 prefix, const rl_h128 *vals, uint32_t n_vals, rl_hkey k, uint64_t *key, uint32_t *check) {
    rl_sip st = rl_sip_init(k);
    st = rl_sip_word(st, prefix.h1);
    st = rl_sip_word(st, prefix.h2);
    for (uint32_t i = 0; i < n_vals; ++i) {
        st = rl_sip_word(st, vals[i].h1);
        st = rl_sip_word(st, vals[i].h2);
    }
    const rl_h128 s = rl_sip_finish(st, (uint64_t)((16u * (n_vals + 1u)) & 0xFFu) << 56);
    *key = s.h1 >= 0xFFFFFFFFFFFFFFFEull ? s.h1 - 2ull : s.h1;
    const uint32_t c = (uint32_t)(s.h2 >> 32);
    *check = c ? c : 1u;
}

#endif /* RL_KEYHASH_H */
