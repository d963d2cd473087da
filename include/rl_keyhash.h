/*
 * rl_keyhash.h — counter keys derived from the CANONICAL KEY BYTES of a counter, the same function on the host and on
 * the device (row f1 of SURVEY.md §8: "hash the canonical key bytes on device").
 *
 * The reference names a counter by the byte string of key_for_counter_v2
 * (limitador/src/storage/keys.rs:220-248): version byte 1 + postcard of
 *     CounterKey { ns: &str, seconds: u64, conditions: Vec<String> (sorted), variables: Vec<(&str, &str)> (sorted by name) }
 * (postcard: LEB128 varints, a string = varint length + bytes, a Vec = varint length + elements).  The engine's table
 * wants a 64-bit key.  The key is a hash of exactly those bytes, computed hierarchically so that the part a LIMIT
 * fixes is hashed once, on the host, when the limit is compiled:
 *
 *     prefix  = 0x01, str(ns), varint(seconds), varint(#conditions), str(condition_i)..., varint(#variables),
 *               str(variable_name_i)...                                  (the canonical bytes WITHOUT the values)
 *     P       = MurmurHash3_x64_128(prefix, seed 0)                     (rl_kh_bytes)                    -- per limit
 *     V_i     = MurmurHash3_x64_128(bytes of value_i, seed 0)           (variables in name order)        -- per request
 *     S       = P;  S = murmur block step(S, V_i.h1, V_i.h2) for every i;  S = murmur finalisation(S, 16 * #values + 1)
 *     key     = S.h1 (the two reserved tags 0xFF..FE / 0xFF..FF folded down by 2)
 *     check   = upper 32 bits of S.h2, never 0
 *
 * (prefix, values) <-> canonical bytes is a bijection (both parse uniquely), so two counters share (key, check) only
 * if the 96 bits collide.  The 64-bit key addresses the cell; the 32-bit check is stored beside it (the cell's spare
 * word) when the cell is created and compared on every later touch: a mismatch is REPORTED (the request is answered
 * RLI_HOST_ONLY and not applied), never merged.  Odds, for N live counters: two of them share a key with probability
 * ~N^2 / 2^65 (N = 10^7: 2.7e-6 over the table's life), in which case the second one is refused; an UNDETECTED merge
 * needs the check to collide as well: ~N^2 / 2^97 (N = 10^7: 6e-16).
 *
 * MurmurHash3_x64_128 is Austin Appleby's public-domain function, restated here from its published description
 * (block step, tail, fmix64); the device reads strings byte by byte, so no alignment is assumed.
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

RL_KH_FN uint64_t rl_kh_rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

RL_KH_FN uint64_t rl_kh_fmix(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}

RL_KH_FN uint64_t rl_kh_mix_k1(uint64_t k1) {
    k1 *= 0x87c37b91114253d5ull;
    k1 = rl_kh_rotl(k1, 31);
    k1 *= 0x4cf5ad432745937full;
    return k1;
}
RL_KH_FN uint64_t rl_kh_mix_k2(uint64_t k2) {
    k2 *= 0x4cf5ad432745937full;
    k2 = rl_kh_rotl(k2, 33);
    k2 *= 0x87c37b91114253d5ull;
    return k2;
}

/* one 16-byte block (k1 = bytes 0..7, k2 = bytes 8..15, little endian) into the state */
RL_KH_FN rl_h128 rl_kh_block(rl_h128 s, uint64_t k1, uint64_t k2) {
    s.h1 ^= rl_kh_mix_k1(k1);
    s.h1 = rl_kh_rotl(s.h1, 27);
    s.h1 += s.h2;
    s.h1 = s.h1 * 5 + 0x52dce729;
    s.h2 ^= rl_kh_mix_k2(k2);
    s.h2 = rl_kh_rotl(s.h2, 31);
    s.h2 += s.h1;
    s.h2 = s.h2 * 5 + 0x38495ab5;
    return s;
}

RL_KH_FN rl_h128 rl_kh_finish(rl_h128 s, uint64_t len) {
    s.h1 ^= len;
    s.h2 ^= len;
    s.h1 += s.h2;
    s.h2 += s.h1;
    s.h1 = rl_kh_fmix(s.h1);
    s.h2 = rl_kh_fmix(s.h2);
    s.h1 += s.h2;
    s.h2 += s.h1;
    return s;
}

RL_KH_FN uint64_t rl_kh_le64(const uint8_t *p, uint32_t n) { /* the first n (<= 8) bytes, little endian */
    uint64_t v = 0;
    for (uint32_t i = 0; i < n; ++i) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

/* MurmurHash3_x64_128(p[0..len), seed) */
RL_KH_FN rl_h128 rl_kh_bytes(const uint8_t *p, uint32_t len, uint64_t seed) {
    rl_h128 s;
    s.h1 = seed;
    s.h2 = seed;
    uint32_t i = 0;
    for (; i + 16 <= len; i += 16) s = rl_kh_block(s, rl_kh_le64(p + i, 8), rl_kh_le64(p + i + 8, 8));
    const uint32_t tail = len - i;
    if (tail > 8) s.h2 ^= rl_kh_mix_k2(rl_kh_le64(p + i + 8, tail - 8));
    if (tail > 0) s.h1 ^= rl_kh_mix_k1(rl_kh_le64(p + i, tail > 8 ? 8 : tail));
    return rl_kh_finish(s, len);
}

/* (key, check) of the counter of a limit with prefix hash `prefix` and the hashes of its variables' values, in
 * variable-name order (n_vals = 0: a limit without variables). */
RL_KH_FN void rl_counter_key(rl_h128 prefix, const rl_h128 *vals, uint32_t n_vals, uint64_t *key, uint32_t *check) {
    rl_h128 s = prefix;
    for (uint32_t i = 0; i < n_vals; ++i) s = rl_kh_block(s, vals[i].h1, vals[i].h2);
    s = rl_kh_finish(s, 16ull * n_vals + 1ull);
    *key = s.h1 >= 0xFFFFFFFFFFFFFFFEull ? s.h1 - 2ull : s.h1;
    const uint32_t c = (uint32_t)(s.h2 >> 32);
    *check = c ? c : 1u;
}

#endif /* RL_KEYHASH_H */
