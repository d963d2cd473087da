"""Pure-Python restatement of include/rl_keyhash.h for the tests: SipHash-2-4 with 128-bit output (Aumasson & Bernstein's
published function; pinned by the paper's 64-bit vector and the reference implementation's first 128-bit vectors in
tests/test_keyhash_cpu.py) and the counter key the hashed mode derives from a counter's canonical key bytes (reference:
limitador/src/storage/keys.rs:209-248, key_for_counter_v2 = version byte 1 + postcard of CounterKey)."""
M = (1 << 64) - 1


def _rotl(x, r):
    return ((x << r) | (x >> (64 - r))) & M


def _round(v):
    v0, v1, v2, v3 = v
    v0 = (v0 + v1) & M
    v1 = _rotl(v1, 13) ^ v0
    v0 = _rotl(v0, 32)
    v2 = (v2 + v3) & M
    v3 = _rotl(v3, 16) ^ v2
    v0 = (v0 + v3) & M
    v3 = _rotl(v3, 21) ^ v0
    v2 = (v2 + v1) & M
    v1 = _rotl(v1, 17) ^ v2
    v2 = _rotl(v2, 32)
    return [v0, v1, v2, v3]


def _init(key, wide=True):
    k0, k1 = key
    v = [k0 ^ 0x736F6D6570736575, k1 ^ 0x646F72616E646F6D, k0 ^ 0x6C7967656E657261, k1 ^ 0x7465646279746573]
    if wide:
        v[1] ^= 0xEE
    return v


def _word(v, m):
    v[3] ^= m
    v = _round(_round(v))
    v[0] ^= m
    return v


def _finish(v, last, wide=True):
    v = _word(v, last)
    v[2] ^= 0xEE if wide else 0xFF
    v = _round(_round(_round(_round(v))))
    h1 = v[0] ^ v[1] ^ v[2] ^ v[3]
    if not wide:
        return h1
    v[1] ^= 0xDD
    v = _round(_round(_round(_round(v))))
    return h1, v[0] ^ v[1] ^ v[2] ^ v[3]


def siphash24(b, key, wide=True):
    """SipHash-2-4 of bytes `b` under key (k0, k1): the 64-bit value, or (h1, h2) of the 128-bit output."""
    v = _init(key, wide)
    i = 0
    while i + 8 <= len(b):
        v = _word(v, int.from_bytes(b[i:i + 8], "little"))
        i += 8
    return _finish(v, ((len(b) & 0xFF) << 56) | int.from_bytes(b[i:], "little"), wide)


def siphash24_words(words, key):
    """rl_kh_words: the 128-bit output over 8-byte little-endian words."""
    v = _init(key)
    for w in words:
        v = _word(v, w)
    return _finish(v, ((8 * len(words)) & 0xFF) << 56)


def varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def pstr(x):
    """postcard string: varint length + bytes"""
    x = x.encode() if isinstance(x, str) else bytes(x)
    return varint(len(x)) + x


def canonical_key_bytes(ns, seconds, conditions, variables):
    """key_for_counter_v2 of a counter without an id (keys.rs:236-241): 0x01 + postcard(CounterKey { ns, seconds,
    conditions (sorted), variables: Vec<(name, value)> (sorted by name: counter.rs variables_for_key) })."""
    out = b"\x01" + pstr(ns) + varint(seconds) + varint(len(conditions)) + b"".join(pstr(c) for c in sorted(conditions))
    out += varint(len(variables))
    for name, value in sorted(variables, key=lambda nv: nv[0].encode() if isinstance(nv[0], str) else nv[0]):
        out += pstr(name) + pstr(value)
    return out


def key_prefix_bytes(ns, seconds, conditions, var_names):
    """The canonical bytes WITHOUT the values (what a limit fixes): include/rl_keyhash.h."""
    enc = [v.encode() if isinstance(v, str) else bytes(v) for v in var_names]
    return (b"\x01" + pstr(ns) + varint(seconds) + varint(len(conditions)) + b"".join(pstr(c) for c in sorted(conditions))
            + varint(len(enc)) + b"".join(pstr(v) for v in sorted(enc)))


def counter_key(ns, seconds, conditions, var_names, values, key):
    """(key, check word) of include/rl_keyhash.h under the ingest's secret `key` = (k0, k1); `values` in variable-NAME order."""
    words = list(siphash24(key_prefix_bytes(ns, seconds, conditions, var_names), key))
    for v in values:
        v = v.encode() if isinstance(v, str) else bytes(v)
        words += list(siphash24(v, key))
    h1, h2 = siphash24_words(words, key)
    return (h1 - 2 if h1 >= M - 1 else h1), ((h2 >> 32) or 1)
