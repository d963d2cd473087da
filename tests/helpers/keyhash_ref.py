"""Pure-Python restatement of include/rl_keyhash.h for the tests: MurmurHash3_x64_128 (Austin Appleby's published
function: block step, tail, fmix64) and the counter key the hashed mode derives from a counter's canonical key bytes
(reference: limitador/src/storage/keys.rs:209-248, key_for_counter_v2 = version byte 1 + postcard of CounterKey)."""
M = (1 << 64) - 1
C1, C2 = 0x87C37B91114253D5, 0x4CF5AD432745937F


def _rotl(x, r):
    return ((x << r) | (x >> (64 - r))) & M


def fmix(k):
    k ^= k >> 33
    k = (k * 0xFF51AFD7ED558CCD) & M
    k ^= k >> 33
    k = (k * 0xC4CEB9FE1A85EC53) & M
    return k ^ (k >> 33)


def _mix1(k):
    return (_rotl((k * C1) & M, 31) * C2) & M


def _mix2(k):
    return (_rotl((k * C2) & M, 33) * C1) & M


def block(h1, h2, k1, k2):
    h1 = (((_rotl(h1 ^ _mix1(k1), 27) + h2) & M) * 5 + 0x52DCE729) & M
    h2 = (((_rotl(h2 ^ _mix2(k2), 31) + h1) & M) * 5 + 0x38495AB5) & M
    return h1, h2


def finish(h1, h2, n):
    h1 ^= n
    h2 ^= n
    h1 = (h1 + h2) & M
    h2 = (h2 + h1) & M
    h1, h2 = fmix(h1), fmix(h2)
    h1 = (h1 + h2) & M
    return h1, (h2 + h1) & M


def murmur3_x64_128(b, seed=0):
    h1 = h2 = seed
    i = 0
    while i + 16 <= len(b):
        h1, h2 = block(h1, h2, int.from_bytes(b[i:i + 8], "little"), int.from_bytes(b[i + 8:i + 16], "little"))
        i += 16
    t = b[i:]
    if len(t) > 8:
        h2 ^= _mix2(int.from_bytes(t[8:], "little"))
    if t:
        h1 ^= _mix1(int.from_bytes(t[:8], "little"))
    return finish(h1, h2, len(b))


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


def counter_key(ns, seconds, conditions, var_names, values):
    """(key, check word) of include/rl_keyhash.h; `values` in variable-NAME order."""
    h1, h2 = murmur3_x64_128(key_prefix_bytes(ns, seconds, conditions, var_names))
    for v in values:
        v = v.encode() if isinstance(v, str) else bytes(v)
        h1, h2 = block(h1, h2, *murmur3_x64_128(v))
    h1, h2 = finish(h1, h2, 16 * len(values) + 1)
    return (h1 - 2 if h1 >= M - 1 else h1), ((h2 >> 32) or 1)
