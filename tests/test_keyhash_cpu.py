"""include/rl_keyhash.h without a GPU: the header as plain C (what a cgo / FFI consumer compiles), the host library's
rli_counter_key, and the pure-Python restatement the GPU tests check the device against (tests/helpers/keyhash_ref.py) —
all three on the published SipHash-2-4 vectors, on every tail length, and on counters of the shapes the hashed key
mode serves.  The canonical bytes themselves are pinned by the one figure the reference's tests hold for them
(limitador/src/storage/keys.rs:416-460: 47 bytes for the counter built there)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import keyhash_ref as ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

K_PAPER = (int.from_bytes(bytes(range(8)), "little"), int.from_bytes(bytes(range(8, 16)), "little"))  # key 00 01 .. 0f
# SipHash-2-4, key 00..0f, message 00 01 .. (n-1): the paper's 64-bit vector (n = 15) and the reference implementation's
# first 128-bit vectors (vectors.h, "vectors_sip128": n = 0, 1), as the 16 output bytes
PAPER_64 = (bytes(range(15)), 0xA129CA6149BE45E5)
REF_128 = [(b"", "a3817f04ba25a8e66df67214c7550293"), (b"\x00", "da87c1d86b99af44347659119b22fc45")]

HARNESS = r"""
#include "rl_keyhash.h"
void kh_bytes(const uint8_t *p, uint32_t len, uint64_t k0, uint64_t k1, uint64_t *out) {
    rl_hkey k = {k0, k1};
    rl_h128 h = rl_kh_bytes(p, len, k);
    out[0] = h.h1;
    out[1] = h.h2;
}
void kh_words(const uint64_t *w, uint32_t n, uint64_t k0, uint64_t k1, uint64_t *out) {
    rl_hkey k = {k0, k1};
    rl_h128 h = rl_kh_words(w, n, k);
    out[0] = h.h1;
    out[1] = h.h2;
}
void kh_counter_key(const uint64_t *prefix, const uint64_t *vals, uint32_t n, uint64_t k0, uint64_t k1, uint64_t *key, uint32_t *check) {
    rl_hkey k = {k0, k1};
    rl_h128 p, v[8];
    p.h1 = prefix[0];
    p.h2 = prefix[1];
    for (uint32_t i = 0; i < n && i < 8; ++i) {
        v[i].h1 = vals[2 * i];
        v[i].h2 = vals[2 * i + 1];
    }
    rl_counter_key(p, v, n, k, key, check);
}
"""


@pytest.fixture(scope="module")
def c_header(tmp_path_factory):
    """The header compiled as C99 by gcc, warnings as errors: nothing in it needs hipcc or C++."""
    d = tmp_path_factory.mktemp("keyhash")
    src, so = d / "kh_harness.c", d / "kh_harness.so"
    src.write_text(HARNESS)
    subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                    str(src), "-o", str(so)], check=True)
    lib = C.CDLL(str(so))
    lib.kh_bytes.argtypes = [C.c_char_p, C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.kh_words.argtypes = [C.POINTER(C.c_uint64), C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.kh_counter_key.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint32, C.c_uint64, C.c_uint64,
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    return lib


def _c_bytes(lib, b, key):
    out = (C.c_uint64 * 2)()
    lib.kh_bytes(bytes(b), len(b), key[0], key[1], out)
    return out[0], out[1]


def test_published_vectors(c_header):
    msg, want = PAPER_64
    assert ref.siphash24(msg, K_PAPER, wide=False) == want  # (the restatement's rounds, constants and padding)
    for b, hexout in REF_128:
        want128 = (int.from_bytes(bytes.fromhex(hexout)[:8], "little"), int.from_bytes(bytes.fromhex(hexout)[8:], "little"))
        assert ref.siphash24(b, K_PAPER) == want128, b
        assert _c_bytes(c_header, b, K_PAPER) == want128, b


def test_every_tail_length_and_key(c_header):
    """0 .. 70 bytes (no word, the 7 tail lengths, several words + tail, lengths whose low byte wraps are out of reach here),
    bytes of every value, three keys; and the words form against the bytes form of the same message."""
    rng = np.random.default_rng(5)
    for n in range(71):
        b = bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
        for key in ((0, 0), K_PAPER, (0xDEADBEEFCAFEF00D, 0x0123456789ABCDEF)):
            assert _c_bytes(c_header, b, key) == ref.siphash24(b, key), (n, key)
            if n % 8 == 0:
                words = [int.from_bytes(b[i:i + 8], "little") for i in range(0, n, 8)]
                out = (C.c_uint64 * 2)()
                c_header.kh_words((C.c_uint64 * max(1, len(words)))(*words), len(words), key[0], key[1], out)
                assert (out[0], out[1]) == ref.siphash24(b, key) == ref.siphash24_words(words, key)


def test_a_different_key_gives_unrelated_keys():
    """What the key is for: the same counter under two secrets has unrelated (key, check) — nothing an attacker
    learns about one deployment's keys, or computes offline, carries over."""
    a = ref.counter_key("shop", 60, ["c"], ["user"], ["alice"], (1, 2))
    b = ref.counter_key("shop", 60, ["c"], ["user"], ["alice"], (1, 3))
    assert a != b and bin(a[0] ^ b[0]).count("1") > 12


def test_canonical_key_bytes_are_the_reference_length():
    """keys.rs:416-460 (`counters_with_id`): namespace "ns_counter:", 1 second, one condition, variable app_id = foo ->
    key_for_counter_v2 is 47 bytes long for the limit without an id."""
    b = ref.canonical_key_bytes("ns_counter:", 1, ["req_method == 'GET'"], [("app_id", "foo")])
    assert len(b) == 47 and b[0] == 1
    # the prefix the limit fixes + the values = the same fields, none dropped
    p = ref.key_prefix_bytes("ns_counter:", 1, ["req_method == 'GET'"], ["app_id"])
    assert len(p) + len(ref.pstr("foo")) == len(b)


def test_counter_key_folds_the_reserved_tags_and_never_gives_check_zero(c_header):
    """rl_counter_key: SipHash-2-4-128 over the digests as 8-byte words, key = h1 (0xFF..FE / 0xFF..FF are the table's
    EMPTY / tombstone tags), check = upper word of h2, never 0 — against the restatement on random states and keys."""
    rng = np.random.default_rng(11)
    for n in range(0, 9):
        for _ in range(40):
            u = lambda m: [int(x) for x in rng.integers(0, 1 << 63, size=m, dtype=np.uint64) * 2 + rng.integers(0, 2, size=m, dtype=np.uint64)]  # noqa: E731
            prefix, vals, key = u(2), u(2 * n), tuple(u(2))
            out_key, chk = C.c_uint64(0), C.c_uint32(0)
            c_header.kh_counter_key((C.c_uint64 * 2)(*prefix), (C.c_uint64 * max(1, 2 * n))(*vals) if n else (C.c_uint64 * 1)(), n,
                                    key[0], key[1], C.byref(out_key), C.byref(chk))
            h1, h2 = ref.siphash24_words(prefix + vals, key)
            assert out_key.value == (h1 - 2 if h1 >= ref.M - 1 else h1)
            assert chk.value == ((h2 >> 32) or 1) and chk.value != 0 and out_key.value < ref.M - 1


def test_host_library_gives_the_key_of_the_canonical_bytes(engine_lib):
    """rli_counter_key (what RLI_KEYS_HASHED installs for simple counters and reports for any counter) against the
    restatement built from the strings alone: namespace, seconds, sorted condition sources, sorted variable sources, the
    values in variable-name order — empty, 16- and 17-byte, long, non-UTF-8 values; 0, 1 and 2 variables."""
    from limitador_amd.ingest import Ingest

    g = Ingest(keys="hashed")
    hkey = g.hash_key
    assert hkey != (0, 0) and Ingest(keys="hashed").hash_key != hkey  # drawn per ingest
    conds = ["descriptors[0]['method'] == 'GET'", "descriptors[0]['path'] != '/admin'"]
    v2 = ["descriptors[0]['user']", "descriptors[0]['app']"]  # (name order: ...['app'] < ...['user'])
    shapes = [("shop", 60, conds, []), ("shop", 3600, conds[:1], v2[:1]), ("shop", 7, [], v2), ("n", 1, [], []),
              ("a-namespace-name-that-is-longer-than-one-hash-block", 86400 * 365, conds[::-1], v2[::-1])]
    ids = [g.add_limit(ns, 100, s, c, v) for ns, s, c, v in shapes]
    assert ids == list(range(len(shapes)))
    values = [b"", b"alice", b"0123456789abcdef", b"0123456789abcdefg", bytes(range(1, 200)), b"\xff\xfe\x00tail"[:2], "ünïcode"]
    for lid, (ns, s, c, v) in zip(ids, shapes):
        nv = len(v)
        for i, a in enumerate(values):
            vals = [a, values[(i + 3) % len(values)]][:nv]
            assert g.counter_key(lid, vals) == ref.counter_key(ns, s, c, v, vals, hkey), (lid, vals)
    # the ORDER of the values is the order of the variable NAMES, however the limit listed them
    assert g.counter_key(ids[2], [b"x", b"y"]) != g.counter_key(ids[2], [b"y", b"x"])
    # a second ingest given the SAME secret derives the same keys (several front-ends in front of one table); the secret
    # cannot change once the limits are compiled
    g2 = Ingest(keys="hashed", hash_key=hkey)
    for ns, s_, c, v in shapes:
        g2.add_limit(ns, 100, s_, c, v)
    assert g2.counter_key(ids[1], [b"alice"]) == g.counter_key(ids[1], [b"alice"])
    from limitador_amd.ingest import IngestError
    with pytest.raises(IngestError):
        g2._check(ingest_symbols()["rli_set_hash_key"](g2._h, 1, 2))
    g.close()
    g2.close()


def ingest_symbols():
    from limitador_amd import ingest

    return ingest.SYMBOLS
