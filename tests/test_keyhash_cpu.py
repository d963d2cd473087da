"""include/rl_keyhash.h without a GPU: the header as plain C (what a cgo / FFI consumer compiles), the host library's
rli_counter_key, and the pure-Python restatement the GPU tests check the device against (tests/helpers/keyhash_ref.py) —
all three on the published MurmurHash3_x64_128 vectors, on every tail length, and on counters of the shapes the hashed key
mode serves.  The canonical bytes themselves are pinned by the one figure the reference's tests hold for them
(limitador/src/storage/keys.rs:416-460: 47 bytes for the counter built there)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import keyhash_ref as ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PUBLISHED = [  # MurmurHash3_x64_128, seed 0 (h1, h2)
    (b"", (0x0000000000000000, 0x0000000000000000)),
    (b"hello", (0xCBD8A7B341BD9B02, 0x5B1E906A48AE1D19)),
    (b"hello, world", (0x342FAC623A5EBC8E, 0x4CDCBC079642414D)),
    (b"The quick brown fox jumps over the lazy dog", (0xE34BBC7BBC071B6C, 0x7A433CA9C49A9347)),
]

HARNESS = r"""
#include "rl_keyhash.h"
void kh_bytes(const uint8_t *p, uint32_t len, uint64_t seed, uint64_t *out) {
    rl_h128 h = rl_kh_bytes(p, len, seed);
    out[0] = h.h1;
    out[1] = h.h2;
}
void kh_counter_key(const uint64_t *prefix, const uint64_t *vals, uint32_t n, uint64_t *key, uint32_t *check) {
    rl_h128 p, v[8];
    p.h1 = prefix[0];
    p.h2 = prefix[1];
    for (uint32_t i = 0; i < n && i < 8; ++i) {
        v[i].h1 = vals[2 * i];
        v[i].h2 = vals[2 * i + 1];
    }
    rl_counter_key(p, v, n, key, check);
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
    lib.kh_bytes.argtypes = [C.c_char_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.kh_counter_key.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    return lib


def _c_bytes(lib, b, seed=0):
    out = (C.c_uint64 * 2)()
    lib.kh_bytes(bytes(b), len(b), seed, out)
    return out[0], out[1]


def test_published_vectors(c_header):
    for b, want in PUBLISHED:
        assert ref.murmur3_x64_128(b) == want, b
        assert _c_bytes(c_header, b) == want, b


def test_every_tail_length_and_seed(c_header):
    """0 .. 70 bytes (no block, the 15 tail lengths, several blocks + tail), bytes of every value, three seeds."""
    rng = np.random.default_rng(5)
    for n in range(71):
        b = bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
        for seed in (0, 1, 0xDEADBEEFCAFEF00D):
            assert _c_bytes(c_header, b, seed) == ref.murmur3_x64_128(b, seed), (n, seed)


def test_canonical_key_bytes_are_the_reference_length():
    """keys.rs:416-460 (`counters_with_id`): namespace "ns_counter:", 1 second, one condition, variable app_id = foo ->
    key_for_counter_v2 is 47 bytes long for the limit without an id."""
    b = ref.canonical_key_bytes("ns_counter:", 1, ["req_method == 'GET'"], [("app_id", "foo")])
    assert len(b) == 47 and b[0] == 1
    # the prefix the limit fixes + the values = the same fields, none dropped
    p = ref.key_prefix_bytes("ns_counter:", 1, ["req_method == 'GET'"], ["app_id"])
    assert len(p) + len(ref.pstr("foo")) == len(b)


def test_counter_key_folds_the_reserved_tags_and_never_gives_check_zero(c_header):
    """rl_counter_key: block steps over the value hashes, the finalisation with 16 * n + 1, key = h1 (0xFF..FE / 0xFF..FF
    are the table's EMPTY / tombstone tags), check = upper word of h2, never 0 — against the restatement on random states."""
    rng = np.random.default_rng(11)
    for n in range(0, 9):
        for _ in range(40):
            prefix = [int(x) for x in rng.integers(0, 1 << 63, size=2, dtype=np.uint64) * 2 + rng.integers(0, 2, size=2, dtype=np.uint64)]
            vals = [int(x) for x in rng.integers(0, 1 << 63, size=2 * n, dtype=np.uint64) * 2 + rng.integers(0, 2, size=2 * n, dtype=np.uint64)]
            key, chk = C.c_uint64(0), C.c_uint32(0)
            c_header.kh_counter_key((C.c_uint64 * 2)(*prefix), (C.c_uint64 * max(1, 2 * n))(*vals) if n else (C.c_uint64 * 1)(), n,
                                    C.byref(key), C.byref(chk))
            h1, h2 = prefix
            for i in range(n):
                h1, h2 = ref.block(h1, h2, vals[2 * i], vals[2 * i + 1])
            h1, h2 = ref.finish(h1, h2, 16 * n + 1)
            assert key.value == (h1 - 2 if h1 >= ref.M - 1 else h1)
            assert chk.value == ((h2 >> 32) or 1) and chk.value != 0 and key.value < ref.M - 1


def test_host_library_gives_the_key_of_the_canonical_bytes(engine_lib):
    """rli_counter_key (what RLI_KEYS_HASHED installs for simple counters and reports for any counter) against the
    restatement built from the strings alone: namespace, seconds, sorted condition sources, sorted variable sources, the
    values in variable-name order — empty, 16- and 17-byte, long, non-UTF-8 values; 0, 1 and 2 variables."""
    from limitador_amd.ingest import Ingest

    g = Ingest(keys="hashed")
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
            assert g.counter_key(lid, vals) == ref.counter_key(ns, s, c, v, vals), (lid, vals)
    # the ORDER of the values is the order of the variable NAMES, however the limit listed them
    assert g.counter_key(ids[2], [b"x", b"y"]) != g.counter_key(ids[2], [b"y", b"x"])
    g.close()
