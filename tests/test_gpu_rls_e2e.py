"""End to end on the wire path (BASELINE.json configs[4] shape): serialized
envoy.service.ratelimit.v3.RateLimitRequests in, serialized RateLimitResponses out — decode, limit matching and
check_and_update on the device, the draft-03 rate-limit headers built by product code (rli_serve_batch) — replayed
batch by batch against the test-side mirror of RateLimiter over the CPU oracle (tests/helpers/limiter.py, whose
response_header is pinned by the reference's vectors: envoy_rls/server.rs:390-397,576-583,644-672 in
tests/scenarios.py).  Then the micro-batcher in front of it, from many threads.  Needs a MI355X."""
import threading

import numpy as np
import pytest

import oracle
from helpers.limiter import Limit, TestsLimiter
from limitador_amd.ingest import UNKNOWN_DOMAIN, Frontend, Ingest
from test_gpu_parity import make_engine  # noqa: F401
from test_ingest_cpu import rls_request

pytestmark = pytest.mark.gpu

NOW = 1_700_000_000_000_000


def _response_class():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

    f = descriptor_pb2.FileDescriptorProto(name="rls_e2e.proto", package="e2e", syntax="proto3")
    hv = descriptor_pb2.DescriptorProto(name="HeaderValue")
    hv.field.add(name="key", number=1, type=9, label=1)
    hv.field.add(name="value", number=2, type=9, label=1)
    resp = descriptor_pb2.DescriptorProto(name="RateLimitResponse")
    resp.field.add(name="overall_code", number=1, type=13, label=1)  # UNKNOWN 0, OK 1, OVER_LIMIT 2
    resp.field.add(name="response_headers_to_add", number=3, type=11, label=3, type_name=".e2e.HeaderValue")
    f.message_type.extend([hv, resp])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("e2e.RateLimitResponse"))


def _limits():
    """4 namespaces x 8 limits: 2 without variables (one generous, one tight), 6 qualified on user / (app, user);
    conditions on method / path; windows 1, 10, 60, 3600 s.  -> [(namespace, max, seconds, [(key, op, value)], [var keys], name)]"""
    out = []
    methods = ["GET", "POST", "PUT"]
    for n in range(4):
        ns = f"ns{n}"
        for j in range(8):
            nv = 0 if j < 2 else (1 if j < 5 else 2)
            conds = [("method", "==" if j % 2 == 0 else "!=", methods[j % 3])]
            if j % 4 == 3:
                conds.append(("path", "!=", "/admin"))
            variables = [] if nv == 0 else (["user"] if nv == 1 else ["app", "user"])
            max_value = 10**6 if j == 0 else [7, 25, 60, 3][(n + j) % 4]
            out.append((ns, max_value, [1, 10, 60, 3600][(n + j) % 4], conds, variables, f"{ns}-limit{j}"))
    return out


def _install(make_engine, keys="exact", **engine_kw):
    engine_kw.setdefault("capacity_cells", 1 << 16)
    engine_kw.setdefault("max_batch_hits", 1 << 15)
    eng = make_engine(**engine_kw)
    g = Ingest(keys=keys)  # the transports' binding: descriptors[0][...]
    model = TestsLimiter(oracle.OracleStorage(), now_us=NOW)
    for ns, mx, secs, conds, variables, name in _limits():
        lid = g.add_limit(ns, mx, secs, [f"descriptors[0]['{k}'] {op} '{v}'" for k, op, v in conds],
                          [f"descriptors[0]['{k}']" for k in variables])
        assert lid >= 0
        g.set_limit_name(lid, name)
        model.add_limit(Limit(ns, mx, secs, [f"{k} {op} '{v}'" for k, op, v in conds], variables, name=name))
    g.install(eng)
    return eng, g, model


# keys: "exact" = host dictionaries + packed ids; "hashed" = the messages decoded on the device, counters keyed by a hash
# of their canonical key bytes (rl_wire.hpp, include/rl_keyhash.h) — same statuses, same response bytes.
@pytest.mark.parametrize("keys", ["exact", "hashed"])
def test_rate_limit_requests_from_the_wire_to_the_wire(make_engine, keys):
    rng = np.random.default_rng(77)
    Resp = _response_class()
    eng, g, model = _install(make_engine, keys)
    methods, paths = ["GET", "POST", "PUT"], ["/", "/admin", "/json"]
    n_ok = n_over = n_unknown = 0
    for batch in range(20):
        msgs, ctxs = [], []
        for _ in range(int(rng.integers(150, 400))):
            r = rng.random()
            domain = "" if r < 0.02 else ("elsewhere" if r < 0.05 else f"ns{int(rng.integers(0, 4))}")
            ctx = {}
            if rng.random() < 0.95:
                ctx["method"] = methods[int(rng.integers(0, 3))]
            if rng.random() < 0.8:
                ctx["path"] = paths[int(rng.integers(0, 3))]
            if rng.random() < 0.9:
                ctx["user"] = f"user{int(rng.zipf(1.4)) % 40}"
            if rng.random() < 0.7:
                ctx["app"] = f"app{int(rng.integers(0, 3))}"
            entries = list(ctx.items())
            rng.shuffle(entries)
            if "user" in ctx and rng.random() < 0.1:  # a repeated key keeps its LAST value (server.rs:122-128)
                entries = [("user", "overwritten")] + entries
            entries.append(("ignored", "x"))
            addend = int(rng.integers(0, 4)) if rng.random() < 0.7 else None  # absent or 0 means 1 (server.rs:131-137)
            msgs.append(rls_request(domain if domain else None, [entries], hits_addend=addend))
            ctxs.append((domain, ctx, addend if addend else 1))
        now = model.now_us
        status, responses = g.serve_batch(eng, msgs, now, with_headers=True)
        for i, (domain, ctx, delta) in enumerate(ctxs):
            m = Resp()
            m.ParseFromString(responses[i])
            if not domain:  # Code::Unknown, server.rs:105-115
                assert status[i] == UNKNOWN_DOMAIN and m.overall_code == 0 and not m.response_headers_to_add
                n_unknown += 1
                continue
            want = model.check_rate_limited_and_update(domain, ctx, delta, True)
            assert status[i] == (1 if want.limited else 0), (batch, i, domain, ctx)
            assert m.overall_code == (2 if want.limited else 1)
            got = [(h.key, h.value) for h in m.response_headers_to_add]
            assert got == sorted(want.response_header().items()), (batch, i, domain, ctx)
            n_ok += not want.limited
            n_over += want.limited
        model.sleep([0.0, 0.3, 1.1, 4.0][batch % 4])  # crosses the 1 s and 10 s windows
        if batch % 5 == 4:
            assert eng.sweep_expired(model.now_us) == model.storage.sweep_expired(model.now_us)
    assert n_ok > 500 and n_over > 500 and n_unknown > 20


@pytest.mark.parametrize("keys", ["exact", "hashed"])
def test_takes_into_account_all_the_descriptors(make_engine, keys):
    """envoy_rls/server.rs:497-592, transcribed: two limits on `descriptors[0].z`, the second one (max 0) also asks for
    `descriptors[1].y == '2'`.  The request's second descriptor carries y = 2: OVER_LIMIT, three headers,
    X-RateLimit-Limit exactly "0, 0;w=60, 10;w=60" (:576-583), remaining "0", reset <= 60 — on the device path, in both
    key modes (VERDICT r05 missing #1: round 5's matcher compiled descriptors[0] only and this test could not run)."""
    Resp = _response_class()
    eng = make_engine(capacity_cells=1 << 12, max_batch_hits=1 << 12)
    g = Ingest(keys=keys)
    assert g.add_limit("test_namespace", 10, 60, ["descriptors[0].x == '1'"], ["descriptors[0].z"]) == 0
    assert g.add_limit("test_namespace", 0, 60, ["descriptors[0].x == '1'", "descriptors[1].y == '2'"], ["descriptors[0].z"]) == 1
    g.install(eng)
    req = rls_request("test_namespace", [[("x", "1"), ("z", "1")], [("y", "2")]], hits_addend=1)
    status, responses = g.serve_batch(eng, [req], NOW, with_headers=True)
    m = Resp()
    m.ParseFromString(responses[0])
    assert status == [1] and m.overall_code == 2  # Code::OverLimit
    hdr = {h.key: h.value for h in m.response_headers_to_add}
    assert len(m.response_headers_to_add) == 3
    assert hdr["X-RateLimit-Limit"] == "0, 0;w=60, 10;w=60"
    assert hdr["X-RateLimit-Remaining"] == "0"
    assert int(hdr["X-RateLimit-Reset"]) <= 60
    # ... and the same request WITHOUT the second descriptor only meets the first limit (the reference's comment at :561-562)
    one = rls_request("test_namespace", [[("x", "1"), ("z", "1")]], hits_addend=1)
    status, responses = g.serve_batch(eng, [one], NOW, with_headers=True)
    m.ParseFromString(responses[0])
    assert status == [0] and m.overall_code == 1
    assert {h.key: h.value for h in m.response_headers_to_add}["X-RateLimit-Limit"] == "10, 10;w=60"
    # the blocked request above counted nothing (all-or-nothing, in_memory.rs:141-153): 9 left of 10 after ONE admitted hit
    assert {h.key: h.value for h in m.response_headers_to_add}["X-RateLimit-Remaining"] == "9"


_MULTI_LIMITS = [
    # (max, seconds, [(descriptor, key, op, value)], [(descriptor, key)], name)
    (6, 60, [(1, "plan", "==", "free")], [(0, "u")], "free-per-user"),
    (9, 10, [(0, "m", "==", "GET"), (1, "plan", "!=", "free")], [(1, "t")], "paid-get-per-tenant"),
    (4, 60, [], [(0, "u"), (0, "a"), (1, "t")], "three-variables"),
    (3, 60, [(2, "r", "==", "eu")], [(0, "u"), (0, "a"), (1, "t"), (1, "plan")], "four-variables"),
    (40, 1, [(1, "t", "!=", "tenant0")], [], "simple-on-second-descriptor"),
]


@pytest.mark.parametrize("keys", ["exact", "hashed"])
def test_every_descriptor_and_up_to_four_variables_from_the_wire_to_the_wire(make_engine, keys):
    """Limits that read descriptors[0], [1] and [2] and carry one to FOUR variables (limit.rs:133-148 resolves any number),
    served from serialized requests with headers, batch by batch against the mirror of RateLimiter over the oracle — the
    request's context there is the same descriptor list, flattened to "d<i>.<key>" names."""
    rng = np.random.default_rng(5)
    Resp = _response_class()
    eng = make_engine(capacity_cells=1 << 14, max_batch_hits=1 << 14)
    g = Ingest(keys=keys)
    model = TestsLimiter(oracle.OracleStorage(), now_us=NOW)
    for mx, secs, conds, variables, name in _MULTI_LIMITS:
        lid = g.add_limit("ns", mx, secs, [f"descriptors[{i}]['{k}'] {op} '{v}'" for i, k, op, v in conds],
                          [f"descriptors[{i}].{k}" for i, k in variables])
        assert lid >= 0
        g.set_limit_name(lid, name)
        model.add_limit(Limit("ns", mx, secs, [f"d{i}.{k} {op} '{v}'" for i, k, op, v in conds],
                              [f"d{i}.{k}" for i, k in variables], name=name))
    g.install(eng)
    n_over = n_ok = 0
    for batch in range(12):
        msgs, ctxs = [], []
        for _ in range(int(rng.integers(100, 250))):
            d0 = {"m": ["GET", "POST"][int(rng.integers(0, 2))], "u": f"user{int(rng.integers(0, 6))}"}
            if rng.random() < 0.7:
                d0["a"] = f"app{int(rng.integers(0, 3))}"
            d1 = {"t": f"tenant{int(rng.integers(0, 3))}"}
            if rng.random() < 0.6:
                d1["plan"] = ["free", "pro"][int(rng.integers(0, 2))]
            descs = [d0, d1]
            if rng.random() < 0.5:
                descs.append({"r": ["eu", "us"][int(rng.integers(0, 2))]})
            wire = [list(d.items()) for d in descs]
            if rng.random() < 0.2:  # a repeated key inside ONE descriptor keeps its last value; the same key elsewhere is another key
                wire[1] = [("t", "overwritten")] + wire[1]
                wire[0] = wire[0] + [("t", "not-the-tenant")]
            msgs.append(rls_request("ns", wire))
            ctxs.append({f"d{i}.{k}": v for i, d in enumerate(descs) for k, v in d.items()})
        status, responses = g.serve_batch(eng, msgs, model.now_us, with_headers=True)
        for i, ctx in enumerate(ctxs):
            want = model.check_rate_limited_and_update("ns", ctx, 1, True)
            m = Resp()
            m.ParseFromString(responses[i])
            assert status[i] == (1 if want.limited else 0), (batch, i, ctx)
            assert [(h.key, h.value) for h in m.response_headers_to_add] == sorted(want.response_header().items()), (batch, i, ctx)
            n_over += want.limited
            n_ok += not want.limited
        model.sleep([0.0, 0.4, 1.2][batch % 3])
    assert n_ok > 200 and n_over > 200


@pytest.mark.parametrize("keys", ["exact", "hashed"])
def test_micro_batcher_in_front_of_the_wire_path(make_engine, keys):
    """64 threads x 40 ShouldRateLimit calls on one tight limit: the batches are aggregated (fewer device batches
    than requests), every request gets its own answer, and exactly max_value of them are OK."""
    Resp = _response_class()
    eng = make_engine(capacity_cells=1 << 12, max_batch_hits=1 << 12)
    g = Ingest(keys=keys)
    assert g.add_limit("shop", 1000, 60, ["descriptors[0]['method'] == 'GET'"], []) == 0
    assert g.add_limit("shop", 10**6, 60, [], ["descriptors[0]['user']"]) == 1
    g.install(eng)
    fe = Frontend(g, eng, max_batch=128, max_delay_us=300, with_headers=True)
    fe.set_clock(NOW)
    ok = [0] * 64
    bad = []

    def worker(t):
        for q in range(40):
            st, resp = fe.should_rate_limit(rls_request("shop", [[("method", "GET"), ("user", f"u{t}")]]))
            m = Resp()
            m.ParseFromString(resp)
            if st not in (0, 1) or m.overall_code != (2 if st else 1) or len(m.response_headers_to_add) != 3:
                bad.append((t, q, st))
            ok[t] += st == 0

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(64)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    batches, requests = fe.stats()
    fe.close()
    assert not bad
    assert sum(ok) == 1000
    assert requests == 64 * 40 and batches < requests
    g.close()


def test_large_batches_are_decoded_and_answered_by_several_threads(make_engine, monkeypatch):
    """rli_serve_batch splits the host side of a batch of more than a thousand messages over threads (decode +
    dictionary encoding before the device call, response bytes after it).  Two identical engines, the same three
    batches of 6000 messages (new dictionary values arriving from all threads at once, unknown domains, a few
    malformed messages): the threaded run must answer byte for byte what the serial one (RLI_THREADS=1) answers."""
    rng = np.random.default_rng(5)
    methods, paths = ["GET", "POST", "PUT"], ["/", "/admin", "/json"]
    batches = []
    for b in range(3):
        msgs = []
        for i in range(6000):
            r = rng.random()
            domain = None if r < 0.02 else ("elsewhere" if r < 0.04 else f"ns{int(rng.integers(0, 4))}")
            entries = [("method", methods[int(rng.integers(0, 3))]), ("path", paths[int(rng.integers(0, 3))]),
                       ("user", f"u{b}-{int(rng.zipf(1.3)) % 3000}"), ("app", f"app{int(rng.integers(0, 3))}")]
            m = rls_request(domain, [entries], hits_addend=int(rng.integers(0, 3)))
            if r > 0.995:
                m = m[: len(m) // 2]  # cut in the middle: malformed
            msgs.append(m)
        batches.append(msgs)
    results = []
    for threads, keys in (("1", "exact"), ("7", "exact"), ("7", "hashed")):
        monkeypatch.setenv("RLI_THREADS", threads)
        eng, g, _model = _install(make_engine, keys)
        out = []
        for b, msgs in enumerate(batches):
            out.append(g.serve_batch(eng, msgs, NOW + b * 700_000, with_headers=bool(b % 2)))
        results.append(out)
    for b in range(3):
        (s1, r1), (s7, r7), (sh, rh) = results[0][b], results[1][b], results[2][b]
        assert s1 == s7, f"batch {b}: statuses"
        assert r1 == r7, f"batch {b}: response bytes"
        # the device's decoder agrees with the host's on every message, malformed ones included
        assert s1 == sh, f"batch {b}: statuses, hashed keys: {[(i, a, c) for i, (a, c) in enumerate(zip(s1, sh)) if a != c][:10]}"
        assert r1 == rh, f"batch {b}: response bytes, hashed keys"
        assert sum(1 for s in s1 if s == 1) > 100 and sum(1 for s in s1 if s == 0) > 100 and any(s < -1 for s in s1)


# The reference's own sandbox fixture, transcribed: limitador-server/sandbox/limits.yaml:1-22 (three variable-less limits
# of one namespace, two conditions each on descriptors[0]['req.method'] / ['req.path']) and
# limitador-server/sandbox/load-test.json (the request its load test sends: GET /json, hits_addend 1).
SANDBOX_LIMITS = [
    ("test_namespace", 10, 60, ["descriptors[0]['req.method'] == 'GET'", "descriptors[0]['req.path'] != '/json'"], []),
    ("test_namespace", 5, 60, ["descriptors[0]['req.method'] == 'POST'", "descriptors[0]['req.path'] != '/json'"], []),
    ("test_namespace", 50000, 10, ["descriptors[0]['req.method'] == 'GET'", "descriptors[0]['req.path'] == '/json'"], []),
]
SANDBOX_REQUEST = ("test_namespace", [[("req.method", "GET"), ("req.path", "/json")]], 1)


@pytest.mark.parametrize("keys", ["exact", "hashed"])
def test_the_reference_sandbox_fixture_through_the_wire_path(make_engine, keys):
    """BASELINE.json configs[0]'s shape as the reference ships it: the limits FILE text through rli_add_limit, the load
    test's request as wire bytes.  GET /json only meets the 50000 / 10 s limit: exactly 50 000 of 50 400 requests inside
    one window are OK, the rest OVER_LIMIT, the headers count down; after the window everything is OK again.  Beside it
    a mix of the other two limits' traffic.  Every status and every response byte against the pinned model + oracle."""
    import re

    Resp = _response_class()
    eng = make_engine(capacity_cells=1 << 12, max_batch_hits=1 << 15)
    g = Ingest(keys=keys)
    model = TestsLimiter(oracle.OracleStorage(), now_us=NOW)
    for ns, mx, secs, conds, variables in SANDBOX_LIMITS:
        assert g.add_limit(ns, mx, secs, conds, variables) >= 0
        # (the model takes the same conditions in its `key op 'value'` spelling)
        model.add_limit(Limit(ns, mx, secs, [re.sub(r"descriptors\[0\]\['([^']+)'\]", r"\1", c) for c in conds], variables))
    g.install(eng)
    domain, descriptors, addend = SANDBOX_REQUEST
    load = rls_request(domain, descriptors, hits_addend=addend)
    ctx = dict(descriptors[0])
    n_ok = n_over = 0
    for batch in range(14):  # 14 x 3600 = 50 400 requests, 1 ms apart: all inside the 10 s window
        now = model.now_us
        status, responses = g.serve_batch(eng, [load] * 3600, now, with_headers=True)
        for i in range(3600):
            want = model.check_rate_limited_and_update(domain, ctx, addend, True)
            assert status[i] == (1 if want.limited else 0), (batch, i)
            if i % 97 == 0 or want.limited != (n_over > 0):
                m = Resp()
                m.ParseFromString(responses[i])
                assert m.overall_code == (2 if want.limited else 1)
                assert [(h.key, h.value) for h in m.response_headers_to_add] == sorted(want.response_header().items())
            n_ok += not want.limited
            n_over += want.limited
        model.sleep(0.001)
    assert (n_ok, n_over) == (50_000, 400)
    # the other two limits of the file: GET elsewhere (10 / 60 s), POST elsewhere (5 / 60 s); POST /json meets none
    mixed = [("GET", "/"), ("POST", "/x"), ("POST", "/json"), ("GET", "/json")] * 6
    msgs = [rls_request(domain, [[("req.method", m_), ("req.path", p_)]], hits_addend=1) for m_, p_ in mixed]
    status, responses = g.serve_batch(eng, msgs, model.now_us, with_headers=True)
    for i, (m_, p_) in enumerate(mixed):
        want = model.check_rate_limited_and_update(domain, {"req.method": m_, "req.path": p_}, 1, True)
        assert status[i] == (1 if want.limited else 0), (i, m_, p_)
        m = Resp()
        m.ParseFromString(responses[i])
        assert [(h.key, h.value) for h in m.response_headers_to_add] == sorted(want.response_header().items())
    assert [status[i] for i in (1, 5, 9, 13, 17, 21)] == [0, 0, 0, 0, 0, 1]  # the sixth POST /x is the first over 5
    # the 10 s window runs out: the load test's request is OK again
    model.sleep(10.5)
    status, _ = g.serve_batch(eng, [load] * 10, model.now_us, with_headers=False)
    assert status == [0] * 10
    g.close()


# ---- hashed keys: what only that mode has -----------------------------------------------------------------------------
def _b(x):
    return x if isinstance(x, bytes) else x.encode()


def rls_request_b(domain, entries, hits_addend=None):
    """rls_request for keys / values given as bytes (any bytes: protobuf strings may hold them)."""
    from test_ingest_cpu import _ld

    msg = _ld(1, _b(domain)) + _ld(2, b"".join(_ld(1, _ld(1, _b(k)) + _ld(2, _b(v))) for k, v in entries))
    if hits_addend is not None:
        msg += _varint((3 << 3) | 0) + _varint(hits_addend)
    return msg


def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def test_hashed_keys_are_the_hash_of_the_canonical_key_bytes(make_engine):
    """The cells the device creates carry exactly the (key, check word) include/rl_keyhash.h defines over the counter's
    canonical key bytes (storage/keys.rs:220-248): recomputed here from the strings and the ingest's secret alone —
    SipHash-2-4-128 of the prefix (version, namespace, seconds, sorted conditions, sorted variable names in postcard) and of
    every value, then of those digests as 8-byte words (tests/helpers/keyhash_ref.py, pinned by the published vectors in
    tests/test_keyhash_cpu.py) — and compared with the table's dump.  A second engine fed the same messages through an
    ingest with ANOTHER secret holds the same counters under unrelated keys."""
    from helpers import keyhash_ref as kh

    def key_of(ns, seconds, conds, var_names, values, hkey=None):
        return kh.counter_key(ns, seconds, conds, var_names, values, hkey or g.hash_key)

    eng = make_engine(capacity_cells=1 << 12, max_batch_hits=1 << 12)
    g = Ingest(keys="hashed")
    conds = ["descriptors[0]['method'] == 'GET'", "descriptors[0]['path'] != '/admin'"]
    v2 = ["descriptors[0]['user']", "descriptors[0]['app']"]
    l0 = g.add_limit("shop", 100, 60, conds, [])
    l1 = g.add_limit("shop", 100, 3600, conds[:1], v2[:1])
    l2 = g.add_limit("shop", 100, 7, [], v2)
    g.install(eng)
    users = [b"alice", b"", b"a-user-name-of-more-than-sixteen-bytes", bytes(range(1, 40))]
    msgs = [rls_request_b("shop", [("method", "GET"), ("path", "/"), ("user", u), ("app", b"app%d" % (i % 2))]) for i, u in enumerate(users)]
    status, _ = g.serve_batch(eng, msgs, NOW)
    assert status == [0] * len(users)
    want = {key_of("shop", 60, conds, [], []): l0 | 0x80000000}
    for i, u in enumerate(users):
        want[key_of("shop", 3600, conds[:1], v2[:1], [u])] = l1
        want[key_of("shop", 7, [], v2, [b"app%d" % (i % 2), u])] = l2  # (variables in NAME order: ...['app'] < ...['user'])
    rows = eng.dump_cells()
    got = {(int(r["key"]), int(r["reserved"])): int(r["limit"]) for r in rows}
    assert {k: v for k, v in got.items() if v & 0x80000000} == {(k[0], 0): v for k, v in want.items() if v & 0x80000000}  # add_counter: no check word
    assert {k: v for k, v in got.items() if not v & 0x80000000} == {k: v for k, v in want.items() if not v & 0x80000000}
    assert g.counter_key(l1, [users[2]]) == key_of("shop", 3600, conds[:1], v2[:1], [users[2]])
    assert g.counter_key(l0) [0] == key_of("shop", 60, conds, [], [])[0]
    # another secret: the same counters (limit, value, expiry), not one key in common
    eng2 = make_engine(capacity_cells=1 << 12, max_batch_hits=1 << 12)
    g2 = Ingest(keys="hashed", hash_key=(g.hash_key[0] ^ 1, g.hash_key[1]))
    for args in (("shop", 100, 60, conds, []), ("shop", 100, 3600, conds[:1], v2[:1]), ("shop", 100, 7, [], v2)):
        g2.add_limit(*args)
    g2.install(eng2)
    assert g2.serve_batch(eng2, msgs, NOW)[0] == status
    rows2 = eng2.dump_cells()
    assert sorted((int(r["limit"]), int(r["value"]), int(r["expiry_us"])) for r in rows) == \
        sorted((int(r["limit"]), int(r["value"]), int(r["expiry_us"])) for r in rows2)
    assert not set(int(k) for k in rows["key"]) & set(int(k) for k in rows2["key"])
    g2.close()
    g.close()


def test_two_counters_that_share_a_key_are_never_merged(make_engine):
    """A cell that already holds ANOTHER counter's check word under the key a message derives (forged here by loading
    such a row: finding a real 64-bit collision is not an option): the message is answered HOST_ONLY, nothing of it is
    applied, the forged cell is untouched, and the other messages of the batch are applied as if it had not been there."""
    from limitador_amd.ingest import HOST_ONLY
    from limitador_amd.wire import CELL_ROW_DTYPE

    eng, g, model = _install(make_engine, "hashed")
    lims = _limits()
    j = next(i for i, L in enumerate(lims) if L[0] == "ns1" and L[4] == ["user"])
    key, chk = g.counter_key(j, ["mallory"])
    row = np.zeros(1, dtype=CELL_ROW_DTYPE)
    row[0] = (key, j, chk ^ 0x5A5A, 3, NOW + 50_000_000)
    eng.load_cells(row)
    ctxs = [("ns1", {"method": m, "path": "/", "user": u, "app": "app0"}) for u in ("bob", "mallory", "carol", "bob") for m in ("PUT", "GET")]
    msgs = [rls_request(d, [list(c.items())]) for d, c in ctxs]
    status, _ = g.serve_batch(eng, msgs, NOW, with_headers=True)
    # every message of mallory that meets limit j is taken out (one per retry), the rest is answered like the model does
    applies = [any(L[0] == d and L[4] == ["user"] and li == j and model_applies(L, c) for li, L in enumerate(lims)) for d, c in ctxs]
    for i, (d, c) in enumerate(ctxs):
        if c["user"] == "mallory" and applies[i]:
            assert status[i] == HOST_ONLY, i
        else:
            want = model.check_rate_limited_and_update(d, c, 1, True)
            assert status[i] == (1 if want.limited else 0), i
    assert sum(s == HOST_ONLY for s in status) == 1 and any(applies)
    after = eng.dump_cells()
    forged = after[after["key"] == key]
    assert len(forged) == 1 and int(forged[0]["value"]) == 3 and int(forged[0]["reserved"]) == chk ^ 0x5A5A
    g.close()


def test_many_colliding_messages_take_one_rerun_and_a_limit_mismatch_is_a_collision(make_engine):
    """(a) 150 messages of one batch whose counters' keys are held by OTHER counters — one re-run per
    colliding message, capped at 64, used to turn that into a failed batch of up to 262 144 requests; the device now names every
    colliding message in one pass and the batch is served with ONE re-run.  (b) A stored cell that shares a message's key
    under another LIMIT id (hashed keys carry the limit in the key, so that is two counters sharing a key) used to come back
    as RL_ERR_KEY_LIMIT and fail the whole batch, again on every later batch carrying that message; it is a collision like
    any other: that message HOST_ONLY, the rest applied.  Both forged by loading rows (a real 64-bit collision is a
    2^32-hash birthday search, cheap for an attacker but not for a test: include/rl_keyhash.h)."""
    from limitador_amd.ingest import HOST_ONLY
    from limitador_amd.wire import CELL_ROW_DTYPE

    eng, g, model = _install(make_engine, "hashed")
    lims = _limits()
    j = next(i for i, L in enumerate(lims) if L[0] == "ns1" and L[4] == ["user"])
    other = next(i for i, L in enumerate(lims) if i != j and L[0] == "ns1")
    victims = [f"victim-{i}" for i in range(150)]
    rows = np.zeros(len(victims) + 1, dtype=CELL_ROW_DTYPE)
    for i, u in enumerate(victims):
        key, chk = g.counter_key(j, [u])
        rows[i] = (key, j, chk ^ (0x1000 + i), 2, NOW + 50_000_000)          # another counter's check word under the key
    key_lm, chk_lm = g.counter_key(j, ["limit-mismatch"])
    rows[-1] = (key_lm, other, chk_lm ^ 0x77, 5, NOW + 50_000_000)            # ... and under ANOTHER limit id
    eng.load_cells(rows)
    before = np.sort(rows, order="key")  # (the forged rows only: the limits' own simple cells are counted on by honest messages)
    users = victims + ["limit-mismatch"] + [f"honest-{i}" for i in range(300)]
    rng = np.random.default_rng(9)
    order = rng.permutation(len(users))
    # (a method / path under which limit j applies, whatever its conditions are in _limits())
    base = next({"method": m, "path": pth, "app": "app0"} for m in ("GET", "POST", "PUT") for pth in ("/", "/admin")
                if model_applies(lims[j], {"method": m, "path": pth, "app": "app0", "user": "x"}))
    ctxs = [("ns1", dict(base, user=users[i])) for i in order]
    msgs = [rls_request(d, [list(c.items())]) for d, c in ctxs]
    for rerun in range(2):  # the second batch meets the same cells again: still served, still without them
        status, _ = g.serve_batch(eng, msgs, NOW + rerun, with_headers=bool(rerun))
        n_host = 0
        for i, (d, c) in enumerate(ctxs):
            applies_j = model_applies(lims[j], c)
            if applies_j and (c["user"].startswith("victim-") or c["user"] == "limit-mismatch"):
                assert status[i] == HOST_ONLY, (i, c["user"], status[i])
                n_host += 1
            else:
                want = model.check_rate_limited_and_update(d, c, 1, bool(rerun))
                assert status[i] == (1 if want.limited else 0), (i, c["user"])
        assert n_host == 151
    after = np.sort(eng.dump_cells(), order="key")
    forged = after[np.isin(after["key"], before["key"])]
    for f in ("key", "limit", "value", "expiry_us", "reserved"):
        assert np.array_equal(forged[f], before[f]), f"a forged cell was touched ({f})"
    g.close()


def model_applies(L, ctx):
    for k, op, v in L[3]:
        if k not in ctx or (ctx[k] == v) != (op == "=="):
            return False
    return all(k in ctx for k in L[4])


def test_both_key_modes_on_one_million_messages(make_engine):
    """The same trace of 2^20 serialized requests (8 batches of 131 072, Zipf users, four namespaces x 8 limits) through
    the dictionary path and through the device decoder with hashed keys: identical statuses, and the two tables hold the
    same counters — the same multiset of (limit, value, expiry); only the keys' spelling differs."""
    rng = np.random.default_rng(2024)
    methods, paths = [b"GET", b"POST", b"PUT"], [b"/", b"/admin", b"/json"]
    n_per, n_batches = 1 << 17, 8
    users = [b"user-%d" % i for i in range(200_000)]
    engines = {k: _install(make_engine, k, capacity_cells=1 << 22, max_batch_hits=1 << 20)[:2] for k in ("exact", "hashed")}
    for b in range(n_batches):
        ns = rng.integers(0, 4, size=n_per)
        me = rng.integers(0, 3, size=n_per)
        pa = rng.integers(0, 3, size=n_per)
        us = rng.zipf(1.2, size=n_per) % len(users)
        ap = rng.integers(0, 3, size=n_per)
        msgs = [rls_request_b(b"ns%d" % ns[i], [(b"method", methods[me[i]]), (b"path", paths[pa[i]]), (b"user", users[us[i]]), (b"app", b"app%d" % ap[i])])
                for i in range(n_per)]
        out = {k: g.serve_batch(eng, msgs, NOW + b * 400_000)[0] for k, (eng, g) in engines.items()}
        assert out["exact"] == out["hashed"], f"batch {b}: {[(i, a, c) for i, (a, c) in enumerate(zip(out['exact'], out['hashed'])) if a != c][:10]}"
        assert 0 in out["exact"] and 1 in out["exact"]
    tables = {}
    for k, (eng, g) in engines.items():
        rows = eng.dump_cells()
        t = np.stack([rows["limit"].astype(np.uint64), rows["value"], rows["expiry_us"]], axis=1)
        tables[k] = t[np.lexsort(t.T[::-1])]
        g.close()
    assert tables["exact"].shape == tables["hashed"].shape and np.array_equal(tables["exact"], tables["hashed"])
    assert len(tables["exact"]) > 100_000


def test_the_device_decoder_agrees_with_the_host_decoder_on_mangled_messages(make_engine):
    """Differential fuzz of the two protobuf readers (csrc/host/ingest.cpp decode_rls; rl_wire.hpp k_wire_count): valid
    messages with junk fields, repeated keys / fields, empty strings, oversized varints, then byte flips, truncations and
    insertions.  Both key modes must give every message the same status (0 / 1 / UNKNOWN_DOMAIN / malformed) — and, since
    statuses include the verdicts, the same counters behind them."""
    rng = np.random.default_rng(31337)
    from test_ingest_cpu import _ld

    def valid():
        entries = [("method", ["GET", "POST", "PUT", ""][int(rng.integers(0, 4))]), ("path", ["/", "/admin"][int(rng.integers(0, 2))]),
                   ("user", "u%d" % int(rng.integers(0, 50))), ("app", "app%d" % int(rng.integers(0, 3)))]
        rng.shuffle(entries)
        if rng.random() < 0.3:
            entries.append(entries[int(rng.integers(0, len(entries)))])  # a repeated key
        m = rls_request("ns%d" % int(rng.integers(0, 5)), [entries, [("method", "PUT")]][: int(rng.integers(1, 3))],
                        hits_addend=[None, 0, 1, 3, 2**32 + 5][int(rng.integers(0, 5))], junk=rng.random() < 0.3)
        if rng.random() < 0.1:
            m = _ld(1, b"ns0") + m  # the domain twice: the last one counts
        if rng.random() < 0.1:
            m += _varint((7 << 3) | 1) + bytes(8) + _varint((8 << 3) | 5) + bytes(4) + _varint((9 << 3) | 0) + _varint(2**63)
        return m

    def mangle(m):
        b = bytearray(m)
        r = rng.random()
        if r < 0.35 and b:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif r < 0.6 and b:
            del b[int(rng.integers(0, len(b))):]
        elif r < 0.8:
            at = int(rng.integers(0, len(b) + 1))
            b[at:at] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 6))).astype(np.uint8))
        elif r < 0.9:
            b = bytearray(b"\xff" * int(rng.integers(1, 12)))  # a varint that never ends
        return bytes(b)

    msgs = [valid() if rng.random() < 0.4 else mangle(valid()) for _ in range(20_000)]
    msgs[0] = b""
    out = {}
    for keys in ("exact", "hashed"):
        eng, g, _model = _install(make_engine, keys)
        res = []
        for b in range(4):
            st, resp = g.serve_batch(eng, msgs[b * 5000:(b + 1) * 5000], NOW + b * 300_000, with_headers=bool(b & 1))
            res.append((st, resp))
        out[keys] = res
        g.close()
    for b in range(4):
        (se, re_), (sh, rh) = out["exact"][b], out["hashed"][b]
        bad = [(i, a, c, msgs[b * 5000 + i].hex()) for i, (a, c) in enumerate(zip(se, sh)) if a != c]
        assert not bad, bad[:5]
        assert re_ == rh, f"batch {b}: response bytes"
    flat = [s for st, _ in out["exact"] for s in st]
    assert sum(1 for s in flat if s == 0) > 2000 and sum(1 for s in flat if s == 1) > 500
    assert sum(1 for s in flat if s == UNKNOWN_DOMAIN) > 20 and sum(1 for s in flat if s not in (0, 1, UNKNOWN_DOMAIN)) > 1000


@pytest.mark.parametrize("keys", ["exact", "hashed"])
def test_responses_built_on_the_device_are_the_bytes_the_host_assembly_builds(make_engine, keys, monkeypatch):
    """rl_resp.hpp against round 4's host assembly from the counters' arrays (RLI_RESP_HOST=1, experiment builds): the same
    batches through two engines, every response byte for byte.  The awkward cases on purpose: names with quotes and a name
    long enough for X-RateLimit-Limit to need a two-byte length (and HeaderValue a two-byte length of its own), counters
    with EQUAL remaining (the stable order of lib.rs:240 decides which one is reported), a limit without a name, max values
    of 1 and 20 digits, requests that derive no counter (no headers at all), no domain, an unknown namespace."""
    specs = [("shop", 10**6, 60, ["descriptors[0]['m'] == 'GET'"], [], 'say "hi"'),
             ("shop", 10**6, 60, ["descriptors[0]['m'] != 'PUT'"], [], None),  # same remaining as the first: order decides
             ("shop", 18446744073709551615, 3600, [], ["descriptors[0]['u']"], "x" * 150),
             ("shop", 5, 1, ["descriptors[0]['m'] == 'GET'"], ["descriptors[0]['u']"], "tight"),
             ("shop", 3, 10, [], ["descriptors[0]['a']", "descriptors[0]['u']"], ""),
             ("solo", 1, 60, [], [], "one")]

    def service():
        eng = make_engine(capacity_cells=1 << 16, max_batch_hits=1 << 16)
        g = Ingest(keys=keys, hash_key=(7, 9))
        for ns, mx, secs, conds, variables, name in specs:
            lid = g.add_limit(ns, mx, secs, conds, variables)
            if name is not None:
                g.set_limit_name(lid, name)
        g.install(eng)
        return eng, g

    (eng_d, g_d), (eng_h, g_h) = service(), service()
    rng = np.random.default_rng(31)
    seen_long = seen_plain = 0
    for batch in range(12):
        msgs = []
        for _ in range(int(rng.integers(200, 500)) if batch != 7 else 9000):  # (one batch above the size from which the
                                                                              # device form is the default)
            r = rng.random()
            domain = None if r < 0.03 else ("nowhere" if r < 0.06 else ("solo" if r < 0.12 else "shop"))
            entries = []
            if rng.random() < 0.9:
                entries.append(("m", ["GET", "POST", "PUT"][int(rng.integers(0, 3))]))
            if rng.random() < 0.85:
                entries.append(("u", f"u{int(rng.integers(0, 30))}"))
            if rng.random() < 0.5:
                entries.append(("a", f"a{int(rng.integers(0, 2))}"))
            msgs.append(rls_request(domain, [entries], hits_addend=int(rng.integers(0, 3))))
        now = NOW + batch * 700_000
        monkeypatch.delenv("RLI_RESP_HOST", raising=False)
        if batch != 7:
            monkeypatch.setenv("RLI_RESP_DEVICE", "1")
        st_d, resp_d = g_d.serve_batch(eng_d, msgs, now, with_headers=True)
        monkeypatch.delenv("RLI_RESP_DEVICE", raising=False)
        monkeypatch.setenv("RLI_RESP_HOST", "1")
        st_h, resp_h = g_h.serve_batch(eng_h, msgs, now, with_headers=True)
        assert st_d == st_h
        for i, (a, b) in enumerate(zip(resp_d, resp_h)):
            assert a == b, (batch, i, a, b)
            seen_long += len(a) > 300
            seen_plain += len(a) == 2
    assert seen_long > 100 and seen_plain > 20
    # a stride the long responses do not fit: they alone are told, the others are served (both assemblies)
    monkeypatch.delenv("RLI_RESP_HOST", raising=False)
    monkeypatch.setenv("RLI_RESP_DEVICE", "1")
    st, resp = g_d.serve_batch(eng_d, msgs, now + 1, with_headers=True, stride=128)
    assert any(s == -102 for s in st) and any(s in (0, 1) and len(r) > 2 for s, r in zip(st, resp))


@pytest.mark.parametrize("n_threads", [2, 4])
def test_two_serving_calls_in_flight_answer_like_one_at_a_time(make_engine, monkeypatch, n_threads):
    """Round 6 (VERDICT r05 missing #3): two — or four, one per serving set — threads call rli_serve_batch on ONE ingest /
    engine at once; the reference serves from N tonic workers behind a shared read lock (envoy_rls/server.rs:238-272,
    in_memory.rs:78).  Each call takes one of the engine's serving sets (its own pinned staging, piece events, response
    snapshot, device copy of the messages); the engine's mutex orders the decisions.  Every thread serves namespaces of its
    own — disjoint counters, so each thread's answers must equal ITS OWN sequential model whatever the interleaving —
    statuses and header bytes, 10 batches each."""
    monkeypatch.setenv("RLI_RESP_DEVICE", "1")  # (small batches: force the device's response kernels — the form that overlaps)
    Resp = _response_class()
    eng, g, _ = _install(make_engine, "hashed")
    methods, paths = ["GET", "POST", "PUT"], ["/", "/admin", "/json"]
    failures = []
    start = threading.Barrier(n_threads)
    per = 4 // n_threads  # (namespaces ns0 .. ns3, _limits)

    def worker(t):
        rng = np.random.default_rng(100 + t)
        model = TestsLimiter(oracle.OracleStorage(), now_us=NOW)
        for ns, mx, secs, conds, variables, name in _limits():
            model.add_limit(Limit(ns, mx, secs, [f"{k} {op} '{v}'" for k, op, v in conds], variables, name=name))
        start.wait()
        for batch in range(10):
            msgs, ctxs = [], []
            for _ in range(int(rng.integers(200, 500))):
                domain = f"ns{per * t + int(rng.integers(0, per))}"
                ctx = {"method": methods[int(rng.integers(0, 3))], "path": paths[int(rng.integers(0, 3))],
                       "user": f"user{int(rng.zipf(1.4)) % 40}"}
                if rng.random() < 0.7:
                    ctx["app"] = f"app{int(rng.integers(0, 3))}"
                msgs.append(rls_request(domain, [list(ctx.items())]))
                ctxs.append((domain, ctx))
            status, responses = g.serve_batch(eng, msgs, NOW, with_headers=True)  # (one clock value: no window ends)
            for i, (domain, ctx) in enumerate(ctxs):
                want = model.check_rate_limited_and_update(domain, ctx, 1, True)
                m = Resp()
                m.ParseFromString(responses[i])
                got = [(h.key, h.value) for h in m.response_headers_to_add]
                if status[i] != (1 if want.limited else 0) or got != sorted(want.response_header().items()):
                    failures.append((t, batch, i, status[i], want.limited, got))
                    return

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not failures, failures[:3]


def test_four_callers_in_flight_at_size_answer_like_each_alone(make_engine):
    """The form the wire path takes when callers are in flight together (four serving sets; the response bytes through a device
    buffer and copy commands issued lazily by the hand-over threads, piece by piece) at a size where a call really has several
    pieces: four threads, one namespace each, three batches of 131 072 messages with headers on ONE ingest / engine — and the same
    batches served one at a time, each thread's on an engine of its own.  Statuses, lengths and every response byte must be equal."""
    import ctypes as C

    n, batches, n_threads = 131072, 3, 4  # (a call of 131 072 messages leaves in five to eight pieces)
    methods, paths = ["GET", "POST", "PUT"], ["/", "/admin", "/json"]

    def messages(t, b):
        rng = np.random.default_rng(1000 * t + b)
        out = []
        for _ in range(n):
            ctx = [("method", methods[int(rng.integers(0, 3))]), ("path", paths[int(rng.integers(0, 3))]),
                   ("user", f"user{int(rng.zipf(1.3)) % 5000}"), ("app", f"app{int(rng.integers(0, 3))}")]
            out.append(rls_request(f"ns{t}", [ctx]))
        return out

    def fresh():
        return _install(make_engine, "hashed", capacity_cells=1 << 21, max_batch_hits=1 << 20)[:2]

    def results(prep):
        lens = np.frombuffer(prep["out_len"], dtype=np.uint32).copy()
        raw = np.frombuffer(prep["out"], dtype=np.uint8).reshape(n, prep["stride"])
        return (np.frombuffer(prep["status"], dtype=np.int32).copy(), lens,
                [raw[i, :lens[i]].tobytes() for i in range(0, n, 97)] + [raw[n - 1, :lens[n - 1]].tobytes()],
                int(np.bitwise_xor.reduce(np.where(np.arange(prep["stride"])[None, :] < lens[:, None], raw, 0).astype(np.uint64).sum(axis=1) * np.arange(1, n + 1, dtype=np.uint64))))

    eng, g = fresh()
    preps = [[g.prepare_batch(messages(t, b), stride=512) for b in range(batches)] for t in range(n_threads)]
    got = [[None] * batches for _ in range(n_threads)]
    failures = []
    start = threading.Barrier(n_threads)

    def worker(t):
        try:
            start.wait()
            for b in range(batches):
                g.serve_prepared(eng, preps[t][b], NOW, with_headers=True)
                got[t][b] = results(preps[t][b])
        except Exception as ex:  # noqa: BLE001
            failures.append((t, repr(ex)))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not failures, failures
    for t in range(n_threads):
        eng1, g1 = fresh()
        for b in range(batches):
            p = g1.prepare_batch(preps[t][b]["keep"], stride=512)
            g1.serve_prepared(eng1, p, NOW, with_headers=True)
            want = results(p)
            assert np.array_equal(got[t][b][0], want[0]), f"thread {t} batch {b}: statuses"
            assert np.array_equal(got[t][b][1], want[1]), f"thread {t} batch {b}: lengths"
            assert got[t][b][2] == want[2], f"thread {t} batch {b}: sampled response bytes"
            assert got[t][b][3] == want[3], f"thread {t} batch {b}: checksum over all response bytes"
        assert int((got[t][-1][0] == 1).sum()) > 100  # (a real allow / deny mix by the last batch)
        g1.close()
        eng1.close()


def test_a_table_is_not_served_under_another_hash_key(make_engine, tmp_path):
    """the hash key names the cells.  An engine that holds hashed counters — filled directly, or reloaded from a
    snapshot (the file's header carries the key's fingerprint, never the key) — refuses the tables of an ingest with another
    key, where accepting them would silently restart every limit; the ingest that was given the first one's key is served and
    goes on counting on the same cells."""
    from limitador_amd.ingest import IngestError

    def ingest(hash_key=None):
        g = Ingest(keys="hashed", hash_key=hash_key)
        assert g.add_limit("ns", 5, 60, [], ["descriptors[0].u"]) == 0
        return g

    eng = make_engine(capacity_cells=1 << 12, max_batch_hits=1 << 12)
    g1 = ingest()
    g1.install(eng)
    req = rls_request("ns", [[("u", "alice")]])
    status, _ = g1.serve_batch(eng, [req] * 3, NOW, with_headers=False)
    assert status == [0, 0, 0]
    other = ingest()  # (a key of its own, drawn at rli_create)
    with pytest.raises(IngestError, match="ANOTHER hash key"):
        other.install(eng)
    path = str(tmp_path / "snap.bin")
    eng.snapshot_save(path)
    eng2 = make_engine(capacity_cells=1 << 12, max_batch_hits=1 << 12)
    eng2.snapshot_load(path)
    with pytest.raises(IngestError, match="ANOTHER hash key"):
        ingest().install(eng2)
    g2 = ingest(hash_key=g1.hash_key)
    g2.install(eng2)
    status, _ = g2.serve_batch(eng2, [req] * 3, NOW, with_headers=False)
    assert status == [0, 0, 1]  # 3 of 5 were counted before the snapshot: two more fit, the sixth does not
