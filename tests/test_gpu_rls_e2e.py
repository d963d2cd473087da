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


def _install(make_engine):
    eng = make_engine(capacity_cells=1 << 16, max_batch_hits=1 << 15)
    g = Ingest()  # the transports' binding: descriptors[0][...]
    model = TestsLimiter(oracle.OracleStorage(), now_us=NOW)
    for ns, mx, secs, conds, variables, name in _limits():
        lid = g.add_limit(ns, mx, secs, [f"descriptors[0]['{k}'] {op} '{v}'" for k, op, v in conds],
                          [f"descriptors[0]['{k}']" for k in variables])
        assert lid >= 0
        g.set_limit_name(lid, name)
        model.add_limit(Limit(ns, mx, secs, [f"{k} {op} '{v}'" for k, op, v in conds], variables, name=name))
    g.install(eng)
    return eng, g, model


def test_rate_limit_requests_from_the_wire_to_the_wire(make_engine):
    rng = np.random.default_rng(77)
    Resp = _response_class()
    eng, g, model = _install(make_engine)
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


def test_micro_batcher_in_front_of_the_wire_path(make_engine):
    """64 threads x 40 ShouldRateLimit calls on one tight limit: the batches are aggregated (fewer device batches
    than requests), every request gets its own answer, and exactly max_value of them are OK."""
    Resp = _response_class()
    eng = make_engine(capacity_cells=1 << 12, max_batch_hits=1 << 12)
    g = Ingest()
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
    for threads in ("1", "7"):
        monkeypatch.setenv("RLI_THREADS", threads)
        eng, g, _model = _install(make_engine)
        out = []
        for b, msgs in enumerate(batches):
            out.append(g.serve_batch(eng, msgs, NOW + b * 700_000, with_headers=bool(b % 2)))
        results.append(out)
    for b in range(3):
        (s1, r1), (s7, r7) = results[0][b], results[1][b]
        assert s1 == s7, f"batch {b}: statuses"
        assert r1 == r7, f"batch {b}: response bytes"
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


def test_the_reference_sandbox_fixture_through_the_wire_path(make_engine):
    """BASELINE.json configs[0]'s shape as the reference ships it: the limits FILE text through rli_add_limit, the load
    test's request as wire bytes.  GET /json only meets the 50000 / 10 s limit: exactly 50 000 of 50 400 requests inside
    one window are OK, the rest OVER_LIMIT, the headers count down; after the window everything is OK again.  Beside it
    a mix of the other two limits' traffic.  Every status and every response byte against the pinned model + oracle."""
    import re

    Resp = _response_class()
    eng = make_engine(capacity_cells=1 << 12, max_batch_hits=1 << 15)
    g = Ingest()
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
