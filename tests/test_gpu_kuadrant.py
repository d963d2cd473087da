"""The Kuadrant RateLimitService through the wire front-end (VERDICT r04 missing #5): CheckRateLimit and Report
(limitador-server/src/envoy_rls/kuadrant_service.rs:27-184) = RateLimiter::is_rate_limited(ns, ctx, 1) and
RateLimiter::update_counters(ns, ctx, hits_addend or 1) (lib.rs:362-423) over the device matcher — rli_serve_batch_op /
rli_frontend_check_rate_limit / rli_frontend_report (include/rl_ingest.h), rl_match_batch_op / rl_wire_match_batch_op
(include/rl_engine.h).  First the reference's own tests of the service (kuadrant_service.rs:208-420 check_rate_limit,
:486-690 report; the one that reads descriptors[1] is out of the device matcher's shapes: such limits are RLI_HOST_ONLY),
then random traffic of the wasm-shim's pattern — check, then report what was let through — against the test-side mirror of
RateLimiter over the CPU oracle, in both key modes, then the micro-batcher with the three methods mixed.  Needs a MI355X."""
import threading

import numpy as np
import pytest

from limitador_amd.ingest import OP_CHECK, OP_CHECK_AND_UPDATE, OP_UPDATE, UNKNOWN_DOMAIN, Frontend, Ingest
from test_gpu_parity import make_engine  # noqa: F401
from test_gpu_rls_e2e import _install, _limits, _response_class
from test_ingest_cpu import rls_request

pytestmark = pytest.mark.gpu

NOW = 1_700_000_000_000_000
OK, OVER_LIMIT, UNKNOWN = 1, 2, 0  # rate_limit_response::Code


def _service(make_engine, keys, limits):
    """[(namespace, max, seconds, [condition source], [variable source])] -> (engine, ingest)"""
    eng = make_engine(capacity_cells=1 << 12, max_batch_hits=1 << 10)
    g = Ingest(keys=keys)
    for ns, mx, secs, conds, variables in limits:
        assert g.add_limit(ns, mx, secs, conds, variables) >= 0
    g.install(eng)
    return eng, g


_RESP = []


def _code(resp_bytes):
    if not _RESP:
        _RESP.append(_response_class())
    m = _RESP[0]()
    m.ParseFromString(resp_bytes)
    assert not m.response_headers_to_add  # neither method adds headers
    return m.overall_code


GET_APP1 = [[("req.method", "GET"), ("app.id", "1")]]
LIMIT = lambda mx: ("test_namespace", mx, 60, ["descriptors[0]['req.method'] == 'GET'"], ["descriptors[0]['app.id']"])  # noqa: E731


@pytest.mark.parametrize("keys", ["exact", "hashed"])
def test_check_rate_limit_vectors_of_the_reference(make_engine, keys):
    # test_returns_ok_correctly (:208-266): a limit of 1 — checking twice is OK twice, a check counts nothing
    eng, g = _service(make_engine, keys, [LIMIT(1)])
    req = rls_request("test_namespace", GET_APP1, hits_addend=1)
    for _ in range(2):
        status, resp = g.serve_batch_op(eng, OP_CHECK, [req], NOW)
        assert status == [0] and _code(resp[0]) == OK
    assert eng.get_counters(0, NOW).shape[0] == 0  # nothing was created either
    # test_returns_overlimit_correctly (:268-322): max 0 — over the limit on the first check
    eng, g = _service(make_engine, keys, [LIMIT(0)])
    status, resp = g.serve_batch_op(eng, OP_CHECK, [req], NOW)
    assert status == [1] and _code(resp[0]) == OVER_LIMIT
    # test_returns_ok_when_no_limits_apply (:324-354), test_returns_unknown_when_domain_is_empty (:356-384)
    eng, g = _service(make_engine, keys, [LIMIT(0)])
    status, resp = g.serve_batch_op(eng, OP_CHECK, [rls_request("another_namespace", [[("req.method", "GET")]], hits_addend=1),
                                                    rls_request("", [[("req.method", "GET")]], hits_addend=1)], NOW)
    assert status == [0, UNKNOWN_DOMAIN] and _code(resp[0]) == OK and _code(resp[1]) == UNKNOWN


@pytest.mark.parametrize("keys", ["exact", "hashed"])
def test_check_rate_limit_takes_into_account_all_the_descriptors(make_engine, keys):
    """kuadrant_service.rs:386-468 (test_takes_into_account_all_the_descriptors): the second limit — max 0 — also asks for
    `descriptors[1].y == '2'`; the request's SECOND descriptor carries y = 2, so the check is OVER_LIMIT.  Without that
    descriptor (or with another y) only the first limit applies and the check is OK."""
    limits = [("test_namespace", 10, 60, ["descriptors[0].x == '1'"], ["descriptors[0].z"]),
              ("test_namespace", 0, 60, ["descriptors[0].x == '1'", "descriptors[1].y == '2'"], ["descriptors[0].z"])]
    eng, g = _service(make_engine, keys, limits)
    both = rls_request("test_namespace", [[("x", "1"), ("z", "1")], [("y", "2")]], hits_addend=1)
    other = rls_request("test_namespace", [[("x", "1"), ("z", "1")], [("y", "3")]], hits_addend=1)
    first_only = rls_request("test_namespace", [[("x", "1"), ("z", "1")]], hits_addend=1)
    status, resp = g.serve_batch_op(eng, OP_CHECK, [both, other, first_only], NOW)
    assert status == [1, 0, 0]
    assert [_code(r) for r in resp] == [OVER_LIMIT, OK, OK]


@pytest.mark.parametrize("keys", ["exact", "hashed"])
def test_report_vectors_of_the_reference(make_engine, keys):
    # report::test_returns_ok_correctly (:486-537): hits_addend 4 on a limit of 10
    eng, g = _service(make_engine, keys, [LIMIT(10)])
    status, resp = g.serve_batch_op(eng, OP_UPDATE, [rls_request("test_namespace", GET_APP1, hits_addend=4)], NOW)
    assert status == [0] and _code(resp[0]) == OK
    rows = eng.get_counters(0, NOW)
    assert rows.shape[0] == 1 and int(rows["value"][0]) == 4
    # what was reported is what a later check sees: 4 + 1 <= 10, and after 6 more 10 + 1 > 10
    assert g.serve_batch_op(eng, OP_CHECK, [rls_request("test_namespace", GET_APP1)], NOW)[0] == [0]
    g.serve_batch_op(eng, OP_UPDATE, [rls_request("test_namespace", GET_APP1, hits_addend=6)], NOW)
    assert g.serve_batch_op(eng, OP_CHECK, [rls_request("test_namespace", GET_APP1)], NOW)[0] == [1]
    # test_going_overlimit_is_ok (:539-590): 20 on a limit of 5 is still OK, and it is counted
    eng, g = _service(make_engine, keys, [LIMIT(5)])
    status, resp = g.serve_batch_op(eng, OP_UPDATE, [rls_request("test_namespace", GET_APP1, hits_addend=20)], NOW)
    assert status == [0] and _code(resp[0]) == OK
    assert int(eng.get_counters(0, NOW)["value"][0]) == 20
    # no limits apply (:592-622) / no domain (:624-652)
    status, resp = g.serve_batch_op(eng, OP_UPDATE, [rls_request("another_namespace", [[("req.method", "GET")]], hits_addend=1),
                                                     rls_request("", [[("req.method", "GET")]], hits_addend=1)], NOW)
    assert status == [0, UNKNOWN_DOMAIN] and _code(resp[0]) == OK and _code(resp[1]) == UNKNOWN
    assert eng.get_counters(0, NOW).shape[0] == 1


def _traffic(rng, n):
    methods, paths = ["GET", "POST", "PUT"], ["/", "/admin", "/json"]
    msgs, ctxs = [], []
    for _ in range(n):
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
        addend = int(rng.integers(0, 5)) if rng.random() < 0.7 else None  # absent or 0 means 1 (kuadrant_service.rs:139-145)
        msgs.append(rls_request(domain if domain else None, [entries], hits_addend=addend))
        ctxs.append((domain, ctx, addend if addend else 1))
    return msgs, ctxs


@pytest.mark.parametrize("keys", ["exact", "hashed"])
def test_check_then_report_against_the_rate_limiter_over_the_oracle(make_engine, keys):
    """The wasm-shim's pattern: CheckRateLimit for a batch of requests, Report for the ones that were let through (with their
    hits_addend) and — every third batch — a batch of plain ShouldRateLimit calls between them.  Status and response of every
    message against is_rate_limited(ns, ctx, 1) / update_counters(ns, ctx, addend) / check_rate_limited_and_update on the
    mirror; at the end every counter of every namespace."""
    rng = np.random.default_rng(4242 if keys == "exact" else 2424)
    eng, g, model = _install(make_engine, keys)
    n_checked = n_over = n_reported = 0
    for batch in range(24):
        msgs, ctxs = _traffic(rng, int(rng.integers(100, 300)))
        status, resp = g.serve_batch_op(eng, OP_CHECK, msgs, model.now_us)
        passed = []
        for i, (domain, ctx, addend) in enumerate(ctxs):
            if not domain:
                assert status[i] == UNKNOWN_DOMAIN and _code(resp[i]) == UNKNOWN
                continue
            want = model.is_rate_limited(domain, ctx, 1)  # (delta 1 whatever the addend: kuadrant_service.rs:62-64)
            assert status[i] == (1 if want.limited else 0), (batch, i, domain, ctx)
            assert _code(resp[i]) == (OVER_LIMIT if want.limited else OK)
            n_checked += 1
            n_over += want.limited
            if not want.limited:
                passed.append(i)
        if passed:
            status, resp = g.serve_batch_op(eng, OP_UPDATE, [msgs[i] for i in passed], model.now_us)
            for k, i in enumerate(passed):
                domain, ctx, addend = ctxs[i]
                model.update_counters(domain, ctx, addend)
                assert status[k] == 0 and _code(resp[k]) == OK
                n_reported += 1
        if batch % 3 == 2:
            msgs, ctxs = _traffic(rng, 120)
            status, resp = g.serve_batch_op(eng, OP_CHECK_AND_UPDATE, msgs, model.now_us)
            for i, (domain, ctx, addend) in enumerate(ctxs):
                if not domain:
                    assert status[i] == UNKNOWN_DOMAIN
                    continue
                want = model.check_rate_limited_and_update(domain, ctx, addend, False)
                assert status[i] == (1 if want.limited else 0), (batch, i, domain, ctx)
        model.sleep([0.0, 0.3, 1.1, 4.0][batch % 4])
    assert n_over > 50 and n_reported > 500, (n_checked, n_over, n_reported)
    # every counter of every limit: what the engine holds is what the mirror's storage holds (both sides add their limits in
    # _limits() order, so limit ids agree; rows are (value_at(now), ttl(now)), keys differ by construction: multisets)
    now = model.now_us
    live = 0
    for lid, (_ns, _mx, _secs, _conds, variables, _name) in enumerate(_limits()):
        wire = lid | (0 if variables else 0x80000000)
        got = sorted((int(r["value"]), int(r["expiry_us"])) for r in eng.get_counters(wire, now))
        want = sorted((int(r["value"]), int(r["expires_in_us"])) for r in model.storage.get_counters(wire, now))
        assert got == want, lid
        live += len(got)
    assert live > 100


def test_the_three_methods_through_one_micro_batcher(make_engine):
    """48 threads: each checks, reports what passed, and now and then calls ShouldRateLimit — on ONE tight limit of 600 per
    minute.  Reports never fail and always count; a check never counts; the final value is exactly what was reported plus
    what ShouldRateLimit admitted; batches were cut at method changes (fewer batches than requests, more than one)."""
    eng = make_engine(capacity_cells=1 << 12, max_batch_hits=1 << 12)
    g = Ingest(keys="hashed")
    assert g.add_limit("shop", 600, 60, ["descriptors[0]['method'] == 'GET'"], []) == 0
    g.install(eng)
    fe = Frontend(g, eng, max_batch=64, max_delay_us=300)
    fe.set_clock(NOW)
    req = rls_request("shop", [[("method", "GET")]], hits_addend=2)
    reported = [0] * 48
    admitted = [0] * 48
    bad = []

    def worker(t):
        for q in range(30):
            st, resp = fe.check_rate_limit(req)
            if st not in (0, 1) or _code(resp) != (OVER_LIMIT if st else OK):
                bad.append(("check", t, q, st))
            if st == 0:
                st2, resp2 = fe.report(req)
                if st2 != 0 or _code(resp2) != OK:
                    bad.append(("report", t, q, st2))
                reported[t] += 2
            if q % 7 == 3:
                st3, resp3 = fe.should_rate_limit(req)
                if st3 not in (0, 1):
                    bad.append(("should", t, q, st3))
                admitted[t] += 2 * (st3 == 0)

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(48)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    batches, requests = fe.stats()
    fe.close()
    assert not bad, bad[:5]
    rows = eng.get_counters(0 | 0x80000000, NOW)  # (a limit without variables: RL_SIMPLE)
    assert rows.shape[0] == 1 and int(rows["value"][0]) == sum(reported) + sum(admitted)
    # checks raced reports (that is the method's nature: is_rate_limited does not reserve), so the limit may be overshot by
    # the reports in flight, never by ShouldRateLimit; every check that saw value + 1 > 600 said OVER_LIMIT
    assert sum(reported) + sum(admitted) >= 600
    assert 1 < batches < requests
    g.close()


def test_alternating_check_and_report_cost_one_delay_per_window_not_one_per_request(make_engine):
    """concurrent clients in the normal Check-then-Report pattern fill the queue with C, R, C, R, ...; a
    batcher that cuts at every method change and waits max_delay_us again for the leftovers serves about one request per
    delay.  The window is now served whole — its same-method runs back to back, one wait — so 16 clients x 24 calls with a
    5 ms delay take a few dozen windows, not 384 delays (1.9 s)."""
    import time

    eng = make_engine(capacity_cells=1 << 12, max_batch_hits=1 << 12)
    g = Ingest(keys="hashed")
    assert g.add_limit("shop", 10**6, 60, [], ["descriptors[0]['user']"]) == 0
    g.install(eng)
    delay_us = 5000
    fe = Frontend(g, eng, max_batch=64, max_delay_us=delay_us)
    fe.set_clock(NOW)
    n_threads, n_calls = 16, 24
    start = threading.Barrier(n_threads)
    bad = []

    def worker(t):
        req = rls_request("shop", [[("user", f"u{t}")]], hits_addend=1)
        start.wait()
        for q in range(n_calls):
            st, resp = (fe.check_rate_limit if (q + t) % 2 == 0 else fe.report)(req)  # neighbours are in opposite phases
            if st != 0 or _code(resp) != OK:
                bad.append((t, q, st))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    t0 = time.perf_counter()
    [t.start() for t in ths]
    [t.join() for t in ths]
    elapsed = time.perf_counter() - t0
    batches, requests = fe.stats()
    windows = fe.windows()
    fe.close()
    assert not bad, bad[:5]
    assert requests == n_threads * n_calls
    assert batches > windows, "mixed windows were served as several same-method runs"
    assert windows <= requests // 3, (windows, requests)          # (a request per delay would be `requests` windows)
    assert elapsed < requests * delay_us * 1e-6 * 0.5, (elapsed, windows, batches)
    # reports counted, checks did not: every user's counter = its reports
    rows = eng.get_counters(0, NOW)
    assert int(rows["value"].sum()) == n_threads * n_calls // 2
    g.close()


def test_match_op_on_dictionary_encoded_requests(make_engine):
    """rl_match_batch_op on the numeric boundary (what a host with its own dictionaries binds): the three methods on the
    same request arrays against the mirror, incl. a request that derives no counter and a namespace without limits."""
    eng, g, model = _install(make_engine, "exact")
    rng = np.random.default_rng(99)
    for step in range(12):
        _msgs, ctxs = _traffic(rng, 200)
        ctxs = [c for c in ctxs if c[0]]
        g.batch_clear()
        for domain, ctx, addend in ctxs:
            g.batch_add(domain, list(ctx.items()), addend)
        b = g.batch()
        op = [OP_CHECK, OP_UPDATE, OP_CHECK_AND_UPDATE][step % 3]
        verdict, limited = eng.match_op(op, b["req_ns"], b["ent_off"], b["ent_key"], b["ent_val"], b["req_delta"], model.now_us)
        for i, (domain, ctx, addend) in enumerate(ctxs):
            if op == OP_CHECK:
                want = model.is_rate_limited(domain, ctx, addend)  # (this boundary checks with the request's own delta)
            elif op == OP_UPDATE:
                model.update_counters(domain, ctx, addend)
                want = None
            else:
                want = model.check_rate_limited_and_update(domain, ctx, addend, False)
            assert bool(verdict[i]) == (want.limited if want else False), (step, i, domain, ctx)
            if want is not None and want.limited:
                names = [n for (_ns, _m, _s, _c, _v, n) in _limits()]
                assert names[int(limited[i])] == want.limit_name
            else:
                assert int(limited[i]) == -1
        model.sleep(0.7)
    for lid, (_ns, _mx, _secs, _conds, variables, _name) in enumerate(_limits()):
        wire = lid | (0 if variables else 0x80000000)
        got = sorted((int(r["value"]), int(r["expiry_us"])) for r in eng.get_counters(wire, model.now_us))
        want = sorted((int(r["value"]), int(r["expires_in_us"])) for r in model.storage.get_counters(wire, model.now_us))
        assert got == want, lid
