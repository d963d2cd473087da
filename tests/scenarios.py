"""The reference's own test vectors for the in-memory counter path, restated
against the TestsLimiter facade.  Each function cites the reference test it replays; assertions
are the reference's assertions.  Run against the CPU oracle (tests/test_oracle_golden.py — this
is what pins the oracle) and against the HIP engine (tests/test_gpu_scenarios.py).

Where the reference sleeps on the wall clock the scenario advances the facade's explicit clock.
"""
from helpers.limiter import Limit

NS = "test_namespace"
GET = "req_method == 'GET'"
POST = "req_method == 'POST'"


def _limit(max_value, seconds=60, conditions=(GET,), variables=("app_id",), ns=NS):
    return Limit(ns, max_value, seconds, conditions, variables)


def _ctx(method="GET", app_id="test_app_id", **extra):
    ctx = {"req_method": method, "app_id": app_id}
    ctx.update(extra)
    return ctx


# limitador/tests/integration_tests.rs:367-394
def delete_limit_also_deletes_associated_counters(rl):
    limit = _limit(10)
    rl.add_limit(limit)
    rl.update_counters(NS, _ctx(app_id="1"), 1)
    rl.delete_limit(limit)
    assert rl.get_counters(NS) == []


# :460-491
def delete_limits_of_a_namespace_also_deletes_counters(rl):
    rl.add_limit(Limit(NS, 5, 60, ["req_method == 'GET'"], ["app_id"]))
    rl.update_counters(NS, {"req_method": "GET", "app_id": "1"}, 1)
    rl.delete_limits(NS)
    assert rl.get_counters(NS) == []


# :493-496
def delete_limits_of_an_empty_namespace_does_nothing(rl):
    rl.delete_limits(NS)


# :493-532
def rate_limited(rl):
    max_hits = 3
    rl.add_limit(_limit(max_hits))
    ctx = _ctx()
    for i in range(max_hits):
        assert not rl.is_rate_limited(NS, ctx, 1).limited, f"Must not be limited after {i}"
        rl.update_counters(NS, ctx, 1)
    assert rl.is_rate_limited(NS, ctx, 1).limited


# :534-574 (identical flow, limit carries an id — id is not part of identity)
def rate_limited_id_counter(rl):
    max_hits = 3
    limit = _limit(max_hits)
    limit.id = "test-rate_limited_id_counter"
    rl.add_limit(limit)
    ctx = _ctx()
    for i in range(max_hits):
        assert not rl.is_rate_limited(NS, ctx, 1).limited
        rl.update_counters(NS, ctx, 1)
    assert rl.is_rate_limited(NS, ctx, 1).limited


# :576-654
def multiple_limits_rate_limited(rl):
    max_hits = 3
    rl.add_limit(_limit(max_hits, conditions=(GET,)))
    rl.add_limit(_limit(max_hits + 1, conditions=(POST,)))
    get_ctx, post_ctx = _ctx("GET"), _ctx("POST")
    for i in range(max_hits):
        assert not rl.is_rate_limited(NS, get_ctx, 1).limited
        assert not rl.is_rate_limited(NS, post_ctx, 1).limited
        rl.check_rate_limited_and_update(NS, get_ctx, 1, False)
        rl.check_rate_limited_and_update(NS, post_ctx, 1, False)
    assert rl.is_rate_limited(NS, get_ctx, 1).limited
    assert not rl.is_rate_limited(NS, post_ctx, 1).limited


# :656-695
def rate_limited_with_delta_higher_than_one(rl):
    rl.add_limit(_limit(10))
    ctx = _ctx()
    for _ in range(2):
        assert not rl.is_rate_limited(NS, ctx, 5).limited
        rl.update_counters(NS, ctx, 5)
    assert rl.is_rate_limited(NS, ctx, 1).limited


# :697-722
def rate_limited_with_delta_higher_than_max(rl):
    rl.add_limit(_limit(10))
    assert rl.is_rate_limited(NS, _ctx(), 11).limited


# :724-769
def takes_into_account_only_vars_of_the_limits(rl):
    max_hits = 3
    rl.add_limit(_limit(max_hits))
    for i in range(max_hits):
        ctx = _ctx(does_not_apply=str(i))
        assert not rl.is_rate_limited(NS, ctx, 1).limited
        rl.update_counters(NS, ctx, 1)
    assert rl.is_rate_limited(NS, _ctx(), 1).limited


# :771-785
def is_rate_limited_returns_false_when_no_limits_in_namespace(rl):
    assert not rl.is_rate_limited(NS, {"req_method": "GET"}, 1).limited


# :787-815
def is_rate_limited_returns_false_when_no_matching_limits(rl):
    rl.add_limit(_limit(0))
    assert not rl.is_rate_limited(NS, _ctx("POST"), 1).limited


# :817-841
def is_rate_limited_applies_limit_if_its_unconditional(rl):
    rl.add_limit(_limit(0, conditions=()))
    assert rl.is_rate_limited(NS, {"app_id": "test_app_id"}, 1).limited


# :843-879 — THE path test
def check_rate_limited_and_update(rl):
    max_hits = 3
    rl.add_limit(_limit(max_hits))
    ctx = _ctx()
    for _ in range(max_hits):
        assert not rl.check_rate_limited_and_update(NS, ctx, 1, False).limited
    assert rl.check_rate_limited_and_update(NS, ctx, 1, False).limited


# :881-929
def check_rate_limited_and_update_load_counters(rl):
    max_hits = 3
    rl.add_limit(_limit(max_hits))
    ctx = _ctx()
    for hit in range(max_hits):
        result = rl.check_rate_limited_and_update(NS, ctx, 1, True)
        assert not result.limited
        assert len(result.counters) == 1
        for counter in result.counters:
            assert counter.expires_in_us // 1_000_000 <= 60
            assert counter.remaining == 3 - (hit + 1)
    result = rl.check_rate_limited_and_update(NS, ctx, 1, True)
    assert result.limited
    assert len(result.counters) == 1
    for counter in result.counters:
        assert counter.expires_in_us // 1_000_000 <= 60
        assert counter.remaining == 0


# :931-959
def check_rate_limited_and_update_returns_true_if_no_limits_apply(rl):
    rl.add_limit(_limit(10))
    assert not rl.check_rate_limited_and_update(NS, _ctx("POST"), 1, False).limited


# :961-987
def check_rate_limited_and_update_applies_limit_if_its_unconditional(rl):
    rl.add_limit(_limit(0, conditions=()))
    assert rl.check_rate_limited_and_update(NS, {"app_id": "test_app_id"}, 1, False).limited


# :989-1039
def get_counters(rl):
    max_hits, hits_app_1, hits_app_2 = 10, 1, 5
    rl.add_limit(_limit(max_hits))
    rl.update_counters(NS, _ctx(app_id="1"), hits_app_1)
    rl.update_counters(NS, _ctx(app_id="2"), hits_app_2)
    assert len(rl.get_limits(NS)) == 1
    counters = rl.get_counters(NS)
    assert len(counters) == 2
    for counter in counters:
        app_id = dict(counter.set_variables)["app_id"]
        if app_id == "1":
            assert counter.remaining == max_hits - hits_app_1
        elif app_id == "2":
            assert counter.remaining == max_hits - hits_app_2
        else:
            raise AssertionError("Unexpected app ID")


# :1041-1049
def get_counters_returns_empty_when_no_limits_in_namespace(rl):
    assert rl.get_counters(NS) == []


# :1051-1071
def get_counters_returns_empty_when_no_counters_in_namespace(rl):
    rl.add_limit(_limit(10))
    assert rl.get_counters(NS) == []


# :1073-1100 (the reference sleeps limit_time + 1 seconds)
def get_counters_does_not_return_expired_ones(rl):
    limit_time = 1
    rl.add_limit(_limit(10, seconds=limit_time))
    rl.update_counters(NS, _ctx(app_id="1"), 1)
    rl.sleep(limit_time + 1)
    assert len(rl.get_counters(NS)) == 0


# :1135-1177
def configure_with_keeps_the_given_limits_and_counters_if_they_exist(rl):
    max_value, hits_to_report = 10, 1
    limit = _limit(max_value)
    rl.add_limit(limit)
    rl.update_counters(NS, _ctx(app_id="1"), hits_to_report)
    rl.configure_with([limit.clone()])
    assert limit in rl.get_limits(NS)
    counters = rl.get_counters(NS)
    assert len(counters) == 1
    assert counters[0].remaining == max_value - hits_to_report


# :1179-1211
def configure_with_deletes_all_except_the_limits_given(rl):
    keep = _limit(10, seconds=1)
    drop = _limit(20, seconds=60)
    rl.add_limit(keep)
    rl.add_limit(drop)
    rl.configure_with([keep.clone()])
    limits = rl.get_limits(NS)
    assert keep in limits and drop not in limits


# :1213-1243
def configure_with_updates_the_limits(rl):
    rl.add_limit(_limit(10))
    rl.configure_with([_limit(20)])
    limits = rl.get_limits(NS)
    assert len(limits) == 1
    assert next(iter(limits)).max_value == 20


# :1245-1283 — max_value and name are not identity
def add_limit_only_adds_if_not_present(rl):
    l1, l2, l3 = _limit(10), _limit(20), _limit(20)
    l3.name = "Name is irrelevant too"
    assert rl.add_limit(l1)
    assert not rl.add_limit(l2)
    assert not rl.add_limit(l3)
    limits = rl.get_limits(NS)
    assert len(limits) == 1
    known = next(iter(limits))
    assert known.max_value == 10 and known.name is None


# limitador/src/lib.rs:760-790 — max 42 -> 50 is seen by the next check, counter untouched
def properly_updates_existing_limits(rl):
    ns = "foo"
    l = Limit(ns, 42, 100, (), ())
    rl.add_limit(l)
    assert next(iter(rl.get_limits(ns))).max_value == 42
    r = rl.check_rate_limited_and_update(ns, {}, 1, True)
    assert r.counters[0].max_value() == 42
    assert r.counters[0].remaining == 41
    l2 = l.clone()
    l2.max_value = 50
    rl.configure_with([l2])
    limits = rl.get_limits(ns)
    assert len(limits) == 1 and next(iter(limits)).max_value == 50
    r = rl.check_rate_limited_and_update(ns, {}, 1, True)
    assert r.counters[0].max_value() == 50
    assert r.counters[0].remaining == 48  # one earlier hit is still counted


# limitador/src/lib.rs:792-817 — delete + re-add starts from a fresh cell
def deletes_qualified_counters(rl):
    ns = "foo"
    l = Limit(ns, 42, 100, (), ("x",))
    ctx = {"x": "a"}
    rl.add_limit(l)
    r = rl.check_rate_limited_and_update(ns, ctx, 1, True)
    assert r.counters[0].remaining == 41
    rl.delete_limit(l)
    rl.add_limit(l)
    r = rl.check_rate_limited_and_update(ns, ctx, 1, True)
    assert r.counters[0].remaining == 41


# limitador/src/storage/in_memory.rs:277-310 — two limits differing only in `seconds`
def counters_for_multiple_limit_per_ns(rl):
    l1 = Limit(NS, 1, 1, (GET,), ("app_id",))
    l2 = Limit(NS, 1, 10, (GET,), ("app_id",))
    rl.add_limit(l1)
    rl.add_limit(l2)
    rl.update_counters(NS, _ctx(app_id="foo"), 1)
    assert len(rl.get_counters(NS)) == 2


# limitador-server/src/envoy_rls/server.rs:337-431 — exact header strings, limit of 1
def envoy_headers_limit_of_one(rl):
    ns = "test_namespace"
    rl.add_limit(Limit(ns, 1, 60, ("req.method == 'GET'",), ("app.id",)))
    ctx = {"req.method": "GET", "app.id": "1"}
    r = rl.check_rate_limited_and_update(ns, ctx, 1, True)
    assert not r.limited
    h = r.response_header()
    assert h["X-RateLimit-Limit"] == "1, 1;w=60"  # server.rs:390-393
    assert h["X-RateLimit-Remaining"] == "0"  # server.rs:394-397
    assert int(h["X-RateLimit-Reset"]) <= 60
    r = rl.check_rate_limited_and_update(ns, ctx, 1, True)
    assert r.limited
    h = r.response_header()
    assert h["X-RateLimit-Limit"] == "1, 1;w=60"
    assert h["X-RateLimit-Remaining"] == "0"


# limitador-server/src/envoy_rls/server.rs:586-681 — hits_addend 6 of 10, then over limit
def envoy_hits_addend(rl):
    ns = "test_namespace"
    rl.add_limit(Limit(ns, 10, 60, ("req.method == 'GET'",), ("app.id",)))
    ctx = {"req.method": "GET", "app.id": "1"}
    r = rl.check_rate_limited_and_update(ns, ctx, 6, True)
    assert not r.limited
    assert r.response_header()["X-RateLimit-Remaining"] == "4"  # server.rs:644-651
    r = rl.check_rate_limited_and_update(ns, ctx, 6, True)
    assert r.limited
    assert r.response_header()["X-RateLimit-Remaining"] == "0"  # server.rs:665-672


# limitador-server/src/envoy_rls/server.rs:520-584 — two limits, both reported, most restrictive first
def envoy_headers_two_limits(rl):
    ns = "test_namespace"
    rl.add_limit(Limit(ns, 10, 60, ("x == '1'",), ("z",)))
    rl.add_limit(Limit(ns, 0, 60, ("x == '1'", "y == '2'"), ("z",)))
    r = rl.check_rate_limited_and_update(ns, {"x": "1", "y": "2", "z": "1"}, 1, True)
    assert r.limited
    h = r.response_header()
    assert h["X-RateLimit-Limit"] == "0, 0;w=60, 10;w=60"  # server.rs:576-579
    assert h["X-RateLimit-Remaining"] == "0"


ALL = [v for k, v in sorted(globals().items()) if callable(v) and getattr(v, "__module__", "") == __name__
       and not k.startswith("_")]
