"""On-device limit matching + key derivation (limitador_amd/csrc/rl_match.hpp) against the test-side
restatement of RateLimiter::counters_that_apply (tests/helpers/limiter.py: Limit.applies,
_counters_that_apply — the code the reference's scenarios pin), then the verdicts of the derived
counters against the CPU oracle.  Needs a MI355X."""
import numpy as np
import pytest

import oracle
from helpers.limiter import Counter, Limit
from limitador_amd.wire import HIT_DTYPE, MATCH_COND_DTYPE, MATCH_LIMIT_DTYPE, RL_SIMPLE
from test_gpu_parity import NOW, SEC, assert_same_state, make_engine  # noqa: F401

pytestmark = pytest.mark.gpu


class Dictionary:
    """Exact string -> dense id (what the ingest side keeps)."""

    def __init__(self):
        self.ids = {}

    def __call__(self, s):
        return self.ids.setdefault(s, len(self.ids))


def compile_table(eng, limits, key_id, val_id):
    """limits: list of helpers.limiter.Limit in table order (sorted by namespace)."""
    from helpers.limiter import _COND

    ns_id = Dictionary()
    rows = np.zeros(len(limits), dtype=MATCH_LIMIT_DTYPE)
    conds = []
    for i, l in enumerate(limits):
        rows[i]["limit"] = i | (0 if l.variables else RL_SIMPLE)
        rows[i]["ns"] = ns_id(l.namespace)
        rows[i]["cond_off"] = len(conds)
        rows[i]["n_cond"] = len(l.conditions)
        for c in l.conditions:
            var, op, lit = _COND.match(c).groups()
            conds.append((key_id(var), 0 if op == "==" else 1, val_id(lit)))
        rows[i]["n_vars"] = len(l.variables)
        for q, v in enumerate(l.variables):  # sorted by name already (BTreeSet)
            rows[i]["var_key"][q] = key_id(v)
    eng.set_limits([(l.max_value, l.seconds) for l in limits])
    eng.set_match_table(rows, np.array(conds, dtype=MATCH_COND_DTYPE).reshape(-1), len(ns_id.ids))
    return ns_id


def expected_counters(limits, ns, ctx):
    """lib.rs:507-522 in table order, then simple first (in_memory.rs:105,121)."""
    out = []
    for i, l in enumerate(limits):
        if l.namespace == ns and l.applies(ctx):
            out.append((i, Counter(l, tuple(sorted((v, ctx[v]) for v in l.variables)))))
    return [c for c in out if not c[1].is_qualified()] + [c for c in out if c[1].is_qualified()]


MATCHERS = {"own_scan": {}, "two_pass": {"RL_MATCH_ONE": "0", "RL_GEN_POST": "0"}, "generic": {"RL_MATCH_GENERIC": "1"}}


@pytest.mark.parametrize("matcher", list(MATCHERS))
@pytest.mark.parametrize("seed,load", [(1, False), (2, True), (3, False)])
def test_match_table_against_counters_that_apply(make_engine, seed, load, matcher, monkeypatch):
    # the slot form of the table (what limit files compile to) with its own scan and a host-mapped status word
    # (k_match_count2 / _scan2 / _fill2) or as count pass + library scan + fill pass (k_match_fast, status through copy
    # commands), and the generic kernel (k_match)
    for k, v in MATCHERS[matcher].items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(seed)
    methods, paths = ["GET", "POST", "PUT"], ["/a", "/b", "/json"]
    limits = []
    for ns in ("ns0", "ns1", "ns2"):
        for j in range(6):
            conds = []
            if rng.random() < 0.7:
                conds.append(f"m {'==' if rng.random() < 0.7 else '!='} '{methods[rng.integers(0, 3)]}'")
            if rng.random() < 0.5:
                conds.append(f"p {'==' if rng.random() < 0.5 else '!='} '{paths[rng.integers(0, 3)]}'")
            variables = [(), ("u",), ("a", "u"), ("a",)][int(rng.integers(0, 4))] if j else ()
            lim = Limit(ns, int(rng.integers(1, 40)), [1, 10, 60][int(rng.integers(0, 3))], conds, variables, name=f"{ns}-{j}")
            if lim not in limits:  # identity = (ns, seconds, conditions, variables)
                limits.append(lim)
    limits.sort(key=lambda l: l.namespace)
    eng = make_engine(capacity_cells=1 << 16, max_batch_hits=1 << 15)
    key_id, val_id = Dictionary(), Dictionary()
    ns_id = compile_table(eng, limits, key_id, val_id)
    orc = oracle.OracleStorage()
    orc.set_limits([(l.max_value, l.seconds) for l in limits])
    n_simple = 0
    for i, l in enumerate(limits):
        if not l.variables:
            eng.add_counter(i | RL_SIMPLE, eng.match_key(i))
            orc.add_counter(i | RL_SIMPLE, eng.match_key(i))
            n_simple += 1
    now = NOW
    for step in range(6):
        n_req = int(rng.integers(1, 3000))
        req_ns, ent_off, ent_key, ent_val, delta = [], [0], [], [], []
        want_hits, want_off = [], [0]
        for _ in range(n_req):
            ns = f"ns{int(rng.integers(0, 3))}" if rng.random() < 0.95 else "ns0"
            ctx = {}
            if rng.random() < 0.9:
                ctx["m"] = methods[rng.integers(0, 3)]
            if rng.random() < 0.8:
                ctx["p"] = paths[rng.integers(0, 3)]
            if rng.random() < 0.8:
                ctx["u"] = f"user{int(rng.zipf(1.5)) % 50}"
            if rng.random() < 0.6:
                ctx["a"] = f"app{int(rng.integers(0, 5))}"
            d = int(rng.integers(0, 4)) if rng.random() < 0.3 else 1
            req_ns.append(ns_id(ns))
            items = list(ctx.items())
            rng.shuffle(items)
            for k, v in items:
                ent_key.append(key_id(k))
                ent_val.append(val_id(v))
            ent_off.append(len(ent_key))
            delta.append(d)
            for i, c in expected_counters(limits, ns, ctx):
                vals = [val_id(v) for _k, v in c.set_variables]
                want_hits.append((eng.match_key(i, vals), i | (0 if c.is_qualified() else RL_SIMPLE), d))
            want_off.append(len(want_hits))
        got = eng.match_and_check(req_ns, ent_off, ent_key, ent_val, delta, now, load_counters=load)
        want = np.zeros(len(want_hits), dtype=HIT_DTYPE)
        for i, h in enumerate(want_hits):
            want[i] = h
        assert np.array_equal(got["req_off"], np.array(want_off, dtype=np.uint32)), "counters per request"
        assert np.array_equal(got["hits"], want), "derived counters (key, limit, delta)"
        v, f, r, e = orc.check_and_update(want, now, req_off=np.array(want_off, dtype=np.uint32), load_counters=load)
        assert np.array_equal(got["verdict"], v)
        lim_of = np.where(f >= 0, want["limit"][np.maximum(f, 0)] & ~np.uint32(RL_SIMPLE), -1) if len(want) else f
        assert np.array_equal(got["limited_limit"], lim_of.astype(np.int32))
        if load:
            assert np.array_equal(got["remaining"], r) and np.array_equal(got["expires_in_us"], e)
        now += int(rng.integers(0, 2 * SEC))
    assert_same_state(eng, orc, n_simple_expected=n_simple)


def test_slot_form_and_generic_matcher_agree_on_long_and_repeated_entries(make_engine, monkeypatch):
    """Requests with up to 14 entries: keys no limit mentions, keys repeated with another value (the first entry
    wins at this level; the RLS ingest has already reduced a repeated key to its last value), entries beyond the
    eight the kernels keep in registers; plus an unknown namespace id.  Differential: both kernels, same arrays."""
    rng = np.random.default_rng(77)
    keys = [f"k{i}" for i in range(6)]
    limits = []
    for ns in ("a", "b"):
        for j in range(10):
            conds = [f"{keys[int(rng.integers(0, 6))]} {'==' if rng.random() < 0.6 else '!='} 'v{int(rng.integers(0, 3))}'"
                     for _ in range(int(rng.integers(0, 3)))]
            variables = tuple(sorted(set(keys[int(rng.integers(0, 6))] for _ in range(int(rng.integers(0, 3)))))) if j else ()
            lim = Limit(ns, int(rng.integers(1, 30)), 60, conds, variables, name=None)
            if lim not in limits:
                limits.append(lim)
    limits.sort(key=lambda l: l.namespace)
    n_req = 4000
    req_ns, ent_off, ent_key, ent_val, delta = [], [0], [], [], []
    key_id, val_id = Dictionary(), Dictionary()
    for k in keys + ["junk0", "junk1", "junk2"]:
        key_id(k)
    for v in range(5):
        val_id(f"v{v}")
    for _ in range(n_req):
        req_ns.append(int(rng.integers(0, 2)))
        for _e in range(int(rng.integers(0, 15))):
            ent_key.append(int(rng.integers(0, 9)))
            ent_val.append(int(rng.integers(0, 5)))
        ent_off.append(len(ent_key))
        delta.append(1)
    results = []
    for matcher in ("generic", "two_pass", "own_scan"):
        for k in ("RL_MATCH_GENERIC", "RL_MATCH_ONE", "RL_GEN_POST"):
            monkeypatch.delenv(k, raising=False)
        for k, v in MATCHERS[matcher].items():
            monkeypatch.setenv(k, v)
        eng = make_engine(capacity_cells=1 << 16, max_batch_hits=1 << 16)
        kid, vid = Dictionary(), Dictionary()
        kid.ids, vid.ids = dict(key_id.ids), dict(val_id.ids)
        compile_table(eng, limits, kid, vid)
        assert kid.ids == key_id.ids and vid.ids == val_id.ids
        for i, l in enumerate(limits):
            if not l.variables:
                eng.add_counter(i | RL_SIMPLE, eng.match_key(i))
        got = [eng.match_and_check(req_ns, ent_off, ent_key, ent_val, delta, NOW + q * SEC) for q in range(2)]
        results.append(got)
        with pytest.raises(Exception):
            eng.match_and_check([5], [0, 0], [], [], [1], NOW)  # unknown namespace id
    for other in results[1:]:
        for a, b in zip(results[0], other):
            assert len(a["hits"]) > n_req
            for field in ("req_off", "hits", "verdict", "limited_limit"):
                assert np.array_equal(a[field], b[field]), field


def test_matcher_offsets_over_hundreds_of_workgroups(make_engine, monkeypatch):
    """300 000 requests = 1172 workgroups: the scan of the workgroups' totals takes two trips of k_match_scan2's loop,
    workgroups with no counters at all sit between busy ones, and a request's offset is its workgroup's base + a scan of
    mask popcounts.  Differential against the library-scan form on the same arrays, every output array."""
    rng = np.random.default_rng(78)
    limits = []
    for ns in ("a", "b", "c"):
        for j in range(8):
            conds = [f"m == 'v{int(rng.integers(0, 3))}'"] if j % 3 == 1 else ([f"p != 'v{int(rng.integers(0, 3))}'"] if j % 3 == 2 else [])
            variables = [(), ("u",), ("a", "u")][j % 3] if j else ()
            lim = Limit(ns, int(rng.integers(1, 2000)), 60, conds, variables, name=None)
            if lim not in limits:
                limits.append(lim)
    limits.append(Limit("d", 5, 60, ["m == 'nothing-sends-this'"], (), name=None))  # namespace d: no counters ever
    limits.sort(key=lambda l: l.namespace)
    n_req = 300_000
    key_id, val_id = Dictionary(), Dictionary()
    for k in ("m", "p", "u", "a"):
        key_id(k)
    for v in range(4000):
        val_id(f"v{v}")
    val_id("nothing-sends-this")
    # long stretches of namespace d (whole workgroups with a total of 0), the rest mixed
    ns = rng.integers(0, 3, size=n_req)
    ns[40_000:90_000] = 3
    ns[200_000:200_300] = 3
    n_ent = rng.integers(0, 5, size=n_req)
    ent_off = np.concatenate([[0], np.cumsum(n_ent)]).astype(np.uint32)
    ent_key = np.concatenate([rng.permutation(4)[:k] for k in n_ent]).astype(np.uint32)
    ent_val = np.where(ent_key < 2, rng.integers(0, 3, size=len(ent_key)), (rng.zipf(1.3, size=len(ent_key)) % 4000)).astype(np.uint32)
    delta = rng.integers(0, 3, size=n_req).astype(np.uint32)
    results = []
    for matcher in ("two_pass", "own_scan"):
        for k in ("RL_MATCH_ONE", "RL_GEN_POST"):
            monkeypatch.delenv(k, raising=False)
        for k, v in MATCHERS[matcher].items():
            monkeypatch.setenv(k, v)
        eng = make_engine(capacity_cells=1 << 21, max_batch_hits=1 << 21)
        kid, vid = Dictionary(), Dictionary()
        kid.ids, vid.ids = dict(key_id.ids), dict(val_id.ids)
        compile_table(eng, limits, kid, vid)
        for i, l in enumerate(limits):
            if not l.variables:
                eng.add_counter(i | RL_SIMPLE, eng.match_key(i))
        results.append([eng.match_and_check(ns.astype(np.uint32), ent_off, ent_key, ent_val, delta, NOW + q * SEC, load_counters=(q == 1))
                        for q in range(3)])
        # a batch that expands to more counters than the engine stages is refused, and the next call is served
        small = make_engine(capacity_cells=1 << 16, max_batch_hits=4096)
        three = [Limit("a", 5, sec, [], (), name=None) for sec in (1, 10, 60)]
        compile_table(small, three, Dictionary(), Dictionary())
        for i in range(3):
            small.add_counter(i | RL_SIMPLE, small.match_key(i))
        zeros = np.zeros(4001, dtype=np.uint32)
        with pytest.raises(Exception):  # 4000 requests x 3 counters
            small.match_and_check(zeros[:4000], zeros, [], [], np.ones(4000, dtype=np.uint32), NOW)
        ok = small.match_and_check(zeros[:1000], zeros[:1001], [], [], np.ones(1000, dtype=np.uint32), NOW)
        assert len(ok["hits"]) == 3000 and int(ok["verdict"].sum()) == 1000 - 5
    for a, b in zip(*results):
        assert len(a["hits"]) > n_req
        for field in ("req_off", "hits", "verdict", "limited_limit"):
            assert np.array_equal(a[field], b[field]), field
    assert np.array_equal(results[0][1]["remaining"], results[1][1]["remaining"])
    assert np.array_equal(results[0][1]["expires_in_us"], results[1][1]["expires_in_us"])


def test_match_table_rejects_what_must_stay_on_the_host(make_engine):
    from limitador_amd.engine import EngineError

    eng = make_engine()
    eng.set_limits([(5, 60)])
    rows = np.zeros(1, dtype=MATCH_LIMIT_DTYPE)
    rows[0]["limit"] = 0  # no variables but RL_SIMPLE missing
    with pytest.raises(EngineError) as e:
        eng.set_match_table(rows, np.zeros(0, dtype=MATCH_COND_DTYPE), 1)
    assert e.value.code == -1
    rows[0]["limit"], rows[0]["n_vars"] = 0, 3  # three variables
    with pytest.raises(EngineError):
        eng.set_match_table(rows, np.zeros(0, dtype=MATCH_COND_DTYPE), 1)
    rows[0]["n_vars"], rows[0]["var_key"] = 1, (0, 0)
    eng.set_match_table(rows, np.zeros(0, dtype=MATCH_COND_DTYPE), 1)
    # a value id beyond 26 bits cannot be packed into an exact key
    with pytest.raises(EngineError) as e:
        eng.match_and_check([0], [0, 1], [0], [1 << 26], [1], NOW)
    assert e.value.code == -1


# The reference's own vectors for Limit::applies (limitador/src/limit.rs:239-348), through the device.
APPLIES_VECTORS = [
    # (conditions, variables, context, applies)                                      limit.rs
    (["x == '5'"], ["y"], {"x": "5", "y": "1"}, True),                              # :239-254 limit_applies
    (["x == '5'"], ["y"], {"x": "1", "y": "1"}, False),                             # :256-271 cond is false
    (["x == '5'"], ["y"], {"a": "1", "y": "1"}, False),                             # :273-289 cond var not set
    (["x == '5'"], ["y"], {"x": "5"}, False),                                       # :291-306 var not set
    (["x == '5'", "y == '2'"], ["z"], {"x": "5", "y": "2", "z": "1"}, True),        # :308-327 all conditions
    (["x == '5'", "y == '2'"], ["z"], {"x": "3", "y": "2", "z": "1"}, False),       # :329-348 one does not
]


@pytest.mark.parametrize("conds,variables,ctx,applies", APPLIES_VECTORS)
def test_reference_applies_vectors_on_device(make_engine, conds, variables, ctx, applies):
    limit = Limit("test_namespace", 10, 60, conds, variables)
    assert limit.applies(ctx) == applies  # the restatement the random test compares against
    eng = make_engine()
    key_id, val_id = Dictionary(), Dictionary()
    ns_id = compile_table(eng, [limit], key_id, val_id)
    items = list(ctx.items())
    got = eng.match_and_check([ns_id("test_namespace")], [0, len(items)], [key_id(k) for k, _ in items],
                              [val_id(v) for _, v in items], [1], NOW)
    assert len(got["hits"]) == (1 if applies else 0)
    assert got["verdict"][0] == 0 and got["limited_limit"][0] == -1
    if applies:
        vals = [val_id(ctx[v]) for v in sorted(variables)]
        assert int(got["hits"][0]["key"]) == eng.match_key(0, vals)


def test_cpp_ingest_drives_the_device_matcher(make_engine):
    """include/rl_ingest.h end to end: limits given as strings are compiled and installed by the C++
    ingest, requests are encoded from strings, rl_match_and_check_batch decides — against the
    string-level counters_that_apply + the oracle."""
    from helpers.match_cpu import limited_limit, match_key, match_requests, random_limits
    from limitador_amd.ingest import Ingest

    rng = np.random.default_rng(41)
    namespaces = ["ns0", "ns1", "ns2"]
    limits = random_limits(rng, namespaces)
    g = Ingest(binding="root")  # the helpers' limits and contexts are the library's: bare identifiers, a HashMap
    for l in limits:
        assert g.add_limit(l.namespace, l.max_value, l.seconds, list(l.conditions), list(l.variables)) >= 0
    t = g.compile()
    eng = make_engine(capacity_cells=1 << 16, max_batch_hits=1 << 15)
    g.install(eng)
    orc = oracle.OracleStorage()
    orc.set_limits([(l.max_value, l.seconds) for l in limits])
    n_simple = 0
    for i, l in enumerate(limits):
        if not l.variables:
            orc.add_counter(i | RL_SIMPLE, match_key(i, []))
            n_simple += 1
    methods, paths = ["GET", "POST", "PUT"], ["/a", "/b", "/json"]
    now = NOW
    for step in range(4):
        g.batch_clear()
        for _ in range(int(rng.integers(1, 2500))):
            ns = namespaces[int(rng.integers(0, 3))] if rng.random() < 0.95 else "nobody"
            ctx = {}
            if rng.random() < 0.9:
                ctx["m"] = methods[rng.integers(0, 3)]
            if rng.random() < 0.8:
                ctx["p"] = paths[rng.integers(0, 3)]
            if rng.random() < 0.8:
                ctx["u"] = f"user{int(rng.zipf(1.5)) % 50}"
            if rng.random() < 0.6:
                ctx["a"] = f"app{int(rng.integers(0, 5))}"
            g.batch_add(ns, list(ctx.items()), int(rng.integers(0, 3)))
        b = g.batch()
        verdict, limited = g.check(eng, now)
        hits, off = match_requests(t["limits"], t["conds"], b["req_ns"], b["ent_off"], b["ent_key"], b["ent_val"],
                                   b["req_delta"])
        wv, wf, _r, _e = orc.check_and_update(hits, now, req_off=off)
        assert np.array_equal(verdict, wv), f"step {step}"
        assert np.array_equal(limited, limited_limit(wf, hits)), f"step {step}"
        now += int(rng.integers(0, 2 * SEC))
    assert_same_state(eng, orc, n_simple_expected=n_simple)
    g.close()
