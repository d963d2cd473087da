"""Test-side CPU restatement of the on-device matcher over dictionary-encoded requests
(limitador_amd/csrc/rl_match.hpp), i.e. of RateLimiter::counters_that_apply (limitador/src/lib.rs:507-522)
= Limit::applies (limit.rs:157-174) + Counter::new / resolve_variables (counter.rs:19-31,
limit.rs:133-148) for `==` / `!=` conditions and plain variables, plus the compile step from
helpers.limiter.Limit objects to the match table.  Checked against the string-level
helpers.limiter.Limit.applies in tests/test_match_cpu.py; used as the owner-side stand-in of the
namespace-sharded path on CPU (tests/test_sharded_requests_gloo.py)."""
import numpy as np

from helpers.limiter import _COND
from limitador_amd.wire import HIT_DTYPE, MATCH_COND_DTYPE, MATCH_LIMIT_DTYPE, RL_SIMPLE

VAL_BITS = 26


class Dictionary:
    """Exact string -> dense id (what the ingest side keeps)."""

    def __init__(self):
        self.ids = {}

    def __call__(self, s):
        return self.ids.setdefault(s, len(self.ids))


def compile_rows(limits, key_id, val_id, ns_id=None):
    """limits: helpers.limiter.Limit list sorted by namespace -> (rows, conds, ns dictionary)."""
    ns_id = ns_id or Dictionary()
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
    return rows, np.array(conds, dtype=MATCH_COND_DTYPE).reshape(-1), ns_id


def match_key(limit_id, values):
    """rl_match_key: limit id + 1 in bits 52.., value ids in bits 0..25 and 26..51."""
    v = list(values) + [0, 0]
    return ((limit_id + 1) << (2 * VAL_BITS)) | ((v[1] if len(values) > 1 else 0) << VAL_BITS) | (v[0] if values else 0)


def match_requests(rows, conds, req_ns, ent_off, ent_key, ent_val, req_delta):
    """-> (hits HIT_DTYPE[], req_off uint32[n+1]): every request's counters in table order, simple first."""
    by_ns = {}
    for i in range(len(rows)):
        by_ns.setdefault(int(rows[i]["ns"]), []).append(i)
    hits, off = [], [0]
    for r in range(len(req_ns)):
        ctx = {}
        for q in range(int(ent_off[r]), int(ent_off[r + 1])):
            ctx.setdefault(int(ent_key[q]), int(ent_val[q]))  # the first entry of a key wins
        simple, qualified = [], []
        for i in by_ns.get(int(req_ns[r]), ()):
            L = rows[i]
            ok = True
            for c in range(int(L["cond_off"]), int(L["cond_off"]) + int(L["n_cond"])):
                v = ctx.get(int(conds[c]["key"]))
                # a condition on a key the request does not carry is false for == and != alike (cel.rs:321-338)
                if v is None or (v == int(conds[c]["value"])) != (int(conds[c]["op"]) == 0):
                    ok = False
                    break
            if not ok:
                continue
            vals = [ctx.get(int(L["var_key"][q])) for q in range(int(L["n_vars"]))]
            if any(v is None for v in vals):  # counter.rs:22-24: no counter without its variables
                continue
            lid = int(L["limit"]) & ~RL_SIMPLE
            (qualified if vals else simple).append((match_key(lid, vals), int(L["limit"]), int(req_delta[r])))
        hits.extend(simple + qualified)
        off.append(len(hits))
    out = np.zeros(len(hits), dtype=HIT_DTYPE)
    for i, h in enumerate(hits):
        out[i] = h
    return out, np.array(off, dtype=np.uint32)


def limited_limit(first_limited, hits):
    """first_limited (index into hits, -1) -> limit id the reference names (Authorization::Limited)."""
    if not len(hits):
        return np.full(len(first_limited), -1, dtype=np.int32)
    f = np.asarray(first_limited)
    return np.where(f >= 0, hits["limit"][np.maximum(f, 0)] & ~np.uint32(RL_SIMPLE), -1).astype(np.int32)


def random_limits(rng, namespaces, per_ns=6):
    from helpers.limiter import Limit

    methods, paths = ["GET", "POST", "PUT"], ["/a", "/b", "/json"]
    limits = []
    for ns in namespaces:
        for j in range(per_ns):
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
    return limits


def random_requests(rng, n_req, namespaces, ns_id, key_id, val_id):
    """-> (contexts as dicts of strings, req_ns, ent_off, ent_key, ent_val, delta) for n_req requests."""
    methods, paths = ["GET", "POST", "PUT"], ["/a", "/b", "/json"]
    ctxs, req_ns, ent_off, ent_key, ent_val, delta = [], [], [0], [], [], []
    for _ in range(n_req):
        ns = namespaces[int(rng.integers(0, len(namespaces)))]
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
        for k, v in items:
            ent_key.append(key_id(k))
            ent_val.append(val_id(v))
        ent_off.append(len(ent_key))
        req_ns.append(ns_id(ns))
        delta.append(int(rng.integers(0, 4)) if rng.random() < 0.3 else 1)
        ctxs.append((ns, ctx))
    a = lambda x: np.array(x, dtype=np.uint32)  # noqa: E731
    return ctxs, a(req_ns), a(ent_off), a(ent_key), a(ent_val), a(delta)
