"""The id-level CPU matcher (tests/helpers/match_cpu.py) against the string-level restatement of
Limit::applies / counters_that_apply (tests/helpers/limiter.py, pinned by the reference's vectors
limit.rs:239-348).  It is the stand-in of the device matcher wherever there is no GPU."""
import numpy as np
import pytest

from helpers.limiter import Counter
from helpers.match_cpu import Dictionary, compile_rows, match_key, match_requests, random_limits, random_requests
from limitador_amd.wire import RL_SIMPLE


def expected(limits, ns, ctx, val_id, delta):
    out = []
    for i, l in enumerate(limits):
        if l.namespace == ns and l.applies(ctx):
            c = Counter(l, tuple(sorted((v, ctx[v]) for v in l.variables)))
            vals = [val_id(v) for _k, v in c.set_variables]
            out.append((c.is_qualified(), (match_key(i, vals), i | (0 if c.is_qualified() else RL_SIMPLE), delta)))
    return [h for q, h in out if not q] + [h for q, h in out if q]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_id_level_matcher_equals_string_level_counters_that_apply(seed):
    rng = np.random.default_rng(seed)
    namespaces = ["ns0", "ns1", "ns2", "ns3"]
    limits = random_limits(rng, namespaces)
    key_id, val_id = Dictionary(), Dictionary()
    rows, conds, ns_id = compile_rows(limits, key_id, val_id)
    ctxs, req_ns, ent_off, ent_key, ent_val, delta = random_requests(rng, 800, namespaces, ns_id, key_id, val_id)
    hits, off = match_requests(rows, conds, req_ns, ent_off, ent_key, ent_val, delta)
    for r, (ns, ctx) in enumerate(ctxs):
        want = expected(limits, ns, ctx, val_id, int(delta[r]))
        got = [(int(h["key"]), int(h["limit"]), int(h["delta"])) for h in hits[off[r]:off[r + 1]]]
        assert got == want, f"request {r}: {ns} {ctx}"
    assert off[-1] == len(hits) and len(hits) > 800  # multi-counter requests


def test_match_key_is_the_engines(engine_lib):
    for lid, vals in [(0, ()), (7, (3,)), (4094, (2**26 - 1, 5)), (12, (0, 2**26 - 1))]:
        v = list(vals) + [0, 0]
        assert engine_lib.rl_match_key(lid, len(vals), v[0], v[1]) == match_key(lid, list(vals))
