"""Test-side mirror of the reference's model + facade, so the parity tests read like
limitador/tests/integration_tests.rs.

  Limit / Counter identity ........ limitador/src/limit.rs:177-214, counter.rs:123-138
  TestsLimiter (RateLimiter) ...... limitador/src/lib.rs:323-523, tests/helpers/tests_limiter.rs
  CheckResult.response_header ..... limitador/src/lib.rs:235-275

It owns everything UPSTREAM of the CounterStorage boundary (matching limits, resolving variables,
interning identities to ids) and drives a storage backend through the engine's wire format.  The
backend is either oracle.OracleStorage (CPU restatement) or limitador_amd.engine.Engine (HIP).
CEL is out of scope: conditions support only `var == 'lit'` / `var != 'lit'`, variables are
plain names — the shapes the reference's storage-level tests use.
"""
import re
from dataclasses import dataclass, field

import numpy as np

from limitador_amd.wire import HIT_DTYPE, RL_SIMPLE

_COND = re.compile(r"^\s*([A-Za-z_][\w.]*)\s*(==|!=)\s*'([^']*)'\s*$")


class Limit:
    def __init__(self, namespace, max_value, seconds, conditions=(), variables=(), name=None, id=None):
        self.namespace = namespace
        self.max_value = int(max_value)
        self.seconds = int(seconds)
        self.conditions = tuple(sorted(set(conditions)))  # BTreeSet<Predicate>
        self.variables = tuple(sorted(set(variables)))  # BTreeSet<Expression>
        self.name = name
        self.id = id

    # limit.rs:177-214 — identity excludes max_value, name and id
    def identity(self):
        return (self.namespace, self.seconds, self.conditions, self.variables)

    def __eq__(self, other):
        return isinstance(other, Limit) and self.identity() == other.identity()

    def __hash__(self):
        return hash(self.identity())

    def clone(self):
        return Limit(self.namespace, self.max_value, self.seconds, self.conditions, self.variables, self.name, self.id)

    # limit.rs:157-174
    def applies(self, ctx):
        for cond in self.conditions:
            m = _COND.match(cond)
            if not m:
                raise ValueError(f"test helper supports only == / != conditions, got {cond!r}")
            var, op, lit = m.groups()
            val = ctx.get(var)
            if op == "==" and val != lit:
                return False
            if op == "!=" and (val is None or val == lit):
                return False
        return all(v in ctx for v in self.variables)


@dataclass
class Counter:
    limit: Limit
    set_variables: tuple  # sorted (name, value) pairs — BTreeMap<String,String>
    remaining: int = None
    expires_in_us: int = None

    def is_qualified(self):  # counter.rs:108-110
        return len(self.set_variables) > 0

    def max_value(self):
        return self.limit.max_value

    def window_secs(self):
        return self.limit.seconds


@dataclass
class CheckResult:
    limited: bool
    counters: list = field(default_factory=list)
    limit_name: str = None

    # lib.rs:235-275
    def response_header(self):
        headers = {}
        self.counters.sort(key=lambda c: c.remaining if c.remaining is not None else c.max_value())
        all_limits = ""
        for c in self.counters:
            all_limits += f", {c.max_value()};w={c.window_secs()}"
            if c.limit.name is not None:
                all_limits += ';name="{}"'.format(c.limit.name.replace('"', "'"))
        if self.counters:
            c = self.counters[0]
            remaining = c.remaining if c.remaining is not None else c.max_value()
            headers["X-RateLimit-Limit"] = f"{c.max_value()}{all_limits}"
            headers["X-RateLimit-Remaining"] = f"{remaining}"
            if c.expires_in_us is not None:
                headers["X-RateLimit-Reset"] = f"{c.expires_in_us // 1_000_000}"
        return headers


class TestsLimiter:
    """RateLimiter over a wire-format storage backend, with an explicit clock."""

    __test__ = False

    def __init__(self, storage, now_us=1_700_000_000_000_000):
        self.storage = storage
        self.now_us = now_us
        self.limits = {}  # namespace -> {identity: Limit}   (Storage.limits, storage/mod.rs:31-34)
        self._limit_ids = {}  # identity -> dense id
        self._rows = []  # per id (max_value, seconds)
        self._counter_keys = {}  # (limit id, set_variables) -> exact u64 key
        self._key_info = {}  # key -> (identity, set_variables)

    # -- clock -------------------------------------------------------------------------------
    def sleep(self, seconds):
        self.now_us += int(seconds * 1_000_000)

    # -- interning ---------------------------------------------------------------------------
    def _limit_id(self, limit):
        ident = limit.identity()
        lid = self._limit_ids.get(ident)
        if lid is None:
            lid = len(self._rows)
            self._limit_ids[ident] = lid
            self._rows.append((limit.max_value, limit.seconds))
            self.storage.set_limits([(limit.max_value, limit.seconds)], first=lid)
        return lid

    def _sync_row(self, limit):
        lid = self._limit_id(limit)
        if self._rows[lid] != (limit.max_value, limit.seconds):
            self._rows[lid] = (limit.max_value, limit.seconds)
            self.storage.set_limits([self._rows[lid]], first=lid)
        return lid

    def _wire_limit(self, limit):
        lid = self._limit_id(limit)
        return lid | (0 if limit.variables else RL_SIMPLE)

    def _key(self, limit, set_variables):
        k = (self._limit_id(limit), set_variables)
        key = self._counter_keys.get(k)
        if key is None:
            # exact, collision-free: a dense sequence number, scrambled so slots spread out
            key = (len(self._counter_keys) + 1) * 0x9E3779B97F4A7C15 % (1 << 63)
            self._counter_keys[k] = key
            self._key_info[key] = (limit.identity(), set_variables)
        return key

    # -- Storage facade (storage/mod.rs:60-152) -------------------------------------------------
    def add_limit(self, limit):
        ns = self.limits.setdefault(limit.namespace, {})
        self._limit_id(limit)
        # add_counter: pre-creates the cell of a limit without variables (in_memory.rs:38-44)
        self.storage.add_counter(self._wire_limit(limit), self._key(limit, ()) if not limit.variables else 0)
        if limit.identity() in ns:
            return False
        ns[limit.identity()] = limit.clone()
        self._sync_row(limit)
        return True

    def update_limit(self, update):  # storage/mod.rs:67-83
        ns = self.limits.get(update.namespace)
        if ns and update.identity() in ns:
            cur = ns[update.identity()]
            if cur.max_value != update.max_value or cur.name != update.name:
                ns[update.identity()] = update.clone()
                self._sync_row(update)
                return True
        return False

    def get_limits(self, namespace):
        return set(self.limits.get(namespace, {}).values())

    def delete_limit(self, limit):  # storage/mod.rs:94-117
        self.storage.delete_counters(self._wire_limit(limit))
        ns = self.limits.get(limit.namespace)
        if ns is not None:
            ns.pop(limit.identity(), None)
            if not ns:
                del self.limits[limit.namespace]

    def delete_limits(self, namespace):
        for limit in list(self.limits.pop(namespace, {}).values()):
            self.storage.delete_counters(self._wire_limit(limit))

    def configure_with(self, limits):  # lib.rs:475-505
        keep = {}
        for l in limits:
            keep.setdefault(l.namespace, {})[l.identity()] = l
        for ns in set(self.limits) | set(keep):
            have = dict(self.limits.get(ns, {}))
            want = keep.get(ns, {})
            for ident, l in have.items():
                if ident not in want:
                    self.delete_limit(l)
            for ident, l in want.items():
                if ident not in have:
                    self.add_limit(l)
            # HashSet::union yields the keep-set's element for identities present in both
            for ident in list(want) + [i for i in have if i not in want]:
                self.update_limit(want.get(ident) or have[ident])

    # -- counters_that_apply (lib.rs:507-522) ----------------------------------------------------
    def _counters_that_apply(self, namespace, ctx):
        out = []
        for limit in self.limits.get(namespace, {}).values():
            if limit.applies(ctx):
                out.append(Counter(limit, tuple(sorted((v, ctx[v]) for v in limit.variables))))
        return out

    def _hits(self, counters, delta):
        hits = np.empty(len(counters), dtype=HIT_DTYPE)
        for i, c in enumerate(counters):
            hits[i] = (self._key(c.limit, c.set_variables), self._wire_limit(c.limit), delta)
        return hits

    # -- RateLimiter (lib.rs:362-464) -------------------------------------------------------------
    def is_rate_limited(self, namespace, ctx, delta):
        counters = self._counters_that_apply(namespace, ctx)
        if not counters:
            return CheckResult(False)
        within = self.storage.is_within_limits(self._hits(counters, delta), self.now_us)
        for c, w in zip(counters, within):  # find_first_limited_counter, lib.rs:387-409
            if not w:
                return CheckResult(True, [], c.limit.name)
        return CheckResult(False)

    def update_counters(self, namespace, ctx, delta):
        counters = self._counters_that_apply(namespace, ctx)
        if counters:
            self.storage.update_counters(self._hits(counters, delta), self.now_us)

    def check_rate_limited_and_update(self, namespace, ctx, delta, load_counters):
        counters = self._counters_that_apply(namespace, ctx)
        if not counters:  # lib.rs:434-440
            return CheckResult(False, counters, None)
        # the storage processes simple counters first, then qualified (in_memory.rs:105,121)
        order = [c for c in counters if not c.is_qualified()] + [c for c in counters if c.is_qualified()]
        hits = self._hits(order, delta)
        req_off = None if len(order) == 1 else np.array([0, len(order)], dtype=np.uint32)
        verdict, first, remaining, expires = self.storage.check_and_update(
            hits, self.now_us, req_off=req_off, load_counters=load_counters)
        if load_counters:
            for i, c in enumerate(order):
                c.remaining = int(remaining[i])
                c.expires_in_us = int(expires[i])
        limited = bool(verdict[0])
        name = order[int(first[0])].limit.name if limited else None
        return CheckResult(limited, order if load_counters else [], name)

    def get_counters(self, namespace):
        out = []
        for limit in self.limits.get(namespace, {}).values():
            rows = self.storage.get_counters(self._wire_limit(limit), self.now_us)
            for r in rows:
                if limit.variables:
                    _ident, set_vars = self._key_info[int(r["key"])]
                else:
                    set_vars = ()
                c = Counter(limit, set_vars)
                c.remaining = (limit.max_value - int(r["value"])) % (1 << 64)  # in_memory.rs:166
                c.expires_in_us = int(r["expires_in_us"] if "expires_in_us" in r.dtype.names else r["expiry_us"])
                out.append(c)
        return out
