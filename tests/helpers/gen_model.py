"""Test-side CPU stand-in for the owner-side phases of ShardedMultiCounterEngine (rl_gen_begin_device ..
rl_gen_commit_device on the HIP engine): the same four phases over a dict table, in plain Python — what one
fixpoint round computes (k_gen_round), what the commit writes (k_gen_commit), restated from in_memory.rs:72-156 and
atomic_expiring_value.rs:19-24,36-42,68-74,87-99.  Used by tests/test_sharded_multi_gloo.py, where two gloo ranks
(no GPU) must reproduce ONE sequential oracle on the concatenated slices."""
import numpy as np
import torch

from limitador_amd.wire import RL_SIMPLE

M64 = (1 << 64) - 1


class ModelGenLocal:
    def __init__(self, rows, simple=()):
        self.rows = [(int(m), int(s) * 1_000_000) for m, s in rows]
        self.table = {}  # key -> [value, expiry_us, limit]
        for limit, key in simple:
            self.table[int(key)] = [0, 0, int(limit) | RL_SIMPLE]  # add_counter: (0, UNIX_EPOCH), in_memory.rs:38-44

    def begin(self, hits, req_id, now_us, load):
        h = hits.cpu().numpy().astype(np.int64).view(np.uint64).reshape(-1, 2)
        self.key = [int(k) for k in h[:, 0]]
        self.limit = [int(x) & 0xFFFFFFFF for x in h[:, 1]]
        self.delta = [int(x) >> 32 for x in h[:, 1]]
        self.req = [int(r) & 0xFFFFFFFF for r in req_id.cpu().numpy()]
        self.now, self.load, self.n = int(now_us), bool(load), len(self.key)
        for k, l in zip(self.key, self.limit):
            if (l & RL_SIMPLE) and k not in self.table:
                raise RuntimeError("missing simple counter")  # in_memory.rs:106-107
        self.last_admitted = None

    def _cell(self, key, limit):
        """(value_at(now), ttl(now), flags) of the cell before the batch."""
        window = self.rows[limit & ~RL_SIMPLE][1]
        c = self.table.get(key)
        if window == 0:
            return 0, 0, "zerowin"
        if c is None:
            return 0, window, "new"
        if c[1] <= self.now:
            return 0, 0, "expired"
        return c[0], c[1] - self.now, "alive"

    def round(self, admitted):
        adm = [1] * self.n if admitted is None else [int(a) for a in admitted.cpu().numpy()]
        self.last_admitted = adm
        flags = np.ones(self.n, dtype=np.uint8)
        self.rem = np.zeros(self.n, dtype=np.uint64)
        self.exp = np.zeros(self.n, dtype=np.uint64)
        run = {}    # key -> [sum of admitted deltas so far, count]
        mine = {}   # (key, request) -> [sum, count] the request itself has already added on this cell
        for i in range(self.n):
            k, l, d, r = self.key[i], self.limit[i], self.delta[i], self.req[i]
            mx, window = self.rows[l & ~RL_SIMPLE]
            s, ttl0, kind = self._cell(k, l)
            tot = run.get(k, [0, 0])
            own = mine.get((k, r), [0, 0])
            # exclusive BY REQUEST: a request's hits on one cell all read the value before any of them is applied
            before_sum, before_cnt = (tot[0] - own[0]) & M64, tot[1] - own[1]
            v = 0 if kind == "zerowin" else (s + before_sum) & M64
            total = (v + d) & M64
            ok = total <= mx
            flags[i] = 1 if ok else 0
            if self.load:
                self.rem[i] = mx - total if ok else 0
                if kind == "zerowin":
                    self.exp[i] = 0
                elif kind in ("alive", "new"):
                    self.exp[i] = ttl0
                else:
                    self.exp[i] = window if before_cnt > 0 else 0
            if adm[i]:
                run[k] = [(tot[0] + d) & M64, tot[1] + 1]
                mine[(k, r)] = [(own[0] + d) & M64, own[1] + 1]
        return torch.from_numpy(flags)

    def count(self, reached):
        self.reached = [1] * self.n if reached is None else [int(x) for x in reached.cpu().numpy()]
        new = {k for k, rch in zip(self.key, self.reached) if rch and k not in self.table}
        return len(new), 1 << 40

    def loaded(self):
        return torch.from_numpy(self.rem.view(np.int64).copy()), torch.from_numpy(self.exp.view(np.int64).copy())

    def commit(self):
        adm = self.last_admitted
        tot = {}
        for i in range(self.n):
            k, l, d = self.key[i], self.limit[i], self.delta[i]
            window = self.rows[l & ~RL_SIMPLE][1]
            if k not in self.table:
                if not self.reached[i]:
                    continue
                self.table[k] = [0, self.now + window, l]  # created at first touch, admitted or not (in_memory.rs:122-127)
                tot[k] = ["fresh", 0, 0]
            if adm[i]:
                t = tot.setdefault(k, ["old", 0, 0])
                t[1] = (t[1] + d) & M64
                t[2] = d
                t.append(True)
        for k, t in tot.items():
            if len(t) <= 3:
                continue  # nothing admitted on this cell
            c = self.table[k]
            window = self.rows[c[2] & ~RL_SIMPLE][1]
            if window == 0:
                c[0], c[1] = t[2], self.now
            elif t[0] == "fresh" or c[1] <= self.now:
                c[0], c[1] = t[1], self.now + window  # update_if_expired: the first admitted hit stores, the rest add
            else:
                c[0] = (c[0] + t[1]) & M64

    def abort(self):
        pass
