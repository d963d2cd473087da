"""The general resolver's ROUND PROTOCOL without a GPU (limitador_amd/csrc/rl_general.hpp + run_general_pass in
rl_engine.hip, round 5): which buffer a kernel reads and writes in which round, and which kernels return at once, restated
launch by launch in Python —

  * two pass-flag buffers, round r writes pass[r & 1]; only FAILURES are stored, the flags start out 1: round 0's are filled by
    k_gen_piece_sum, round r's by k_gen_admit(r), request by request (RL_GEN_PASS_PREFILL);
  * k_gen_admit(r) = per request the AND of pass[(r - 1) & 1]; it is where the fixpoint is seen (changed[slot] = 0: every
    kernel enqueued behind it in the group returns at once — they are enqueued BLIND, a group at a time);
  * the first group is as long as the pass before needed (gen_rounds_hint), later groups GEN_ROUNDS_ENQ_MORE; the host clears
    the status block between groups;
  * `remaining` is stored once behind a group by k_gen_load, from A.admitted and last_round (RL_GEN_LOAD_DEFERRED);
  * k_gen_final reads pass[last_round & 1];
  * round 6: the "admitted set changed" flag of a round is 16 words (GenStatus::changed[slot][16 x 32]) — a k_gen_admit
    workgroup that saw a difference stores 1 into word (workgroup % 16), readers OR the words; there is no k_gen_admit_fold
    launch any more.

What is checked: whatever the group lengths, the final verdicts / first failing hit / `remaining` equal the sequential
reference — in_memory.rs:72-156 applied request by request (all-or-nothing across a request's counters, every counter of a
request reads the value before any of them is applied) — and the number of rounds that really ran does not depend on how they
were grouped.  The arithmetic of one round is the plain definition (value + admitted deltas before the hit, by request); the
kernels' scans are covered on the GPU (tests/test_gpu_parity.py)."""
import numpy as np
import pytest

M64 = (1 << 64) - 1


def sequential(reqs, cells, limits):
    """in_memory.rs:72-156, request by request.  reqs: [[(cell, delta), ...]]; cells: {cell: value}; limits: {cell: max}.
    -> (verdict per request, first failing hit position per request or -1, remaining per hit as read before the update)"""
    value = dict(cells)
    verdict, first, rem = [], [], []
    for hits in reqs:
        f, rr = -1, []
        for j, (c, d) in enumerate(hits):
            v = value.get(c, 0)
            ok = v + d <= limits[c]
            rr.append(limits[c] - (v + d) if ok else 0)
            if not ok and f < 0:
                f = j
        if f < 0:
            for c, d in hits:
                value[c] = value.get(c, 0) + d
        verdict.append(0 if f < 0 else 1)
        first.append(f)
        rem.append(rr)
    return verdict, first, rem


class Device:
    """The state the kernels share, and the kernels — each a function that may `return` at once like its HIP original."""

    def __init__(self, reqs, cells, limits):
        self.reqs, self.cells, self.limits = reqs, cells, limits
        self.off = np.cumsum([0] + [len(r) for r in reqs])
        self.n = int(self.off[-1])
        self.hit = [(r, j, c, d) for r, hits in enumerate(reqs) for j, (c, d) in enumerate(hits)]
        self.by_cell = sorted(range(self.n), key=lambda i: (self.hit[i][2], i))  # k_gen_sort: by cell, trace order inside
        rng = np.random.default_rng(1)
        self.pass_ = [rng.integers(0, 2, self.n).astype(np.uint8), rng.integers(0, 2, self.n).astype(np.uint8)]  # garbage
        self.admitted = rng.integers(0, 2, len(reqs)).astype(np.uint8)  # garbage until an admit wrote it
        self.remaining = np.full(self.n, 0xDEAD, dtype=np.uint64)
        self.clear_status()
        self.rounds_run_total = 0
        self.launches = 0

    def clear_status(self):  # hipMemsetAsync(d_gst, 0): between groups
        self.changed = [[0] * 16 for _ in range(40)]  # [slot][workgroup % 16]
        self.last_round = 0
        self.last_slot = 0
        self.rounds_run = 0

    # -- one round's arithmetic: hit i passes iff value + the admitted deltas before it (other requests') + its own fits --
    def _scan(self, use_admitted):
        run = {}
        out_pass = np.ones(self.n, dtype=np.uint8)
        out_rem = np.zeros(self.n, dtype=np.uint64)
        seen = {}  # (cell, request) -> what the request itself already added on the cell (exclusive by request)
        for i in self.by_cell:
            r, _j, c, d = self.hit[i]
            own = seen.get((c, r), 0)
            v = self.cells.get(c, 0) + run.get(c, 0) - own
            ok = v + d <= self.limits[c]
            out_pass[i] = 1 if ok else 0
            out_rem[i] = self.limits[c] - (v + d) if ok else 0
            if (not use_admitted) or self.admitted[r]:
                run[c] = run.get(c, 0) + d
                seen[(c, r)] = own + d
        return out_pass, out_rem

    def k_gen_piece_sum(self, rnd, check_slot):
        self.launches += 1
        if check_slot and not any(self.changed[check_slot]):
            return
        if rnd == 0:
            self.pass_[0][:] = 1  # round 0's flags start out "passes"

    def k_gen_round(self, rnd, check_slot, write_slot):
        self.launches += 1
        if check_slot and not any(self.changed[check_slot]):
            return
        self.last_round = rnd
        self.rounds_run += 1
        if check_slot == 0:  # no k_gen_admit of this group before this round: the next round always follows
            self.last_slot = write_slot
            self.changed[write_slot][0] = 1
        p, _ = self._scan(use_admitted=rnd != 0)
        cur = self.pass_[rnd & 1]
        for i in range(self.n):
            if not p[i]:
                cur[i] = 0  # only the failures are stored

    def k_gen_admit(self, rnd, check_slot, write_slot):
        self.launches += 1
        if rnd == 0:
            return
        if check_slot and not any(self.changed[check_slot]):
            return
        self.last_slot = write_slot
        self.last_round = rnd - 1
        prev, nxt = self.pass_[(rnd - 1) & 1], self.pass_[rnd & 1]
        for r in range(len(self.reqs)):
            b, e = int(self.off[r]), int(self.off[r + 1])
            adm = 1 if all(prev[b:e]) else 0
            before = 1 if rnd == 1 else int(self.admitted[r])
            if adm != before:  # (the workgroup of request r — 4 requests per "workgroup" here — says so in ITS word)
                self.changed[write_slot][(r // 4) % 16] = 1
            self.admitted[r] = adm
            nxt[b:e] = 1  # this round's flags start out "passes"

    def k_gen_load(self):
        self.launches += 1
        _, rem = self._scan(use_admitted=self.last_round != 0)
        self.remaining[:] = rem

    def k_gen_final(self):
        p = self.pass_[self.last_round & 1]
        verdict, first = [], []
        for r in range(len(self.reqs)):
            b, e = int(self.off[r]), int(self.off[r + 1])
            f = next((q - b for q in range(b, e) if not p[q]), -1)
            verdict.append(0 if f < 0 else 1)
            first.append(f)
        return verdict, first


def run_pass(dev, first_group, more):
    """run_general_pass's loop: groups of rounds enqueued blind, the status block read between groups.  The LAST round of a
    group is only its k_gen_admit (round 6): a converged pass needs nothing else of it; if the admitted set still moves, the next
    group opens with that round's scan, unconditionally."""
    rnd = 0
    tail_open = False
    while True:
        n_enq = first_group if rnd == 0 else more
        for q in range(n_enq):
            admit_done = tail_open and q == 0
            check = 0 if q == 0 else q
            if rnd > 0 and not admit_done:
                dev.k_gen_admit(rnd, check, q + 1)
            if rnd > 0 and q + 1 == n_enq:
                tail_open = True
                break
            tail_open = False
            run_if = 0 if (rnd == 0 or admit_done) else q + 1
            dev.k_gen_piece_sum(rnd, run_if)
            dev.k_gen_round(rnd, run_if, q + 1)
            rnd += 1
        dev.k_gen_load()
        verdict, first = dev.k_gen_final()
        converged = not any(dev.changed[dev.last_slot])  # (k_gen_commit's test)
        dev.rounds_run_total += dev.rounds_run
        if converged:
            return verdict, first
        assert rnd < 200, "no convergence"
        dev.clear_status()


def workload(seed, n_req=60, n_cells=12, chain=False):
    rng = np.random.default_rng(seed)
    limits = {c: int(rng.integers(3, 12)) for c in range(n_cells)}
    cells = {c: int(rng.integers(0, 4)) for c in range(n_cells) if rng.random() < 0.5}
    reqs = []
    for r in range(n_req):
        k = int(rng.integers(1, 4))
        cs = rng.choice(n_cells, size=k, replace=rng.random() < 0.2)  # now and then the same cell twice in one request
        reqs.append([(int(c), int(rng.integers(1, 3))) for c in cs])
    if chain:
        # requests that depend on each other many deep: request i fits on cell i only if request i - 1 (which also hits
        # cell i) was REFUSED, and it is refused iff request i - 2 was admitted ... — one more round per link
        reqs, limits, cells = [], {}, {}
        for i in range(12):
            limits[i] = 1
        for i in range(11):
            reqs.append([(i, 1), (i + 1, 1)])
    return reqs, cells, limits


@pytest.mark.parametrize("seed", range(12))
def test_any_grouping_of_blind_rounds_gives_the_sequential_answer(seed):
    reqs, cells, limits = workload(seed, chain=seed >= 10)
    want_v, want_f, want_rem = sequential(reqs, cells, limits)
    ran = set()
    for first_group, more in [(3, 6), (2, 2), (2, 3), (5, 2), (12, 6), (4, 2)]:
        dev = Device(reqs, cells, limits)
        got_v, got_f = run_pass(dev, first_group, more)
        assert got_v == want_v, (seed, first_group, more)
        assert got_f == want_f, (seed, first_group, more)
        flat = [x for rr in want_rem for x in rr]
        # `remaining` of a hit is what the reference reports for the requests it decides the same way: every hit of an
        # ADMITTED request, and of a refused one every hit up to its first failing one is read from the same state
        for r, hits in enumerate(reqs):
            b = int(dev.off[r])
            for j in range(len(hits)):
                assert int(dev.remaining[b + j]) == flat[b + j], (seed, first_group, more, r, j)
        ran.add(dev.rounds_run_total)
    assert len(ran) == 1, ran  # the rounds that really ran do not depend on the grouping


def test_the_hint_saves_the_blind_rounds_it_is_for():
    """A pass that needs R rounds: with the first group = R + 1 (what the pass before reported) nothing is enqueued in vain
    except the one round that sees the fixpoint; with 3 + 6 a second group goes out and most of it returns at once."""
    reqs, cells, limits = workload(10, chain=True)
    dev = Device(reqs, cells, limits)
    run_pass(dev, 3, 6)
    need = dev.rounds_run_total
    assert need >= 5
    launches_36 = dev.launches
    dev2 = Device(reqs, cells, limits)
    run_pass(dev2, need + 1, 6)
    assert dev2.rounds_run_total == need
    assert dev2.launches < launches_36
