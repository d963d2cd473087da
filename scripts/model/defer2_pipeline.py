"""Model of the two-stream pipeline's HOST sequencing (limitador_amd/csrc/rl_engine.hip: submit_k1_bucketed, collect_k1_bucketed,
flush_one / flush_pending_apply / poll_pending_apply, wait_done) with and without RL_DEFER2, over a device of two in-order
streams — to check the order and the dependencies the engine promises, and to say what the held-back replay is worth.

Device: the partition stream runs P(p) (needs R(p - 3) complete: always true when it is enqueued, the host has collected that
batch), the replay stream runs R(p) (needs P(p) complete).  A kernel starts `GAP` us after the one before it on its stream ended
(3.6 us: scripts/microbench/kernel_gap2.hip) — `GAP_WAIT` (9.0 us) if a wait command sits in front of it, i.e. if the host did
NOT see the event complete when it enqueued the kernel — and not before its own enqueue + `LAUNCH` us.
Host: the bench's loop (three batches in flight): submit 0, 1, 2; then collect(i), submit(i + 3).  Every host action takes
`HOST` us; a collect spins until the replay's completion reaches the host, and (RL_DEFER2) polls the held-back replays' partition
events every `POLL` us while it spins.

Checked on every run: each replay goes out exactly once, in batch order; R(p) never starts before P(p) has ended; P(p) never
starts before R(p - 3) has ended; every collect returns.  Reported: us per step, wait commands per replay."""
import random

GAP, GAP_WAIT, LAUNCH, HOST, POLL = 3.6, 9.0, 4.0, 1.5, 2.0


class Sim:
    def __init__(self, defer2, t_part, t_replay, seed=0, jitter=0.15, depth_limit=3):
        self.defer2 = defer2
        self.rnd = random.Random(seed)
        self.tp, self.tr, self.jit = t_part, t_replay, jitter
        self.now = 0.0  # host clock
        self.p_free = 0.0  # partition stream: end of its last kernel
        self.r_free = 0.0
        self.p_end, self.r_end, self.r_start, self.p_start = {}, {}, {}, {}
        self.pend = None
        self.pend_old = None
        self.sub = self.col = 0
        self.launched = []
        self.waits = 0

    def dur(self, t):
        return t * (1.0 + self.rnd.uniform(-self.jit, self.jit))

    # ---- device ----
    def enqueue_partition(self, p):
        if p >= 3:
            assert p - 3 in self.r_end and self.r_end[p - 3] <= self.now, "partition enqueued before the replay of p - 3 was collected"
        start = max(self.p_free + GAP, self.now + LAUNCH)
        self.p_start[p] = start
        self.p_end[p] = start + self.dur(self.tp)
        self.p_free = self.p_end[p]

    def event_parted_complete(self, p):  # what hipEventQuery answers at host time `now`
        return self.p_end[p] <= self.now

    def flush_one(self, which):
        p = getattr(self, which)
        setattr(self, which, None)
        assert p == (self.launched[-1] + 1 if self.launched else 0), ("replays out of order", p, self.launched)
        wait = not self.event_parted_complete(p)
        self.waits += wait
        start = max(self.r_free + (GAP_WAIT if wait else GAP), self.now + LAUNCH, self.p_end[p])
        assert start >= self.p_end[p]
        self.r_start[p] = start
        self.r_end[p] = start + self.dur(self.tr)
        self.r_free = self.r_end[p]
        self.launched.append(p)
        self.now += HOST

    def flush_all(self):
        if self.pend_old is not None:
            self.flush_one("pend_old")
        if self.pend is not None:
            self.flush_one("pend")

    def poll(self):
        if self.pend_old is not None:
            if not self.event_parted_complete(self.pend_old):
                return
            self.flush_one("pend_old")
        if self.pend is not None and self.event_parted_complete(self.pend):
            self.flush_one("pend")

    # ---- host ----
    def submit(self):
        assert self.sub - self.col < 3
        p = self.sub
        self.now += HOST
        self.enqueue_partition(p)
        if self.pend_old is not None:
            self.flush_one("pend_old")
        if self.pend is not None:
            if self.defer2 and not self.event_parted_complete(self.pend):
                self.pend_old, self.pend = self.pend, None
            else:
                self.flush_one("pend")
        self.pend = p
        self.sub += 1

    def collect(self):
        k = self.col
        if self.pend_old == k:
            self.flush_one("pend_old")
        if self.pend == k:
            self.flush_all()
        elif self.pend is not None or self.pend_old is not None:
            self.poll()
        assert k in self.r_end, "collect would wait for a replay that was never enqueued"
        while self.now < self.r_end[k]:
            step = min(POLL, self.r_end[k] - self.now) if self.defer2 else self.r_end[k] - self.now
            self.now += step
            if self.defer2 and (self.pend is not None or self.pend_old is not None):
                self.poll()
        self.now += HOST
        self.col += 1

    def run(self, steps):
        for _ in range(min(3, steps)):
            self.submit()
        for i in range(steps):
            self.collect()
            if self.sub < steps:
                self.submit()
        assert self.launched == list(range(steps))
        for p in range(steps):
            assert self.r_start[p] >= self.p_end[p]
            if p >= 3:
                assert self.p_start[p] >= self.r_end[p - 3]
        return self.now / steps, self.waits / steps


    def run_random(self, steps, p_submit=0.6):
        """Any interleaving the API allows: submit while fewer than three are in flight, collect while any is; the host idles
        a random while between two calls (a caller that does other things)."""
        while self.col < steps:
            can_submit = self.sub < steps and self.sub - self.col < 3
            can_collect = self.sub > self.col
            self.now += self.rnd.choice((0.0, 0.0, 5.0, 40.0, 200.0))
            if can_submit and (not can_collect or self.rnd.random() < p_submit):
                self.submit()
            else:
                self.collect()
        assert self.launched == list(range(steps)) and self.pend is None and self.pend_old is None
        for p in range(steps):
            assert self.r_start[p] >= self.p_end[p]
            if p >= 3:
                assert self.p_start[p] >= self.r_end[p - 3]


def compare(t_part=36.0, t_replay=36.5, steps=200, seed=1):
    a = Sim(False, t_part, t_replay, seed).run(steps)
    b = Sim(True, t_part, t_replay, seed).run(steps)
    return a, b


if __name__ == "__main__":
    for steps in (20, 200):
        (s0, w0), (s1, w1) = compare(steps=steps)
        print(f"{steps} steps: tree {s0:.1f} us per step, {w0:.2f} wait commands per replay; RL_DEFER2 {s1:.1f} us, {w1:.2f}")
