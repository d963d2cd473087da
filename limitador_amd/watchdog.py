"""A progress-based watchdog for multi-rank runs (bench.py --gpus N).

A routed run has many places where one rank can wait for another for ever (a rendezvous, a communicator bring-up, a
collective a peer never enters).  The watchdog ends THIS process when nothing has called `kick()` for `limit_s` seconds —
"no progress", not "the run is long": every phase boundary and every collected slice re-arms it, and a phase that is known
to be slow on a fresh box (the first RCCL initialisation pages in a few hundred MB of device code and has been seen to
take 400 s, profiles/r05a_defer2.md §3) names its own, longer limit with `kick(phase, limit_s=...)`.

No dependency on torch or on the engine: tests/test_watchdog_cpu.py drives it with a fake clock.
"""
import os
import sys
import threading
import time


class Watchdog:
    def __init__(self, limit_s=300.0, on_stuck=None, clock=time.monotonic, poll_s=1.0, name="bench.py"):
        self.default_limit_s = float(limit_s)
        self.clock = clock
        self.poll_s = poll_s
        self.name = name
        self.on_stuck = on_stuck if on_stuck is not None else self._exit
        self._lock = threading.Lock()
        self._phase = "start"
        self._limit = self.default_limit_s
        self._last = clock()
        self._kicks = 0
        self._stop = threading.Event()
        self._thread = None
        self.fired = False

    # -- what the run calls -------------------------------------------------------------------------------------
    def kick(self, phase=None, limit_s=None):
        """Progress was made.  `phase` names what runs now (it is what the exit message quotes); `limit_s` is the time
        THIS phase may take without another kick (default: the watchdog's own limit)."""
        with self._lock:
            self._last = self.clock()
            self._kicks += 1
            if phase is not None:
                self._phase = phase
                self._limit = self.default_limit_s if limit_s is None else float(limit_s)
            elif limit_s is not None:
                self._limit = float(limit_s)

    def start(self):
        if self._thread is None:
            self._thread = threading.Thread(target=self._run, name="watchdog", daemon=True)
            self._thread.start()
        return self

    def cancel(self):
        self._stop.set()

    # -- the check (also called directly by the unit test, with a fake clock) --------------------------------------
    def state(self):
        with self._lock:
            return {"phase": self._phase, "limit_s": self._limit, "idle_s": self.clock() - self._last, "kicks": self._kicks}

    def check(self):
        """-> True (and fires on_stuck once) when the current phase has made no progress for longer than its limit."""
        s = self.state()
        if s["idle_s"] <= s["limit_s"] or self.fired:
            return False
        self.fired = True
        self.on_stuck(s)
        return True

    def _run(self):
        while not self._stop.wait(self.poll_s):
            if self.check():
                return

    def _exit(self, s):
        sys.stderr.write(f"{self.name}: no progress for {s['idle_s']:.0f} s in phase '{s['phase']}' "
                         f"(limit {s['limit_s']:.0f} s, {s['kicks']} progress marks before it): giving up\n")
        sys.stderr.flush()
        os._exit(3)
