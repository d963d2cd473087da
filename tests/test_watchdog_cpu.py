"""limitador_amd/watchdog.py with a fake clock: 'no progress for N seconds', not 'N seconds since the start'
(bench.py armed an absolute 300-s deadline at process start)."""
import threading
import time

from limitador_amd.watchdog import Watchdog


class Clock:
    def __init__(self):
        self.t = 1000.0

    def __call__(self):
        return self.t


def test_a_400_second_init_followed_by_progress_does_not_exit():
    clk, fired = Clock(), []
    wd = Watchdog(limit_s=300.0, on_stuck=fired.append, clock=clk)
    wd.kick("rl_sharded_create_rccl", limit_s=900.0)  # a bring-up phase names its own limit
    clk.t += 400.0
    assert not wd.check() and not fired
    wd.kick("warm-up steps")  # back to the default: 300 s WITHOUT progress
    for _ in range(50):  # 50 collected slices, 100 s apart: 5000 s of run, never 300 s idle
        clk.t += 100.0
        assert not wd.check()
        wd.kick()
    assert not fired
    clk.t += 299.0
    assert not wd.check()
    clk.t += 2.0
    assert wd.check() and len(fired) == 1
    assert fired[0]["phase"] == "warm-up steps" and fired[0]["limit_s"] == 300.0 and fired[0]["idle_s"] > 300.0
    assert not wd.check(), "fires once"


def test_an_init_that_never_ends_is_still_caught_by_its_own_limit():
    clk, fired = Clock(), []
    wd = Watchdog(limit_s=300.0, on_stuck=fired.append, clock=clk)
    wd.kick("init_process_group", limit_s=900.0)
    clk.t += 899.0
    assert not wd.check()
    clk.t += 2.0
    assert wd.check() and fired[0]["phase"] == "init_process_group"


def test_a_kick_without_a_phase_keeps_the_phase_and_its_limit():
    clk, fired = Clock(), []
    wd = Watchdog(limit_s=10.0, on_stuck=fired.append, clock=clk)
    wd.kick("pre-populate", limit_s=50.0)
    clk.t += 40.0
    wd.kick()
    clk.t += 40.0
    assert not wd.check()
    assert wd.state()["phase"] == "pre-populate" and wd.state()["limit_s"] == 50.0 and wd.state()["kicks"] == 2


def test_the_thread_fires_on_a_real_clock_and_cancel_stops_it():
    fired = threading.Event()
    wd = Watchdog(limit_s=0.15, on_stuck=lambda s: fired.set(), poll_s=0.02).start()
    for _ in range(10):  # progress for 0.5 s: no exit
        time.sleep(0.05)
        wd.kick()
    assert not fired.is_set()
    assert fired.wait(2.0)
    wd2 = Watchdog(limit_s=0.1, on_stuck=lambda s: fired.clear(), poll_s=0.02).start()
    wd2.cancel()
    time.sleep(0.3)
    assert fired.is_set()
