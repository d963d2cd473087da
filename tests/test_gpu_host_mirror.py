"""The reference's own scenarios (tests/scenarios.py) replayed through the C++ host mirror of
`CounterStorage` (include/rl_storage.h): strings in, Authorization out, everything below the trait
boundary native.  Needs a MI355X (the mirror has no CPU backend)."""
import threading

import pytest

import scenarios
from helpers.host_limiter import HostMirrorLimiter

pytestmark = pytest.mark.gpu


@pytest.fixture()
def host():
    from limitador_amd.host_storage import HostStorage

    h = HostStorage(capacity_cells=1 << 12, max_batch_hits=1 << 12)
    yield h
    h.close()


@pytest.mark.parametrize("scenario", scenarios.ALL, ids=lambda f: f.__name__)
def test_reference_scenarios_through_the_host_mirror(host, scenario):
    scenario(HostMirrorLimiter(host))


@pytest.mark.parametrize("scenario", [s for s in scenarios.ALL if "check_rate_limited" in s.__name__],
                         ids=lambda f: f.__name__)
def test_reference_scenarios_through_the_micro_batcher(host, scenario):
    scenario(HostMirrorLimiter(host, batched=True))


def test_micro_batcher_coalesces_concurrent_callers(host):
    """64 threads x 50 requests against one limit of 1000: exactly 1000 admitted, far fewer device
    batches than requests."""
    host.set_clock(1_700_000_000_000_000)
    limit = ("ns", 1000, 60, ("req_method == 'GET'",), ("app_id",), None)
    admitted = []
    lock = threading.Lock()

    def worker():
        ok = 0
        for _ in range(50):
            limited, _idx, _ = host.check_and_update([(limit, (("app_id", "x"),))], 1, False, batched=True)
            ok += 0 if limited else 1
        with lock:
            admitted.append(ok)

    threads = [threading.Thread(target=worker) for _ in range(64)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert sum(admitted) == 1000
    batches, requests = host.batcher_stats()
    assert requests == 64 * 50 and batches < requests


def test_errors_carry_the_engine_status(host):
    from limitador_amd.host_storage import StorageError

    host.set_clock(1_700_000_000_000_000)
    simple = ("ns", 5, 60, (), (), None)  # never add_counter'ed: the reference panics (in_memory.rs:107)
    with pytest.raises(StorageError) as e:
        host.check_and_update([(simple, ())], 1, False)
    assert e.value.code == -5


def test_a_delta_beyond_32_bits_is_answered_like_the_reference(host):
    """The trait's delta is a u64 (storage/mod.rs:283-288): `value + delta <= max_value` with a delta of 2^40
    is Limited, not an error (in_memory.rs:259-264; HTTP /check_and_report takes a u64 delta), and a limit
    that large admits it with exact 64-bit arithmetic."""
    host.set_clock(1_700_000_000_000_000)
    small = (("ns", 5, 60, (), ("u",), None), (("u", "1"),))
    huge = (("ns", (1 << 63), 60, (), ("u",), None), (("u", "1"),))
    limited, idx, _ = host.check_and_update([small], 1 << 40, False)
    assert limited and idx == 0
    assert host.is_within_limits(*small, 1 << 40) is False
    assert host.is_within_limits(*huge, 1 << 40) is True
    limited, _idx, out = host.check_and_update([huge], (1 << 62) + 5, True)
    assert not limited and out[0][0] == (1 << 63) - (1 << 62) - 5
    limited, _idx, out = host.check_and_update([huge], (1 << 62), True)   # 2^62 + 5 + 2^62 > 2^63
    assert limited and out[0][0] == 0
    host.update_counter(*huge, (1 << 40))
    limited, _idx, out = host.check_and_update([huge], 1, True)
    assert not limited and out[0][0] == (1 << 63) - ((1 << 62) + 5 + (1 << 40) + 1)


def test_the_mirror_forgets_the_identities_of_swept_counters(host):
    """Descriptor values are caller-controlled: the interning tables must not grow without bound.  A TTL sweep
    (on its own every `sweep_after` new counters, or explicit) drops the expired qualified cells and their
    interned identities; a counter that comes back afterwards starts from a fresh cell, like after a moka
    eviction in the reference."""
    t0 = 1_700_000_000_000_000
    host.set_clock(t0)
    host.set_sweep_after(0)
    lim = ("ns", 3, 1, (), ("u",), None)
    for i in range(500):
        limited, _i, _l = host.check_and_update([(lim, (("u", f"user{i}"),))], 1, False)
        assert not limited
    assert host.interned_counters() == 500
    host.set_clock(t0 + 2_000_000)  # every 1-second window has ended
    assert host.sweep_expired() == 500
    assert host.interned_counters() == 0
    # automatic: once 100 new counters have been interned since the last sweep, the next call sweeps first
    host.set_sweep_after(100)
    for i in range(150):  # (the sweep after the 100th finds nothing expired yet)
        host.check_and_update([(lim, (("u", f"again{i}"),))], 1, False)
    assert host.interned_counters() == 150
    host.set_clock(t0 + 5_000_000)  # their windows have ended
    for i in range(101):
        host.check_and_update([(lim, (("u", f"third{i}"),))], 1, False)
    assert host.interned_counters() == 101  # the 150 expired ones are gone
    # a swept counter starts over
    for _ in range(3):
        limited, _i, _l = host.check_and_update([(lim, (("u", "again0"),))], 1, False)
        assert not limited
    limited, _i, _l = host.check_and_update([(lim, (("u", "again0"),))], 1, False)
    assert limited
