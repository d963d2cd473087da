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
