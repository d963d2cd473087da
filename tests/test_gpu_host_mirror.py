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
    with pytest.raises(StorageError) as e:
        host.check_and_update([(("ns", 5, 60, (), ("u",), None), (("u", "1"),))], 1 << 40, False)
    assert e.value.code == -1
