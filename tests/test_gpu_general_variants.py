"""The forms the general resolver and the wire path's response kernels had BEFORE round 5's changes stay in the tree behind
experiment switches (they are what the new forms are measured against: scripts/exp/r14f…r14m.sh) — and stay parity-tested:

  RL_GEN_PASS_PREFILL=0   k_gen_round stores every pass flag (default: the flags start out 1, only failures are stored)
  RL_GEN_LOAD_DEFERRED=0  remaining / expires_in stored by every round (default: once behind a group, k_gen_load)
  RL_GEN_CARRY_REQ=0      k_gen_sort gathers every record's request through the record's index (default: it travels with
                          the partitioned record, k_bkt_scatter's b_req)
  RL_RESP_DIRECT=0        k_resp<true> writes a device buffer, copy commands carry it to the host (default: it writes the
                          engine's pinned host staging itself)
  RL_RESP_BLIND=0         the host reads the responses' total before their kernels go out (default: they are enqueued behind
                          the offsets' copy, the staging sized from a bound)

Each switch re-runs tests that take every branch it touches (multi-counter requests with and without load_counters, long
buckets, the wire path with headers through the device's response kernels).  Experiment builds only (the suite's default
library); needs a MI355X."""
import pytest

import test_gpu_parity as P
import test_gpu_rls_e2e as E
from test_gpu_parity import make_engine  # noqa: F401

pytestmark = pytest.mark.gpu

GEN = [{"RL_GEN_PASS_PREFILL": "0"}, {"RL_GEN_LOAD_DEFERRED": "0"}, {"RL_GEN_CARRY_REQ": "0"},
       {"RL_GEN_PASS_PREFILL": "0", "RL_GEN_LOAD_DEFERRED": "0", "RL_GEN_CARRY_REQ": "0"}]
RESP = [{"RL_RESP_DIRECT": "0"}, {"RL_RESP_BLIND": "0"}, {"RL_RESP_PIECES": "3", "RL_RESP_WRITERS": "5"}]
ids = lambda e: "+".join(f"{k[3:]}={v}" for k, v in e.items())  # noqa: E731


@pytest.mark.parametrize("env", GEN, ids=ids)
@pytest.mark.parametrize("load", [False, True], ids=["noload", "load_counters"])
def test_general_resolver_forms_against_the_oracle(make_engine, monkeypatch, env, load):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    P.test_random_multi_counter_requests(make_engine, 21, load)
    P.test_large_multi_counter_batches_against_the_oracle(make_engine, monkeypatch, 8, load)


@pytest.mark.parametrize("env", GEN[-1:] + RESP, ids=ids)
@pytest.mark.parametrize("keys", ["exact", "hashed"])
def test_wire_path_forms_against_the_mirror_and_the_host_assembly(make_engine, monkeypatch, env, keys):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("RLI_RESP_DEVICE", "1")  # (the batches of these tests are small: force the device's response kernels)
    E.test_rate_limit_requests_from_the_wire_to_the_wire(make_engine, keys)
    E.test_responses_built_on_the_device_are_the_bytes_the_host_assembly_builds(make_engine, keys, monkeypatch)
