"""Not collected by the suite (it lives outside tests/): a key-sharded step on two in-process ranks, then a FAILING assert with
everything still open — does pytest hang on its way out?
usage: timeout -s ABRT 90 python -X faulthandler -m pytest scripts/debug/test_exit_hang_probe.py -x -q -p no:cacheprovider"""
import os
import sys
import threading

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from limitador_amd import sharded_abi  # noqa: E402
from limitador_amd import workloads as W  # noqa: E402
from limitador_amd.engine import Engine  # noqa: E402
from limitador_amd.wire import HIT_DTYPE  # noqa: E402


def test_fails_with_everything_open():
    dev = torch.device("cuda", 0)
    world, n_req = 2, 7_000
    engines = []
    for r in range(world):
        e = Engine(capacity_cells=1 << (18 if r == 0 or os.environ.get("PROBE_FULL") != "1" else 9), max_batch_hits=1 << 18)
        e.set_limits([(100_000, 60), (3, 60)])
        engines.append(e)
    group = sharded_abi.LocalGroup(world)
    ranks = [sharded_abi.Sharded(engines[r], world, r, 1 << 18, transport=group.transport(r)) for r in range(world)]
    h = np.zeros(2 * n_req, dtype=HIT_DTYPE)
    h["key"][0::2] = 0x123456789ABC
    h["key"][1::2] = W.splitmix64(np.arange(n_req, dtype=np.uint64) % np.uint64(40_000)) & np.uint64(0x3FFFFFFFFFFFFFFF)
    h["limit"][1::2] = 1
    h["delta"] = 1
    off = torch.arange(0, 2 * n_req + 1, 2, dtype=torch.int32, device=dev)
    t = torch.from_numpy(h.view(np.int64).reshape(-1, 2).copy()).to(dev)
    out, errors = {}, []

    def run(r):
        v = torch.zeros(n_req, dtype=torch.uint8, device=dev)
        f = torch.zeros(n_req, dtype=torch.int32, device=dev)
        try:
            ranks[r].check_requests(t.data_ptr(), 2 * n_req, off.data_ptr(), n_req, W.NOW0_US, v.data_ptr(), False, f.data_ptr())
            out[r] = "applied"
        except Exception as ex:
            errors.append((r, repr(ex)))

    th = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=60)
    assert not errors and not out, (errors, out)
