"""bench.py's contract, the parts that need no GPU: the command line the driver uses, the defaults it relies on, and the
`cpu_baseline` leg — the oracle timed on the host's cores on a bounded sample — with the fields the bench line carries."""
import importlib
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_the_drivers_command_line_and_the_defaults(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup) == (1, 20, 5)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    d = b.parse()
    # no flags: one GPU, BASELINE.json configs[2] (10 M keys, 1 M-hit Zipf-0.99 batches), a run of minutes at most
    assert (d.gpus, d.keys, d.batch, d.zipf) == (1, 10_000_000, 1_000_000, 0.99)
    assert 1 <= d.steps <= 1000 and 0 <= d.warmup <= 100 and 0 < d.cpu_seconds <= 30


def test_cpu_baseline_leg_on_a_small_sample():
    b = _bench()
    args = types.SimpleNamespace(keys=20_000, batch=5_000, zipf=0.99)
    r = b.cpu_baseline(args, 0.4)
    assert r["unit"] == "decisions/s" and r["kind"] == "port" and r["value"] > 0
    assert 1 <= r["cores"] <= (os.cpu_count() or 1)
    assert "oracle/limitador_oracle.c" in r["sample"] and "5000 hits" in r["sample"]
    if r["cores"] > 1:
        assert r["single_thread"] > 0 and str(r["cores"]) in r["threads_tried"]
