"""The host sequencing of the two-stream pipeline, with and without RL_DEFER2 (a replay held back across one more submit and
sent out from the collect's spin once its partition is seen complete), as a model over two in-order device streams
(scripts/model/defer2_pipeline.py restates submit_k1_bucketed / collect_k1_bucketed / flush_one / poll_pending_apply of
limitador_amd/csrc/rl_engine.hip).  Whatever the kernel times and however the caller interleaves submits and collects: every
replay goes out exactly once and in batch order, never before its partition has ended; a partition never starts before the replay
three batches back has ended; no collect waits for a replay that was not enqueued; nothing is left held back at the end."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("defer2_model", os.path.join(ROOT, "scripts", "model", "defer2_pipeline.py"))
model = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(model)


@pytest.mark.parametrize("defer2", [False, True])
@pytest.mark.parametrize("t_part,t_replay", [(36.0, 36.5), (10.0, 60.0), (70.0, 20.0), (1.0, 1.0), (45.0, 45.0)])
def test_the_bench_loop_keeps_order_and_dependencies(defer2, t_part, t_replay):
    for seed in range(20):
        for steps in (1, 2, 3, 4, 7, 20, 60):
            model.Sim(defer2, t_part, t_replay, seed=seed, jitter=0.5).run(steps)


@pytest.mark.parametrize("defer2", [False, True])
def test_any_interleaving_of_submits_and_collects(defer2):
    for seed in range(300):
        t_part, t_replay = [(36.0, 36.5), (10.0, 60.0), (70.0, 20.0), (3.0, 3.0)][seed % 4]
        model.Sim(defer2, t_part, t_replay, seed=seed, jitter=0.6).run_random(40, p_submit=(0.3, 0.6, 0.9)[seed % 3])


def test_held_back_replays_go_out_without_wait_commands():
    """With the kernel times of round 4 the held-back form takes the wait command off (nearly) every replay and is not slower."""
    (s0, w0), (s1, w1) = model.compare(steps=200)
    assert w1 < 0.1 and w1 < w0 and s1 <= s0 + 0.1
