#!/usr/bin/env python
"""bench.py — rate-limit decisions/s of the check_and_update hot path on MI355X.

A step = one pass of CounterStorage::check_and_update over one batch of synthetic hits that is
already resident in HBM (BASELINE.json configs[2]: 10 M keys, Zipf-0.99, 1 M single-counter
requests per batch, max 1000 / 60 s fixed window, delta 1).  With N GPUs the key space is
hash-sharded (N x 10 M keys, every rank ingests its own 1 M-hit slice: weak scaling) and each step
includes the descriptor all-to-all and the verdict all-to-all over RCCL.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, timed with HIP events on the
engine's stream inside the timed region; `cpu_baseline` is the CPU oracle (a C restatement of the
reference path, kind "port") timed on a bounded sample of the same workload, rank 0, N == 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Algorithmic bytes per (request x counter), SURVEY.md §8(d): 16 B descriptor read + 24 B cell read
# (key, value, expiry) + 8 B value write-back + 1 B verdict = 49 B.  k_bkt_step is the kernel that
# reads the descriptor and the cell, writes the value back and emits the verdict, so all 49 B are
# its algorithmic bytes; the partition kernel (k_bkt_part) only reorders the batch —
# ranking traffic is overhead, not algorithmic — and is charged 0.
ALGO_BYTES = {"apply": 49, "part": 0}
NOT_KERNELS = ("apply_gap", "part_slack")  # timing slots that are intervals between kernels
ALGO_BYTES_TOTAL = 49
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# roofline.random_line_frac: what the replay is actually bound by.  The two inputs come from the last evidence
# visit's profiles/random_line.json (scripts/gpu_final_r06.sh + scripts/summarize_sq.py: the replay's TCP -> TCC requests per
# launch, the random-access slope microbenchmark of the same visit); the constants below are round 4's, used only if that
# file is missing.
RANDOM_TRANSACTIONS_PER_1M_ZIPF = 1.45e6  # profiles/r04h_replay_memory_path.md (configs[2], 1 M-hit Zipf-0.99 batch)
RANDOM_LINE_RATE = 57e9                   # random 32-byte cell reads per second, chip-wide: profiles/r04a_random_slope.txt


def _random_line_model():
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "random_line.json")))
        return float(j["transactions_per_launch"]), float(j["lines_per_s"]), {"file": "profiles/random_line.json", "profile": j.get("source"),
                                                                                "commit": j.get("commit")}
    except Exception:
        return RANDOM_TRANSACTIONS_PER_1M_ZIPF, RANDOM_LINE_RATE, {"file": None, "profile": "round-4 constants in bench.py",
                                                                   "commit": None}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--keys", type=int, default=10_000_000, help="keys per GPU")
    ap.add_argument("--batch", type=int, default=1_000_000, help="hits per GPU per step")
    ap.add_argument("--zipf", type=float, default=0.99, help="0 => uniform keys")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget; 0 disables")
    ap.add_argument("--secondary", type=int, default=1, help="0: skip the `secondary` block (other shapes of BASELINE.json)")
    ap.add_argument("--cap-mult", type=float, default=1.0, help="scale the table capacity (experiments)")
    ap.add_argument("--depth", type=int, default=3, choices=(1, 2, 3),
                    help="batches in flight on one GPU: N = submit batch k+N-1 before collecting batch k "
                         "(rl_check_and_update_submit_device / _collect; with 2+ the partition of the next batch "
                         "overlaps k_bkt_step of this one), 1 = one blocking call per batch; "
                         "the routed path keeps 3 ingress slices in flight at depth >= 2 (ShardedEngine)")
    ap.add_argument("--timing-mode", type=int, default=3, choices=(0, 2, 3),
                    help="HIP events in the timed region: 2 = k_bkt_step of every batch, 3 = k_bkt_part and k_bkt_step "
                         "of every fourth batch, 0 = none (roofline then comes from the breakdown pass)")
    ap.add_argument("--sharded-impl", choices=("torch", "abi"), default=os.environ.get("RL_SHARDED_IMPL", "abi"),
                    help="routed step driven by the C entry with its own RCCL communicator (include/rl_sharded.h: 105 us per "
                         "1 M-hit slice at world 1) or by torch.distributed from Python (limitador_amd/sharded.py: 181 us).  "
                         "RL_SHARDED_IMPL=torch selects the latter.  A watchdog (limitador_amd/watchdog.py) ends a routed run "
                         "that makes no PROGRESS for --stall-seconds instead of leaving the ranks at a collective: every phase "
                         "boundary, every loaded chunk and every collected slice re-arms it; communicator bring-up has its own, "
                         "longer limit (--init-seconds).")
    ap.add_argument("--stall-seconds", type=float, default=300.0, help="routed runs: seconds without progress before the watchdog exits (rc 3)")
    ap.add_argument("--init-seconds", type=float, default=900.0,
                    help="routed runs: limit for ONE bring-up phase (rendezvous, RCCL initialisation: the first one on a fresh box "
                         "pages in librccl's device code and has taken 400 s)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the routed (all-to-all) data path even with one rank (exercises the N>1 code on one GPU)")
    return ap.parse_args(argv)


def _cpu_batches(args):
    import numpy as np

    from limitador_amd import workloads as W

    rng = np.random.default_rng(W.SEED)
    cdf = W.zipf_cdf(args.keys, args.zipf) if args.zipf > 0 else None
    n = min(args.batch, 1_000_000)
    return n, [W.zipf_batch(args.keys, n, rng, cdf) if cdf is not None else W.uniform_batch(args.keys, n, rng)
               for _ in range(3)]


def _cpu_shard(args, n_shards, shard):
    """One oracle table holding shard `shard` of the universe (keys are assigned by a hash of the key)."""
    import numpy as np

    import oracle
    from limitador_amd import workloads as W

    orc = oracle.OracleStorage()
    orc.set_limits([(W.MAX_VALUE, W.WINDOW_S)])
    chunk = 1 << 21
    for lo in range(0, args.keys, chunk):
        c = W.universe_rows(args.keys, lo=lo, hi=min(args.keys, lo + chunk))
        if n_shards > 1:
            c = c[(W.splitmix64(c["key"]) % np.uint64(n_shards)) == shard]
        orc.load_cells(c["key"], c["limit"], c["value"], c["expiry_us"])
    return orc


def cpu_baseline(args, budget_s):
    """The CPU oracle (C restatement of the reference path) on the same workload shape.

    `value`: every host core, keys hash-sharded over one single-threaded table per thread (valid for
    single-counter requests: cells are independent) — each thread replays, in trace order,
    the hits of its shard; the partition of the batch by shard is NOT timed.  This is an upper bound for a
    CPU implementation of these semantics: the reference itself is one shared concurrent map and pays CEL
    evaluation, allocation and cache bookkeeping per request on top.  `single_thread`: the same oracle,
    one thread, whole batch (the parity reference)."""
    import threading

    import numpy as np

    from limitador_amd import workloads as W

    n, batches = _cpu_batches(args)
    # ---- one thread -----------------------------------------------------------------------------
    orc = _cpu_shard(args, 1, 0)
    done, spent, now, i = 0, 0.0, W.NOW0_US, 0
    while spent < budget_s / 2 and i < 64:
        h = batches[i % len(batches)]
        t0 = time.perf_counter()
        orc.check_and_update(h, now, want_first_limited=False)
        spent += time.perf_counter() - t0
        done += n
        now += 1000
        i += 1
    orc.close()
    single = done / spent
    # ---- every host core ----------------------------------------------------------------------------
    logical = os.cpu_count() or 1
    try:
        import psutil

        physical = psutil.cpu_count(logical=False) or logical
    except Exception:
        physical = logical
    if logical < 2:
        return {"value": single, "unit": "decisions/s", "cores": 1, "kind": "port",
                "sample": f"{i} batches x {n} hits, single thread, oracle/limitador_oracle.c"}
    best = None
    tried = {}
    tries = sorted({c for c in (32, 64, physical, logical) if c <= logical})
    for cores in tries:
        r = _cpu_sharded(args, cores, batches, n, budget_s / 2 / len(tries))
        tried[str(cores)] = r[0]
        if best is None or r[0] > best[0]:
            best = (r[0], cores, r[1])
    return {"value": best[0], "unit": "decisions/s", "cores": best[1], "host_physical_cores": physical,
            "host_logical_cpus": logical, "threads_tried": tried, "kind": "port", "single_thread": single,
            # `value` is the BEST of the thread counts tried (the box is shared: the per-count rates are not monotonic from
            # run to run — threads_tried has every one of them, spread = max / min over the counts)
            "best_of": True, "spread_over_thread_counts": (max(tried.values()) / max(1e-9, min(tried.values()))) if tried else None,
            "sample": f"{best[2]} batches x {n} hits over {best[1]} threads (keys hash-sharded, one table per thread, "
                      f"partition not timed) + {i} batches on one thread; {args.keys} keys, zipf {args.zipf}; "
                      f"oracle/limitador_oracle.c (hash-map table, no CEL/moka/tracing)"}


def _cpu_sharded(args, cores, batches, n, budget_s):
    """decisions/s of `cores` oracle tables, one thread each, keys hash-sharded -> (rate, batches timed)."""
    import threading

    import numpy as np

    import oracle
    from limitador_amd import workloads as W

    # the universe once, partitioned by owner (a hash of the key), one oracle table per thread
    uni = W.universe_rows(args.keys)
    owner = W.splitmix64(uni["key"]) % np.uint64(cores)
    order = np.argsort(owner, kind="stable")
    uni = uni[order]
    bounds = np.searchsorted(owner[order], np.arange(cores + 1, dtype=np.uint64))
    shards = [None] * cores

    def build(t):
        o = oracle.OracleStorage()
        o.set_limits([(W.MAX_VALUE, W.WINDOW_S)])
        c = uni[bounds[t]:bounds[t + 1]]
        o.load_cells(c["key"], c["limit"], c["value"], c["expiry_us"])
        shards[t] = o

    ths = [threading.Thread(target=build, args=(t,)) for t in range(cores)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    del uni
    parts = []
    for h in batches:
        owner = W.splitmix64(h["key"]) % np.uint64(cores)
        order = np.argsort(owner, kind="stable")
        hs = h[order]
        bounds = np.searchsorted(owner[order], np.arange(cores + 1, dtype=np.uint64))
        parts.append([np.ascontiguousarray(hs[bounds[t]:bounds[t + 1]]) for t in range(cores)])
    # one C thread per shard (oracle.bench_sharded: pthreads, a barrier between batches): a short run to size the
    # timed one to the budget
    warm = oracle.bench_sharded(shards, parts, 2, W.NOW0_US)
    reps = int(max(4, min(2000, budget_s / max(warm / 2, 1e-6))))
    sec = oracle.bench_sharded(shards, parts, reps, W.NOW0_US + 10_000)
    for o in shards:
        o.close()
    return n * reps / sec, reps


def secondary(args, eng, dev, gen):
    """The other shapes of BASELINE.json next to the headline (N = 1 only): each is a measured rate of the same
    build, none of them is `value`."""
    import numpy as np
    import torch

    import oracle
    from limitador_amd import workloads as W
    from limitador_amd.engine import Engine

    out = {}
    now = [W.NOW0_US + 10_000_000]

    def run_device(e, batches, n, steps, depth=3, jump_every=0, jump_us=0):
        v = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(4)]
        torch.cuda.synchronize()
        pending = 0
        t0 = time.perf_counter()
        for i in range(steps):
            e.submit_device(batches[i % len(batches)].data_ptr(), n, now[0], v[i & 3].data_ptr())
            now[0] += jump_us if jump_every and i % jump_every == jump_every - 1 else 1000
            if pending == depth - 1:
                e.collect()
            else:
                pending += 1
        while pending:
            e.collect()
            pending -= 1
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    # -- a long run of the headline workload (1000 steps over 100 distinct batches)
    cdf = W.torch_zipf_cdf(args.keys, dev, args.zipf) if args.zipf > 0 else None
    many = [W.torch_batch(args.keys, args.batch, dev, gen, cdf) for _ in range(100)]
    run_device(eng, many, args.batch, 20)
    dt = run_device(eng, many, args.batch, 1000)
    out["headline_1000_steps"] = {"decisions_per_s": args.batch * 1000 / dt, "ms_per_step": dt}
    # -- the same workload with the windows ENDING inside the run: the clock jumps 40 s behind every 8th batch (windows
    #    of 60 s, pre-populated expiries 30-60 s ahead), so every touched cell is reset again and again
    #    (atomic_expiring_value.rs:36-42,87-99: value = delta, expiry = now + ttl — a 16-byte write-back instead of 8)
    #    and the hot keys leave saturation; the denials of the last batch say where the run ended
    run_device(eng, many, args.batch, 16, jump_every=8, jump_us=40_000_000)
    vlast = torch.empty(args.batch, dtype=torch.uint8, device=dev)
    dt = run_device(eng, many, args.batch, 200, jump_every=8, jump_us=40_000_000)
    eng.check_and_update_device(many[0].data_ptr(), args.batch, now[0], vlast.data_ptr())
    out["headline_with_expiry"] = {"decisions_per_s": args.batch * 200 / dt, "ms_per_step": dt / 200 * 1e3,
                                   "clock": "+1 ms per batch, +40 s behind every 8th (windows of 60 s)",
                                   "denied_in_a_batch_after_the_run": int(vlast.sum().item())}
    # -- the headline at the OTHER table sizing (ADVICE r04 / VERDICT r04 #9): rounds 1-3 sized the table at load 0.30
    #    (2^25 cells, 1.07 GB for 10 M keys), round 4 on at 0.15 (2^26 cells, 2.1 GB: shorter probe chains).  Both belong in
    #    the record: same keys, same batches, same 20-step window as the driver's line, on a second engine of half the
    #    capacity (`--cap-mult 0.5`, run as the headline, gives the same figure)
    try:
        cap2 = eng.stats()["capacity_cells"] // 2
        e2 = Engine(capacity_cells=cap2, max_batch_hits=args.batch, device=eng.device)
        e2.set_limits([(W.MAX_VALUE, W.WINDOW_S)])
        rows = W.torch_universe_rows(args.keys, dev)
        torch.cuda.synchronize()
        for lo in range(0, rows.shape[0], 1 << 20):
            part = rows[lo:lo + (1 << 20)].contiguous()
            e2.load_cells_device(part.data_ptr(), part.shape[0])
        del rows
        torch.cuda.synchronize()
        run_device(e2, many, args.batch, 12)
        dt20 = run_device(e2, many[12:], args.batch, 20)
        dt200 = run_device(e2, many, args.batch, 200)
        out["headline_load_0.30"] = {"table_capacity_cells": cap2, "table_bytes": cap2 * 32, "table_load": round(args.keys / cap2, 3),
                                     "decisions_per_s_20_steps": args.batch * 20 / dt20, "ms_per_step_20_steps": dt20 / 20 * 1e3,
                                     "decisions_per_s_200_steps": args.batch * 200 / dt200, "ms_per_step_200_steps": dt200 / 200 * 1e3,
                                     "note": "the round-1..3 sizing of the same workload; the headline line is at "
                                             f"load {round(args.keys / (2 * cap2), 3)}"}
        e2.close()
    except Exception as ex:  # noqa: BLE001
        out["headline_load_0.30"] = {"error": str(ex)[:200]}
    del many
    # -- uniform keys, same table and batch size (no hot keys: every hit reads and writes a cell)
    uni = [W.torch_batch(args.keys, args.batch, dev, gen, None) for _ in range(10)]
    run_device(eng, uni, args.batch, 10)
    dt = run_device(eng, uni, args.batch, 50)
    out["uniform_10M_keys_1M_hits"] = {"decisions_per_s": args.batch * 50 / dt, "ms_per_step": dt / 50 * 1e3}
    # -- the boundary handing over HOST buffers (rl_check_and_update_batch: 21 B/hit over PCIe, pageable memory)
    rng = np.random.default_rng(W.SEED)
    cdf_np = W.zipf_cdf(args.keys, args.zipf) if args.zipf > 0 else None
    hb = [W.zipf_batch(args.keys, args.batch, rng, cdf_np) if cdf_np is not None else W.uniform_batch(args.keys, args.batch, rng)
          for _ in range(2)]
    eng.check_and_update(hb[0], now[0], want_first_limited=False)
    t0 = time.perf_counter()
    for i in range(6):
        eng.check_and_update(hb[i & 1], now[0], want_first_limited=False)
        now[0] += 1000
    dt = time.perf_counter() - t0
    out["host_buffers_1M_hits"] = {"decisions_per_s": args.batch * 6 / dt, "ms_per_call": dt / 6 * 1e3,
                                   "note": "rl_check_and_update_batch from pageable host arrays, PCIe inclusive"}
    # the same call with the caller's arrays pinned in place once (rl_host_register: what a binding does with the
    # staging buffers it reuses): the copies are DMA from / into the caller's pages
    try:
        vout = np.empty(args.batch, dtype=np.uint8)
        for a in (hb[0], hb[1], vout):
            eng.host_register(a)
        eng.check_and_update(hb[0], now[0], want_first_limited=False, verdict_out=vout)
        now[0] += 1000
        t0 = time.perf_counter()
        for i in range(6):
            eng.check_and_update(hb[i & 1], now[0], want_first_limited=False, verdict_out=vout)
            now[0] += 1000
        dt = time.perf_counter() - t0
        for a in (hb[0], hb[1], vout):
            eng.host_unregister(a)
        out["host_buffers_1M_hits_registered"] = {"decisions_per_s": args.batch * 6 / dt, "ms_per_call": dt / 6 * 1e3,
                                                  "note": "the same arrays after rl_host_register (pinned in place), PCIe inclusive"}
    except Exception as ex:  # noqa: BLE001
        out["host_buffers_1M_hits_registered"] = {"error": str(ex)[:200]}
    del uni, hb
    # -- BASELINE.json configs[1]: 1 M keys, uniform 64 k-hit batches
    e1 = Engine(capacity_cells=1 << 22, max_batch_hits=1 << 16, device=eng.device)
    e1.set_limits([(W.MAX_VALUE, W.WINDOW_S)])
    rows = W.torch_universe_rows(1 << 20, dev)
    torch.cuda.synchronize()
    e1.load_cells_device(rows.data_ptr(), rows.shape[0])
    b1 = [W.torch_batch(1 << 20, 1 << 16, dev, gen, None) for _ in range(50)]
    run_device(e1, b1, 1 << 16, 50)
    dt = run_device(e1, b1, 1 << 16, 500)
    out["configs1_1M_keys_uniform_64k_hits"] = {"decisions_per_s": (1 << 16) * 500 / dt, "us_per_batch": dt / 500 * 1e6}
    e1.close()
    del rows, b1
    # -- BASELINE.json configs[0]: 3 limits / 1 namespace, 10 k sequential check_and_update calls, in the reference's two
    #    shapes (SURVEY.md §8d "Config #1"), the single-thread oracle beside each:
    #    A  limitador/benches/bench.rs:526-570 — three limits of one namespace with identical conditions,
    #       max_value = u64::MAX, seconds = i * 60 + 10: every call meets all three (k = 3) and is ADMITTED (three
    #       write-backs per call), through the C++ mirror of the trait (rls_check_and_update, one request per call);
    #    B  limitador-server/sandbox/limits.yaml:1-22 as written (two conditions per limit) through the wire path
    #       (rli_serve_batch, one serialized RateLimitRequest per call): a GET /json request meets ONLY the 50000 / 10 s
    #       limit (k = 1); the clock advances 5 ms per call, so the window restarts every 2000 calls, and hits_addend 30
    #       makes the tail of every window OVER_LIMIT (2000 x 30 > 50000).
    try:
        from limitador_amd.host_storage import HostStorage

        U64 = (1 << 64) - 1
        hs = HostStorage(capacity_cells=1 << 12, max_batch_hits=1 << 10, device=eng.device)
        hs.set_clock(W.NOW0_US)
        lims = [("0", U64, i * 60 + 10, ("cond_0 == '1'",), (), None) for i in range(3)]
        for la in lims:
            hs.add_counter(la)
        ctrs = [(la, ()) for la in lims]
        hs.check_and_update_repeat(ctrs, 1, 200)
        sec, limited = hs.check_and_update_repeat(ctrs, 1, 10_000)
        hs.close()
        orc = oracle.OracleStorage()
        orc.set_limits([(U64, i * 60 + 10) for i in range(3)])
        hits = np.zeros(30_600, dtype=oracle.HIT_DTYPE)  # the same 200 warm-up calls first, so that the counts compare
        for q in range(3):
            orc.add_counter(q | oracle.SIMPLE_FLAG)
            hits["key"][q::3], hits["limit"][q::3], hits["delta"][q::3] = 7_000_000 + q, q | oracle.SIMPLE_FLAG, 1
        off = (np.arange(10_201) * 3).astype(np.uint32)
        orc.check_and_update(hits[:600], W.NOW0_US, req_off=off[:201])
        t0 = time.perf_counter()
        v, _f, _r, _e = orc.check_and_update(hits[600:], W.NOW0_US, req_off=off[:10_001])
        osec = time.perf_counter() - t0
        orc.close()
        out["configs0_bench_rs_shape_10k_sequential_calls"] = {
            "gpu_calls_per_s": 10_000 / sec, "gpu_us_per_call": sec / 10_000 * 1e6, "gpu_limited": int(limited),
            "cpu_oracle_calls_per_s": 10_000 / osec, "cpu_limited": int(v.sum()), "counters_per_call": 3,
            "note": "limitador/benches/bench.rs:526-570 shape (3 limits, identical conditions, max u64::MAX, seconds i*60+10): "
                    "every call admitted, three cells written per call; one request per call through the C++ mirror of the "
                    "trait (rls_check_and_update), answered by a lingering k_gen_serve through a host-mapped mailbox "
                    "(RL_SERVE=0: one launch per call)"}
    except Exception as ex:  # the host mirror is optional plumbing for this leg
        out["configs0_bench_rs_shape_10k_sequential_calls"] = {"error": str(ex)[:200]}
    # -- where the device overtakes the CPU: ONE blocking call from host arrays (PCIe inclusive, the
    #    boundary a binding that does not pipeline uses) against ONE thread of the C oracle on the same hits, batch size by
    #    batch size; 1 M keys (Zipf-0.99), single-counter requests.  crossover_hits_per_call = the smallest size from which
    #    the device call is never slower again.  (Pipelined, device-resident batches are the headline; per-request calls:
    #    configs0_* above.)
    try:
        nk = 1 << 20
        ex_ = Engine(capacity_cells=1 << 22, max_batch_hits=1 << 18, device=eng.device)
        ox = oracle.OracleStorage()
        ex_.set_limits([(W.MAX_VALUE, W.WINDOW_S)])
        ox.set_limits([(W.MAX_VALUE, W.WINDOW_S)])
        cells = W.universe_rows(nk)
        ex_.load_cells(cells)
        ox.load_cells(cells["key"], cells["limit"], cells["value"], cells["expiry_us"])
        cdf_x = W.zipf_cdf(nk, args.zipf if args.zipf > 0 else 0.99)
        rng_x = np.random.default_rng(W.SEED + 5)
        rows, t_now = [], W.NOW0_US
        # (the device idles while the oracle's table is loaded above, and the first ~50 ms of tiny launches after an idle
        # stretch run at idle clocks: 200 us per 16-hit call in the first row, whatever size came first — ramp up first)
        hw = W.zipf_batch(nk, 64, rng_x, cdf_x)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.15:
            ex_.check_and_update(hw, t_now, want_first_limited=False)
        for n in (16, 64, 256, 1024, 4096, 16384, 65536, 262144):
            hx = [W.zipf_batch(nk, n, rng_x, cdf_x) for _ in range(4)]
            reps = max(4, min(200, 400_000 // n))
            for _ in range(20):  # (the first size used to carry the engine's first tiny launches: 219 us per 16-hit call)
                ex_.check_and_update(hx[0], t_now, want_first_limited=False)
            t0 = time.perf_counter()
            for i in range(reps):
                ex_.check_and_update(hx[i & 3], t_now + 1 + i, want_first_limited=False)
            g_us = (time.perf_counter() - t0) / reps * 1e6
            ox.check_and_update(hx[0], t_now)
            t0 = time.perf_counter()
            for i in range(reps):
                ox.check_and_update(hx[i & 3], t_now + 1 + i)
            c_us = (time.perf_counter() - t0) / reps * 1e6
            t_now += 1000
            rows.append({"hits_per_call": n, "gpu_us_per_call": round(g_us, 2), "cpu_1_thread_us_per_call": round(c_us, 2)})
        ex_.close()
        ox.close()
        cross = None
        for i, r in enumerate(rows):
            if all(q["gpu_us_per_call"] <= q["cpu_1_thread_us_per_call"] for q in rows[i:]):
                cross = r["hits_per_call"]
                break
        out["crossover_blocking_host_calls_vs_one_cpu_thread"] = {
            "rows": rows, "crossover_hits_per_call": cross,
            "note": "rl_check_and_update_batch from pageable host arrays, one call at a time (no pipelining), against "
                    "lo_check_and_update_batch on one host thread; below the crossover the CPU answers sooner"}
    except Exception as ex:  # noqa: BLE001
        out["crossover_blocking_host_calls_vs_one_cpu_thread"] = {"error": str(ex)[:200]}
    try:
        from limitador_amd.ingest import Ingest

        def pb_str(field, b):  # (a length-delimited protobuf field; every length here is < 128)
            return bytes([field << 3 | 2, len(b)]) + b

        def rls_request(domain, entries, hits_addend):
            d = b"".join(pb_str(1, pb_str(1, k.encode()) + pb_str(2, v.encode())) for k, v in entries)
            return pb_str(1, domain.encode()) + pb_str(2, d) + bytes([3 << 3, hits_addend])

        e0 = Engine(capacity_cells=1 << 12, max_batch_hits=1 << 10, device=eng.device)
        g = Ingest()
        sandbox = [("test_namespace", 10, 60, ["descriptors[0]['req.method'] == 'GET'", "descriptors[0]['req.path'] != '/json'"]),
                   ("test_namespace", 5, 60, ["descriptors[0]['req.method'] == 'POST'", "descriptors[0]['req.path'] != '/json'"]),
                   ("test_namespace", 50000, 10, ["descriptors[0]['req.method'] == 'GET'", "descriptors[0]['req.path'] == '/json'"])]
        for ns, mx, secs, conds in sandbox:
            g.add_limit(ns, mx, secs, conds, [])
        g.install(e0)
        ADD, STEP_US = 30, 5000
        prep = g.prepare_batch([rls_request("test_namespace", [("req.method", "GET"), ("req.path", "/json")], ADD)])
        t_now = W.NOW0_US
        for _ in range(200):
            g.serve_prepared(e0, prep, t_now)
            t_now += STEP_US
        gpu_limited = 0
        t0 = time.perf_counter()
        for _ in range(10_000):
            g.serve_prepared(e0, prep, t_now)
            gpu_limited += prep["status"][0] == 1
            t_now += STEP_US
        sec_b = time.perf_counter() - t0
        g.close()
        e0.close()
        orc = oracle.OracleStorage()
        orc.set_limits([(10, 60), (5, 60), (50000, 10)])
        orc.add_counter(2 | oracle.SIMPLE_FLAG)
        hits = np.zeros(10_200, dtype=oracle.HIT_DTYPE)
        hits["key"], hits["limit"], hits["delta"] = 7_000_002, 2 | oracle.SIMPLE_FLAG, ADD
        nows = (W.NOW0_US + np.arange(10_200, dtype=np.uint64) * np.uint64(STEP_US)).astype(np.uint64)
        offb = np.arange(10_201, dtype=np.uint32)
        orc.check_and_update(hits[:200], W.NOW0_US, req_off=offb[:201], req_now_us=nows[:200])
        t0 = time.perf_counter()
        v, _f, _r, _e = orc.check_and_update(hits[200:], W.NOW0_US, req_off=offb[:10_001], req_now_us=nows[200:])
        osec = time.perf_counter() - t0
        orc.close()
        out["configs0_sandbox_limits_10k_sequential_calls"] = {
            "gpu_calls_per_s": 10_000 / sec_b, "gpu_us_per_call": sec_b / 10_000 * 1e6, "gpu_limited": int(gpu_limited),
            "cpu_oracle_calls_per_s": 10_000 / osec, "cpu_limited": int(v.sum()), "counters_per_call": 1,
            "note": "limitador-server/sandbox/limits.yaml:1-22 as written, a GET /json RateLimitRequest (hits_addend 30) per "
                    "rli_serve_batch call: wire decode + device matching (only the 50000 / 10 s limit applies) + "
                    "check_and_update + response bytes; clock + 5 ms per call, the window restarts every 2000 calls; the "
                    "figure includes the ctypes call from Python (the oracle's is the bare cell arithmetic, no matching)"}
    except Exception as ex:
        out["configs0_sandbox_limits_10k_sequential_calls"] = {"error": str(ex)[:200]}
    # -- BASELINE.json configs[4] shape on one GPU: 4 namespaces x 8 limits, 1 M requests -> ~3.1 M counters per call,
    #    limit matching + key derivation on the device + the multi-counter resolver (scripts/bench_match.py, own process)
    try:
        import subprocess

        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "bench_match.py"),
                            "--steps", "10"], capture_output=True, text=True, timeout=180)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        m = json.loads(line)
        gbps = m["counters_per_batch"] * ALGO_BYTES_TOTAL / (m["ms_per_step"] * 1e-3) / 1e9
        out["configs4_shape_match_and_check_1M_requests"] = {
            "requests_per_s": m["requests_per_s"], "ms_per_call": m["ms_per_step"], "counters_per_call": m["counters_per_batch"],
            # the same 49 B per (request x counter) as the headline, over the whole call (matcher + resolver)
            "achieved_GBps_49B": gbps, "frac_of_8TBps": gbps / HBM_PEAK_GBPS,
            "note": "rl_match_and_check_batch_device: matcher (k_match_count2 / scan2 / fill2) + general resolver "
                    "(round 1: 2.03 ms per call, round 5: 0.426-0.438; kernel by kernel with PMC traffic: "
                    "profiles/r06a_general_kernels.md, r06f_general_kernels.md)"}
    except Exception as ex:
        out["configs4_shape_match_and_check_1M_requests"] = {"error": str(ex)[:200]}
    # -- multi-counter requests whose counters are sharded BY KEY (SURVEY.md 8e "k > 1"): rl_sharded_check_requests_device at world 1
    #    over the library's own RCCL communicator (the protocol's own cost: exchanges, blind round groups, the veto gather, the gated
    #    commit; scripts/bench_sharded_requests.py, own process, bounded: a first RCCL bring-up on a fresh box can take minutes;
    #    profiles/r06_key_sharded.md)
    try:
        import subprocess

        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "bench_sharded_requests.py"),
                            "262144", "50", "rccl"], capture_output=True, text=True, timeout=120)
        m = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        out["key_sharded_multi_counter_262144_requests_x3"] = {
            "ms_per_step": m["ms_per_step"], "requests_per_s": m["requests_per_s"], "counters_per_step": m["counters_per_step"],
            "rounds": m["rounds"][-1], "note": m["what"] + " (round 5: 0.455 ms; profiles/r06_key_sharded.md; the in-process "
                                               "transport of the tests synchronises twice per exchange: 0.43 ms)"}
    except Exception as ex:
        out["key_sharded_multi_counter_262144_requests_x3"] = {"error": str(ex)[:200]}
    # -- the wire path: serialized RateLimitRequests -> verdicts -> RateLimitResponse bytes, per batch size,
    #    with the host's dictionaries and with the messages decoded and the keys hashed on the device (scripts/bench_rls.py)
    try:
        import subprocess

        wp = {}
        for keys in ("exact", "hashed"):
            r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "bench_rls.py"), keys],
                               capture_output=True, text=True, timeout=240)
            m = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            wp[keys] = {n: dict({"codes_only_ms": v["codes_only"]["p50_ms"], "with_headers_ms": v["with_headers"]["p50_ms"],
                                 "requests_per_s": v["codes_only"]["requests_per_s"]},
                                **({"kuadrant_check_ms": v["kuadrant_check"]["p50_ms"], "kuadrant_report_ms": v["kuadrant_report"]["p50_ms"]}
                                   if "kuadrant_check" in v else {}),
                                # two rli_serve_batch calls in flight on one ingest / engine (round 6): sustained ms per batch
                                **({"with_headers_two_in_flight_ms_per_batch": v["with_headers_two_in_flight"]["ms_per_batch_sustained"],
                                    "with_headers_two_in_flight_requests_per_s": v["with_headers_two_in_flight"]["requests_per_s"],
                                    "with_headers_two_in_flight_call_p50_ms": v["with_headers_two_in_flight"]["call_p50_ms"]}
                                   if "with_headers_two_in_flight" in v else {}),
                                **({"with_headers_four_in_flight_ms_per_batch": v["with_headers_four_in_flight"]["ms_per_batch_sustained"],
                                    "with_headers_four_in_flight_requests_per_s": v["with_headers_four_in_flight"]["requests_per_s"],
                                    "with_headers_four_in_flight_call_p50_ms": v["with_headers_four_in_flight"]["call_p50_ms"]}
                                   if "with_headers_four_in_flight" in v else {}))
                        for n, v in m["sizes"].items() if n in ("256", "32768", "262144")}
        out["wire_path_rli_serve_batch"] = dict(wp, note="p50 of the C call per batch of N serialized messages (4 namespaces x 8 limits, "
                                                "Zipf users); exact = host dictionaries + packed ids, hashed = RLI_KEYS_HASHED "
                                                "(messages decoded on the device, keys = hash of the canonical key bytes); with "
                                                "headers the response bytes are built on the device from 4096 messages on "
                                                "(rl_resp.hpp); kuadrant_check / kuadrant_report = rli_serve_batch_op, the Kuadrant "
                                                "service's CheckRateLimit (is_rate_limited, read-only) and Report (update_counters); "
                                                "with_headers_two / four_in_flight_* = two / four threads calling rli_serve_batch back to back on "
                                                "the same ingest / engine, each call on one of the engine's four serving sets "
                                                "(profiles/r06_wire_two_in_flight.md)")
    except Exception as ex:
        out["wire_path_rli_serve_batch"] = {"error": str(ex)[:200]}
    # -- the streaming maintenance kernels over the headline's table (LAST: they change it): a sweep that finds nothing
    #    expired is a pure scan of the table (the kernel expected near the HBM roofline); a sweep as a
    #    command between two batches in flight; a compaction (rehash of every live cell into a fresh table)
    try:
        table_bytes = eng.stats()["capacity_cells"] * 32
        eng.sweep_expired(W.NOW0_US)  # (warm)
        t0 = time.perf_counter()
        for _ in range(5):
            eng.sweep_expired(W.NOW0_US)
        dt = (time.perf_counter() - t0) / 5
        b2 = [W.torch_batch(args.keys, args.batch, dev, gen, None) for _ in range(2)]
        v2 = [torch.empty(args.batch, dtype=torch.uint8, device=dev) for _ in range(2)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.submit_device(b2[0].data_ptr(), args.batch, now[0], v2[0].data_ptr())
        eng.sweep_expired_submit(W.NOW0_US)
        eng.submit_device(b2[1].data_ptr(), args.batch, now[0] + 1000, v2[1].data_ptr())
        eng.collect()
        swept = eng.sweep_expired_collect()
        eng.collect()
        torch.cuda.synchronize()
        dt_mix = time.perf_counter() - t0
        t0 = time.perf_counter()
        eng.compact()
        dt_c = time.perf_counter() - t0
        # a sweep that FINDS something: the clock jumps 100 s (every window of the table is over), seven uniform batches
        # restart the windows of the ~half of the cells they touch, and the sweep right behind them removes the other
        # half; then compact the table the tombstones are in (in place: k_compact_mark / k_compact_shift)
        now[0] += 100_000_000
        b7 = [W.torch_batch(args.keys, args.batch, dev, gen, None) for _ in range(7)]
        run_device(eng, b7, args.batch, 7)
        live0 = eng.stats()["live_cells"]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        removed = eng.sweep_expired(now[0])
        dt_x = time.perf_counter() - t0
        t0 = time.perf_counter()
        eng.compact()
        dt_xc = time.perf_counter() - t0
        live1 = eng.stats()["live_cells"]
        slots = table_bytes // 32
        out["sweep_and_compact_10M_keys"] = {
            "sweep_ms": dt * 1e3, "sweep_GBps": table_bytes / dt / 1e9, "sweep_frac_of_8TBps": table_bytes / dt / 1e9 / HBM_PEAK_GBPS,
            "sweep_GBps_24B_per_slot": slots * 24 / dt / 1e9,
            "batch_sweep_batch_in_flight_ms": dt_mix * 1e3, "swept_between_batches": int(swept),
            "compact_ms": dt_c * 1e3, "table_bytes": table_bytes,
            "expired_sweep": {"live_before": int(live0), "removed": int(removed), "removed_frac": removed / max(live0, 1),
                              "live_after_compact": int(live1), "sweep_ms": dt_x * 1e3,
                              "sweep_GBps_24B_per_slot": slots * 24 / dt_x / 1e9, "sweep_GBps_32B_cells": table_bytes / dt_x / 1e9,
                              "compact_ms": dt_xc * 1e3},
            "note": "rl_sweep_expired (host call incl. its status read-back) over the 32-byte cells (SURVEY.md 8d counts 24 B "
                    "per slot: both rates given); rl_sweep_expired_submit between two 1 M-hit batches in flight; rl_compact of a "
                    "table without tombstones; expired_sweep: clock + 100 s, seven uniform 1 M-hit batches restart the windows of the "
                    "cells they touch, the sweep removes every other cell (expiry <= now) and — tombstones now exceed capacity / 8 — compacts "
                    "the table in place inside the same call (expired_sweep.sweep_ms = scan + 5 M tombstone stores + compaction); "
                    "expired_sweep.compact_ms: rl_compact of the already compact table"}
    except Exception as ex:
        out["sweep_and_compact_10M_keys"] = {"error": str(ex)[:200]}
    return out


_RCCL_PROBE = r"""
import os, sys, socket, datetime
import torch, torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
# (under torch.distributed.run the parent's environment says "use the agent's store": this probe is a world of its own)
for k in [k for k in os.environ if k.startswith("TORCHELASTIC_") or k in ("GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE")]:
    os.environ.pop(k)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                  RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
dev = torch.device("cuda", int(sys.argv[1])); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=200))
t = torch.ones(1024, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
dist.destroy_process_group(); print("rccl ok")
"""


class GpuPlatform:
    """Everything main() needs from the machine.  The product run uses this class as it is; tests/test_bench_sharded_cpu.py
    runs the same main() at world 2 over gloo with a CPU stand-in (same methods) so that the routed run's ORCHESTRATION —
    phases, watchdog, fall-back agreement, timing protocol, the JSON line — is exercised without a GPU."""

    dist_backend = "nccl"

    def __init__(self, local_rank):
        import torch

        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a MI355X: the engine has no CPU path")
        torch.cuda.set_device(local_rank)
        self.local_rank = local_rank
        self.device = torch.device("cuda", local_rank)

    def sync(self):
        import torch

        torch.cuda.synchronize()

    def make_engine(self, capacity_cells, max_batch_hits):
        from limitador_amd.engine import Engine

        return Engine(capacity_cells=capacity_cells, max_batch_hits=max_batch_hits, device=self.local_rank)

    def warm_collectives(self, wd, attempts=2, timeout_s=240.0):
        """The first RCCL initialisation of a fresh box, in a PROCESS OF ITS OWN that can be killed (what
        tests/conftest.py::rccl_ready does for the suite): a world-1 communicator + one all-reduce on this rank's GPU.
        Afterwards the box's page cache holds librccl's device code, and the in-process bring-ups that follow (torch's
        communicator, then rl_sharded_create_rccl, which binds to the copy torch mapped) take seconds.  -> (ok, note)."""
        import subprocess

        note = ""
        for attempt in range(attempts):
            wd.kick(f"RCCL warm-up in a subprocess (attempt {attempt + 1})", limit_s=timeout_s + 60)
            t0 = time.perf_counter()
            try:
                p = subprocess.run([sys.executable, "-c", _RCCL_PROBE, str(self.local_rank)], stdout=subprocess.PIPE,
                                   stderr=subprocess.STDOUT, text=True, timeout=timeout_s,
                                   env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
                if p.returncode == 0 and "rccl ok" in p.stdout:
                    return True, f"{time.perf_counter() - t0:.1f} s (attempt {attempt + 1})"
                note = p.stdout.strip()[-200:]
            except subprocess.TimeoutExpired:
                note = f"no world-1 communicator within {timeout_s:.0f} s"
        return False, note

    def make_torch_sharded(self, eng, group, max_local_hits):
        from limitador_amd.sharded import ShardedEngine

        return ShardedEngine(eng, group, self.device, max_local_hits=max_local_hits)

    def abi_unique_id(self):
        from limitador_amd import sharded_abi

        return sharded_abi.unique_id()

    def make_abi_sharded(self, eng, world, rank, max_slice_hits, unique_id):
        from limitador_amd import sharded_abi

        return sharded_abi.Sharded(eng, world, rank, max_slice_hits, unique_id=unique_id)


def main(argv=None, platform=None):
    args = parse(argv)
    import datetime

    import torch
    import torch.distributed as dist

    from limitador_amd import workloads as W
    from limitador_amd.watchdog import Watchdog

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    plat = platform if platform is not None else GpuPlatform(local_rank)
    dev = plat.device
    sharded = world > 1 or args.force_sharded
    # Progress-based: `wd.kick(phase)` at every phase boundary, every loaded chunk, every collected slice.  A single-GPU
    # unrouted run has nobody to wait for and runs without it.
    wd = Watchdog(limit_s=args.stall_seconds, name=f"bench.py rank {rank}")
    phases = {}  # phase -> seconds (rank 0's; on the line as config.bringup_s for routed runs)
    t_phase = [time.perf_counter(), "start"]

    def phase(name, limit_s=None):
        t = time.perf_counter()
        phases[t_phase[1]] = round(phases.get(t_phase[1], 0.0) + t - t_phase[0], 3)
        t_phase[0], t_phase[1] = t, name
        wd.kick(name, limit_s)

    if sharded:
        wd.start()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        warm_note = None
        if hasattr(plat, "warm_collectives"):
            ok, warm_note = plat.warm_collectives(wd)
            if not ok:  # not fatal: the in-process bring-up below gets its own --init-seconds
                sys.stderr.write(f"bench.py rank {rank}: RCCL warm-up did not finish ({warm_note}); trying in-process\n")
        phase("init_process_group", args.init_seconds)
        kw = {"device_id": dev} if dev.type == "cuda" else {}
        dist.init_process_group(plat.dist_backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=max(60.0, args.init_seconds)), **kw)
        phase("first collective", args.init_seconds)
        t = torch.ones(1, device=dev)
        dist.all_reduce(t)  # (with device_id the communicator exists already; without it this creates it)
        plat.sync()

    n_keys_total = args.keys * world
    # 32-byte cells at load <= 0.15: 10 M keys -> 2^26 cells = 2.1 GB of the GPU's 288 (215 bytes of table per key; the
    # reference keeps a Counter — limit, set variables, strings — and an AtomicExpiringValue per key).  The capacity is the
    # deployment's choice (rl_config.capacity_cells); what it buys here, measured on MI355X (profiles/r04h_replay_memory_path.md):
    # load 0.30 / 0.15 / 0.075 (2^25 / 2^26 / 2^27 cells) -> replay launch 39.1 / 36.4 / 36.0 us, step (20 steps) 50.7 / 48.3 /
    # 48.2 us.  --cap-mult 0.5 is the round-3 sizing.
    cap = 1 << (int(n_keys_total / world * 4.4 * args.cap_mult - 1).bit_length())
    max_batch = int(args.batch * 2) if sharded else args.batch
    phase("engine")
    eng = plat.make_engine(cap, max_batch)
    eng.set_limits([(W.MAX_VALUE, W.WINDOW_S)])

    # ---- pre-populate this rank's shard -----------------------------------------------------
    # The universe is key(i) = splitmix64(i), i < N x keys; a key's owner is a hash of the KEY (owner_of, rl_cell.hpp), so a
    # rank finds its share by filtering.  It walks the index range in pieces of 4 M (generated, filtered and loaded one
    # at a time: 128 MB of temporaries whatever N is) and starts at its own N-th of the range, so that the ranks do not
    # all hash the same piece at the same moment.  Every piece is progress for the watchdog.
    phase("pre-populate")
    piece = 1 << 22
    n_pieces = (n_keys_total + piece - 1) // piece
    first = (rank * n_pieces) // world
    chunk = 1 << 20
    keep = None
    if world > 1:
        from limitador_amd.sharded import owner_mask

        keep = lambda k: owner_mask(k, eng.hash_seed, world, rank)  # noqa: E731
    plat.sync()  # (the engine loads on its own stream)
    n_loaded = 0
    for j in range(n_pieces):
        lo = ((first + j) % n_pieces) * piece
        rows = W.torch_universe_rows(n_keys_total, dev, keep=keep, lo=lo, hi=min(n_keys_total, lo + piece))
        plat.sync()
        for c in range(0, rows.shape[0], chunk):
            part = rows[c:c + chunk].contiguous()
            eng.load_cells_device(part.data_ptr(), part.shape[0])
        n_loaded += rows.shape[0]
        plat.sync()
        del rows
        wd.kick()

    # ---- synthetic batches, resident in HBM before the timed region --------------------------
    phase("batches")
    gen = torch.Generator(device=dev).manual_seed(W.SEED + rank)
    cdf = W.torch_zipf_cdf(n_keys_total, dev, args.zipf) if args.zipf > 0 else None
    total_steps = args.warmup + args.steps
    batches = [W.torch_batch(n_keys_total, args.batch, dev, gen, cdf) for _ in range(total_steps)]
    del cdf
    verdict = torch.empty(args.batch, dtype=torch.uint8, device=dev)
    verdicts = [verdict] + [torch.empty(args.batch, dtype=torch.uint8, device=dev) for _ in range(3)]
    plat.sync()

    sharded_note = ""
    if sharded and args.sharded_impl == "abi":
        # The C entry with its own RCCL communicator.  If ANY rank cannot bring it up (librccl not found or mapped twice,
        # ncclCommInitRank failing on this node's topology, ...) EVERY rank falls back to the torch.distributed driver —
        # the same routed step, driven from Python — and the line says so in config.parallelism.
        from limitador_amd import sharded_abi

        phase("rl_sharded_create_rccl", args.init_seconds)
        sh, why = None, ""
        idt = torch.zeros(sharded_abi.UNIQUE_ID_BYTES, dtype=torch.uint8, device=dev)
        try:
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(plat.abi_unique_id()), dtype=torch.uint8))
        except Exception as ex:  # rank 0 could not even make an id: broadcast zeros, everybody fails below
            why = f"unique_id: {ex}"
        dist.broadcast(idt, 0)
        plat.sync()
        if not why and not bool(idt.any().item()):
            why = "rank 0 could not make a communicator id"
        if not why:
            try:
                sh = plat.make_abi_sharded(eng, world, rank, args.batch, bytes(idt.cpu().numpy()))
            except Exception as ex:
                why = f"rl_sharded_create_rccl: {ex}"
        ok = torch.tensor([1 if sh is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if sh is not None:
                sh.close()
                sh = None
            sys.stderr.write(f"bench.py rank {rank}: falling back to --sharded-impl torch ({why or 'another rank failed'})\n")
            args.sharded_impl = "torch"
            sharded_note = " [fell back from the C-ABI router: " + (str(why)[:160] or "another rank failed") + "]"
    host_submit = [0.0, 0]
    if sharded and args.sharded_impl == "abi":
        from limitador_amd import sharded_abi

        pending = [0]
        if args.depth == 1:
            def step(i, now):
                sh.check_and_update(batches[i].data_ptr(), args.batch, now, verdict.data_ptr())
                wd.kick()
        else:
            def step(i, now):
                sh.submit(batches[i].data_ptr(), args.batch, now, verdicts[i & 3].data_ptr())
                if sh.in_flight == sharded_abi.MAX_IN_FLIGHT:  # (four: the host enqueues a slice ahead of the device)
                    sh.collect()
                    wd.kick()
    elif sharded:
        phase("torch.distributed router")
        sh = plat.make_torch_sharded(eng, dist.group.WORLD, args.batch)
        pending = [0]
        if args.depth == 1:
            def step(i, now):
                sh.check_and_update(batches[i], now, verdict)
                wd.kick()
        else:
            # three ingress slices in flight per rank (routed / applied / returned, see ShardedEngine);
            # every slice has completed when drain() returns
            def step(i, now):
                sh.submit(batches[i], now, verdicts[i % 3])
                if sh.in_flight == 3:
                    sh.collect()
                    wd.kick()
    elif args.depth == 1:
        def step(i, now):
            eng.check_and_update_device(batches[i].data_ptr(), args.batch, now, verdict.data_ptr())
    else:
        # `depth` batches in flight: batch i is enqueued before the host waits for batch i-depth+1, so the
        # device never idles between batches and the partition of batch i+1 (own stream) overlaps k_bkt_step
        # of batch i.  Batches are applied in submission order; every step's batch has completed when the
        # timed region ends (drain()).
        pending = [0]

        def step(i, now):
            t_s = time.perf_counter()
            eng.submit_device(batches[i].data_ptr(), args.batch, now, verdicts[i & 3].data_ptr())
            host_submit[0] += time.perf_counter() - t_s
            host_submit[1] += 1
            if pending[0] == args.depth - 1:
                eng.collect()
            else:
                pending[0] += 1

    def drain():
        if sharded:
            while sh.in_flight:
                sh.collect()
                wd.kick()
            if args.sharded_impl == "abi":
                sh.sync()
        elif args.depth >= 2:
            while pending[0]:
                eng.collect()
                pending[0] -= 1

    now = W.NOW0_US
    phase("warm-up steps")
    for i in range(args.warmup):
        step(i, now)
        now += 1000
    drain()
    # Per-kernel breakdown, OUTSIDE the timed region: every kernel of a batch launched with its own start / stop
    # events, one blocking call per batch (so each kernel runs alone).  The warm-up batches are replayed.
    phase("per-kernel breakdown pass")
    eng.kernel_timing(1)
    eng.kernel_timing_read(reset=True)
    # (a dozen batches, cycling through the warm-up ones: the hot-key set and its promotion threshold adapt over a few
    # generations of three batches each, and a cold engine's first three batches run without any hot set)
    for j in range(12 if args.warmup else 0):
        i = j % max(1, min(args.warmup, 5))
        if sharded:
            step(i, now)
        else:  # one blocking call per batch: every kernel runs alone (no overlap with the next batch's partition)
            eng.check_and_update_device(batches[i].data_ptr(), args.batch, now, verdict.data_ptr())
        now += 1000
    drain()
    kt_all = eng.kernel_timing_read(reset=True)
    # Timed region: events around the dominant kernel only (k_bkt_step), for the roofline.
    eng.kernel_timing(args.timing_mode)
    denied = 0
    phase("timed region")
    plat.sync()
    if sharded:
        dist.barrier()
        plat.sync()
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        step(i, now)
        now += 1000
    drain()
    plat.sync()
    if sharded:
        dist.barrier()
        plat.sync()
    dt = time.perf_counter() - t0
    phase("report")
    kt = eng.kernel_timing_read(reset=True)
    eng.kernel_timing(0)
    if args.depth == 1:
        last = verdict
    else:
        last = (verdicts[(total_steps - 1) % 3] if (sharded and args.sharded_impl != "abi") else verdicts[(total_steps - 1) & 3])
    denied = int(last.sum().item())
    if sharded:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        st = eng.stats()
        decisions = args.batch * world * args.steps
        launches = max(1, kt["launches"])
        timed_all = {k: v / launches for k, v in kt["ms"].items() if v != 0}  # the timed region
        timed = {k: v for k, v in timed_all.items() if k not in NOT_KERNELS and v > 0}
        per = {k: v / max(1, kt_all["launches"]) for k, v in kt_all["ms"].items() if v > 0 and k not in NOT_KERNELS}  # breakdown pass
        per_alone = dict(per)
        per.update(timed)
        hits_per_launch = st["hits"] / max(1, st["batches"])
        # the dominant kernel for the roofline is the one that does the algorithmic work (k_bkt_step: the replay); the
        # partition kernel beside it is overhead and is reported in `pipeline`
        dom = "apply"
        dom_gbps = ALGO_BYTES[dom] * hits_per_launch / (per[dom] * 1e-3) / 1e9 if per.get(dom, 0) > 0 else 0.0
        pipe_ms = sum(per.values())
        kname = {"apply": "k_bkt_step", "part": "k_bkt_part"}.get(dom, "k_" + dom)
        # HBM bytes per launch of that kernel from the PMC passes of the last profiling visit
        # (scripts/gpu_profile.sh -> scripts/summarize_prof.py -> profiles/traffic.json): counters cannot
        # be collected inside this run, so the figure is the committed one for this workload or null.
        traffic = pipeline_traffic = l2_hit = traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        default_workload = (args.keys, args.batch, args.zipf) == (10_000_000, 1_000_000, 0.99)
        if os.path.exists(tpath) and default_workload and world == 1:
            try:
                tj = json.load(open(tpath))
                tk = tj["kernels"]
                traffic_source = {"profile": tj.get("source"), "commit": tj.get("commit"), "file": "profiles/traffic.json"}
                traffic = tk.get(kname, {}).get("hbm_bytes_per_launch")
                l2_hit = tk.get(kname, {}).get("tcc_hit_rate")
                # the whole batch pipeline (partition + hot state + replay), for the traffic / algorithmic ratio
                pipeline_traffic = sum((v.get("hbm_bytes_per_launch") or 0.0) for k, v in tk.items()
                                       if k.startswith("k_bkt_part") or k == "k_bkt_step")
            except Exception:
                traffic = pipeline_traffic = l2_hit = None
        out = {
            "metric": "rate-limit decisions/sec", "value": decisions / dt, "unit": "decisions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "configs[2]: 10M keys/GPU, Zipf-0.99 1M-hit batch/GPU, fixed window 1000/60s, "
                                   "delta=1, single-counter requests" if (args.keys, args.batch, args.zipf) == (10_000_000, 1_000_000, 0.99)
                       else f"{args.keys} keys/GPU, zipf {args.zipf}, {args.batch}-hit batch/GPU",
                       "keys_per_gpu": args.keys, "batch_per_gpu": args.batch, "zipf_s": args.zipf,
                       "table_capacity_cells": cap, "cell_bytes": 32, "table_bytes": cap * 32,
                       "table_load": round(args.keys / cap, 3),
                       "parallelism": (f"hash-sharded x{world}, RCCL all-to-all" + (" behind the C ABI (rl_sharded_*)" if args.sharded_impl == "abi" else " (torch.distributed)") + sharded_note) if sharded else "single GPU",
                       "batches_in_flight": args.depth if not sharded or args.depth == 1 else (4 if args.sharded_impl == "abi" else 3),
                       "overlap": "partition of batches k+1, k+2 (own stream) beside k_bkt_step of batch k" if (not sharded and args.depth >= 2 and os.environ.get("RL_OVERLAP", "1") != "0") else "none",
                       "denied_in_last_batch": denied,
                       # routed runs: seconds rank 0 spent in every phase before the report (bring-up included) and what the
                       # killable RCCL warm-up took; the watchdog's limits that were in force
                       "bringup_s": (dict(phases, rccl_warmup=warm_note, stall_seconds=args.stall_seconds,
                                          init_seconds=args.init_seconds, cells_loaded_rank0=n_loaded) if sharded else None),
                       # every engine knob the process saw (rl_engine_create reads RL_*, the ingest layer RLI_*)
                       "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith(("RL_", "RLI_"))}},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": dom_gbps, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": dom_gbps / HBM_PEAK_GBPS, "traffic": traffic,
                         # SURVEY.md §8(d)'s three figures: algorithmic GB/s (`achieved`), counter-derived GB/s, L2 hit rate
                         "counter_GBps": (traffic / (per[dom] * 1e-3) / 1e9) if (traffic and per.get(dom, 0) > 0) else None,
                         "l2_hit_rate": l2_hit,
                         # traffic / counter_GBps / l2_hit_rate are NOT measured in this run: they are the committed PMC
                         # passes of the profiling visit named here (scripts/gpu_profile.sh + summarize_prof.py)
                         "traffic_measured_at": traffic_source,
                         "pipeline_traffic": pipeline_traffic,
                         "pipeline_traffic_over_algorithmic": (pipeline_traffic / (ALGO_BYTES_TOTAL * hits_per_launch))
                         if pipeline_traffic else None,
                         "algorithmic_bytes_per_hit": ALGO_BYTES[dom], "hits_per_launch": hits_per_launch,
                         "avg_launch_ms": per.get(dom, 0.0),
                         "avg_launch_ms_alone": per_alone.get(dom),
                         # The honest end-to-end: the same algorithmic bytes over the STEP (partition, gaps and pipeline fill
                         # included), not over the dominant kernel's own launch.
                         "step_frac": (ALGO_BYTES_TOTAL * args.batch / (dt / args.steps) / 1e9 / HBM_PEAK_GBPS) if world == 1 else None,
                         # ... and the reachable ceiling: the replay is bound by random TRANSACTIONS, not bytes.  Per 1 M-hit
                         # Zipf-0.99 batch it issues ~1.45 M of them (record-gather lines ~350 k, cell reads ~500 k, write-backs
                         # ~400 k, one-byte verdict stores ~200 k: profiles/r04h_replay_memory_path.md, from the TCP / TCC
                         # counters); the chip's measured random-line rate is 57 G/s (profiles/r04a_random_slope.txt).
                         # random_line_frac = (transactions / 57 G/s) / the kernel's launch time: 1.0 = at the fabric's line rate.
                         "random_line_frac": ((_random_line_model()[0] / _random_line_model()[1]) / (per[dom] * 1e-3))
                         if (default_workload and per.get(dom, 0) > 0) else None,
                         "random_line_model": {"transactions_per_launch": _random_line_model()[0], "lines_per_s": _random_line_model()[1],
                                               "measured_at": _random_line_model()[2]}
                         if default_workload else None,
                         "timed_with": (f"HIP events on {kt['launches']} of the {args.steps} launches of the timed region: "
                                        "the launch carries its own start / stop events (hipExtLaunchKernelGGL), no marker "
                                        "commands around the kernel; the partition of the next batches runs beside it on a "
                                        "second stream; avg_launch_ms_alone: the same kernel in the breakdown pass, one "
                                        "blocking call per batch") if dom in timed
                         else "HIP events in the breakdown pass before the timed region"},
            "pipeline": {"kernel_ms_per_batch": per, "kernel_ms_per_batch_alone": per_alone,
                         "kernel_ms_per_batch_in_pipeline": timed,
                         "apply_stream_idle_ms_per_batch": timed_all.get("apply_gap"),
                         "partition_done_before_apply_ms": timed_all.get("part_slack"), "device_ms_per_batch": pipe_ms,
                         "achieved_GBps_49B": ALGO_BYTES_TOTAL * hits_per_launch / (pipe_ms * 1e-3) / 1e9 if pipe_ms else 0.0,
                         "ordered_hits_per_batch": st["ordered_hits"] / max(1, st["batches"]),
                         "host_submit_us_per_batch": (host_submit[0] / max(1, host_submit[1]) * 1e6)
                         if (not sharded and args.depth >= 2) else None},
        }
        if world == 1 and not args.force_sharded and args.secondary:
            out["secondary"] = secondary(args, eng, dev, gen)
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if sharded and args.sharded_impl == "abi":
        sh.close()
    wd.cancel()
    eng.close()
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
