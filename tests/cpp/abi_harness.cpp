// abi_harness.cpp — the C ABI of include/rl_engine.h driven from C++ with no Python in the loop (what a
// non-Python host of the library looks like), every result checked against the C oracle
// (oracle/limitador_oracle.h: the restatement of limitador/src/storage/in_memory.rs).  Test infrastructure:
// built and run by tests/test_gpu_cpp_harness.py on the GPU box.
//   g++ -O2 -std=c++17 abi_harness.cpp -I include -I oracle -L limitador_amd/lib -lrl_engine -L oracle -llimitador_oracle
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "limitador_oracle.h"
#include "rl_engine.h"

#define CHECK(cond, ...)                                  \
    do {                                                  \
        if (!(cond)) {                                    \
            std::fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); \
            std::fprintf(stderr, __VA_ARGS__);            \
            std::fprintf(stderr, "\n");                   \
            std::exit(1);                                 \
        }                                                 \
    } while (0)
#define RL(call)                                                                                   \
    do {                                                                                           \
        int32_t rc_ = (call);                                                                      \
        CHECK(rc_ == RL_OK, "%s -> %d (%s)", #call, rc_, rl_last_error(eng));                      \
    } while (0)

static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

int main() {
    const uint64_t NOW = 1700000000000000ull, SEC = 1000000ull;
    rl_config cfg{};
    cfg.device = 0;
    cfg.max_batch_hits = 1u << 17;
    cfg.capacity_cells = 1u << 17;
    cfg.max_limits = 16;
    rl_engine* eng = nullptr;
    int32_t rc = rl_engine_create(&cfg, &eng);
    CHECK(rc == RL_OK, "rl_engine_create -> %d", rc);
    // limits: 0, 1 without variables (simple), 2.. with
    const rl_limit_row rows[] = {{100000, 60}, {500, 1}, {30, 10}, {4, 1}, {1000, 60}, {7, 0}, {~0ull, 3600}};
    const uint32_t n_limits = sizeof(rows) / sizeof(rows[0]);
    RL(rl_limits_set(eng, 0, rows, n_limits));
    lo_storage* orc = lo_storage_new();
    std::vector<lo_limit_row> orows(n_limits);
    for (uint32_t i = 0; i < n_limits; ++i) orows[i] = lo_limit_row{rows[i].max_value, rows[i].seconds};
    const uint64_t simple_key[2] = {50000001ull, 50000002ull};
    for (uint32_t l = 0; l < 2; ++l) {
        RL(rl_add_counter(eng, l | RL_SIMPLE, simple_key[l]));
        CHECK(lo_add_counter(orc, l, 0) == 0, "lo_add_counter");
    }
    std::mt19937_64 rng(2024);
    uint64_t now = NOW;
    uint64_t total_hits = 0, total_limited = 0;
    for (int step = 0; step < 12; ++step) {
        // ---- a batch of multi-counter requests (simple counters first), every third one with load_counters,
        //      every fourth with the trait's u64 deltas
        const uint32_t n_req = 200 + (uint32_t)(rng() % 3000);
        std::vector<rl_hit> hits;
        std::vector<uint32_t> off{0};
        std::vector<uint64_t> delta64;
        for (uint32_t r = 0; r < n_req; ++r) {
            const uint32_t d = (rng() % 5 == 0) ? (uint32_t)(rng() % 4) : 1u;
            const uint64_t user = rng() % 7 ? rng() % 40 : rng() % 4000;
            for (uint32_t l = 0; l < 2; ++l)
                if (rng() % 2) hits.push_back(rl_hit{simple_key[l], l | RL_SIMPLE, d});
            const uint32_t k = (uint32_t)(rng() % 4);
            for (uint32_t q = 0; q < k; ++q) {
                const uint32_t l = 2 + (uint32_t)(rng() % (n_limits - 2));
                hits.push_back(rl_hit{splitmix64(l * 1000003ull + user) >> 1, l, d});
            }
            off.push_back((uint32_t)hits.size());
            delta64.push_back(rng() % 50 == 0 ? (1ull << 40) + d : (uint64_t)d);  // now and then a delta beyond u32: Limited
        }
        const bool load = step % 3 == 2, big = step % 4 == 3;
        const size_t nh = hits.size();
        std::vector<uint8_t> v(n_req), wv(n_req);
        std::vector<int32_t> f(n_req), wf(n_req);
        std::vector<uint64_t> rem(nh + 1), exp(nh + 1), wrem(nh + 1), wexp(nh + 1);
        if (big)
            RL(rl_check_and_update_batch_ex(eng, hits.data(), (uint32_t)nh, off.data(), n_req, delta64.data(), nullptr, now, load,
                                            v.data(), f.data(), rem.data(), exp.data()));
        else
            RL(rl_check_and_update_batch(eng, hits.data(), (uint32_t)nh, off.data(), n_req, now, load, v.data(), f.data(),
                                         rem.data(), exp.data()));
        CHECK(lo_check_and_update_batch_ex(orc, orows.data(), n_limits, reinterpret_cast<const lo_hit*>(hits.data()), nh, off.data(),
                                           n_req, big ? delta64.data() : nullptr, nullptr, now, load, wv.data(), wf.data(),
                                           wrem.data(), wexp.data()) == 0, "oracle batch");
        CHECK(std::memcmp(v.data(), wv.data(), n_req) == 0, "step %d: verdicts differ", step);
        CHECK(std::memcmp(f.data(), wf.data(), n_req * sizeof(int32_t)) == 0, "step %d: first_limited differs", step);
        if (load) {
            CHECK(std::memcmp(rem.data(), wrem.data(), nh * 8) == 0, "step %d: remaining differs", step);
            CHECK(std::memcmp(exp.data(), wexp.data(), nh * 8) == 0, "step %d: expires_in differs", step);
        }
        total_hits += nh;
        for (uint32_t r = 0; r < n_req; ++r) total_limited += v[r];
        // ---- is_within_limits / update_counter on a slice of the same hits
        const uint32_t ns = (uint32_t)std::min<size_t>(nh, 300);
        std::vector<uint8_t> w(ns), ww(ns);
        RL(rl_is_within_limits_batch(eng, hits.data(), ns, now, w.data()));
        CHECK(lo_is_within_limits_batch(orc, orows.data(), n_limits, reinterpret_cast<const lo_hit*>(hits.data()), ns, now, ww.data()) == 0,
              "oracle within");
        CHECK(std::memcmp(w.data(), ww.data(), ns) == 0, "step %d: is_within_limits differs", step);
        if (step % 5 == 1) {
            RL(rl_update_counter_batch(eng, hits.data(), ns, now));
            CHECK(lo_update_counter_batch(orc, orows.data(), n_limits, reinterpret_cast<const lo_hit*>(hits.data()), ns, now) == 0,
                  "oracle update");
        }
        now += (step % 2) ? SEC / 3 : 2 * SEC;
        if (step == 6) {  // an explicit sweep, replayed into the oracle
            uint64_t removed = 0;
            RL(rl_sweep_expired(eng, now, &removed));
            CHECK(removed == lo_sweep_expired(orc, now), "sweep: %" PRIu64 " removed", removed);
        }
        if (step == 8) {  // delete_counters of one limit (in_memory.rs:190-195,241-257)
            RL(rl_delete_counters(eng, 2));
            lo_delete_counters_of_limit(orc, 2, 1);
        }
    }
    // ---- get_counters of a limit, the whole table, a snapshot file round trip -------------------------------------
    std::vector<rl_cell_row> got(1u << 17);
    uint64_t n_got = 0;
    RL(rl_get_counters(eng, 4, now, got.data(), got.size(), &n_got));
    {
        std::vector<lo_counter_row> want(got.size());
        const size_t n_want = lo_get_counters(orc, 4, 1, now, want.data(), want.size());
        CHECK(n_got == n_want, "get_counters: %" PRIu64 " rows, oracle %zu", n_got, n_want);
        uint64_t sum_got = 0, sum_want = 0;  // (row order is unspecified on both sides)
        for (uint64_t i = 0; i < n_got; ++i) sum_got += got[i].value * 31 + got[i].expiry_us + got[i].key;
        for (size_t i = 0; i < n_want; ++i) sum_want += want[i].value * 31 + want[i].expires_in_us + want[i].key;
        CHECK(sum_got == sum_want, "get_counters: rows differ");
    }
    uint64_t n_all = 0;
    RL(rl_dump_cells(eng, got.data(), got.size(), &n_all));
    uint64_t n_qual = 0;
    for (uint64_t i = 0; i < n_all; ++i) {
        lo_cell c;
        uint32_t lim = 0;
        if (got[i].limit & RL_SIMPLE) {
            CHECK(lo_peek_simple(orc, got[i].limit & ~RL_SIMPLE, &c) == 1, "simple cell %u unknown to the oracle", got[i].limit);
        } else {
            ++n_qual;
            CHECK(lo_peek_qualified(orc, got[i].key, &c, &lim) == 1, "cell %" PRIx64 " unknown to the oracle", got[i].key);
            CHECK(lim == got[i].limit, "limit of cell %" PRIx64, got[i].key);
        }
        CHECK(c.value == got[i].value && c.expiry_us == got[i].expiry_us, "cell %" PRIx64 ": (%" PRIu64 ", %" PRIu64 ") vs oracle (%" PRIu64 ", %" PRIu64 ")",
              got[i].key, got[i].value, got[i].expiry_us, (uint64_t)c.value, (uint64_t)c.expiry_us);
    }
    CHECK(n_qual == lo_num_qualified(orc), "%" PRIu64 " qualified cells, oracle %zu", n_qual, lo_num_qualified(orc));
    const char* path = "/tmp/rl_abi_harness.snap";
    RL(rl_snapshot_save(eng, path));
    rl_engine* eng2 = nullptr;
    cfg.capacity_cells = 1u << 16;
    cfg.hash_seed = 12345;
    CHECK(rl_engine_create(&cfg, &eng2) == RL_OK, "second engine");
    CHECK(rl_snapshot_load(eng2, path) == RL_OK, "snapshot load: %s", rl_last_error(eng2));
    std::vector<rl_cell_row> got2(1u << 17);
    uint64_t n2 = 0;
    CHECK(rl_dump_cells(eng2, got2.data(), got2.size(), &n2) == RL_OK && n2 == n_all, "snapshot: %" PRIu64 " of %" PRIu64 " cells", n2, n_all);
    std::remove(path);
    // ---- errors are loud and leave the table alone -----------------------------------------------------------
    rl_hit bad{123456789ull, 15u, 1u};  // unknown limit id
    uint8_t bv = 0;
    int32_t bf = 0;
    uint32_t boff[2] = {0, 1};
    CHECK(rl_check_and_update_batch(eng, &bad, 1, boff, 1, now, 0, &bv, &bf, nullptr, nullptr) == RL_ERR_INVALID, "unknown limit id must be refused");
    rl_stats_t st{};
    RL(rl_stats(eng, &st));
    CHECK(st.live_cells == n_all, "a refused call changed the table");
    rl_engine_destroy(eng2);
    rl_engine_destroy(eng);
    lo_storage_free(orc);
    std::printf("abi_harness ok: %" PRIu64 " hits in 12 batches, %" PRIu64 " requests limited, %" PRIu64 " cells\n", total_hits,
                total_limited, n_all);
    return 0;
}
