/*
 * rl_storage.h — C view of the C++ host mirror of Limitador's `CounterStorage`
 * (limitador_amd/csrc/host/gpu_counter_storage.hpp).
 *
 * The reference trait (limitador/src/storage/mod.rs:279-292) speaks Limit / Counter objects made
 * of strings; the engine ABI (rl_engine.h) speaks exact numeric ids.  The host mirror is the layer
 * a Rust `GpuStorage` would be: it interns Limit identity
 * (limitador/src/limit.rs:177-214: namespace, seconds, conditions, variables — NOT max_value/name)
 * and Counter identity (limitador/src/counter.rs:123-138: limit + resolved variables), keeps the
 * engine's limit table in sync with the request-side max_value/seconds (counter.rs:64-66,76-78),
 * orders a request's counters simple-first (in_memory.rs:105,121) and maps results back.
 * This header exists so tests (ctypes) and non-C++ callers can drive that C++ code.
 *
 * Every function returns 0 or a negative rl_status (rl_engine.h); rls_last_error() has the text.
 */
#ifndef RL_STORAGE_H
#define RL_STORAGE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rls_storage rls_storage;
typedef struct rls_batcher rls_batcher;

/* Limit (limitador/src/limit.rs:34-48).  conditions/variables are the source strings. */
typedef struct {
    const char *namespace_;
    uint64_t max_value;
    uint64_t seconds;
    const char *const *conditions;
    uint32_t n_conditions;
    const char *const *variables;
    uint32_t n_variables;
    const char *name; /* may be NULL */
} rls_limit;

/* Counter (limitador/src/counter.rs:10-17): a limit plus its resolved variables; remaining /
 * expires_in are outputs (counter.rs:96-106). */
typedef struct {
    rls_limit limit;
    const char *const *var_names;
    const char *const *var_values;
    uint32_t n_vars;
    uint32_t has_remaining;
    uint64_t remaining;
    uint64_t expires_in_us;
    uint32_t has_expires_in;
    uint32_t reserved;
} rls_counter;

/* InMemoryStorage::new(cache_size) (in_memory.rs:205-212): capacity_cells plays cache_size. */
int32_t rls_storage_create(uint64_t capacity_cells, uint32_t max_batch_hits, int32_t device,
                           rls_storage **out);
void rls_storage_destroy(rls_storage *s);
const char *rls_last_error(const rls_storage *s);
/* Test clock: now_us != 0 pins the clock every call reads; 0 returns to the system clock. */
void rls_set_clock(rls_storage *s, uint64_t now_us);

/* trait CounterStorage, method for method */
int32_t rls_is_within_limits(rls_storage *s, const rls_counter *counter, uint64_t delta,
                             int32_t *within);
int32_t rls_add_counter(rls_storage *s, const rls_limit *limit);
int32_t rls_update_counter(rls_storage *s, const rls_counter *counter, uint64_t delta);
/* *limited: 0 = Authorization::Ok, 1 = Limited; *limited_idx: index into counters of the counter
 * whose limit name the reference reports (Authorization::Limited(name)), -1 when Ok. */
int32_t rls_check_and_update(rls_storage *s, rls_counter *counters, uint32_t n, uint64_t delta,
                             int32_t load_counters, int32_t *limited, int32_t *limited_idx);
/* get_counters: emit() is called once per live counter of the given limits. */
typedef void (*rls_emit_fn)(void *user, uint32_t limit_index, const char *const *var_names,
                            const char *const *var_values, uint32_t n_vars, uint64_t remaining,
                            uint64_t expires_in_us);
int32_t rls_get_counters(rls_storage *s, const rls_limit *limits, uint32_t n_limits,
                         rls_emit_fn emit, void *user);
int32_t rls_delete_counters(rls_storage *s, const rls_limit *limits, uint32_t n_limits);
int32_t rls_clear(rls_storage *s);

/* No reference analogue: drop every qualified cell whose window has ended and forget its interned identity
 * (stands in for moka's capacity eviction, in_memory.rs:205-212, which is what bounds the reference's memory
 * under caller-controlled descriptor values).  It runs by itself once `n_new_counters` new counters have been
 * interned since the last sweep (default 2^20; 0 = never). */
int32_t rls_sweep_expired(rls_storage *s, uint64_t *n_removed);
/* exception barrier self-test of THIS library (rl_engine.h: rl_abi_selftest; kinds 1-4) */
int32_t rls_abi_selftest(int32_t kind);
void rls_set_sweep_after(rls_storage *s, uint64_t n_new_counters);
uint64_t rls_interned_counters(const rls_storage *s);

/* Micro-batching aggregator: many threads call rls_batcher_check_and_update concurrently; a
 * dispatcher closes a batch at max_batch requests or max_delay_us, stamps it with ONE clock value,
 * runs it as one rl_check_and_update_batch and wakes the callers. */
int32_t rls_batcher_create(rls_storage *s, uint32_t max_batch, uint32_t max_delay_us,
                           rls_batcher **out);
void rls_batcher_destroy(rls_batcher *b);
int32_t rls_batcher_check_and_update(rls_batcher *b, rls_counter *counters, uint32_t n,
                                     uint64_t delta, int32_t load_counters, int32_t *limited,
                                     int32_t *limited_idx);
/* batches dispatched / requests carried so far */
void rls_batcher_stats(rls_batcher *b, uint64_t *batches, uint64_t *requests);

/* Measurement helper (bench.py, BASELINE.json configs[0]): `iterations` sequential check_and_update calls with
 * the same counters, timed in native code — the per-request cost of the trait boundary without a Python loop
 * around it.  elapsed_ns: wall time of the loop; n_limited: calls answered Limited. */
int32_t rls_check_and_update_repeat(rls_storage *s, rls_counter *counters, uint32_t n, uint64_t delta,
                                    uint32_t iterations, uint64_t *elapsed_ns, uint32_t *n_limited);

#ifdef __cplusplus
}
#endif
#endif
