/*
 * limitador_oracle.h — CPU restatement of Limitador's InMemoryStorage hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (limitador_amd/, include/,
 * bench.py's GPU leg) may include, link or call this.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * What it restates (paths relative to the reference tree, limitador/src/):
 *   storage/atomic_expiring_value.rs:12-46,55-99,151-158  -> lo_cell_*
 *   storage/in_memory.rs:20-35    is_within_limits        -> lo_is_within_limits
 *   storage/in_memory.rs:38-44    add_counter             -> lo_add_counter
 *   storage/in_memory.rs:47-69    update_counter          -> lo_update_counter
 *   storage/in_memory.rs:72-156   check_and_update        -> lo_check_and_update
 *   storage/in_memory.rs:159-187  get_counters            -> lo_get_counters
 *   storage/in_memory.rs:190-195,241-257 delete_counters  -> lo_delete_counters_of_limit
 *   storage/in_memory.rs:198-201  clear                   -> lo_clear
 *   storage/in_memory.rs:259-264  counter_is_within_limits-> static helper
 *
 * Deviations, all deliberate and stated:
 *   - Every SystemTime::now() read inside one call (in_memory.rs:49,83;
 *     atomic_expiring_value.rs:27,72) is collapsed into the explicit `now_us`
 *     argument of that call.
 *   - Identity is interned upstream: a Limit (limit.rs:177-214 identity) is a
 *     u32 `limit`, a qualified Counter (counter.rs:123-138 identity) is an exact
 *     u64 `key`.  The oracle never hashes strings.
 *   - u64 `value + delta` wraps (Rust release-build behaviour of in_memory.rs:88,261;
 *     fetch_add always wraps).
 *   - get_counters (in_memory.rs:159-187) is restated PER LIMIT: the reference walks the limit set,
 *     and for every limit returns the simple cell of that limit plus — through one shared pass over
 *     the qualified cache filtered by `limits.contains(counter.limit())` — the qualified cells of
 *     every limit in the set; lo_get_counters(limit) returns the cells of ONE limit and the caller
 *     unions them over the set, which yields the same HashSet<Counter>.  The reference also computes
 *     `remaining = max_value - value` unchecked (:166,178: wraps in release, panics in debug when a
 *     limit's max_value was lowered below the stored value); the oracle reports value_at(now) and
 *     leaves that subtraction to the caller (tests/helpers/limiter.py, the host mirror), which wraps.
 *   - moka's capacity eviction (in_memory.rs:208-210) is not modelled: parity is
 *     unpinned above cache_size (no reference test exceeds it).  Eviction exists
 *     only as the explicit events lo_evict / lo_sweep_expired.
 *
 * Pinning: tests/test_oracle_golden.py replays the reference's own unit and
 * integration vectors for this path against this file.  The Rust
 * reference itself cannot be built in this image (no rustc/cargo), so there is
 * no oracle/_ref.
 */
#ifndef LIMITADOR_ORACLE_H
#define LIMITADOR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* AtomicExpiringValue: exactly 16 bytes (atomic_expiring_value.rs:239-244). */
typedef struct {
    uint64_t value;
    uint64_t expiry_us; /* microseconds since the epoch */
} lo_cell;

uint64_t lo_cell_value_at(const lo_cell *c, uint64_t when_us);
uint64_t lo_cell_update(lo_cell *c, uint64_t delta, uint64_t ttl_us, uint64_t when_us);
uint64_t lo_cell_ttl_us(const lo_cell *c, uint64_t now_us);

/* One Counter as the storage sees it (counter.rs:10-17). */
typedef struct {
    uint64_t key;       /* exact identity of a qualified counter; ignored when !qualified */
    uint32_t limit;     /* interned Limit identity */
    uint32_t qualified; /* counter.rs:108-110 */
    uint64_t max_value; /* counter.rs:64-66, always request-side */
    uint64_t seconds;   /* counter.rs:76-78 window = seconds */
    /* outputs, only written when load_counters (counter.rs:96-106) */
    uint64_t remaining;
    uint64_t expires_in_us;
    uint32_t has_remaining;
    uint32_t has_expires_in;
} lo_counter;

typedef struct lo_storage lo_storage;

lo_storage *lo_storage_new(void);
void lo_storage_free(lo_storage *s);

/* Return codes */
#define LO_OK 0
#define LO_LIMITED 1
#define LO_ERR_MISSING_SIMPLE (-2) /* in_memory.rs:107 `.unwrap()` would panic */
#define LO_ERR_NOMEM (-3)

int lo_is_within_limits(lo_storage *s, const lo_counter *c, uint64_t delta, uint64_t now_us,
                        int *within);
int lo_add_counter(lo_storage *s, uint32_t limit, int limit_has_variables);
int lo_update_counter(lo_storage *s, const lo_counter *c, uint64_t delta, uint64_t now_us);
/* Returns LO_OK / LO_LIMITED; *limited_idx = index (into ctrs) of the counter whose
 * limit name the reference would report, or -1. */
int lo_check_and_update(lo_storage *s, lo_counter *ctrs, size_t n, uint64_t delta,
                        int load_counters, uint64_t now_us, int64_t *limited_idx);

/* get_counters for ONE limit: every stored cell of that limit whose ttl > 0.
 * Writes up to cap rows, returns the total number of matching rows. */
typedef struct {
    uint64_t key;
    uint32_t limit;
    uint32_t qualified;
    uint64_t value;         /* value_at(now) */
    uint64_t expires_in_us; /* ttl(now) > 0 */
} lo_counter_row;
size_t lo_get_counters(lo_storage *s, uint32_t limit, int limit_has_variables, uint64_t now_us,
                       lo_counter_row *out, size_t cap);
void lo_delete_counters_of_limit(lo_storage *s, uint32_t limit, int limit_has_variables);
void lo_clear(lo_storage *s);

/* Explicit eviction events (no reference analogue; SURVEY.md §7 hard part 3). */
int lo_evict(lo_storage *s, uint64_t key);
size_t lo_sweep_expired(lo_storage *s, uint64_t now_us);

/* Introspection for parity checks of final table state. */
size_t lo_num_qualified(const lo_storage *s);
int lo_peek_qualified(const lo_storage *s, uint64_t key, lo_cell *out, uint32_t *limit);
int lo_peek_simple(const lo_storage *s, uint32_t limit, lo_cell *out);
/* Every qualified cell, raw (full-size final-state comparisons): writes up to cap, returns the total. */
size_t lo_dump_qualified(const lo_storage *s, uint64_t *keys, uint32_t *limits, uint64_t *values,
                         uint64_t *expiries, size_t cap);
/* Bulk load of qualified cells (snapshot restore); used to pre-populate tables. */
int lo_load_qualified(lo_storage *s, const uint64_t *keys, const uint32_t *limits,
                      const uint64_t *values, const uint64_t *expiries, size_t n);

/* ------------------------------------------------------------------------------------------
 * Batch driver: replays a batch in the engine's wire format (include/rl_engine.h: rl_hit,
 * rl_limit_row) through lo_check_and_update, request by request, in index order.  This is
 * the definition of what the GPU engine must reproduce bit for bit.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t key;
    uint32_t limit; /* bit 31 set = simple (no variables) counter */
    uint32_t delta;
} lo_hit;
typedef struct {
    uint64_t max_value;
    uint64_t seconds;
} lo_limit_row;
#define LO_SIMPLE_FLAG 0x80000000u

/* req_off: n_req+1 offsets into hits (NULL => every hit is its own request, n_req == n_hits).
 * Within a request the delta is hits[first].delta (the reference has one delta per request).
 * verdict[n_req]: 0 ok / 1 limited; first_limited[n_req]: hit index or -1;
 * remaining/expires_in_us [n_hits] may be NULL unless load_counters.
 * Returns 0 or a negative LO_ERR code. */
int lo_check_and_update_batch(lo_storage *s, const lo_limit_row *limits, size_t n_limits,
                              const lo_hit *hits, size_t n_hits, const uint32_t *req_off,
                              size_t n_req, uint64_t now_us, int load_counters, uint8_t *verdict,
                              int32_t *first_limited, uint64_t *remaining,
                              uint64_t *expires_in_us);
/* Same; req_delta[n_req] (may be NULL): the request's u64 delta (in_memory.rs:75) instead of the
 * 32-bit wire field; req_now_us[n_req] (may be NULL): the clock value each request reads
 * (in_memory.rs:83), instead of one now_us for the whole batch. */
int lo_check_and_update_batch_ex(lo_storage *s, const lo_limit_row *limits, size_t n_limits,
                                 const lo_hit *hits, size_t n_hits, const uint32_t *req_off,
                                 size_t n_req, const uint64_t *req_delta, const uint64_t *req_now_us,
                                 uint64_t now_us, int load_counters, uint8_t *verdict,
                                 int32_t *first_limited, uint64_t *remaining,
                                 uint64_t *expires_in_us);
int lo_is_within_limits_batch(lo_storage *s, const lo_limit_row *limits, size_t n_limits,
                              const lo_hit *hits, size_t n_hits, uint64_t now_us,
                              uint8_t *within);
int lo_update_counter_batch(lo_storage *s, const lo_limit_row *limits, size_t n_limits,
                            const lo_hit *hits, size_t n_hits, uint64_t now_us);

/* CrCounterValue<A> (limitador/src/storage/distributed/cr_counter_value.rs:10-149) with explicit clocks: the checker
 * of rl_merge_cells / rl_export_local.  Actors are u32; lo_cr_from_values = From<(SystemTime, BTreeMap)>. */
typedef struct lo_cr lo_cr;
lo_cr *lo_cr_new(uint32_t ourselves, uint64_t max_value, uint64_t expiry_us);
lo_cr *lo_cr_from_values(uint64_t expiry_us, const uint32_t *actors, const uint64_t *values, size_t n);
void lo_cr_free(lo_cr *c);
uint64_t lo_cr_expiry_us(const lo_cr *c);
uint64_t lo_cr_local_value(const lo_cr *c);
uint64_t lo_cr_read_at(const lo_cr *c, uint64_t when_us);
void lo_cr_inc_at(lo_cr *c, uint64_t increment, uint64_t window_us, uint64_t when_us);
void lo_cr_inc_actor_at(lo_cr *c, uint32_t actor, uint64_t increment, uint64_t window_us, uint64_t when_us);
void lo_cr_merge_at(lo_cr *c, const lo_cr *other, uint64_t when_us);

/* Multi-threaded replay of single-counter batches over hash-sharded storages (bench.py's cpu_baseline leg):
 * thread t owns shards[t] and replays parts[(r % n_distinct) * n_shards + t] for batch r, with a barrier
 * between batches.  Returns the wall seconds of the `reps` batches, < 0 on error. */
double lo_bench_sharded(lo_storage **shards, size_t n_shards, const lo_limit_row *limits, size_t n_limits,
                        const lo_hit *const *parts, const size_t *part_n, size_t n_distinct, size_t reps,
                        uint64_t now0);

#ifdef __cplusplus
}
#endif
#endif
