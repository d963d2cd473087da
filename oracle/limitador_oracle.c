/*
 * limitador_oracle.c — see limitador_oracle.h.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain C restatement of limitador/src/storage/{in_memory.rs,atomic_expiring_value.rs};
 * every function cites the reference lines it follows.  Parity is PINNED by
 * tests/test_oracle_golden.py (the reference's own vectors) except for
 * moka capacity eviction, which is unpinned and not modelled.
 */
#define _GNU_SOURCE
#include "limitador_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------
 * AtomicExpiringValue (atomic_expiring_value.rs)
 * ------------------------------------------------------------------------------------- */

/* AtomicExpiryTime::expired_at, :76-79 — note `<=`: expiry == when is expired (:195-200). */
static int expired_at(const lo_cell *c, uint64_t when_us) { return c->expiry_us <= when_us; }

/* value_at, :19-24 */
uint64_t lo_cell_value_at(const lo_cell *c, uint64_t when_us) {
    if (expired_at(c, when_us)) return 0;
    return c->value;
}

/* update, :36-42 with update_if_expired, :87-99 (sequential case: the CAS always succeeds). */
uint64_t lo_cell_update(lo_cell *c, uint64_t delta, uint64_t ttl_us, uint64_t when_us) {
    if (c->expiry_us <= when_us) {
        c->expiry_us = when_us + ttl_us;
        c->value = delta;
        return delta;
    }
    c->value += delta; /* fetch_add wraps */
    return c->value;
}

/* AtomicExpiryTime::ttl, :68-74 — duration_since(now).unwrap_or(ZERO). */
uint64_t lo_cell_ttl_us(const lo_cell *c, uint64_t now_us) {
    return c->expiry_us > now_us ? c->expiry_us - now_us : 0;
}

/* ---------------------------------------------------------------------------------------
 * InMemoryStorage state (in_memory.rs:13-16)
 *   simple_limits      : BTreeMap<Limit, AtomicExpiringValue>   -> array indexed by limit id
 *   qualified_counters : moka Cache<Counter, Arc<AEV>>          -> open-addressing map by key
 * ------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t key;
    lo_cell cell;
    uint32_t limit;
    uint32_t state; /* 0 empty, 1 live, 2 tombstone */
} qslot;

struct lo_storage {
    /* simple */
    lo_cell *simple;
    uint8_t *simple_present;
    size_t simple_cap;
    /* qualified */
    qslot *q;
    size_t q_cap; /* power of two */
    size_t q_live;
    size_t q_used; /* live + tombstones */
};

static uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

lo_storage *lo_storage_new(void) {
    lo_storage *s = (lo_storage *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->q_cap = 1024;
    s->q = (qslot *)calloc(s->q_cap, sizeof(qslot));
    if (!s->q) {
        free(s);
        return NULL;
    }
    return s;
}

void lo_storage_free(lo_storage *s) {
    if (!s) return;
    free(s->simple);
    free(s->simple_present);
    free(s->q);
    free(s);
}

static int simple_reserve(lo_storage *s, uint32_t limit) {
    if ((size_t)limit < s->simple_cap) return 0;
    size_t ncap = s->simple_cap ? s->simple_cap : 16;
    while (ncap <= (size_t)limit) ncap *= 2;
    lo_cell *nc = (lo_cell *)realloc(s->simple, ncap * sizeof(lo_cell));
    if (!nc) return LO_ERR_NOMEM;
    s->simple = nc;
    uint8_t *np = (uint8_t *)realloc(s->simple_present, ncap);
    if (!np) return LO_ERR_NOMEM;
    s->simple_present = np;
    memset(s->simple + s->simple_cap, 0, (ncap - s->simple_cap) * sizeof(lo_cell));
    memset(s->simple_present + s->simple_cap, 0, ncap - s->simple_cap);
    s->simple_cap = ncap;
    return 0;
}

static lo_cell *simple_get(lo_storage *s, uint32_t limit) {
    if ((size_t)limit >= s->simple_cap || !s->simple_present[limit]) return NULL;
    return &s->simple[limit];
}

static qslot *q_find(const lo_storage *s, uint64_t key) {
    size_t mask = s->q_cap - 1;
    size_t i = (size_t)mix64(key) & mask;
    for (;;) {
        qslot *e = &s->q[i];
        if (e->state == 0) return NULL;
        if (e->state == 1 && e->key == key) return e;
        i = (i + 1) & mask;
    }
}

static int q_grow(lo_storage *s, size_t want_cap) {
    qslot *old = s->q;
    size_t old_cap = s->q_cap;
    qslot *nq = (qslot *)calloc(want_cap, sizeof(qslot));
    if (!nq) return LO_ERR_NOMEM;
    s->q = nq;
    s->q_cap = want_cap;
    s->q_used = 0;
    s->q_live = 0;
    size_t mask = want_cap - 1;
    for (size_t j = 0; j < old_cap; j++) {
        if (old[j].state != 1) continue;
        size_t i = (size_t)mix64(old[j].key) & mask;
        while (nq[i].state) i = (i + 1) & mask;
        nq[i] = old[j];
        s->q_used++;
        s->q_live++;
    }
    free(old);
    return 0;
}

/* moka get_with / get_with_by_ref: insert-if-absent (in_memory.rs:51-56,122-127). */
static qslot *q_get_or_insert(lo_storage *s, uint64_t key, uint32_t limit, lo_cell init) {
    qslot *e = q_find(s, key);
    if (e) return e;
    if ((s->q_used + 1) * 2 > s->q_cap) {
        size_t want = s->q_cap;
        while ((s->q_live + 1) * 2 > want) want *= 2;
        if ((s->q_live + 1) * 4 > want) want *= 2;
        if (q_grow(s, want)) return NULL;
    }
    size_t mask = s->q_cap - 1;
    size_t i = (size_t)mix64(key) & mask;
    while (s->q[i].state == 1) i = (i + 1) & mask; /* stops at empty or tombstone */
    if (s->q[i].state == 0) s->q_used++;
    s->q[i].key = key;
    s->q[i].limit = limit;
    s->q[i].cell = init;
    s->q[i].state = 1;
    s->q_live++;
    return &s->q[i];
}

static void q_remove(lo_storage *s, qslot *e) {
    e->state = 2;
    s->q_live--;
}

/* ---------------------------------------------------------------------------------------
 * CounterStorage for InMemoryStorage
 * ------------------------------------------------------------------------------------- */

/* counter_is_within_limits, in_memory.rs:259-264 (value + delta wraps in release builds). */
static int counter_is_within_limits(const lo_counter *c, uint64_t current, uint64_t delta) {
    return (uint64_t)(current + delta) <= c->max_value;
}

/* is_within_limits, in_memory.rs:20-35: missing cell reads as 0 (unwrap_or_default). */
int lo_is_within_limits(lo_storage *s, const lo_counter *c, uint64_t delta, uint64_t now_us,
                        int *within) {
    uint64_t value = 0;
    if (c->qualified) {
        qslot *e = q_find(s, c->key);
        if (e) value = lo_cell_value_at(&e->cell, now_us);
    } else {
        lo_cell *cell = simple_get(s, c->limit);
        if (cell) value = lo_cell_value_at(cell, now_us);
    }
    *within = c->max_value >= (uint64_t)(value + delta);
    return LO_OK;
}

/* add_counter, in_memory.rs:38-44: only limits WITHOUT variables get a pre-created
 * Default cell = (0, UNIX_EPOCH) (atomic_expiring_value.rs:151-158); entry().or_default()
 * keeps an existing cell. */
int lo_add_counter(lo_storage *s, uint32_t limit, int limit_has_variables) {
    if (limit_has_variables) return LO_OK;
    int rc = simple_reserve(s, limit);
    if (rc) return rc;
    if (!s->simple_present[limit]) {
        s->simple[limit].value = 0;
        s->simple[limit].expiry_us = 0;
        s->simple_present[limit] = 1;
    }
    return LO_OK;
}

/* update_counter, in_memory.rs:47-69 */
int lo_update_counter(lo_storage *s, const lo_counter *c, uint64_t delta, uint64_t now_us) {
    uint64_t window_us = c->seconds * 1000000ULL;
    if (c->qualified) {
        lo_cell init = {0, now_us + window_us}; /* :52-54 */
        qslot *e = q_get_or_insert(s, c->key, c->limit, init);
        if (!e) return LO_ERR_NOMEM;
        lo_cell_update(&e->cell, delta, window_us, now_us); /* :57 */
    } else {
        int rc = simple_reserve(s, c->limit);
        if (rc) return rc;
        if (!s->simple_present[c->limit]) { /* Entry::Vacant, :60-62 */
            s->simple[c->limit].value = delta;
            s->simple[c->limit].expiry_us = now_us + window_us;
            s->simple_present[c->limit] = 1;
        } else { /* Entry::Occupied, :63-65 */
            lo_cell_update(&s->simple[c->limit], delta, window_us, now_us);
        }
    }
    return LO_OK;
}

/* check_and_update, in_memory.rs:72-156 */
int lo_check_and_update(lo_storage *s, lo_counter *ctrs, size_t n, uint64_t delta,
                        int load_counters, uint64_t now_us, int64_t *limited_idx) {
    int64_t first_limited = -1; /* :79 */
    *limited_idx = -1;
    /* counter_values_to_update / qualified_counter_values_to_updated (:80-82): the cells are
     * re-resolved below in the same two-pass order instead of being kept as references. */
    enum { STACK_N = 64 };
    size_t stack_idx[STACK_N];
    size_t *touched = n <= STACK_N ? stack_idx : (size_t *)malloc(n * sizeof(size_t));
    if (!touched) return LO_ERR_NOMEM;
    size_t n_touched = 0;
    int rc = LO_OK;

    /* `touched` records counter indices, not cell pointers: a later insert in the same request
     * can grow (move) the qualified table, so cells are looked up again in the update loop. */
    for (int pass = 0; pass < 2; pass++) { /* pass 0: simple (:105-118); pass 1: qualified (:121-139) */
        for (size_t i = 0; i < n; i++) {
            lo_counter *c = &ctrs[i];
            if ((c->qualified != 0) != (pass == 1)) continue;
            lo_cell *cell;
            if (!c->qualified) {
                cell = simple_get(s, c->limit); /* :106-107 .unwrap() */
                if (!cell) {
                    rc = LO_ERR_MISSING_SIMPLE;
                    goto done;
                }
            } else {
                uint64_t window_us = c->seconds * 1000000ULL;
                lo_cell init = {0, now_us + window_us}; /* :122-127: created BEFORE the verdict */
                qslot *e = q_get_or_insert(s, c->key, c->limit, init);
                if (!e) {
                    rc = LO_ERR_NOMEM;
                    goto done;
                }
                cell = &e->cell;
            }
            uint64_t value = lo_cell_value_at(cell, now_us);
            /* process_counter closure, :85-102 */
            int limited_here = 0;
            if (load_counters) {
                uint64_t sum = value + delta; /* wraps */
                int has = c->max_value >= sum; /* checked_sub */
                c->remaining = has ? c->max_value - sum : 0; /* unwrap_or_default */
                c->has_remaining = 1;
                if (first_limited < 0 && !has) first_limited = (int64_t)i; /* :90-94 */
            }
            if (!counter_is_within_limits(c, value, delta)) limited_here = 1; /* :96-100 */
            if (limited_here && !load_counters) { /* :109-113, :129-133 */
                *limited_idx = (int64_t)i;
                rc = LO_LIMITED;
                goto done;
            }
            if (load_counters) { /* :114-116, :134-136 */
                c->expires_in_us = lo_cell_ttl_us(cell, now_us);
                c->has_expires_in = 1;
            }
            touched[n_touched++] = i;
        }
    }

    if (first_limited >= 0) { /* :141-143 */
        *limited_idx = first_limited;
        rc = LO_LIMITED;
        goto done;
    }

    /* Update counters, :146-153: simple ones first, then qualified, each in Vec order. */
    for (size_t t = 0; t < n_touched; t++) {
        lo_counter *c = &ctrs[touched[t]];
        lo_cell *cell;
        if (!c->qualified) {
            cell = simple_get(s, c->limit);
        } else {
            qslot *e = q_find(s, c->key);
            cell = e ? &e->cell : NULL;
        }
        if (!cell) {
            rc = LO_ERR_MISSING_SIMPLE;
            goto done;
        }
        lo_cell_update(cell, delta, c->seconds * 1000000ULL, now_us);
    }

done:
    if (touched != stack_idx) free(touched);
    return rc;
}

/* get_counters, in_memory.rs:159-187 for one limit: cells with ttl > 0 only (:168,:180). */
size_t lo_get_counters(lo_storage *s, uint32_t limit, int limit_has_variables, uint64_t now_us,
                       lo_counter_row *out, size_t cap) {
    size_t n = 0;
    if (!limit_has_variables) {
        lo_cell *cell = simple_get(s, limit);
        if (cell && lo_cell_ttl_us(cell, now_us) > 0) {
            if (n < cap) {
                out[n].key = 0;
                out[n].limit = limit;
                out[n].qualified = 0;
                out[n].value = lo_cell_value_at(cell, now_us);
                out[n].expires_in_us = lo_cell_ttl_us(cell, now_us);
            }
            n++;
        }
        return n;
    }
    for (size_t i = 0; i < s->q_cap; i++) {
        qslot *e = &s->q[i];
        if (e->state != 1 || e->limit != limit) continue;
        uint64_t ttl = lo_cell_ttl_us(&e->cell, now_us);
        if (ttl == 0) continue;
        if (n < cap) {
            out[n].key = e->key;
            out[n].limit = limit;
            out[n].qualified = 1;
            out[n].value = lo_cell_value_at(&e->cell, now_us);
            out[n].expires_in_us = ttl;
        }
        n++;
    }
    return n;
}

/* delete_counters_of_limit, in_memory.rs:241-257 */
void lo_delete_counters_of_limit(lo_storage *s, uint32_t limit, int limit_has_variables) {
    if (!limit_has_variables) {
        if ((size_t)limit < s->simple_cap) s->simple_present[limit] = 0;
        return;
    }
    for (size_t i = 0; i < s->q_cap; i++)
        if (s->q[i].state == 1 && s->q[i].limit == limit) q_remove(s, &s->q[i]);
}

/* clear, in_memory.rs:198-201: ONLY simple_limits is emptied; the moka cache is untouched. */
void lo_clear(lo_storage *s) {
    if (s->simple_cap) memset(s->simple_present, 0, s->simple_cap);
}

int lo_evict(lo_storage *s, uint64_t key) {
    qslot *e = q_find(s, key);
    if (!e) return 0;
    q_remove(s, e);
    return 1;
}

size_t lo_sweep_expired(lo_storage *s, uint64_t now_us) {
    size_t removed = 0;
    for (size_t i = 0; i < s->q_cap; i++) {
        if (s->q[i].state == 1 && s->q[i].cell.expiry_us <= now_us) {
            q_remove(s, &s->q[i]);
            removed++;
        }
    }
    return removed;
}

size_t lo_num_qualified(const lo_storage *s) { return s->q_live; }

int lo_peek_qualified(const lo_storage *s, uint64_t key, lo_cell *out, uint32_t *limit) {
    qslot *e = q_find(s, key);
    if (!e) return 0;
    if (out) *out = e->cell;
    if (limit) *limit = e->limit;
    return 1;
}

size_t lo_dump_qualified(const lo_storage *s, uint64_t *keys, uint32_t *limits, uint64_t *values,
                         uint64_t *expiries, size_t cap) {
    size_t n = 0;
    for (size_t i = 0; i < s->q_cap; i++) {
        const qslot *e = &s->q[i];
        if (e->state != 1) continue;
        if (n < cap) {
            keys[n] = e->key;
            limits[n] = e->limit;
            values[n] = e->cell.value;
            expiries[n] = e->cell.expiry_us;
        }
        n++;
    }
    return n;
}

int lo_peek_simple(const lo_storage *s, uint32_t limit, lo_cell *out) {
    if ((size_t)limit >= s->simple_cap || !s->simple_present[limit]) return 0;
    if (out) *out = s->simple[limit];
    return 1;
}

int lo_load_qualified(lo_storage *s, const uint64_t *keys, const uint32_t *limits,
                      const uint64_t *values, const uint64_t *expiries, size_t n) {
    size_t want = s->q_cap;
    while ((s->q_live + n) * 2 > want) want *= 2;
    if (want != s->q_cap && q_grow(s, want)) return LO_ERR_NOMEM;
    for (size_t i = 0; i < n; i++) {
        lo_cell init = {values[i], expiries[i]};
        qslot *e = q_get_or_insert(s, keys[i], limits[i], init);
        if (!e) return LO_ERR_NOMEM;
        e->cell = init;
        e->limit = limits[i];
    }
    return LO_OK;
}

/* ---------------------------------------------------------------------------------------
 * Batch drivers (wire format of include/rl_engine.h)
 * ------------------------------------------------------------------------------------- */
static int hit_to_counter(const lo_limit_row *limits, size_t n_limits, const lo_hit *h,
                          lo_counter *c) {
    uint32_t id = h->limit & ~LO_SIMPLE_FLAG;
    if ((size_t)id >= n_limits) return -1;
    memset(c, 0, sizeof(*c));
    c->key = h->key;
    c->limit = id;
    c->qualified = (h->limit & LO_SIMPLE_FLAG) ? 0 : 1;
    c->max_value = limits[id].max_value;
    c->seconds = limits[id].seconds;
    return 0;
}

int lo_check_and_update_batch(lo_storage *s, const lo_limit_row *limits, size_t n_limits,
                              const lo_hit *hits, size_t n_hits, const uint32_t *req_off,
                              size_t n_req, uint64_t now_us, int load_counters, uint8_t *verdict,
                              int32_t *first_limited, uint64_t *remaining,
                              uint64_t *expires_in_us) {
    return lo_check_and_update_batch_ex(s, limits, n_limits, hits, n_hits, req_off, n_req, NULL, NULL, now_us,
                                        load_counters, verdict, first_limited, remaining, expires_in_us);
}

/* The same with the reference's full-width arguments: req_delta[r] (may be NULL) is the request's
 * u64 delta (in_memory.rs:75 `delta: u64`) instead of the 32-bit wire field; req_now_us[r] (may be
 * NULL) is the clock value request r reads (in_memory.rs:83 reads it once per call). */
int lo_check_and_update_batch_ex(lo_storage *s, const lo_limit_row *limits, size_t n_limits,
                                 const lo_hit *hits, size_t n_hits, const uint32_t *req_off,
                                 size_t n_req, const uint64_t *req_delta, const uint64_t *req_now_us,
                                 uint64_t now_us, int load_counters, uint8_t *verdict,
                                 int32_t *first_limited, uint64_t *remaining,
                                 uint64_t *expires_in_us) {
    enum { STACK_N = 64 };
    lo_counter stack_ctrs[STACK_N];
    for (size_t r = 0; r < n_req; r++) {
        size_t b = req_off ? req_off[r] : r;
        size_t e = req_off ? req_off[r + 1] : r + 1;
        size_t k = e - b;
        if (e > n_hits || e < b) return -1;
        if (k == 0) { /* lib.rs:434-440: no counters => not limited, storage not called */
            verdict[r] = 0;
            if (first_limited) first_limited[r] = -1;
            continue;
        }
        lo_counter *ctrs = k <= STACK_N ? stack_ctrs : (lo_counter *)malloc(k * sizeof(lo_counter));
        if (!ctrs) return LO_ERR_NOMEM;
        for (size_t j = 0; j < k; j++)
            if (hit_to_counter(limits, n_limits, &hits[b + j], &ctrs[j])) {
                if (ctrs != stack_ctrs) free(ctrs);
                return -1;
            }
        int64_t lim = -1;
        int rc = lo_check_and_update(s, ctrs, k, req_delta ? req_delta[r] : (uint64_t)hits[b].delta, load_counters,
                                     req_now_us ? req_now_us[r] : now_us, &lim);
        if (rc < 0) {
            if (ctrs != stack_ctrs) free(ctrs);
            return rc;
        }
        verdict[r] = (uint8_t)(rc == LO_LIMITED);
        if (first_limited) first_limited[r] = lim < 0 ? -1 : (int32_t)(b + (size_t)lim);
        if (load_counters) {
            for (size_t j = 0; j < k; j++) {
                if (remaining) remaining[b + j] = ctrs[j].has_remaining ? ctrs[j].remaining : 0;
                if (expires_in_us)
                    expires_in_us[b + j] = ctrs[j].has_expires_in ? ctrs[j].expires_in_us : 0;
            }
        }
        if (ctrs != stack_ctrs) free(ctrs);
    }
    return 0;
}

int lo_is_within_limits_batch(lo_storage *s, const lo_limit_row *limits, size_t n_limits,
                              const lo_hit *hits, size_t n_hits, uint64_t now_us,
                              uint8_t *within) {
    for (size_t i = 0; i < n_hits; i++) {
        lo_counter c;
        if (hit_to_counter(limits, n_limits, &hits[i], &c)) return -1;
        int w = 0;
        lo_is_within_limits(s, &c, hits[i].delta, now_us, &w);
        within[i] = (uint8_t)w;
    }
    return 0;
}

int lo_update_counter_batch(lo_storage *s, const lo_limit_row *limits, size_t n_limits,
                            const lo_hit *hits, size_t n_hits, uint64_t now_us) {
    for (size_t i = 0; i < n_hits; i++) {
        lo_counter c;
        if (hit_to_counter(limits, n_limits, &hits[i], &c)) return -1;
        int rc = lo_update_counter(s, &c, hits[i].delta, now_us);
        if (rc < 0) return rc;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Multi-threaded replay for bench.py's cpu_baseline leg: n_shards storages, one thread each (keys are
 * hash-sharded by the caller: valid for single-counter requests, whose cells are independent), `reps`
 * batches, batch r = parts[(r % n_distinct) * n_shards + t] for thread t; a barrier between batches (a batch
 * is complete when its slowest shard is).  Returns the wall seconds of the whole replay, < 0 on error.
 * ---------------------------------------------------------------------------------------- */
#include <pthread.h>
#include <time.h>

typedef struct {
    lo_storage *s;
    const lo_limit_row *limits;
    size_t n_limits;
    const lo_hit *const *parts;
    const size_t *part_n;
    size_t n_shards, n_distinct, reps, t;
    uint64_t now0;
    pthread_barrier_t *bar;
    int rc;
} lo_bench_arg;

static void *lo_bench_worker(void *p) {
    lo_bench_arg *a = (lo_bench_arg *)p;
    size_t cap = 0;
    for (size_t d = 0; d < a->n_distinct; d++)
        if (a->part_n[d * a->n_shards + a->t] > cap) cap = a->part_n[d * a->n_shards + a->t];
    uint8_t *verdict = (uint8_t *)malloc(cap ? cap : 1);
    if (!verdict) a->rc = LO_ERR_NOMEM;
    pthread_barrier_wait(a->bar); /* start */
    for (size_t r = 0; r < a->reps; r++) {
        const size_t q = (r % a->n_distinct) * a->n_shards + a->t;
        if (!a->rc && a->part_n[q]) {
            int rc = lo_check_and_update_batch(a->s, a->limits, a->n_limits, a->parts[q], a->part_n[q], NULL,
                                               a->part_n[q], a->now0 + 1000 * r, 0, verdict, NULL, NULL, NULL);
            if (rc < 0) a->rc = rc;
        }
        pthread_barrier_wait(a->bar);
    }
    free(verdict);
    return NULL;
}

double lo_bench_sharded(lo_storage **shards, size_t n_shards, const lo_limit_row *limits, size_t n_limits,
                        const lo_hit *const *parts, const size_t *part_n, size_t n_distinct, size_t reps,
                        uint64_t now0) {
    pthread_barrier_t bar;
    if (!n_shards || !n_distinct || pthread_barrier_init(&bar, NULL, (unsigned)n_shards + 1)) return -1.0;
    pthread_t *th = (pthread_t *)malloc(n_shards * sizeof(pthread_t));
    lo_bench_arg *args = (lo_bench_arg *)malloc(n_shards * sizeof(lo_bench_arg));
    if (!th || !args) return -1.0;
    for (size_t t = 0; t < n_shards; t++) {
        args[t] = (lo_bench_arg){shards[t], limits, n_limits, parts, part_n, n_shards, n_distinct, reps, t, now0, &bar, 0};
        if (pthread_create(&th[t], NULL, lo_bench_worker, &args[t])) return -1.0;
    }
    struct timespec t0, t1;
    pthread_barrier_wait(&bar);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (size_t r = 0; r < reps; r++) pthread_barrier_wait(&bar);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    int rc = 0;
    for (size_t t = 0; t < n_shards; t++) {
        pthread_join(th[t], NULL);
        if (args[t].rc) rc = args[t].rc;
    }
    pthread_barrier_destroy(&bar);
    free(th);
    free(args);
    if (rc) return -1.0;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ------------------------------------------------------------------------------------------
 * CrCounterValue<A> (limitador/src/storage/distributed/cr_counter_value.rs:10-149) restated with explicit
 * clocks: our own value + a map actor -> value of the others, one expiry.  Pinned by the reference's own
 * vectors (:176-303) in tests/test_oracle_golden.py; the checker of rl_merge_cells / rl_export_local.
 * ---------------------------------------------------------------------------------------- */
#define LO_CR_MAX_OTHERS 16
struct lo_cr {
    uint32_t ourselves;
    uint64_t max_value;
    uint64_t value;
    uint64_t expiry_us;
    uint32_t other_actor[LO_CR_MAX_OTHERS];
    uint64_t other_value[LO_CR_MAX_OTHERS];
    size_t n_others;
};

lo_cr *lo_cr_new(uint32_t ourselves, uint64_t max_value, uint64_t expiry_us) { /* :20-28, expiry = now + window */
    lo_cr *c = (lo_cr *)calloc(1, sizeof(lo_cr));
    if (!c) return NULL;
    c->ourselves = ourselves;
    c->max_value = max_value;
    c->expiry_us = expiry_us;
    return c;
}
/* From<(SystemTime, BTreeMap<A, u64>)> (:163-173): ourselves = A::default() = 0, value 0, others = the map */
lo_cr *lo_cr_from_values(uint64_t expiry_us, const uint32_t *actors, const uint64_t *values, size_t n) {
    lo_cr *c = lo_cr_new(0, 0, expiry_us);
    if (!c || n > LO_CR_MAX_OTHERS) return c;
    for (size_t i = 0; i < n; i++) {
        c->other_actor[i] = actors[i];
        c->other_value[i] = values[i];
    }
    c->n_others = n;
    return c;
}
void lo_cr_free(lo_cr *c) { free(c); }
uint64_t lo_cr_expiry_us(const lo_cr *c) { return c->expiry_us; }
uint64_t lo_cr_local_value(const lo_cr *c) { return c->value; } /* local_values(), :131-141 */

uint64_t lo_cr_read_at(const lo_cr *c, uint64_t when_us) { /* :38-47 */
    if (c->expiry_us <= when_us) return 0;
    uint64_t sum = c->value;
    for (size_t i = 0; i < c->n_others; i++) sum += c->other_value[i];
    return sum;
}
static int cr_update_if_expired(lo_cr *c, uint64_t ttl_us, uint64_t when_us) { /* atomic_expiring_value.rs:87-99 */
    if (c->expiry_us <= when_us) {
        c->expiry_us = when_us + ttl_us;
        return 1;
    }
    return 0;
}
void lo_cr_inc_at(lo_cr *c, uint64_t increment, uint64_t window_us, uint64_t when_us) { /* :53-59 */
    if (cr_update_if_expired(c, window_us, when_us)) c->value = increment;
    else c->value += increment;
}
static uint64_t *cr_other(lo_cr *c, uint32_t actor, int create) {
    for (size_t i = 0; i < c->n_others; i++)
        if (c->other_actor[i] == actor) return &c->other_value[i];
    if (!create || c->n_others >= LO_CR_MAX_OTHERS) return NULL;
    c->other_actor[c->n_others] = actor;
    c->other_value[c->n_others] = 0;
    return &c->other_value[c->n_others++];
}
void lo_cr_inc_actor_at(lo_cr *c, uint32_t actor, uint64_t increment, uint64_t window_us, uint64_t when_us) { /* :65-76 */
    if (actor == c->ourselves) {
        lo_cr_inc_at(c, increment, window_us, when_us);
        return;
    }
    uint64_t *v = cr_other(c, actor, 1);
    if (!v) return;
    if (cr_update_if_expired(c, window_us, when_us)) *v = increment;
    else *v += increment;
}
void lo_cr_merge_at(lo_cr *c, const lo_cr *other, uint64_t when_us) { /* :81-113 */
    const uint64_t expiry = other->expiry_us; /* into_inner(): (expiry, others + (ourselves -> value)) */
    if (!(expiry > when_us)) return;
    /* AtomicExpiryTime::merge_at, atomic_expiring_value.rs:113-130: the earliest expiry still in the future */
    if (expiry < c->expiry_us && expiry > when_us) c->expiry_us = expiry;
    if (c->expiry_us <= when_us) { /* reset(expiry), :144-149 */
        c->expiry_us = expiry;
        c->value = 0;
        c->n_others = 0;
    }
    const uint64_t ourselves = c->value;
    for (size_t i = 0; i <= other->n_others; i++) {
        const uint32_t actor = i < other->n_others ? other->other_actor[i] : other->ourselves;
        const uint64_t other_value = i < other->n_others ? other->other_value[i] : other->value;
        if (actor == c->ourselves) {
            if (other_value > ourselves) c->value += other_value - ourselves;
        } else {
            uint64_t *known = cr_other(c, actor, 0);
            if (!known) {
                if (other_value > 0) {
                    uint64_t *v = cr_other(c, actor, 1);
                    if (v) *v = other_value;
                }
            } else if (other_value > *known) {
                *known = other_value;
            }
        }
    }
}
