/*
 * rl_engine.h — C ABI of the MI355X rate-limit counter engine.
 *
 * This is the drop-in boundary for ONE path of Kuadrant/limitador: the in-memory counter
 * storage behind `trait CounterStorage` (reference limitador/src/storage/mod.rs:279-292),
 * whose hot method is `InMemoryStorage::check_and_update`
 * (limitador/src/storage/in_memory.rs:72-156).  The reference has no FFI for this path today
 * (backends are compiled-in Rust modules, storage/mod.rs:10-24); these entry points are what a
 * `GpuStorage: CounterStorage` Rust module would bind with `extern "C"` (INTEGRATION.md shows
 * the binding).  Plain pointers and sizes only; no C++ or torch types cross this line.
 *
 * Semantics contract (all entry points): the result of a batch call is bit-identical to
 * applying the reference method to the batch's requests ONE AT A TIME IN INDEX ORDER, with every
 * `SystemTime::now()` the reference would read during the batch (in_memory.rs:49,83;
 * atomic_expiring_value.rs:27,72) replaced by the call's single `now_us`.
 *
 * Identity: the reference keys a counter by (Limit identity, resolved variables)
 * (counter.rs:123-138, limit.rs:177-214) — strings.  The caller interns that identity to an
 * exact 64-bit `key` (unique per counter, simple or qualified; two different counters must
 * never share a key) and the Limit identity to a dense `limit` id.  The engine hashes `key`
 * to a slot on the device but stores and compares the full 64-bit key, so there is no
 * fingerprint-collision error mode.  Keys 0xFFFFFFFFFFFFFFFE/F are reserved.
 *
 * Threading: calls on one engine are serialised internally (one mutex, one HIP stream); any
 * thread may call.  Errors: every function returns RL_OK (0) or a negative rl_status;
 * rl_last_error() gives the message; rl_status_is_transient() maps to StorageErr.transient
 * (storage/mod.rs:312-339).
 */
#ifndef RL_ENGINE_H
#define RL_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rl_engine rl_engine;

typedef enum {
    RL_OK = 0,
    RL_ERR_INVALID = -1,        /* bad argument (null pointer, size, unknown limit id, reserved key) */
    RL_ERR_DEVICE = -2,         /* HIP runtime error (transient) */
    RL_ERR_NO_DEVICE = -3,      /* no gfx950 device / HIP unavailable: the engine never falls back to CPU */
    RL_ERR_TABLE_FULL = -4,     /* live cells would exceed the table's load bound; grow or sweep */
    RL_ERR_MISSING_SIMPLE = -5, /* a simple counter has no pre-created cell: the reference panics
                                   here (in_memory.rs:107 `.unwrap()`) */
    RL_ERR_KEY_LIMIT = -6,      /* a hit's limit id differs from the one stored with its key */
    RL_ERR_BATCH_TOO_LARGE = -7,
    RL_ERR_NOMEM = -8,
    RL_ERR_BUSY = -9,           /* batches are in flight (rl_check_and_update_submit_device): collect them first */
    RL_ERR_KEY_COLLISION = -10, /* hashed keys (rl_wire_*): two counters share a 64-bit key; nothing was applied */
    RL_ERR_INTERNAL = -11       /* a C++ exception was stopped at the boundary (rl_last_internal_error); see below */
} rl_status;

/* bit 31 of a limit id marks a counter of a limit WITHOUT variables ("simple", the
 * `simple_limits` BTreeMap of in_memory.rs:14); without it the counter is "qualified"
 * (the moka cache of in_memory.rs:15). */
#define RL_SIMPLE 0x80000000u
#define RL_LIMIT_ID(x) ((x) & 0x7FFFFFFFu)

/* rl_config.flags: double the table (rl_resize) instead of answering RL_ERR_TABLE_FULL when a call
 * finds it past its occupancy bound — only possible while no batch is in flight. */
#define RL_CFG_AUTO_GROW 1u

typedef struct {
    int32_t device;          /* HIP device ordinal */
    uint32_t max_batch_hits; /* largest n_hits one call may carry */
    uint64_t capacity_cells; /* table slots; rounded up to a power of two.  Live cells + tombstones are bounded
                                by 3/4 of it before a batch, 15/16 after (RL_ERR_TABLE_FULL beyond).  Replaces moka's
                                `cache_size` (in_memory.rs:205-212) but never evicts silently. */
    uint32_t max_limits;     /* rows of the limit table */
    uint32_t flags;          /* RL_CFG_* */
    uint64_t hash_seed;
} rl_config;

/* One row per interned Limit: the request-side attributes the reference reads from
 * `Counter.limit` (counter.rs:64-66,76-78).  max_value is NOT stored with a cell
 * (limit.rs:207-214: not part of identity), so updating a row is what
 * Storage::update_limit (storage/mod.rs:67-83) does. */
typedef struct {
    uint64_t max_value;
    uint64_t seconds;
} rl_limit_row;

/* One (request x counter) record: 16 bytes, the unit the hot kernels stream. */
typedef struct {
    uint64_t key;   /* exact counter identity */
    uint32_t limit; /* limit id | RL_SIMPLE */
    uint32_t delta; /* hits to add; one delta per request in the reference (in_memory.rs:75):
                       for a multi-counter request every hit carries the request's delta */
} rl_hit;

/* A stored cell as reported by rl_get_counters / rl_dump_cells. */
typedef struct {
    uint64_t key;
    uint32_t limit; /* limit id | RL_SIMPLE */
    uint32_t reserved;
    uint64_t value;     /* rl_get_counters: value_at(now); rl_dump_cells: raw value */
    uint64_t expiry_us; /* rl_get_counters: ttl(now) in us (>0); rl_dump_cells: raw expiry */
} rl_cell_row;

typedef struct {
    uint64_t capacity_cells;
    uint64_t live_cells;
    uint64_t tombstones;
    uint64_t batches;
    uint64_t hits;
    uint64_t ordered_hits;     /* hits that needed trace-order resolution */
    uint64_t ordered_batches;  /* batches that went through the general resolver (multi-counter / load_counters / u64 deltas) */
    uint64_t probe_steps;      /* reserved (0 unless built with RL_PROBE_STATS) */
    uint64_t rebuilds;
} rl_stats_t;

/* ---- lifecycle ------------------------------------------------------------------------ */
/* InMemoryStorage::new (in_memory.rs:205-212). */
int32_t rl_engine_create(const rl_config *cfg, rl_engine **out);
void rl_engine_destroy(rl_engine *e);
const char *rl_last_error(const rl_engine *e);
int32_t rl_status_is_transient(int32_t status);
int32_t rl_stats(rl_engine *e, rl_stats_t *out);
/* The device ordinal and max_batch_hits the engine was created with (either pointer may be NULL). */
int32_t rl_engine_info(rl_engine *e, int32_t *device, uint32_t *max_batch_hits);

/* Host arrays the caller reuses from call to call (the binding's batch staging: hits in, verdicts out) can be PINNED
 * in place once: the copies of every later host-buffer call that reads or writes inside [ptr, ptr + bytes) are then
 * plain DMA from / into the caller's own pages instead of the runtime's pageable path (staging copies on a CPU thread).
 * No reference analogue (the reference has no device).  The range must stay mapped until rl_host_unregister(ptr) —
 * unregister before freeing or reallocating the array.  Returns RL_ERR_INVALID if the runtime refuses the range. */
int32_t rl_host_register(rl_engine *e, void *ptr, uint64_t bytes);
int32_t rl_host_unregister(rl_engine *e, void *ptr);

/* ---- limits --------------------------------------------------------------------------- */
/* Upload rows [first, first+n) of the limit table (max_value / seconds of interned limits). */
int32_t rl_limits_set(rl_engine *e, uint32_t first, const rl_limit_row *rows, uint32_t n);
/* CounterStorage::add_counter (in_memory.rs:38-44): for a limit WITHOUT variables pre-create
 * its cell as (0, UNIX_EPOCH) unless it exists; a no-op for limits with variables.
 * `limit` carries RL_SIMPLE when the limit has no variables; `key` is that counter's key. */
int32_t rl_add_counter(rl_engine *e, uint32_t limit, uint64_t key);

/* ---- the hot path --------------------------------------------------------------------- */
/* CounterStorage::check_and_update (in_memory.rs:72-156) for a batch of requests.
 *   hits[n_hits]        request x counter records, requests contiguous, trace order.
 *   req_off[n_req+1]    CSR offsets of requests into hits; NULL => every hit is its own
 *                       request (n_req must equal n_hits).  Within a request, hits must be in
 *                       the reference's processing order: simple counters first, then
 *                       qualified, each in Vec order (in_memory.rs:105,121).
 *   now_us              the clock value of this batch (us since the epoch).
 *   load_counters       as the reference flag: fill remaining/expires_in for every hit and
 *                       defer the early return (in_memory.rs:87-95,110-116,130-136).
 *   verdict[n_req]      0 = Authorization::Ok, 1 = Authorization::Limited.
 *   first_limited[n_req] (may be NULL) index into hits of the counter whose limit name the
 *                       reference reports, -1 when Ok.
 *   remaining[n_hits], expires_in_us[n_hits] (may be NULL unless load_counters): the values
 *                       set_remaining / set_expires_in receive (counter.rs:96-106).
 * Host pointers; the call copies in, runs the kernels, copies out and returns when done.
 * A call with ONE request of a few counters — the trait's check_and_update called request by request — or a micro-batch
 * of up to 64 counters / 64 requests launches nothing when a server kernel is lingering on the engine's stream (k_gen_serve: a host-mapped mailbox in, tagged
 * write-through stores out; ~10 us per call instead of ~30).  The server leaves by itself RL_SERVE_LINGER_US (200) after
 * the last request and every other entry point sends it away before touching the device, so nothing waits for it longer
 * than that; RL_SERVE=0 turns it off (one launch per call).  Same kernel body as the one-launch path, same results. */
int32_t rl_check_and_update_batch(rl_engine *e, const rl_hit *hits, uint32_t n_hits,
                                  const uint32_t *req_off, uint32_t n_req, uint64_t now_us,
                                  int32_t load_counters, uint8_t *verdict, int32_t *first_limited,
                                  uint64_t *remaining, uint64_t *expires_in_us);
/* The same with the reference's full-width arguments:
 *   req_delta[n_req]   (may be NULL) the request's delta as the trait's u64 (`delta: u64`, in_memory.rs:75;
 *                      storage/mod.rs:283-288) — it replaces the 32-bit wire field of the request's hits, so a
 *                      delta beyond 2^32 is answered like the reference answers it (Limited / wrapping add,
 *                      in_memory.rs:259-264), never with an error;
 *   req_now_us[n_req]  (may be NULL) the clock value each request reads (in_memory.rs:83 reads it once per
 *                      call) instead of one now_us for the whole batch: the requests are applied as
 *                      consecutive runs that share a clock value. */
int32_t rl_check_and_update_batch_ex(rl_engine *e, const rl_hit *hits, uint32_t n_hits,
                                     const uint32_t *req_off, uint32_t n_req, const uint64_t *req_delta,
                                     const uint64_t *req_now_us, uint64_t now_us, int32_t load_counters,
                                     uint8_t *verdict, int32_t *first_limited, uint64_t *remaining,
                                     uint64_t *expires_in_us);
/* Same, every pointer a DEVICE pointer on the engine's device (the rate quoted by bench.py).
 * Work is enqueued on the engine's stream and the call returns after the batch has completed
 * (its status — errors, cells created — is read back before the call returns). */
int32_t rl_check_and_update_batch_device(rl_engine *e, const rl_hit *d_hits, uint32_t n_hits,
                                         const uint32_t *d_req_off, uint32_t n_req,
                                         uint64_t now_us, int32_t load_counters,
                                         uint8_t *d_verdict, int32_t *d_first_limited,
                                         uint64_t *d_remaining, uint64_t *d_expires_in_us);

/* The same for single-counter requests (every hit its own request, no load_counters), split in two so
 * that a feeder (a micro-batcher) can keep the device busy: _submit enqueues the batch on the engine's
 * stream and returns; _collect waits for the OLDEST submitted batch and returns ITS status (verdicts
 * and first_limited of that batch are then complete for work ordered after the batch on the engine's stream;
 * a reader on another stream, or the host, synchronises with that stream first).  At most three batches may
 * be in flight; they are applied in submission order, so the sequential contract holds across them, and the
 * partition of one overlaps the replay of the one before (two streams inside the engine).  d_verdict and
 * d_first_limited belong to the engine from _submit to the matching _collect (the partition pass writes the default
 * answer into them, the decision pass the denials): every element is defined at _collect.  The replay of a batch is
 * enqueued when the NEXT command is submitted or the batch itself is collected (rl_engine_record_event flushes it too).
 * While batches are in flight:
 *   - rl_is_within_limits_batch(_ex) and rl_get_counters are served — the reference serves them under the read lock
 *     check_and_update itself holds (in_memory.rs:20-35,78,159-187): they are enqueued behind every batch submitted
 *     so far, wait for that, and leave the batches in flight;
 *   - rl_sweep_expired_submit joins the pipeline as a command of its own (below);
 *   - every other entry point (the mutating ones, which take the reference's WRITE lock, in_memory.rs:40,48,199,243;
 *     dumps, snapshots, resize) returns RL_ERR_BUSY. */
int32_t rl_check_and_update_submit_device(rl_engine *e, const rl_hit *d_hits, uint32_t n_hits, uint64_t now_us,
                                          uint8_t *d_verdict, int32_t *d_first_limited);
int32_t rl_check_and_update_collect(rl_engine *e);
/* _submit with the caller's event (hipEvent_t) recorded behind the batch's REPLAY — whenever the engine enqueues it (the
 * next submit, the batch's collect, or rl_engine_flush): what a host that chains its own device work behind the verdicts
 * (the multi-GPU router's verdict exchange) waits on, without forcing the replay out early.  An event that has not been
 * recorded yet does not hold a stream that waits on it: call rl_engine_flush before waiting on the event of a batch
 * that may still be the last one submitted. */
int32_t rl_check_and_update_submit_device_ev(rl_engine *e, const rl_hit *d_hits, uint32_t n_hits, uint64_t now_us,
                                             uint8_t *d_verdict, int32_t *d_first_limited, void *done_event);
/* Enqueue whatever the engine is holding back (the replay of the batch submitted last). */
int32_t rl_engine_flush(rl_engine *e);

/* CounterStorage::is_within_limits (in_memory.rs:20-35), one verdict per hit, read-only:
 * within[i] = max_value >= value_at(now) + delta; a missing cell reads as 0. */
int32_t rl_is_within_limits_batch(rl_engine *e, const rl_hit *hits, uint32_t n_hits,
                                  uint64_t now_us, uint8_t *within);
/* CounterStorage::update_counter (in_memory.rs:47-69) applied to hits[0..n) in order:
 * find-or-create then AtomicExpiringValue::update; never checks the limit. */
int32_t rl_update_counter_batch(rl_engine *e, const rl_hit *hits, uint32_t n_hits,
                                uint64_t now_us);
/* Both with the trait's `delta: u64` per hit (delta[n_hits], may be NULL: the 32-bit wire field). */
int32_t rl_is_within_limits_batch_ex(rl_engine *e, const rl_hit *hits, uint32_t n_hits,
                                     const uint64_t *delta, uint64_t now_us, uint8_t *within);
int32_t rl_update_counter_batch_ex(rl_engine *e, const rl_hit *hits, uint32_t n_hits,
                                   const uint64_t *delta, uint64_t now_us);

/* ---- the rest of the CounterStorage surface -------------------------------------------- */
/* CounterStorage::get_counters (in_memory.rs:159-187) for one limit: every cell of that limit
 * with ttl(now) > 0.  Writes up to cap rows; *n_out = total matching rows. */
int32_t rl_get_counters(rl_engine *e, uint32_t limit, uint64_t now_us, rl_cell_row *out,
                        uint64_t cap, uint64_t *n_out);
/* CounterStorage::delete_counters for one limit (in_memory.rs:190-195,241-257). */
int32_t rl_delete_counters(rl_engine *e, uint32_t limit);
/* CounterStorage::clear (in_memory.rs:198-201): removes ONLY simple cells — the reference
 * leaves the qualified cache untouched, and so does this. */
int32_t rl_clear(rl_engine *e);

/* ---- no reference analogue ------------------------------------------------------------- */
/* TTL sweep: drop every QUALIFIED cell with expiry <= now (an explicit eviction event; it
 * replaces moka's capacity eviction and is replayed into the oracle by the tests), then
 * compact the table if tombstones exceed 1/8 of capacity. */
int32_t rl_sweep_expired(rl_engine *e, uint64_t now_us, uint64_t *n_removed);
/* The same, reporting the swept cells (up to cap rows, raw value / expiry; *n_removed = all of them): what a
 * host that interns identities needs in order to forget the keys of the cells that are gone. */
int32_t rl_sweep_expired_rows(rl_engine *e, uint64_t now_us, rl_cell_row *out, uint64_t cap, uint64_t *n_removed);
/* The sweep as a STREAM-ORDERED command of the batch pipeline (BASELINE.json configs[4]: "concurrent expiry sweep"):
 * enqueued behind every batch submitted so far and in front of every later one, without draining the pipeline — it
 * takes one of the three in-flight slots, like a batch of rl_check_and_update_submit_device, and is collected in
 * submission order: rl_sweep_expired_collect (or rl_check_and_update_collect, which drops the count) when it is the
 * oldest command in flight.  For the counter table the result is the same as a blocking rl_sweep_expired called at that
 * point of the sequence; the table is not compacted by it (tombstones are counted, the next blocking sweep / rl_compact
 * compacts), and the PEER tables of rl_merge_cells are left alone (the blocking sweep also drops their entries of windows
 * that are over; a peer entry's expiry is checked wherever it is read, so a stale one changes no answer — it only keeps
 * its slot until the next blocking sweep). */
int32_t rl_sweep_expired_submit(rl_engine *e, uint64_t now_us);
int32_t rl_sweep_expired_collect(rl_engine *e, uint64_t *n_removed);
/* Force a compaction: the tombstones go and the gaps they leave are closed IN PLACE (every probe cluster rebuilt inside
 * its own slots, the table read once: k_compact_bounds / k_compact_seg) — no second table, nothing that can fail half-way. */
int32_t rl_compact(rl_engine *e);
/* Rehash the live cells into a table of capacity_cells (rounded up to a power of two, >= 1024): how a
 * caller answers RL_ERR_TABLE_FULL — or shrinks after a sweep — without losing a counter.  Refused
 * (RL_ERR_INVALID) if the live cells would fill the new table beyond one half.  Old and new table are
 * both resident while it runs. */
int32_t rl_resize(rl_engine *e, uint64_t capacity_cells);
/* Snapshot: bulk insert / overwrite cells, and dump every live cell (raw value/expiry). */
int32_t rl_load_cells(rl_engine *e, const rl_cell_row *rows, uint64_t n);
int32_t rl_load_cells_device(rl_engine *e, const rl_cell_row *d_rows, uint64_t n);
int32_t rl_dump_cells(rl_engine *e, rl_cell_row *out, uint64_t cap, uint64_t *n_out);

/* ---- snapshot files, cross-node merge (no CounterStorage analogue) ------------- */
/* The limit table and every live cell to / from a file: the resume path of this in-memory engine.  Loading
 * into a fresh engine (same hash_seed not required: keys are re-hashed) gives a table with the same cells. */
int32_t rl_snapshot_save(rl_engine *e, const char *path);
int32_t rl_snapshot_load(rl_engine *e, const char *path);
/* CrCounterValue::merge_at (limitador/src/storage/distributed/cr_counter_value.rs:81-113) for what ONE remote
 * actor reports: rows[i] = (key, limit, that actor's OWN value, the expiry of its window) — the triples of
 * CrCounterValue::local_values (:131-141).  A cell's value is the sum over the actors, like read_at (:38-47); per
 * actor only the largest value seen in the current window counts (:96-110); expired rows are ignored (:83); the
 * earliest future expiry wins (:84); a cell that is expired at now restarts from the row (:85-87).  actor ==
 * self_actor: another replica's memory of OUR value, which only counts if it is larger (:91-95).  Actor ids are
 * 0..7; a key must not appear twice in one call.
 * A window restarted by a LOCAL update follows CrCounterValue::inc_at (cr_counter_value.rs:53-59): our own value becomes
 * the increment, the peers' contributions to the window that ended stay (`others` is only cleared by a merge's reset,
 * :85-87,144-149) — the first hit after the expiry reads 0, every later one increment + that stale part, and a later report
 * of a peer only counts if it exceeds its stale figure (:96-110).  That rule lives in the general resolver: from the first
 * rl_merge_cells on, every check_and_update / update_counter of this engine takes it (exact, slower), and the pipelined
 * single-counter entry (rl_check_and_update_submit_device) answers RL_ERR_INVALID — an engine without peer state is not
 * affected.  tests/test_gpu_merge.py walks the restart number by number against the oracle's CrCounterValue. */
int32_t rl_merge_cells(rl_engine *e, uint32_t self_actor, uint32_t actor, const rl_cell_row *rows, uint64_t n,
                       uint64_t now_us);
/* local_values() of every live, unexpired cell: (key, limit, OUR part of the value, expiry) — what a node
 * sends to its peers (the input of their rl_merge_cells).  Writes up to cap rows; *n_out = all of them. */
int32_t rl_export_local(rl_engine *e, uint64_t now_us, rl_cell_row *out, uint64_t cap, uint64_t *n_out);

/* ---- upstream of the trait: limit matching and key derivation on the device --------------------- */
/* RateLimiter::counters_that_apply (lib.rs:507-522) = Limit::applies (limit.rs:157-174) +
 * Counter::new / resolve_variables (counter.rs:19-31, limit.rs:133-148) for limits whose conditions are
 * `descriptors[0]['k'] == 'v'` / `!= 'v'` and whose variables are `descriptors[0]['k']` (at most two).
 * Strings are dictionary-encoded by the caller (exact ids < 2^26); a request is its namespace id and the
 * (key id, value id) entries of descriptors[0].  A condition on an absent key is false for both
 * operators (limit/cel.rs:321-338); a limit with an absent variable yields no counter
 * (limit/cel.rs:176-191).  The counters of a request come out in match-table order, simple counters
 * first (in_memory.rs:105,121), keyed by rl_match_key — packed ids, injective, hence exact. */
typedef struct {
    uint32_t key;   /* descriptor key id */
    uint32_t op;    /* 0: ==   1: != */
    uint32_t value; /* value id */
} rl_match_cond;
typedef struct {
    uint32_t limit;      /* limit id (< 4095, a row of rl_limits_set) | RL_SIMPLE iff n_vars == 0 */
    uint32_t ns;         /* namespace id; the table must be sorted by it */
    uint32_t cond_off;   /* first condition of this limit in conds[] */
    uint32_t n_cond;
    uint32_t n_vars;     /* 0..2 */
    uint32_t var_key[2]; /* descriptor key ids of the variables, in variable-name order */
    uint32_t pad;
} rl_match_limit;
int32_t rl_match_table_set(rl_engine *e, const rl_match_limit *limits, uint32_t n_limits,
                           const rl_match_cond *conds, uint32_t n_conds, uint32_t n_namespaces);
/* The same with limits of up to EIGHT variables (limit.rs:133-148 resolves any number): a row with n_vars > 2 keeps the
 * descriptor key ids of its variables, in variable-name order, in more_vars[var_key[0] .. var_key[0] + n_vars).  Such a
 * table is served by the hashed-key wire path only (rl_wire_match_and_check_batch / rl_wire_serve_batch: every variable's
 * value is hashed into the counter's key); the dictionary entries (rl_match_and_check_batch*) refuse it, because the
 * packed exact key has room for two value ids (rl_match_key) — a caller with exact keys folds the variables of such a
 * limit into one synthetic variable, as include/rl_ingest.h's rli_compile does. */
int32_t rl_match_table_set_ex(rl_engine *e, const rl_match_limit *limits, uint32_t n_limits,
                              const rl_match_cond *conds, uint32_t n_conds, uint32_t n_namespaces,
                              const uint32_t *more_vars, uint32_t n_more_vars);
/* The key the device derives for (limit, variable values): what rl_add_counter must be given for a
 * limit without variables, and what identifies a counter in rl_get_counters / rl_dump_cells rows. */
uint64_t rl_match_key(uint32_t limit_id, uint32_t n_vars, uint32_t v0, uint32_t v1);
/* counters_that_apply + check_and_update for a batch of requests, applied in index order with one
 * clock value.  Requests: req_ns[n_req], req_delta[n_req], CSR ent_off[n_req+1] into ent_key / ent_val.
 * Out: verdict[n_req]; limited_limit[n_req] (may be NULL): limit id of the first limited counter, -1 when
 * Ok or when no limit applies (lib.rs:434-440); optionally the derived counters themselves
 * (req_off_out[n_req+1], hits_out[<= hits_cap], *n_hits_out) and, with load_counters, remaining /
 * expires_in_us per derived counter.  Host pointers. */
int32_t rl_match_and_check_batch(rl_engine *e, const uint32_t *req_ns, const uint32_t *ent_off,
                                 const uint32_t *ent_key, const uint32_t *ent_val, const uint32_t *req_delta,
                                 uint32_t n_req, uint64_t now_us, int32_t load_counters, uint8_t *verdict,
                                 int32_t *limited_limit, uint32_t *req_off_out, rl_hit *hits_out,
                                 uint32_t hits_cap, uint32_t *n_hits_out, uint64_t *remaining,
                                 uint64_t *expires_in_us);
/* Same with device pointers for the request arrays, verdict and limited_limit. */
int32_t rl_match_and_check_batch_device(rl_engine *e, const uint32_t *d_req_ns, const uint32_t *d_ent_off,
                                        const uint32_t *d_ent_key, const uint32_t *d_ent_val,
                                        const uint32_t *d_req_delta, uint32_t n_req, uint64_t now_us,
                                        int32_t load_counters, uint8_t *d_verdict, int32_t *d_limited_limit,
                                        uint32_t *n_hits_out);

/* The other two RateLimiter methods over the same matcher — what the Kuadrant RateLimitService splits ShouldRateLimit
 * into (limitador-server/src/envoy_rls/kuadrant_service.rs:27-184: CheckRateLimit, then Report once the response is known):
 *   RL_OP_CHECK_AND_UPDATE  check_rate_limited_and_update (lib.rs:425-464): rl_match_and_check_batch without load_counters;
 *   RL_OP_CHECK             is_rate_limited (lib.rs:362-409): a request's counters in counters_that_apply's order, each
 *                           through is_within_limits (in_memory.rs:20-35) with the request's delta; verdict[i] = 1 and
 *                           limited_limit[i] = the limit of the FIRST counter that does not fit (find_first_limited_counter).
 *                           Nothing is written: no cell is created, no value moves;
 *   RL_OP_UPDATE            update_counters (lib.rs:411-423): every derived counter takes the request's delta through
 *                           update_counter (in_memory.rs:47-69: no limit test, missing cells are created — simple ones too);
 *                           verdict[i] = 0, limited_limit[i] = -1.
 * Applied in index order with one clock value like every batch entry point; host pointers. */
#define RL_OP_CHECK_AND_UPDATE 0
#define RL_OP_CHECK 1
#define RL_OP_UPDATE 2
int32_t rl_match_batch_op(rl_engine *e, int32_t op, const uint32_t *req_ns, const uint32_t *ent_off,
                          const uint32_t *ent_key, const uint32_t *ent_val, const uint32_t *req_delta, uint32_t n_req,
                          uint64_t now_us, uint8_t *verdict, int32_t *limited_limit);

/* ---- the wire path without host dictionaries (limitador_amd/csrc/rl_wire.hpp; row f1 of SURVEY.md 8) ------------------
 * Serialized envoy.service.ratelimit.v3.RateLimitRequest messages are decoded ON THE DEVICE (what ShouldRateLimit does
 * per call, envoy_rls/server.rs:97-137), matched against the table of rl_match_table_set (lib.rs:507-522), and every
 * counter is keyed by a hash of its CANONICAL KEY BYTES (storage/keys.rs:220-248) — include/rl_keyhash.h says which
 * bytes, which hash, and how a collision of two 64-bit keys is detected (the cell's 32-bit check word) instead of
 * merged.  Strings the TABLE names are compared byte by byte on the device; the host keeps no per-request state.
 *
 * rl_wire_table_set: the strings behind the ids of the installed match table (call after rl_match_table_set; the table
 * must have the slot form: at most 8 distinct descriptor keys, 64 limits per namespace) — all inside `blob`
 * (<= 8192 bytes): ns[namespace id] (ns[0] is the namespace without limits), keys[key id] — the key's own bytes, and in
 * bits 24..31 of its `len` the index i of the descriptor the key is read from (the reference binds the whole list
 * `descriptors`, one map per descriptor: envoy_rls/server.rs:121-137; `descriptors[1].y` is key "y" with i = 1) —
 * vals[value id] for every id the conditions use (<= 512), limit_prefix[2 * limit id ..] = rl_kh_bytes of the limit's canonical prefix, and
 * hash_key[2] = the 128-bit secret every hash of this path is keyed with (include/rl_keyhash.h: the prefixes must have
 * been hashed under the same key; whoever else derives keys for this table — another front-end, a restart — needs it).
 * The engine remembers a FINGERPRINT of the key (never the key), rl_snapshot_save writes it into the file's header and
 * rl_snapshot_load brings it back: a table that holds cells is not given a table set under another key — RL_ERR_INVALID,
 * where a silent acceptance would orphan every hashed counter (all limits start from zero, the old cells linger). */
typedef struct {
    uint32_t off, len; /* a string: blob[off .. off + len) */
} rl_wire_str;
int32_t rl_wire_table_set(rl_engine *e, const uint8_t *blob, uint32_t blob_len, const rl_wire_str *ns, uint32_t n_ns,
                          const rl_wire_str *keys, uint32_t n_keys, const rl_wire_str *vals, uint32_t n_vals,
                          const uint64_t *limit_prefix, uint32_t n_limits, const uint64_t *hash_key);
/* counters_that_apply + check_and_update for n messages, applied in index order with one clock value: message i is
 * wire[msg_off[i] .. msg_off[i + 1]) (host pointers; msg_off[0] = 0).  Out, per message: status[i] = 0, or
 * -101 (no domain: Code::Unknown, envoy_rls/server.rs:105-115; the message derives no counter) or RL_ERR_INVALID
 * (malformed: no counter); verdict[i]; limited_limit[i] (may be NULL); the derived counters as in
 * rl_match_and_check_batch.  Returns RL_ERR_KEY_COLLISION when two counters of the batch, or a counter of the batch and
 * a stored one, share a 64-bit key with different check words (a limit id that differs under one key is the same thing:
 * hashed keys carry the limit in the key): NOTHING was applied, status[i] = -103 for EVERY message that derives a
 * counter which is not the one its key belongs to (the word stored in the key's cell decides; without a cell, the
 * batch's first counter of that key) and *collided_message is the index of one of them — the caller answers those
 * messages on its exact path (or drops them) and calls again without them: one re-run, however many collide. */
/* Pinned host buffers that belong to the engine, by slot (0 .. 4 RL_SERVE_SETS - 1 — four per serving set, see rl_wire_serve_batch_set: set s
 * owns slots 4 s .. 4 s + 3; grown on demand — a later call for the same slot may move
 * it — and freed with the engine): where a host layer builds the arrays it hands to the host-pointer entry points and
 * receives their results, so that every copy is plain DMA instead of the runtime's pageable path (a fresh 60 MB result
 * array per call cost the wire path 20 ms of page faults and staged copies).  Slots 4 s + 2 and 4 s + 3 are also where
 * rl_match_serve_batch / rl_wire_serve_batch (s = 0) / rl_wire_serve_batch_set leave their responses.  No reference analogue. */
int32_t rl_host_staging(rl_engine *e, uint32_t slot, uint64_t bytes, void **out);
int32_t rl_wire_match_and_check_batch(rl_engine *e, const uint8_t *wire, const uint32_t *msg_off, uint32_t n,
                                      uint64_t now_us, int32_t load_counters, uint8_t *verdict, int32_t *limited_limit,
                                      int32_t *status, uint32_t *req_off_out, rl_hit *hits_out, uint32_t hits_cap,
                                      uint32_t *n_hits_out, uint64_t *remaining, uint64_t *expires_in_us,
                                      int64_t *collided_message);
/* ---- the answer as RateLimitResponse bytes, built on the device (limitador_amd/csrc/rl_resp.hpp) -------------------------
 * ShouldRateLimit's reply (envoy_rls/server.rs:176-206) is a serialized RateLimitResponse per request: overall_code, and
 * with RateLimitHeaders::DraftVersion03 the three headers of CheckResult::response_header (lib.rs:235-275) over the
 * request's loaded counters.  These entries decide the batch like rl_match_and_check_batch / rl_wire_match_and_check_batch
 * do (load_counters = with_headers) and hand back the responses — request i's bytes are resp[resp_off[i] .. resp_off[i + 1])
 * — instead of the counters, their remaining / expires_in and the request offsets: what crosses PCIe is the answer.
 *
 * rl_resp_table_set: what limit `id` contributes to X-RateLimit-Limit, frag[id] inside blob — the text
 * `, {max};w={seconds}[;name="{name}"]` (lib.rs:246-262), formatted once per limit by the host layer that knows the names;
 * a limit id beyond n_limits contributes `, 0;w=0`.  Call after rl_limits_set, again whenever a limit's max / name changes.
 *
 * A request with status != 0 (rl_wire_serve_batch: no domain, malformed, -103) has an empty response: for -101 that IS the
 * reply (Code::Unknown = 0 is the proto3 default, server.rs:105-115), the others have none.  *resp_off ([n + 1]) and *resp
 * point into pinned memory of the ENGINE (rl_host_staging slots 2 and 3, sized by the batch's own total): valid until the
 * next serving call on the engine or a rl_host_staging call for those slots.  The other pointers are host pointers.
 *
 * flags: RL_SERVE_HEADERS — the draft-03 headers (else overall_code only); RL_SERVE_ASYNC — return as soon as verdicts,
 * statuses and offsets are on the host, with the response BYTES still travelling (in order, in up to 8 copies):
 * rl_serve_wait(e, upto) returns once resp[0 .. upto) has arrived, so a host layer that hands the responses on (scatters
 * them into its callers' buffers) works on the first ones while the last ones cross PCIe — for 262 144 responses the copy
 * is 1 ms of a 3.8 ms call and the scatter 0.8.  rl_serve_wait takes no lock and may be called from several threads; every
 * byte must have been waited for before the next call on the engine. */
#define RL_SERVE_HEADERS 1u
#define RL_SERVE_ASYNC 2u
int32_t rl_resp_table_set(rl_engine *e, const uint8_t *blob, uint32_t blob_len, const rl_wire_str *frag, uint32_t n_limits);
/* 1 once rl_resp_table_set has been called on THIS engine (a host layer that remembers "I sent the fragments to engine X"
 * by address asks the engine itself: a destroyed engine's address may come back with the next one). */
int32_t rl_resp_table_ready(rl_engine *e);
int32_t rl_match_serve_batch(rl_engine *e, const uint32_t *req_ns, const uint32_t *ent_off, const uint32_t *ent_key,
                             const uint32_t *ent_val, const uint32_t *req_delta, uint32_t n_req, uint64_t now_us,
                             uint32_t flags, uint8_t *verdict, const uint32_t **resp_off, const uint8_t **resp);
int32_t rl_wire_serve_batch(rl_engine *e, const uint8_t *wire, const uint32_t *msg_off, uint32_t n, uint64_t now_us,
                            uint32_t flags, uint8_t *verdict, int32_t *status, const uint32_t **resp_off,
                            const uint8_t **resp, int64_t *collided_message);
int32_t rl_serve_wait(rl_engine *e, uint64_t upto);
/* SEVERAL serving calls in flight (the reference serves from N tonic workers at once, envoy_rls/server.rs:238-272, behind a
 * shared read lock, in_memory.rs:78).  Everything a serving call leaves behind for the host — the pinned staging of its
 * offsets and bytes, the events of its pieces, the device copy of its messages — exists RL_SERVE_SETS times, by `set`, and the
 * bytes' kernels run on a stream of their own from a snapshot of what they read; so while the host still waits for and hands
 * on the responses of the call on set s (rl_serve_wait_set(e, s, upto)), other threads may pack, copy in and decide the next
 * batches with the other sets.  Four sets: a call's own latency (pack, copy-in, decide, the responses' way over PCIe, the
 * hand-over) is ≈ 4.5 ms for 262 144 messages of which the engine is held ≈ 1–2 ms.  While another set's bytes are still on
 * their way, a call's own bytes leave as copy commands issued one piece at a time by rl_serve_wait_set.  The
 * engine's mutex serialises the calls themselves (decisions are applied in the order the
 * calls enter); a set must not be used again before every byte of its previous call has been waited for.
 * rl_wire_serve_batch = set 0. */
#define RL_SERVE_SETS 4
int32_t rl_wire_serve_batch_set(rl_engine *e, uint32_t set, const uint8_t *wire, const uint32_t *msg_off, uint32_t n,
                                uint64_t now_us, uint32_t flags, uint8_t *verdict, int32_t *status,
                                const uint32_t **resp_off, const uint8_t **resp, int64_t *collided_message);
int32_t rl_serve_wait_set(rl_engine *e, uint32_t set, uint64_t upto);

/* rl_match_batch_op for serialized messages (RL_OP_* above).  RL_OP_CHECK is the Kuadrant CheckRateLimit: every counter is
 * checked with delta 1 whatever the message's hits_addend says (kuadrant_service.rs:62-64); a message whose counter's key
 * belongs to another counter gets status[i] = -103 and no verdict (nothing to re-run: the call wrote nothing).  RL_OP_UPDATE
 * is Report: hits_addend (0 means 1, kuadrant_service.rs:139-145) is added to every derived counter; a key collision
 * refuses the call like rl_wire_match_and_check_batch does (RL_ERR_KEY_COLLISION, nothing applied, -103 names the messages). */
int32_t rl_wire_match_batch_op(rl_engine *e, int32_t op, const uint8_t *wire, const uint32_t *msg_off, uint32_t n,
                               uint64_t now_us, uint8_t *verdict, int32_t *limited_limit, int32_t *status,
                               int64_t *collided_message);

/* ---- the general resolver in phases: admission decided by the host ----------------------- */
/* For requests whose counters live on SEVERAL engines (key-sharded multi-counter requests): the per-request AND
 * of InMemoryStorage::check_and_update (in_memory.rs:141-153) then spans engines, so each fixpoint round goes
 * through the host.  One pass at a time; every other entry point answers RL_ERR_BUSY until commit / abort.
 *   begin   the engine's share of the hits, in GLOBAL trace order, each with its request's id (any u32; equal ids =
 *           same request: hits of one request on one cell all read the value before any of them is applied).
 *           Validates, sorts by cell, reads the cells.  Nothing is written to the table.
 *   round   d_admitted[i] != 0: hit i's request is admitted so far (NULL: all are — the first round).  d_pass[i] =
 *           hit i fits on top of the admitted hits before it on its cell; with load_counters also remaining /
 *           expires_in of the hit (in_memory.rs:87-95,114-116).  The host ANDs the flags per request, across
 *           engines, and calls again until the admitted set no longer changes.  (d_pass is first overwritten with 1s
 *           and then the failures are stored: it must not be the same array as d_admitted.)
 *   count   d_reached[i] != 0: the request's walk got to hit i (in_memory.rs:109-113,129-133; NULL: all hits).
 *           -> cells the pass would create, and how many the table still takes: the host decides for ALL engines.
 *   commit  applies the admitted hits of the LAST round and creates the reached new cells; ends the pass.
 *   abort   ends the pass, nothing applied. */
int32_t rl_gen_begin_device(rl_engine *e, const rl_hit *d_hits, const uint32_t *d_req_id, uint32_t n_hits,
                            uint64_t now_us, int32_t load_counters);
int32_t rl_gen_round_device(rl_engine *e, const uint8_t *d_admitted, uint8_t *d_pass, uint64_t *d_remaining,
                            uint64_t *d_expires_in_us);
int32_t rl_gen_count_device(rl_engine *e, const uint8_t *d_reached, uint32_t *n_new, uint64_t *room);
int32_t rl_gen_commit_device(rl_engine *e);
int32_t rl_gen_abort(rl_engine *e);
/* A caller that owns the engine's stream (rl_engine_set_stream(e, stream, 1)) and orders what reads d_pass by that stream or by
 * events recorded on it: with on != 0, rl_gen_round_device returns with its kernels enqueued instead of waiting for them,
 * and rl_gen_begin_device does not wait for its sort either — what the sort has to say (an error bit, hash buckets that
 * overflowed) comes out of the next call that stops for the device anyway: rl_gen_count_device (overflow: RL_ERR_BUSY, the
 * pass is closed and is to be begun again) or rl_gen_commit_gated_device; the rounds in between compute nothing from an
 * unusable sort.  (The multi-GPU router's step.) */
int32_t rl_gen_set_async(rl_engine *e, int32_t on);
/* Count and commit of an async pass with NO host stop in between — all engines of a job apply the step or none does, decided
 * on the devices:
 *   count_async   enqueues the count and leaves this engine's veto word at d_veto: 1 = the cells the pass creates do not
 *                 fit, 2 = an error bit, 4 = hash buckets overflowed, 8 = *d_also != 0 (the caller's own reason — "my last
 *                 round still changed the admitted set"; d_also may be NULL).  0 = fine by me.
 *   (the caller gathers every engine's word on the device: n_veto words, veto_stride words apart, at d_veto)
 *   commit_gated  enqueues the commit, which applies the pass only if all n_veto words are zero, copies the words to
 *                 h_veto (pinned host memory, n_veto * veto_stride words) and stops ONCE.  *committed = 1: applied, the
 *                 pass is closed.  0 with RL_OK: some engine vetoed (h_veto says who and why) — the pass is still open:
 *                 more rounds and another count_async / commit_gated, or rl_gen_abort.  An error of THIS engine's pass
 *                 closes it and is returned (RL_ERR_BUSY: overflow — begin again); its veto word told the others. */
int32_t rl_gen_count_async_device(rl_engine *e, const uint8_t *d_reached, const uint32_t *d_also, uint32_t *d_veto);
int32_t rl_gen_commit_gated_device(rl_engine *e, const uint32_t *d_veto, uint32_t *h_veto, uint32_t n_veto,
                                   uint32_t veto_stride, uint32_t *committed);

/* ---- multi-GPU routing helpers (device pointers, engine's stream) ----------------------- */
/* Owner shard of a key for a world of `world` shards (any world >= 1). */
uint32_t rl_owner_of(uint64_t key, uint64_t hash_seed, uint32_t world);
/* Stable partition of d_hits by owner shard: d_out holds the hits grouped by owner
 * (owner 0 first), each group in original order; d_perm[j] = original index of d_out[j];
 * d_counts[world] = group sizes.  Blocks until done (unless rl_engine_set_stream gave the engine the
 * caller's stream). */
int32_t rl_route_partition_device(rl_engine *e, const rl_hit *d_hits, uint32_t n_hits,
                                  uint32_t world, rl_hit *d_out, uint32_t *d_perm,
                                  uint32_t *d_counts);
/* d_dst[d_perm[j]] = d_src[j] for j < n (returns verdict bytes to ingress order). */
int32_t rl_unpermute_u8_device(rl_engine *e, const uint8_t *d_src, const uint32_t *d_perm,
                               uint32_t n, uint8_t *d_dst);
/* Both on a stream of the caller's (hipStream_t on the engine's device): enqueue and return, never block.
 * The router keeps one scratch block per engine, so route calls of one engine belong on ONE stream
 * (include/rl_sharded.h puts them on its exchange stream, beside the engine's batches on another). */
int32_t rl_route_partition_stream(rl_engine *e, void *stream, const rl_hit *d_hits, uint32_t n_hits,
                                  uint32_t world, rl_hit *d_out, uint32_t *d_perm, uint32_t *d_counts);
/* Up to four device-to-device copies as ONE kernel launch on `stream` (a router's own segments of an exchange).  Device
 * pointers; host-mapped pinned memory counts as device memory. */
typedef struct {
    void *dst;
    const void *src;
    uint64_t bytes;
} rl_copy_seg;
int32_t rl_copy_segments_stream(rl_engine *e, void *stream, const rl_copy_seg *segs, uint32_t n);
int32_t rl_unpermute_u8_stream(rl_engine *e, void *stream, const uint8_t *d_src, const uint32_t *d_perm,
                               uint32_t n, uint8_t *d_dst);

/* Ingress side of the key-sharded MULTI-counter step (include/rl_sharded.h: rl_sharded_check_requests_device).  A
 * request's counters live on several owners; the all-or-nothing rule (in_memory.rs:141-153) is the AND of its hits'
 * pass flags, taken on the rank the request entered.  Hits travel in ROUTED order (rl_route_partition_stream: d_perm[j] =
 * ingress index of routed hit j).  All enqueue on the caller's stream and return.
 *   ids      d_req_of_hit[i] = request of ingress hit i (from the CSR offsets); d_req_id_sorted[j] = base + request of
 *            routed hit j — the id rl_gen_begin_device wants beside every hit (base: requests of the ranks before)
 *   round    pass flags back from the owners, routed order -> ingress order (d_pass_home, scratch) -> per request
 *            d_adm (in/out), d_first (first failing hit, ingress index, or -1: in_memory.rs:90-99), d_verdict;
 *            *d_changed |= 1 if the admitted set differs from the previous round's (first_round: from "all admitted");
 *            d_adm_sorted[j] = admission of routed hit j's request, for the owners' next round
 *   reached  d_reached_sorted[j] = the request's walk got to routed hit j (it stops at its first limited counter unless
 *            the values are loaded: in_memory.rs:109-113,129-133) */
int32_t rl_req_ids_stream(rl_engine *e, void *stream, const uint32_t *d_req_off, uint32_t n_req, uint32_t n_hits,
                          uint32_t base, const uint32_t *d_perm, uint32_t *d_req_of_hit, uint32_t *d_req_id_sorted);
int32_t rl_req_round_stream(rl_engine *e, void *stream, const uint8_t *d_pass_sorted, const uint32_t *d_perm,
                            const uint32_t *d_req_off, const uint32_t *d_req_of_hit, uint32_t n_req, uint32_t n_hits,
                            int32_t first_round, uint8_t *d_pass_home, uint8_t *d_adm, int32_t *d_first,
                            uint8_t *d_verdict, uint32_t *d_changed, uint8_t *d_adm_sorted);
int32_t rl_req_reached_stream(rl_engine *e, void *stream, const int32_t *d_first, const uint32_t *d_req_of_hit,
                              const uint32_t *d_perm, uint32_t n_hits, uint8_t *d_reached_sorted);
/* d_dst[d_perm[j]] = d_src[j], 8-byte elements (remaining / expires_in back to ingress order). */
int32_t rl_unpermute_u64_stream(rl_engine *e, void *stream, const uint64_t *d_src, const uint32_t *d_perm, uint32_t n,
                                uint64_t *d_dst);

/* The HIP stream (hipStream_t) the engine launches on, for callers that order their own work
 * against it, and a per-kernel timing hook used by bench.py (HIP events on that stream). */
void *rl_engine_stream(rl_engine *e);
/* Launch on the caller's stream instead (external = 1; `stream` may be NULL, the default stream), or go
 * back to the engine's own (external = 0).  For callers that chain
 * their own device work (collectives, copies) around the engine's: with an external stream the
 * routing helpers below enqueue and return without blocking (the caller's stream order is the
 * synchronisation), and submit / collect give the same for the hot path. */
int32_t rl_engine_set_stream(rl_engine *e, void *stream, int32_t external);
/* Ordering against the engine's OWN streams without giving up their overlap: the batch submitted NEXT
 * (rl_check_and_update_submit_device*) reads its inputs only after `event` (a hipEvent_t recorded by the caller, e.g.
 * after the exchange that filled the batch's buffers) — call it right in front of that submit; rl_engine_record_event
 * records `event` behind everything submitted so far (verdicts complete). */
int32_t rl_engine_wait_event(rl_engine *e, void *event);
int32_t rl_engine_record_event(rl_engine *e, void *event);
/* HIP-event timing of the kernels of the single-counter hot path: a timed launch carries its own
 * start / stop events (hipExtLaunchKernelGGL — no marker commands around the kernel, the interval is
 * the kernel's own begin..end on its stream).  enable = 0 off (completion is a sequence word the last
 * workgroup stores into host-mapped memory), 1 both kernels (k_bkt_part, k_bkt_apply) of every batch, 2 only the
 * dominant kernel, k_bkt_apply, 3 both kernels of every fourth batch.  rl_kernel_timing_read copies
 * the milliseconds accumulated per slot since the last reset into ms[RL_TIMING_SLOTS] and the number
 * of timed batches into *launches. */
enum {
    RL_T_PART = 0,           /* k_bkt_part: batch validation + single-pass stable partition (tile-local runs) */
    RL_T_APPLY_GAP = 1,      /* idle time of the apply stream between two k_bkt_apply (timed launches, two streams) */
    RL_T_PART_SLACK = 2,     /* k_bkt_part's end to the start of the k_bkt_apply that consumes it */
    RL_T_APPLY = 3,          /* k_bkt_apply: probe, decide, commit — the dominant kernel */
    RL_T_RESERVED4 = 4,
    RL_TIMING_SLOTS = 8
};
int32_t rl_kernel_timing(rl_engine *e, int32_t enable);
int32_t rl_kernel_timing_read(rl_engine *e, double *ms, uint64_t *launches, int32_t reset);

/* No C++ exception crosses this boundary (the reference's errors are values, storage/mod.rs:312-339, and its in-memory
 * path never fails, in_memory.rs:72-156): every entry point of librl_engine.so, librl_storage.so and librl_sharded.so
 * that can allocate behind the call ends in the same barrier (limitador_amd/csrc/rl_abi_guard.h).  std::bad_alloc comes
 * back as RL_ERR_NOMEM, anything else as RL_ERR_INTERNAL; rl_last_internal_error() is the calling thread's last such
 * message ("entry point: what()").  The device table is only written by kernels of a batch the call had already
 * validated, so the counters are as the call found them or as the batch made them.
 * rl_abi_selftest throws behind the barrier — kind 1 std::bad_alloc, 2 std::length_error, 3 an object that is no
 * std::exception, 4 a vector resized to 2^58 elements (a real failed allocation), 5 a reserve beyond max_size — and
 * returns what the barrier made of it; it touches no engine and needs no device (tests/test_abi_barrier.py). */
const char *rl_last_internal_error(void);
int32_t rl_abi_selftest(int32_t kind);
int32_t rl_abi_caught(const char *fn, const char *what, int32_t status); /* (the barrier's landing pad; not for callers) */

#ifdef __cplusplus
}
#endif
#endif /* RL_ENGINE_H */
