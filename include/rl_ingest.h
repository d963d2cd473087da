/*
 * rl_ingest.h — host-side ingest for the on-device limit matcher (rl_match_and_check_batch):
 * what sits between Limitador's transport (limitador-server/src/envoy_rls/server.rs:91-208: a
 * RateLimitRequest = domain + descriptors of string key/value entries + hits_addend) and the engine's
 * dictionary-encoded request arrays.  C view of limitador_amd/csrc/host/ingest.cpp.
 *
 *   limits    Limit::new(namespace, max_value, seconds, conditions, variables) (limit.rs:54-76) with
 *             conditions   <key ref> == 'value'   |   <key ref> != 'value'
 *             and variables  <key ref>  (at most eight per limit), where a key ref is what the caller's Context
 *             can resolve (rli_set_binding):
 *               RLI_BIND_DESCRIPTORS (default: the transports bind only the list `descriptors` — the WHOLE list, one
 *                   map per descriptor: envoy_rls/server.rs:121-137, http_api/server.rs:140-141):
 *                   descriptors[N]['key'] | descriptors[N]["key"] | descriptors[N].ident      for N = 0 .. 63
 *                   (`descriptors[1].y == '2'`: envoy_rls/server.rs:520, kuadrant_service.rs:415)
 *               RLI_BIND_ROOT (library callers, Context::from(HashMap), limit/cel.rs:81-96,153-156):  ident
 *             -> rows of rl_limits_set + the compiled match table of rl_match_table_set.  Anything else —
 *             a bare identifier under RLI_BIND_DESCRIPTORS included: it is unbound there and the reference
 *             never applies such a limit — is CEL: RLI_HOST_ONLY, the caller keeps the limit on its own
 *             evaluation path.
 *             Identity is (namespace, seconds, conditions, variables) ON THE SOURCE TEXT of the expressions
 *             (limit.rs:177-214; Predicate / Expression compare by source): two spellings of one predicate
 *             are two limits with two counters; adding a limit with a known identity returns its id and
 *             refreshes max_value (what update_limit does).
 *             A limit with more than two variables: with RLI_KEYS_HASHED the device hashes every variable's value into
 *             the counter's key; with RLI_KEYS_EXACT the packed key has room for two value ids, so the ingest interns the
 *             TUPLE of the values (exact, like every string) and the limit reads that one id on the device.
 *   requests  namespace + the entries of every descriptor some limit reads + delta -> req_ns / CSR (ent_key, ent_val) /
 *             req_delta (an entry's key id stands for the pair (descriptor index, key)).
 *             Strings are interned exactly (two different strings never share an id).  A namespace
 *             without limits maps to namespace id 0, which never has limits: no counter, not limited
 *             (lib.rs:434-440).  Values first seen in a request are interned on the fly: an id the
 *             table does not know simply matches no `==` and every `!=`.
 *
 * Every function returns >= 0 or a negative rl_status (rl_engine.h) unless stated otherwise.
 */
#ifndef RL_INGEST_H
#define RL_INGEST_H
#include <stdint.h>

#include "rl_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rli_ingest rli_ingest;

#define RLI_HOST_ONLY (-100)      /* a condition or variable the device matcher does not evaluate */
#define RLI_RESPONSE_TOO_LARGE (-102) /* rli_serve_batch: the request was decided and counted, its serialized response
                                     * does not fit out_stride (rli_last_error names the size); every other
                                     * request of the batch has its response */
#define RLI_UNKNOWN_DOMAIN (-101) /* RateLimitRequest without a domain: the reference answers Code::Unknown
                                     without consulting the storage (envoy_rls/server.rs:105-115) */

int32_t rli_create(rli_ingest **out);
void rli_destroy(rli_ingest *g);
const char *rli_last_error(const rli_ingest *g);

/* -> limit id (row of rl_limits_set), RLI_HOST_ONLY, or a negative rl_status. */
int32_t rli_add_limit(rli_ingest *g, const char *namespace_, uint64_t max_value, uint64_t seconds,
                      const char *const *conditions, uint32_t n_conditions, const char *const *variables,
                      uint32_t n_variables);
/* Build the tables (sorted by namespace id; namespace 0 is the empty one).  Call after the last
 * rli_add_limit and before the accessors below / rli_install. */
int32_t rli_compile(rli_ingest *g);
uint32_t rli_n_limits(const rli_ingest *g);
uint32_t rli_n_conds(const rli_ingest *g);
uint32_t rli_n_namespaces(const rli_ingest *g);
const rl_limit_row *rli_limit_rows(const rli_ingest *g);  /* [n_limits], indexed by limit id */
const rl_match_limit *rli_match_limits(const rli_ingest *g); /* [n_limits], table order */
const rl_match_cond *rli_match_conds(const rli_ingest *g);   /* [n_conds] */
/* rl_limits_set + rl_match_table_set + rl_add_counter for every limit without variables. */
int32_t rli_install(rli_ingest *g, rl_engine *e);

/* The request batch under construction. */
void rli_batch_clear(rli_ingest *g);
/* -> index of the request in the batch.  keys/values: the entries of descriptors[0], in order. */
int32_t rli_batch_add(rli_ingest *g, const char *namespace_, const char *const *keys, const char *const *values,
                      uint32_t n_entries, uint32_t delta);
/* The same with the whole descriptor list: descriptor d's entries are keys / values [desc_off[d], desc_off[d + 1])
 * (desc_off[n_descriptors] = the number of entries); inside one descriptor a repeated key keeps its last value. */
int32_t rli_batch_add_descriptors(rli_ingest *g, const char *namespace_, uint32_t n_descriptors, const uint32_t *desc_off,
                                  const char *const *keys, const char *const *values, uint32_t delta);
/* The same from the wire: one serialized envoy.service.ratelimit.v3.RateLimitRequest
 * (rls.proto:38-53: domain = 1, descriptors = 2, hits_addend = 3; ratelimit.proto:65-95: entries = 1 of
 * (key = 1, value = 2)), interpreted like ShouldRateLimit does (envoy_rls/server.rs:97-137): namespace =
 * domain, the context is one map per descriptor — its entries with the LAST value of a repeated key
 * (HashMap::insert) — hits_addend 0 means 1.  -> request index, RLI_UNKNOWN_DOMAIN, or RL_ERR_INVALID for a malformed
 * message (nothing is added).  Malformed is what prost — the reference's decoder — answers a decode error for: truncated
 * or overlong fields, a KNOWN field (domain, descriptors, hits_addend, entries, Entry.key, Entry.value) that arrives with
 * another wire type than its declared one, a `string` (domain, key, value) that is not UTF-8.  Unknown fields are skipped
 * by wire type like prost does.  Still more lenient than prost, on purpose: descriptors no limit reads and the nested
 * messages this path never reads (RateLimitOverride, HitsAddend) are skipped by wire type without being validated — a
 * message that is malformed only THERE is served where the reference would have refused it.  The device reader
 * (limitador_amd/csrc/rl_wire.hpp, RLI_KEYS_HASHED) has the same rules. */
int32_t rli_batch_add_rls(rli_ingest *g, const uint8_t *msg, uint32_t len);
/* What the caller's Context binds (see the header comment); before the first rli_add_limit. */
#define RLI_BIND_DESCRIPTORS 0
#define RLI_BIND_ROOT 1
int32_t rli_set_binding(rli_ingest *g, int32_t binding);
/* How counters are keyed, before the first request (default RLI_KEYS_EXACT):
 *   RLI_KEYS_EXACT   the host interns every namespace / key / value string (exact dictionaries under a reader-writer
 *                    lock) and the device packs ids into an injective 64-bit key (rl_match_key);
 *   RLI_KEYS_HASHED  the host keeps NO per-request state: rli_serve_batch hands the serialized messages to the device,
 *                    which decodes them, compares the strings the limits name byte by byte, and keys every counter by
 *                    a hash of its canonical key bytes (storage/keys.rs:220-248; include/rl_keyhash.h says which bytes,
 *                    which hash, and the odds).  The 64-bit key addresses the cell, a 32-bit check word stored with the
 *                    cell is compared on every touch: two counters that share a key are never merged — the message
 *                    that derives the second one is answered RLI_HOST_ONLY (nothing of it is applied; the other messages
 *                    of the batch are).  Needs the slot form of the match table (<= 8 distinct descriptor keys, <= 64
 *                    limits per namespace, <= 512 condition literals, <= 8 KB of table strings): rli_install answers
 *                    RLI_HOST_ONLY otherwise.  rli_batch_add* / rli_check stay dictionary calls (RLI_KEYS_EXACT only). */
#define RLI_KEYS_EXACT 0
#define RLI_KEYS_HASHED 1
int32_t rli_set_key_mode(rli_ingest *g, int32_t mode);
/* The 128-bit secret RLI_KEYS_HASHED keys its hash with (SipHash-2-4-128, include/rl_keyhash.h: without it a caller who
 * chooses descriptor values cannot aim at another counter's key).  rli_create draws it from the OS; a host whose table
 * outlives the ingest or is fed by several ingests — a restart that reloads a snapshot, several front-ends in front of one
 * engine, the ranks of a sharded deployment — reads it once (rli_hash_key) and gives it to the others (rli_set_hash_key,
 * before the limits are compiled): it names the cells.  Keep it as secret as the table's contents. */
int32_t rli_set_hash_key(rli_ingest *g, uint64_t k0, uint64_t k1);
int32_t rli_hash_key(const rli_ingest *g, uint64_t out[2]);
/* The key (and check word) RLI_KEYS_HASHED gives the counter of `limit_id` with these variable values (in variable-name
 * order; n_values = the limit's number of variables): what rl_get_counters / rl_dump_cells rows of that mode carry. */
int32_t rli_counter_key(rli_ingest *g, uint32_t limit_id, const char *const *values, const uint32_t *value_lens,
                        uint32_t n_values, uint64_t *key, uint32_t *check);
/* Most distinct descriptor values the dictionary may hold (default 2^24, at most 2^26: value ids travel in 26
 * bits).  At the cap a request that carries a value never seen before answers RLI_HOST_ONLY (nothing is added;
 * requests made of known values go on): descriptor values are caller-controlled and must not grow host state
 * without bound (the reference bounds its counters with moka's cache_size, in_memory.rs:205-212). */
int32_t rli_set_value_cap(rli_ingest *g, uint32_t cap);
uint32_t rli_batch_n_requests(const rli_ingest *g);
uint32_t rli_batch_n_entries(const rli_ingest *g);
const uint32_t *rli_batch_req_ns(const rli_ingest *g);    /* [n_requests] */
const uint32_t *rli_batch_req_delta(const rli_ingest *g); /* [n_requests] */
const uint32_t *rli_batch_ent_off(const rli_ingest *g);   /* [n_requests + 1] */
const uint32_t *rli_batch_ent_key(const rli_ingest *g);   /* [n_entries] */
const uint32_t *rli_batch_ent_val(const rli_ingest *g);   /* [n_entries] */
/* counters_that_apply + check_and_update for the batch (rl_match_and_check_batch): verdict[n_requests],
 * limited_limit[n_requests] (limit id of the first limited counter, -1 otherwise; may be NULL). */
int32_t rli_check(rli_ingest *g, rl_engine *e, uint64_t now_us, uint8_t *verdict, int32_t *limited_limit);

/* The answer on the wire: a serialized RateLimitResponse carrying overall_code (rls.proto:62-71,182) the way
 * ShouldRateLimit builds it without rate-limit headers (envoy_rls/server.rs:176-206): verdict 0 -> OK,
 * 1 -> OVER_LIMIT, RLI_UNKNOWN_DOMAIN -> UNKNOWN.  Writes at most 2 bytes into out, returns the length. */
uint32_t rli_rls_response(int32_t verdict, uint8_t out[2]);

/* Limit.name (limit.rs:107-113): not part of the identity, only reported (the `name="..."` of X-RateLimit-Limit). */
int32_t rli_set_limit_name(rli_ingest *g, uint32_t limit_id, const char *name);

/* ShouldRateLimit for a batch of serialized RateLimitRequests (envoy_rls/server.rs:91-208), applied in index
 * order with one clock value: decode (rli_batch_add_rls), counters_that_apply + check_and_update on the device,
 * and the serialized RateLimitResponse of each request in out + i * out_stride (out_len[i] bytes):
 *   overall_code = 1: OK / OVER_LIMIT; UNKNOWN (no domain) is the empty message;
 *   with_headers: the draft-03 headers the reference adds with RateLimitHeaders::DraftVersion03
 *   (response_headers_to_add = 3, sorted by key): X-RateLimit-Limit `{max}, {max};w={secs}[;name="{name}"]...`
 *   over the request's counters sorted by remaining, X-RateLimit-Remaining, X-RateLimit-Reset of the most
 *   restrictive one (CheckResult::response_header, lib.rs:235-275; the counters are loaded: load_counters).
 * With headers, a batch of 4096 messages or more has the DEVICE build the response bytes (rl_engine.h: rl_wire_serve_batch /
 * rl_match_serve_batch, csrc/rl_resp.hpp; what each limit contributes to X-RateLimit-Limit is sent to the engine once per
 * change of a limit's max / name) and this call only hands them on while they arrive; smaller batches are assembled on the
 * host from the counters' arrays.  Same bytes either way.
 * May be called from several threads at once on one ingest / engine: with RLI_KEYS_HASHED and device-built responses up to
 * RL_SERVE_SETS (rl_engine.h) calls are in flight together — each takes a serving set of the engine (its own pinned
 * staging, rl_host_staging slots 4 s .. 4 s + 3, device copy of the messages and piece events): one packs while one copies
 * in / decides and the others' responses cross PCIe and are handed on; decisions are applied in the order the calls enter
 * the engine.  Every other form of the call waits for the engine to itself.
 * status[i]: 0 OK, 1 OVER_LIMIT, RLI_UNKNOWN_DOMAIN, RLI_HOST_ONLY (the value dictionary is at its cap: the caller
 * evaluates this request itself; no response is produced), RLI_RESPONSE_TOO_LARGE (counted, but its response does not
 * fit out_stride: out_len[i] = 0), or another value < 0 for a malformed message. */
int32_t rli_serve_batch(rli_ingest *g, rl_engine *e, const uint8_t *const *msgs, const uint32_t *lens, uint32_t n,
                        uint64_t now_us, int32_t with_headers, uint8_t *out, uint32_t out_stride, uint32_t *out_len,
                        int32_t *status);

/* The Kuadrant RateLimitService over the same messages (limitador-server/src/envoy_rls/kuadrant_service.rs:27-184), which
 * splits ShouldRateLimit in two — op is one of rl_engine.h's RL_OP_*:
 *   RL_OP_CHECK   CheckRateLimit (:27-106) = RateLimiter::is_rate_limited(namespace, ctx, 1) (lib.rs:362-409): every counter
 *                 the message derives is checked with delta 1 WHATEVER hits_addend says (:62-64), in counters_that_apply's
 *                 order, the first one that does not fit limits the request; nothing is written.  status 0 OK / 1 OVER_LIMIT;
 *   RL_OP_UPDATE  Report (:108-184) = RateLimiter::update_counters(namespace, ctx, hits_addend or 1) (lib.rs:411-423): every
 *                 derived counter takes the addend, no limit is tested; status 0, the response is always OK (:171-181);
 *   RL_OP_CHECK_AND_UPDATE  rli_serve_batch without headers.
 * Responses carry overall_code only (neither method adds headers); no domain -> RLI_UNKNOWN_DOMAIN and the empty message
 * (:37-47, :118-128).  Everything else — index order, one clock value, the statuses of undecodable / RLI_HOST_ONLY messages,
 * one call at a time per engine — as rli_serve_batch. */
int32_t rli_serve_batch_op(rli_ingest *g, rl_engine *e, int32_t op, const uint8_t *const *msgs, const uint32_t *lens,
                           uint32_t n, uint64_t now_us, uint8_t *out, uint32_t out_stride, uint32_t *out_len,
                           int32_t *status);

/* The micro-batcher of the wire path: thread-safe, blocking; concurrent callers are aggregated into one
 * rli_serve_batch, closed at max_batch requests or max_delay_us after its first request arrived (the latency
 * budget), stamped with one clock value.  -> the request's status (as rli_serve_batch), its response in resp. */
typedef struct rli_frontend rli_frontend;
int32_t rli_frontend_create(rli_ingest *g, rl_engine *e, uint32_t max_batch, uint32_t max_delay_us, int32_t with_headers,
                            rli_frontend **out);
void rli_frontend_destroy(rli_frontend *f);
void rli_frontend_set_clock(rli_frontend *f, uint64_t now_us); /* tests: a fixed clock instead of the system's */
int32_t rli_frontend_should_rate_limit(rli_frontend *f, const uint8_t *msg, uint32_t len, uint8_t *resp,
                                       uint32_t resp_cap, uint32_t *resp_len);
/* The Kuadrant methods through the same micro-batcher (rli_serve_batch_op).  A device batch is one method: a window of
 * mixed traffic is served as its consecutive same-method runs, back to back under one clock value and ONE wait
 * (max_delay_us is paid per window, not per run), and arrival order is kept across the runs, so a Report queued behind a
 * CheckRateLimit is applied behind it. */
int32_t rli_frontend_check_rate_limit(rli_frontend *f, const uint8_t *msg, uint32_t len, uint8_t *resp, uint32_t resp_cap,
                                      uint32_t *resp_len);
int32_t rli_frontend_report(rli_frontend *f, const uint8_t *msg, uint32_t len, uint8_t *resp, uint32_t resp_cap,
                            uint32_t *resp_len);
void rli_frontend_stats(rli_frontend *f, uint64_t *batches, uint64_t *requests); /* device calls, requests */
uint64_t rli_frontend_windows(rli_frontend *f); /* windows served (one wait of at most max_delay_us each) */
/* exception barrier self-test of THIS library (rl_engine.h: rl_abi_selftest; kinds 1-4) */
int32_t rli_abi_selftest(int32_t kind);

/* Dictionary look-ups (tests, diagnostics): id of a string, -1 if it was never interned. */
int64_t rli_key_id(const rli_ingest *g, const char *key);
int64_t rli_value_id(const rli_ingest *g, const char *value);
int64_t rli_namespace_id(const rli_ingest *g, const char *namespace_);

#ifdef __cplusplus
}
#endif
#endif /* RL_INGEST_H */
