// gpu_counter_storage.hpp — C++ host mirror of Limitador's `CounterStorage`
// (limitador/src/storage/mod.rs:279-292) over the engine's C ABI (include/rl_engine.h).
//
// This is the layer a Rust `GpuStorage: CounterStorage` would be: it owns
// everything between the reference's string-typed model and the engine's numeric wire format —
//   * Limit identity (limit.rs:177-214: namespace, seconds, conditions, variables — NOT max_value,
//     name or id) -> dense limit id; the engine's limit table row (max_value, seconds) follows the
//     request-side values the reference reads from Counter.limit (counter.rs:64-66,76-78);
//   * Counter identity (counter.rs:123-138: limit + resolved variables) -> exact 64-bit key;
//   * the order the storage walks a request's counters: simple first, then qualified, each in Vec
//     order (in_memory.rs:105,121);
//   * mapping results back: Authorization::Limited(name of the first limited counter),
//     set_remaining / set_expires_in (counter.rs:96-106), get_counters' remaining = max - value
//     (in_memory.rs:166,178);
// and the micro-batching aggregator that turns per-request calls into device batches.
// Method names, argument meaning and error behaviour follow the trait; no verdict is computed here.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../../include/rl_engine.h"

namespace rls {

struct Limit {  // limitador/src/limit.rs:34-48
    std::string ns;
    uint64_t max_value = 0;
    uint64_t seconds = 0;
    std::vector<std::string> conditions;  // kept sorted + unique: BTreeSet<Predicate>
    std::vector<std::string> variables;   // kept sorted + unique: BTreeSet<Expression>
    std::string name;
    bool has_name = false;
    std::string identity() const;  // canonical encoding of (ns, seconds, conditions, variables)
};

struct Counter {  // limitador/src/counter.rs:10-17
    Limit limit;
    std::vector<std::pair<std::string, std::string>> set_variables;  // sorted: BTreeMap<String,String>
    bool has_remaining = false, has_expires_in = false;
    uint64_t remaining = 0, expires_in_us = 0;
    bool is_qualified() const { return !set_variables.empty(); }  // counter.rs:108-110
};

struct Authorization {  // storage/mod.rs:26-29
    bool limited = false;
    int limited_idx = -1;  // index into the caller's Vec<Counter> of the counter whose name is reported
};

// StorageErr (storage/mod.rs:312-339): message + transient flag; code = the engine's rl_status.
struct StorageErr {
    int code = 0;
    std::string msg;
    bool transient = false;
};

class GpuCounterStorage {
public:
    // InMemoryStorage::new(cache_size) (in_memory.rs:205-212)
    static int create(uint64_t capacity_cells, uint32_t max_batch_hits, int device, GpuCounterStorage** out);
    ~GpuCounterStorage();

    // trait CounterStorage
    int is_within_limits(const Counter& counter, uint64_t delta, bool* within);                 // :20-35
    int add_counter(const Limit& limit);                                                         // :38-44
    int update_counter(const Counter& counter, uint64_t delta);                                  // :47-69
    int check_and_update(std::vector<Counter>& counters, uint64_t delta, bool load_counters,
                         Authorization* auth);                                                   // :72-156
    int get_counters(const std::vector<Limit>& limits, std::vector<std::pair<uint32_t, Counter>>* out);  // :159-187
    int delete_counters(const std::vector<Limit>& limits);                                       // :190-195
    int clear();                                                                                 // :198-201

    // One request of a micro-batch: counters in the CALLER's order; results are mapped back.
    struct Request {
        std::vector<Counter>* counters;
        uint64_t delta;
        bool load_counters;
        Authorization auth;
        int rc = 0;
    };
    // Applies the requests in index order with ONE clock value (the batch semantics of rl_engine.h).
    int check_and_update_many(std::vector<Request*>& reqs);

    // Drop every qualified cell whose window has ended and forget the interned identity of each one (no
    // reference analogue: it stands in for moka's capacity eviction, in_memory.rs:205-212, which bounds the
    // reference's memory; it never changes a decision on an unexpired counter).  Runs by itself once
    // `sweep_after` new counters have been interned since the last sweep (set_sweep_after; 0 = never).
    int sweep_expired(uint64_t* n_removed);
    void set_sweep_after(uint64_t n_new_counters) { sweep_after_ = n_new_counters; }
    size_t interned_counters() const { return by_key_.size(); }

    const StorageErr& last_error() const { return err_; }
    void set_clock(uint64_t now_us) { fixed_now_us_ = now_us; }
    uint64_t now_us() const;
    rl_engine* engine() { return eng_; }

private:
    GpuCounterStorage() = default;
    int fail(int rc);                            // records the engine's message, returns rc
    int fail_invalid(const std::string& msg);    // RL_ERR_INVALID raised on this side
    uint32_t limit_id(const Limit& l);           // interns; uploads / refreshes the limit row
    uint32_t wire_limit(const Limit& l) { return limit_id(l) | (l.variables.empty() ? RL_SIMPLE : 0u); }
    uint64_t key_of(uint32_t id, const Counter& c);
    int to_hit(const Counter& c, uint64_t delta, rl_hit* out);

    rl_engine* eng_ = nullptr;
    std::mutex mu_;  // guards the interning tables (the engine serialises its own calls)
    StorageErr err_;
    uint64_t fixed_now_us_ = 0;
    uint32_t max_batch_ = 0;
    uint64_t key_seq_ = 0;
    uint64_t sweep_after_ = 1u << 20, new_since_sweep_ = 0;
    int maybe_sweep();  // (mu_ held)
    int sweep_locked(uint64_t* n_removed);
    static std::string counter_ident(uint32_t id, const std::vector<std::pair<std::string, std::string>>& sorted_vars);
    std::unordered_map<std::string, uint32_t> limit_ids_;
    std::vector<rl_limit_row> rows_;
    std::vector<Limit> limit_of_id_;
    std::unordered_map<std::string, uint64_t> counter_keys_;  // (limit id, set_variables) -> key
    std::unordered_map<uint64_t, std::pair<uint32_t, std::vector<std::pair<std::string, std::string>>>> by_key_;
};

// Micro-batching aggregator: many threads call check_and_update concurrently; a dispatcher closes
// a batch at max_batch requests or max_delay_us after the first one arrived, stamps it with one
// clock value, runs it as device batches and wakes the callers.  (What an AsyncCounterStorage
// implementation — storage/mod.rs:294-310 — would do with a oneshot channel per request.)
class MicroBatcher {
public:
    MicroBatcher(GpuCounterStorage* s, uint32_t max_batch, uint32_t max_delay_us);
    ~MicroBatcher();
    int check_and_update(std::vector<Counter>& counters, uint64_t delta, bool load_counters, Authorization* auth);
    void stats(uint64_t* batches, uint64_t* requests);

private:
    struct Slot {
        GpuCounterStorage::Request req;
        bool done = false;
    };
    void run();
    GpuCounterStorage* s_;
    uint32_t max_batch_, max_delay_us_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<Slot*> queue_;
    bool stop_ = false;
    uint64_t n_batches_ = 0, n_requests_ = 0;
    std::thread worker_;
};

}  // namespace rls
