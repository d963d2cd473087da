// gpu_counter_storage.cpp — see gpu_counter_storage.hpp.  Also the C view (include/rl_storage.h).
#include "gpu_counter_storage.hpp"

#include <algorithm>
#include <chrono>
#include <cstring>

#include "../../../include/rl_storage.h"
#include "../rl_abi_guard.h"
#include <stdexcept>
#include <vector>

namespace rls {

namespace {
void append_field(std::string& s, const std::string& f) {
    // length-prefixed so that no choice of strings can make two identities collide
    s += std::to_string(f.size());
    s += ':';
    s += f;
}
std::vector<std::string> sorted_unique(std::vector<std::string> v) {
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    return v;
}
}  // namespace

std::string Limit::identity() const {  // limit.rs:177-184 (Hash) / :207-214 (PartialEq)
    std::string s;
    append_field(s, ns);
    s += std::to_string(seconds);
    s += '|';
    for (const auto& c : sorted_unique(conditions)) append_field(s, c);
    s += '|';
    for (const auto& v : sorted_unique(variables)) append_field(s, v);
    return s;
}

int GpuCounterStorage::create(uint64_t capacity_cells, uint32_t max_batch_hits, int device, GpuCounterStorage** out) {
    *out = nullptr;
    rl_config cfg{};
    cfg.device = device;
    cfg.max_batch_hits = max_batch_hits;
    cfg.capacity_cells = capacity_cells;
    cfg.max_limits = 4096;
    cfg.flags = RL_CFG_AUTO_GROW;  // the reference's storage never refuses a counter: grow, do not fail
    cfg.hash_seed = 0x9E3779B97F4A7C15ull;
    rl_engine* e = nullptr;
    const int rc = rl_engine_create(&cfg, &e);
    if (rc) return rc;  // RL_ERR_NO_DEVICE without a MI355X: there is no CPU implementation behind this
    auto* s = new GpuCounterStorage();
    s->eng_ = e;
    s->max_batch_ = max_batch_hits ? max_batch_hits : (1u << 20);
    *out = s;
    return RL_OK;
}

GpuCounterStorage::~GpuCounterStorage() {
    if (eng_) rl_engine_destroy(eng_);
}

uint64_t GpuCounterStorage::now_us() const {
    if (fixed_now_us_) return fixed_now_us_;
    using namespace std::chrono;  // SystemTime::now().duration_since(UNIX_EPOCH), atomic_expiring_value.rs:62-66
    return (uint64_t)duration_cast<microseconds>(system_clock::now().time_since_epoch()).count();
}

int GpuCounterStorage::fail(int rc) {
    err_.code = rc;
    err_.msg = rl_last_error(eng_);
    err_.transient = rl_status_is_transient(rc) != 0;
    return rc;
}

int GpuCounterStorage::fail_invalid(const std::string& msg) {
    err_.code = RL_ERR_INVALID;
    err_.msg = msg;
    err_.transient = false;
    return RL_ERR_INVALID;
}

uint32_t GpuCounterStorage::limit_id(const Limit& l) {
    const std::string ident = l.identity();
    auto it = limit_ids_.find(ident);
    uint32_t id;
    if (it == limit_ids_.end()) {
        id = (uint32_t)rows_.size();
        limit_ids_.emplace(ident, id);
        rows_.push_back(rl_limit_row{l.max_value, l.seconds});
        limit_of_id_.push_back(l);
        rl_limits_set(eng_, id, &rows_[id], 1);
    } else {
        id = it->second;
        // max_value is not part of a limit's identity: the reference reads it from the request-side
        // Counter.limit (counter.rs:64-66), so the row follows whatever the caller carries
        // (Storage::update_limit, storage/mod.rs:67-83; pinned by lib.rs:760-790).
        if (rows_[id].max_value != l.max_value) {
            rows_[id].max_value = l.max_value;
            limit_of_id_[id].max_value = l.max_value;
            rl_limits_set(eng_, id, &rows_[id], 1);
        }
        if (l.has_name) {
            limit_of_id_[id].name = l.name;
            limit_of_id_[id].has_name = true;
        }
    }
    return id;
}

std::string GpuCounterStorage::counter_ident(uint32_t id, const std::vector<std::pair<std::string, std::string>>& vars) {
    std::string k = std::to_string(id);
    k += '|';
    for (const auto& kv : vars) {
        append_field(k, kv.first);
        append_field(k, kv.second);
    }
    return k;
}

uint64_t GpuCounterStorage::key_of(uint32_t id, const Counter& c) {
    auto vars = c.set_variables;
    std::sort(vars.begin(), vars.end());
    std::string k = counter_ident(id, vars);
    auto it = counter_keys_.find(k);
    if (it != counter_keys_.end()) return it->second;
    // exact and collision-free: a dense sequence number, scrambled (an odd multiplier is a bijection
    // on u64) so that slots spread out; the two reserved tags are skipped
    uint64_t key;
    do {
        key = ++key_seq_ * 0x9E3779B97F4A7C15ull;
    } while (key >= 0xFFFFFFFFFFFFFFFEull);
    counter_keys_.emplace(std::move(k), key);
    by_key_.emplace(key, std::make_pair(id, std::move(vars)));
    if (c.is_qualified()) ++new_since_sweep_;  // (only qualified cells are ever swept)
    return key;
}

int GpuCounterStorage::sweep_locked(uint64_t* n_removed) {
    // size the row buffer for every qualified cell that could be swept: the interned counters
    std::vector<rl_cell_row> rows(by_key_.size());
    uint64_t n = 0;
    if (int rc = rl_sweep_expired_rows(eng_, now_us(), rows.data(), rows.size(), &n)) return fail(rc);
    const uint64_t got = n < rows.size() ? n : rows.size();
    for (uint64_t q = 0; q < got; ++q) {
        auto it = by_key_.find(rows[q].key);
        if (it == by_key_.end()) continue;
        counter_keys_.erase(counter_ident(it->second.first, it->second.second));
        by_key_.erase(it);
    }
    new_since_sweep_ = 0;
    if (n_removed) *n_removed = n;
    return RL_OK;
}

int GpuCounterStorage::sweep_expired(uint64_t* n_removed) {
    std::lock_guard<std::mutex> g(mu_);
    return sweep_locked(n_removed);
}

int GpuCounterStorage::maybe_sweep() {
    if (!sweep_after_ || new_since_sweep_ < sweep_after_) return RL_OK;
    return sweep_locked(nullptr);
}

int GpuCounterStorage::to_hit(const Counter& c, uint64_t delta, rl_hit* out) {
    const uint32_t id = limit_id(c.limit);
    out->limit = id | (c.is_qualified() ? 0u : RL_SIMPLE);
    out->key = key_of(id, c);
    // the 32-bit wire field; a delta beyond it travels in the call's u64 delta array (the trait's `delta: u64`)
    out->delta = delta > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)delta;
    return RL_OK;
}

int GpuCounterStorage::is_within_limits(const Counter& counter, uint64_t delta, bool* within) {
    std::lock_guard<std::mutex> g(mu_);
    rl_hit h;
    if (int rc = to_hit(counter, delta, &h)) return rc;
    uint8_t w = 0;
    if (int rc = rl_is_within_limits_batch_ex(eng_, &h, 1, delta > 0xFFFFFFFFull ? &delta : nullptr, now_us(), &w)) return fail(rc);
    *within = w != 0;
    return RL_OK;
}

int GpuCounterStorage::add_counter(const Limit& limit) {
    std::lock_guard<std::mutex> g(mu_);
    const uint32_t id = limit_id(limit);
    if (!limit.variables.empty()) return RL_OK;  // in_memory.rs:39: only limits without variables
    Counter c;
    c.limit = limit;
    if (int rc = rl_add_counter(eng_, id | RL_SIMPLE, key_of(id, c))) return fail(rc);
    return RL_OK;
}

int GpuCounterStorage::update_counter(const Counter& counter, uint64_t delta) {
    std::lock_guard<std::mutex> g(mu_);
    rl_hit h;
    if (int rc = to_hit(counter, delta, &h)) return rc;
    if (int rc = rl_update_counter_batch_ex(eng_, &h, 1, delta > 0xFFFFFFFFull ? &delta : nullptr, now_us())) return fail(rc);
    return RL_OK;
}

int GpuCounterStorage::check_and_update(std::vector<Counter>& counters, uint64_t delta, bool load_counters,
                                        Authorization* auth) {
    Request r{&counters, delta, load_counters, {}, 0};
    std::vector<Request*> one{&r};
    const int rc = check_and_update_many(one);
    *auth = r.auth;
    return rc ? rc : r.rc;
}

int GpuCounterStorage::check_and_update_many(std::vector<Request*>& reqs) {
    std::lock_guard<std::mutex> g(mu_);
    if (int rc = maybe_sweep()) return rc;  // keeps the interning tables (caller-controlled strings) bounded
    const uint64_t now = now_us();
    size_t begin = 0;
    while (begin < reqs.size()) {
        // a device batch = a run of requests with the same load_counters flag (it is a property of the
        // whole engine call), cut at max_batch hits; runs are applied in order, so the result is still
        // the reference applied to the requests one at a time
        const bool load = reqs[begin]->load_counters;
        std::vector<rl_hit> hits;
        std::vector<uint64_t> deltas;  // per request, as the trait's u64
        bool big_delta = false;
        std::vector<uint32_t> off{0};
        std::vector<std::vector<size_t>> order;  // per request: caller index of each hit
        size_t end = begin;
        bool all_single = true;
        while (end < reqs.size() && reqs[end]->load_counters == load) {
            Request* r = reqs[end];
            auto& cs = *r->counters;
            if (hits.size() + cs.size() > max_batch_ && end > begin) break;
            std::vector<size_t> ord;  // simple counters first, then qualified (in_memory.rs:105,121)
            for (size_t i = 0; i < cs.size(); ++i)
                if (!cs[i].is_qualified()) ord.push_back(i);
            for (size_t i = 0; i < cs.size(); ++i)
                if (cs[i].is_qualified()) ord.push_back(i);
            for (size_t i : ord) {
                rl_hit h;
                if (int rc = to_hit(cs[i], r->delta, &h)) return rc;
                hits.push_back(h);
            }
            off.push_back((uint32_t)hits.size());
            deltas.push_back(r->delta);
            big_delta = big_delta || r->delta > 0xFFFFFFFFull;
            all_single = all_single && cs.size() == 1;
            order.push_back(std::move(ord));
            ++end;
        }
        const uint32_t n_req = (uint32_t)(end - begin);
        std::vector<uint8_t> verdict(n_req);
        std::vector<int32_t> first(n_req);
        std::vector<uint64_t> rem(load ? hits.size() : 0), exp(load ? hits.size() : 0);
        const int rc = rl_check_and_update_batch_ex(eng_, hits.data(), (uint32_t)hits.size(),
                                                    (all_single && !load) ? nullptr : off.data(), n_req,
                                                    big_delta ? deltas.data() : nullptr, nullptr, now, load ? 1 : 0,
                                                    verdict.data(), first.data(), load ? rem.data() : nullptr,
                                                    load ? exp.data() : nullptr);
        if (rc) return fail(rc);
        for (uint32_t q = 0; q < n_req; ++q) {
            Request* r = reqs[begin + q];
            auto& cs = *r->counters;
            const auto& ord = order[q];
            if (load)
                for (size_t j = 0; j < ord.size(); ++j) {  // counter.rs:96-106
                    cs[ord[j]].remaining = rem[off[q] + j];
                    cs[ord[j]].has_remaining = true;
                    cs[ord[j]].expires_in_us = exp[off[q] + j];
                    cs[ord[j]].has_expires_in = true;
                }
            r->auth.limited = verdict[q] != 0;
            r->auth.limited_idx = verdict[q] ? (int)ord[(size_t)(first[q] - (int32_t)off[q])] : -1;
        }
        begin = end;
    }
    return RL_OK;
}

int GpuCounterStorage::get_counters(const std::vector<Limit>& limits,
                                    std::vector<std::pair<uint32_t, Counter>>* out) {
    std::lock_guard<std::mutex> g(mu_);
    const uint64_t now = now_us();
    for (uint32_t li = 0; li < limits.size(); ++li) {
        const Limit& l = limits[li];
        const uint32_t wire = wire_limit(l);
        uint64_t n = 0;
        if (int rc = rl_get_counters(eng_, wire, now, nullptr, 0, &n)) return fail(rc);
        std::vector<rl_cell_row> rows(n);
        if (n)
            if (int rc = rl_get_counters(eng_, wire, now, rows.data(), n, &n)) return fail(rc);
        for (const auto& r : rows) {
            Counter c;
            c.limit = l;
            auto it = by_key_.find(r.key);
            if (it != by_key_.end()) c.set_variables = it->second.second;
            c.remaining = l.max_value - r.value;  // in_memory.rs:166,178 (wrapping, as the release build)
            c.has_remaining = true;
            c.expires_in_us = r.expiry_us;  // ttl(now) > 0: cells with ttl == 0 are hidden by the engine
            c.has_expires_in = true;
            out->emplace_back(li, std::move(c));
        }
    }
    return RL_OK;
}

int GpuCounterStorage::delete_counters(const std::vector<Limit>& limits) {
    std::lock_guard<std::mutex> g(mu_);
    for (const Limit& l : limits)
        if (int rc = rl_delete_counters(eng_, wire_limit(l))) return fail(rc);
    return RL_OK;
}

int GpuCounterStorage::clear() {
    std::lock_guard<std::mutex> g(mu_);
    if (int rc = rl_clear(eng_)) return fail(rc);  // simple cells only, in_memory.rs:198-201
    return RL_OK;
}

// ---- MicroBatcher -------------------------------------------------------------------------------
MicroBatcher::MicroBatcher(GpuCounterStorage* s, uint32_t max_batch, uint32_t max_delay_us)
    : s_(s), max_batch_(max_batch ? max_batch : 1), max_delay_us_(max_delay_us), worker_([this] { run(); }) {}

MicroBatcher::~MicroBatcher() {
    {
        std::lock_guard<std::mutex> g(mu_);
        stop_ = true;
    }
    cv_work_.notify_all();
    worker_.join();
}

int MicroBatcher::check_and_update(std::vector<Counter>& counters, uint64_t delta, bool load_counters,
                                   Authorization* auth) {
    Slot slot;
    slot.req = GpuCounterStorage::Request{&counters, delta, load_counters, {}, 0};
    std::unique_lock<std::mutex> lk(mu_);
    queue_.push_back(&slot);
    if (queue_.size() == 1 || queue_.size() >= max_batch_) cv_work_.notify_one();
    cv_done_.wait(lk, [&] { return slot.done; });
    *auth = slot.req.auth;
    return slot.req.rc;
}

void MicroBatcher::run() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
        cv_work_.wait(lk, [&] { return stop_ || !queue_.empty(); });
        if (stop_ && queue_.empty()) return;
        // the batch closes max_delay_us after its first request arrived, or when it is full
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(max_delay_us_);
        cv_work_.wait_until(lk, deadline, [&] { return stop_ || queue_.size() >= max_batch_; });
        std::vector<Slot*> batch;
        batch.swap(queue_);
        lk.unlock();
        int rc;
        try {  // (no entry point's barrier above this thread: the callers get the status, the worker lives on)
            std::vector<GpuCounterStorage::Request*> reqs;
            reqs.reserve(batch.size());
            for (Slot* s : batch) reqs.push_back(&s->req);
            rc = s_->check_and_update_many(reqs);
        } catch (const std::bad_alloc&) {
            rc = rl_abi_caught("MicroBatcher worker", "std::bad_alloc (host memory exhausted)", RL_ERR_NOMEM);
        } catch (...) {
            rc = rl_abi_caught("MicroBatcher worker", "C++ exception", RL_ERR_INTERNAL);
        }
        lk.lock();
        ++n_batches_;
        n_requests_ += batch.size();
        for (Slot* s : batch) {
            if (rc) s->req.rc = rc;
            s->done = true;
        }
        cv_done_.notify_all();
    }
}

void MicroBatcher::stats(uint64_t* batches, uint64_t* requests) {
    std::lock_guard<std::mutex> g(mu_);
    *batches = n_batches_;
    *requests = n_requests_;
}

}  // namespace rls

// ---- C view (include/rl_storage.h) -----------------------------------------------------------------
struct rls_storage {
    rls::GpuCounterStorage* s;
    std::string create_err;
};
struct rls_batcher {
    rls::MicroBatcher* b;
    rls_storage* owner;
};

namespace {
rls::Limit to_limit(const rls_limit* l) {
    rls::Limit out;
    out.ns = l->namespace_ ? l->namespace_ : "";
    out.max_value = l->max_value;
    out.seconds = l->seconds;
    for (uint32_t i = 0; i < l->n_conditions; ++i) out.conditions.emplace_back(l->conditions[i]);
    for (uint32_t i = 0; i < l->n_variables; ++i) out.variables.emplace_back(l->variables[i]);
    std::sort(out.conditions.begin(), out.conditions.end());
    out.conditions.erase(std::unique(out.conditions.begin(), out.conditions.end()), out.conditions.end());
    std::sort(out.variables.begin(), out.variables.end());
    out.variables.erase(std::unique(out.variables.begin(), out.variables.end()), out.variables.end());
    if (l->name) {
        out.name = l->name;
        out.has_name = true;
    }
    return out;
}
rls::Counter to_counter(const rls_counter* c) {
    rls::Counter out;
    out.limit = to_limit(&c->limit);
    for (uint32_t i = 0; i < c->n_vars; ++i) out.set_variables.emplace_back(c->var_names[i], c->var_values[i]);
    std::sort(out.set_variables.begin(), out.set_variables.end());
    return out;
}
void copy_back(const rls::Counter& src, rls_counter* dst) {
    dst->has_remaining = src.has_remaining;
    dst->remaining = src.remaining;
    dst->has_expires_in = src.has_expires_in;
    dst->expires_in_us = src.expires_in_us;
}
}  // namespace

extern "C" {

// behind the same barrier as every entry point of this library (../rl_abi_guard.h): proves THIS library was built
// with it (tests/test_abi_barrier.py, no GPU needed)
int32_t rls_abi_selftest(int32_t kind) try {
    std::vector<uint64_t> unwound(16, 1ull);
    switch (kind) {
        case 1: throw std::bad_alloc();
        case 2: throw std::length_error("rls_abi_selftest: std::length_error");
        case 3: throw 42;
        case 4: {
            std::vector<uint64_t> v;
            v.resize((size_t)1 << 58);
            return (int32_t)v.size();
        }
        default: return RL_OK;
    }
} RL_ABI_CATCH

int32_t rls_storage_create(uint64_t capacity_cells, uint32_t max_batch_hits, int32_t device, rls_storage** out) try {
    if (!out) return RL_ERR_INVALID;
    *out = nullptr;
    rls::GpuCounterStorage* s = nullptr;
    const int rc = rls::GpuCounterStorage::create(capacity_cells, max_batch_hits, device, &s);
    if (rc) return rc;
    *out = new rls_storage{s, {}};
    return RL_OK;
} RL_ABI_CATCH

void rls_storage_destroy(rls_storage* s) {
    if (!s) return;
    delete s->s;
    delete s;
}

const char* rls_last_error(const rls_storage* s) { return s ? s->s->last_error().msg.c_str() : "null storage"; }

void rls_set_clock(rls_storage* s, uint64_t now_us) {
    if (s) s->s->set_clock(now_us);
}

int32_t rls_is_within_limits(rls_storage* s, const rls_counter* counter, uint64_t delta, int32_t* within) try {
    if (!s || !counter || !within) return RL_ERR_INVALID;
    bool w = false;
    const int rc = s->s->is_within_limits(to_counter(counter), delta, &w);
    *within = w ? 1 : 0;
    return rc;
} RL_ABI_CATCH

int32_t rls_add_counter(rls_storage* s, const rls_limit* limit) try {
    if (!s || !limit) return RL_ERR_INVALID;
    return s->s->add_counter(to_limit(limit));
} RL_ABI_CATCH

int32_t rls_update_counter(rls_storage* s, const rls_counter* counter, uint64_t delta) try {
    if (!s || !counter) return RL_ERR_INVALID;
    return s->s->update_counter(to_counter(counter), delta);
} RL_ABI_CATCH

int32_t rls_check_and_update(rls_storage* s, rls_counter* counters, uint32_t n, uint64_t delta, int32_t load_counters,
                             int32_t* limited, int32_t* limited_idx) try {
    if (!s || (n && !counters) || !limited) return RL_ERR_INVALID;
    std::vector<rls::Counter> cs;
    for (uint32_t i = 0; i < n; ++i) cs.push_back(to_counter(&counters[i]));
    rls::Authorization a;
    const int rc = s->s->check_and_update(cs, delta, load_counters != 0, &a);
    if (rc) return rc;
    for (uint32_t i = 0; i < n; ++i) copy_back(cs[i], &counters[i]);
    *limited = a.limited ? 1 : 0;
    if (limited_idx) *limited_idx = a.limited_idx;
    return RL_OK;
} RL_ABI_CATCH

int32_t rls_get_counters(rls_storage* s, const rls_limit* limits, uint32_t n_limits, rls_emit_fn emit, void* user) try {
    if (!s || (n_limits && !limits) || !emit) return RL_ERR_INVALID;
    std::vector<rls::Limit> ls;
    for (uint32_t i = 0; i < n_limits; ++i) ls.push_back(to_limit(&limits[i]));
    std::vector<std::pair<uint32_t, rls::Counter>> out;
    const int rc = s->s->get_counters(ls, &out);
    if (rc) return rc;
    for (const auto& pc : out) {
        std::vector<const char*> names, values;
        for (const auto& kv : pc.second.set_variables) {
            names.push_back(kv.first.c_str());
            values.push_back(kv.second.c_str());
        }
        emit(user, pc.first, names.data(), values.data(), (uint32_t)names.size(), pc.second.remaining,
             pc.second.expires_in_us);
    }
    return RL_OK;
} RL_ABI_CATCH

int32_t rls_delete_counters(rls_storage* s, const rls_limit* limits, uint32_t n_limits) try {
    if (!s || (n_limits && !limits)) return RL_ERR_INVALID;
    std::vector<rls::Limit> ls;
    for (uint32_t i = 0; i < n_limits; ++i) ls.push_back(to_limit(&limits[i]));
    return s->s->delete_counters(ls);
} RL_ABI_CATCH

int32_t rls_clear(rls_storage* s) try {
    return s ? s->s->clear() : RL_ERR_INVALID;
} RL_ABI_CATCH

int32_t rls_sweep_expired(rls_storage* s, uint64_t* n_removed) try {
    return s ? s->s->sweep_expired(n_removed) : RL_ERR_INVALID;
} RL_ABI_CATCH

void rls_set_sweep_after(rls_storage* s, uint64_t n_new_counters) {
    if (s) s->s->set_sweep_after(n_new_counters);
}

uint64_t rls_interned_counters(const rls_storage* s) { return s ? (uint64_t)s->s->interned_counters() : 0; }

int32_t rls_batcher_create(rls_storage* s, uint32_t max_batch, uint32_t max_delay_us, rls_batcher** out) try {
    if (!s || !out) return RL_ERR_INVALID;
    *out = new rls_batcher{new rls::MicroBatcher(s->s, max_batch, max_delay_us), s};
    return RL_OK;
} RL_ABI_CATCH

void rls_batcher_destroy(rls_batcher* b) {
    if (!b) return;
    delete b->b;
    delete b;
}

int32_t rls_batcher_check_and_update(rls_batcher* b, rls_counter* counters, uint32_t n, uint64_t delta,
                                     int32_t load_counters, int32_t* limited, int32_t* limited_idx) try {
    if (!b || (n && !counters) || !limited) return RL_ERR_INVALID;
    std::vector<rls::Counter> cs;
    for (uint32_t i = 0; i < n; ++i) cs.push_back(to_counter(&counters[i]));
    rls::Authorization a;
    const int rc = b->b->check_and_update(cs, delta, load_counters != 0, &a);
    if (rc) return rc;
    for (uint32_t i = 0; i < n; ++i) copy_back(cs[i], &counters[i]);
    *limited = a.limited ? 1 : 0;
    if (limited_idx) *limited_idx = a.limited_idx;
    return RL_OK;
} RL_ABI_CATCH

int32_t rls_check_and_update_repeat(rls_storage* s, rls_counter* counters, uint32_t n, uint64_t delta, uint32_t iterations,
                                    uint64_t* elapsed_ns, uint32_t* n_limited) try {
    if (!s || (n && !counters) || !elapsed_ns) return RL_ERR_INVALID;
    std::vector<rls::Counter> cs;
    for (uint32_t i = 0; i < n; ++i) cs.push_back(to_counter(&counters[i]));
    uint32_t limited = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t it = 0; it < iterations; ++it) {
        rls::Authorization a;
        const int rc = s->s->check_and_update(cs, delta, false, &a);
        if (rc) return rc;
        limited += a.limited ? 1u : 0u;
    }
    *elapsed_ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    if (n_limited) *n_limited = limited;
    return RL_OK;
} RL_ABI_CATCH

void rls_batcher_stats(rls_batcher* b, uint64_t* batches, uint64_t* requests) {
    uint64_t nb = 0, nr = 0;
    if (b) b->b->stats(&nb, &nr);
    if (batches) *batches = nb;
    if (requests) *requests = nr;
}

}  // extern "C"
