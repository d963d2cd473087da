// ingest.cpp — host-side ingest for the on-device limit matcher (include/rl_ingest.h): compiles Limits
// into the match table of rl_match_table_set and dictionary-encodes requests for
// rl_match_and_check_batch.  What it restates from the reference:
//   Limit identity                     limitador/src/limit.rs:177-214
//   conditions / variables as sets     limitador/src/limit.rs:34-48 (BTreeSet)
//   "no limits -> not limited"         limitador/src/lib.rs:434-440
//   request shape                      limitador-server/src/envoy_rls/server.rs:91-208
// The predicate shapes are the ones rl_match.hpp evaluates; anything else is CEL and stays with the
// caller (RLI_HOST_ONLY).
#include "../../../include/rl_ingest.h"
#include "../../../include/rl_keyhash.h"
#include "../rl_abi_guard.h"
#include <stdexcept>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <random>
#include <shared_mutex>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

// experiment / diagnostics switches exist only in -DRL_EXPERIMENT builds (see rl_engine.hip)
#ifdef RL_EXPERIMENT
#define RL_EXP_ENV(name) getenv(name)
#else
#define RL_EXP_ENV(name) (static_cast<const char*>(nullptr))
#endif

namespace {

struct Dictionary {  // exact string -> dense id
    std::unordered_map<std::string, uint32_t> ids;
    uint32_t intern(const std::string& s) {
        auto it = ids.find(s);
        if (it != ids.end()) return it->second;
        const uint32_t id = (uint32_t)ids.size();
        ids.emplace(s, id);
        return id;
    }
    int64_t find(const std::string& s) const {
        auto it = ids.find(s);
        return it == ids.end() ? -1 : (int64_t)it->second;
    }
};

struct Cond {
    std::string key;
    uint32_t op;  // 0 ==, 1 !=
    std::string value;
    bool operator<(const Cond& o) const { return std::tie(key, op, value) < std::tie(o.key, o.op, o.value); }
    bool operator==(const Cond& o) const { return key == o.key && op == o.op && value == o.value; }
};

struct LimitSpec {
    std::string ns;
    uint64_t max_value, seconds;
    std::vector<Cond> conds;        // parsed, in the order of cond_src
    std::vector<std::string> vars;  // descriptor keys, in the order of var_src
    // Identity is on the SOURCE expressions, like the reference's (limit.rs:177-214: namespace, seconds and
    // the BTreeSets of Predicate / Expression, which compare by their source text): two spellings of one
    // predicate are two limits with two counters.
    std::vector<std::string> cond_src, var_src;  // sorted, unique
    std::string name;                            // Limit.name: not identity, only reported (limit.rs:107-113)
    bool has_name = false;
    using Identity = std::tuple<std::string, uint64_t, std::vector<std::string>, std::vector<std::string>>;
    Identity identity() const { return Identity(ns, seconds, cond_src, var_src); }
};

void skip_ws(const std::string& s, size_t& i) {
    while (i < s.size() && std::isspace((unsigned char)s[i])) ++i;
}

// 'text' or "text" -> text
bool parse_quoted(const std::string& s, size_t& i, std::string* out) {
    skip_ws(s, i);
    if (i >= s.size() || (s[i] != '\'' && s[i] != '"')) return false;
    const char q = s[i++];
    const size_t b = i;
    while (i < s.size() && s[i] != q) ++i;
    if (i >= s.size()) return false;
    *out = s.substr(b, i - b);
    ++i;
    return true;
}

// How a limit's expressions reach the request's strings depends on what the caller's Context binds:
//   RLI_BIND_DESCRIPTORS (default; both transports: envoy_rls/server.rs:136-137, http_api/server.rs:140-141 bind
//       ONLY the list `descriptors`):   descriptors[0]['key'] | descriptors[0]["key"] | descriptors[0].ident
//       A bare `x == '1'` or `req.method == 'GET'` references an unbound variable there and never applies, and
//       `descriptors[0].a.b` is nested member access, not the key "a.b".
//   RLI_BIND_ROOT (library callers, Context::from(HashMap): limit/cel.rs:81-96,153-156 binds every key as a
//       root variable; what limit.rs:239-348 tests with):   ident
// Everything else is CEL the device matcher does not restate: RLI_HOST_ONLY.
static bool cel_ident(const std::string& s, size_t& i, std::string* out) {
    const size_t b = i;  // [A-Za-z_][A-Za-z0-9_]*
    while (i < s.size() && (std::isalnum((unsigned char)s[i]) || s[i] == '_')) ++i;
    if (i == b || std::isdigit((unsigned char)s[b])) return false;
    if (i < s.size() && (s[i] == '.' || s[i] == '[' || s[i] == '(')) return false;  // nested access / call: CEL
    *out = s.substr(b, i - b);
    return true;
}
// The name a (descriptor index, key) pair goes by in the dictionaries and tables: the key itself for descriptors[0] (the
// common case: no string is built per entry), "\x1f<index>\x1f<key>" for the others — and for a descriptors[0] key that
// itself starts with \x1f, so that the mapping stays injective.
constexpr uint32_t MAX_DESC_INDEX = 63;  // descriptors[0..63] may be read by limits (a bit mask of the ones that are)
std::string desc_key_name(uint32_t desc, const char* key, size_t len) {
    if (desc == 0 && (len == 0 || key[0] != '\x1f')) return std::string(key, len);
    std::string o = "\x1f" + std::to_string(desc) + "\x1f";
    o.append(key, len);
    return o;
}
// ... and back: -> (descriptor index, offset of the key's own bytes)
void desc_key_split(const std::string& name, uint32_t* desc, size_t* key_off) {
    *desc = 0;
    *key_off = 0;
    if (name.empty() || name[0] != '\x1f') return;
    const size_t e = name.find('\x1f', 1);
    if (e == std::string::npos) return;
    *desc = (uint32_t)strtoul(name.substr(1, e - 1).c_str(), nullptr, 10);
    *key_off = e + 1;
}

bool parse_key_ref(const std::string& s, size_t& i, std::string* key, int binding) {
    skip_ws(s, i);
    if (binding == RLI_BIND_ROOT) return cel_ident(s, i, key);
    // descriptors[N] — the transports bind the whole list (envoy_rls/server.rs:121-137: one map per descriptor), so a limit
    // may read any of them: `descriptors[1].y == '2'` (envoy_rls/server.rs:520, kuadrant_service.rs:415)
    static const std::string pre = "descriptors[";
    if (s.compare(i, pre.size(), pre) != 0) return false;
    i += pre.size();
    skip_ws(s, i);
    const size_t d0 = i;
    while (i < s.size() && std::isdigit((unsigned char)s[i])) ++i;
    if (i == d0 || i - d0 > 3) return false;
    const uint32_t desc = (uint32_t)strtoul(s.substr(d0, i - d0).c_str(), nullptr, 10);
    if (desc > MAX_DESC_INDEX) return false;
    skip_ws(s, i);
    if (i >= s.size() || s[i] != ']') return false;
    ++i;
    skip_ws(s, i);
    std::string k;
    if (i < s.size() && s[i] == '[') {
        ++i;
        if (!parse_quoted(s, i, &k)) return false;
        skip_ws(s, i);
        if (i >= s.size() || s[i] != ']') return false;
        ++i;
    } else {
        if (i >= s.size() || s[i] != '.') return false;
        ++i;
        if (!cel_ident(s, i, &k)) return false;
    }
    *key = desc_key_name(desc, k.data(), k.size());
    return true;
}

bool parse_condition(const std::string& s, Cond* c, int binding) {
    size_t i = 0;
    if (!parse_key_ref(s, i, &c->key, binding)) return false;
    skip_ws(s, i);
    if (s.compare(i, 2, "==") == 0) c->op = 0;
    else if (s.compare(i, 2, "!=") == 0) c->op = 1;
    else return false;
    i += 2;
    if (!parse_quoted(s, i, &c->value)) return false;
    skip_ws(s, i);
    return i == s.size();
}

bool parse_variable(const std::string& s, std::string* key, int binding) {
    size_t i = 0;
    if (!parse_key_ref(s, i, key, binding)) return false;
    skip_ws(s, i);
    return i == s.size();
}

// ---- protobuf wire format, just enough for RateLimitRequest ------------------------------------------
struct Wire {
    const uint8_t* p;
    const uint8_t* end;
    bool varint(uint64_t* v) {
        uint64_t r = 0;
        for (int shift = 0; shift < 64 && p < end; shift += 7) {
            const uint8_t b = *p++;
            r |= (uint64_t)(b & 0x7F) << shift;
            if (!(b & 0x80)) {
                *v = r;
                return true;
            }
        }
        return false;
    }
    bool bytes(Wire* sub) {  // length-delimited payload
        uint64_t n;
        if (!varint(&n) || n > (uint64_t)(end - p)) return false;
        sub->p = p;
        sub->end = p + n;
        p += n;
        return true;
    }
    bool skip(uint32_t wire_type) {
        uint64_t v;
        Wire w;
        switch (wire_type) {
            case 0: return varint(&v);
            case 1: if (end - p < 8) return false; p += 8; return true;
            case 2: return bytes(&w);
            case 5: if (end - p < 4) return false; p += 4; return true;
            default: return false;  // groups are not used by these messages
        }
    }
    bool done() const { return p >= end; }
};

// What prost — the reference's decoder (envoy_rls/server.rs, tonic + prost) — rejects and a lenient reader would not
// (ADVICE r04): a KNOWN field with another wire type than its declared one is a DecodeError ("invalid wire type"), and a
// `string` field must be valid UTF-8 (Rust's str::from_utf8: no overlong forms, no surrogates, nothing above U+10FFFF).
// A message the reference would have answered with a gRPC decode error must not create a counter here.  (What stays more
// lenient than prost, documented in rl_ingest.h: descriptors no limit reads and the nested messages this path does
// not read — RateLimitOverride, HitsAddend — are skipped by wire type without being validated.)
bool utf8_ok(const uint8_t* p, const uint8_t* end) {
    while (p < end) {
        const uint8_t b = *p++;
        if (b < 0x80) continue;
        uint32_t need;
        uint8_t lo = 0x80, hi = 0xBF;
        if (b >= 0xC2 && b <= 0xDF) need = 1;
        else if (b >= 0xE0 && b <= 0xEF) {
            need = 2;
            if (b == 0xE0) lo = 0xA0;
            if (b == 0xED) hi = 0x9F;
        } else if (b >= 0xF0 && b <= 0xF4) {
            need = 3;
            if (b == 0xF0) lo = 0x90;
            if (b == 0xF4) hi = 0x8F;
        } else return false;
        if ((uint32_t)(end - p) < need) return false;
        if (*p < lo || *p > hi) return false;
        ++p;
        for (uint32_t q = 1; q < need; ++q, ++p)
            if ((*p & 0xC0) != 0x80) return false;
    }
    return true;
}

// RateLimitDescriptor.Entry { key = 1; value = 2 }
bool parse_entry(Wire w, std::string* key, std::string* value) {
    while (!w.done()) {
        uint64_t tag;
        if (!w.varint(&tag)) return false;
        const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
        Wire sub;
        if (field == 1 || field == 2) {
            if (wt != 2 || !w.bytes(&sub) || !utf8_ok(sub.p, sub.end)) return false;
            (field == 1 ? key : value)->assign(reinterpret_cast<const char*>(sub.p), (size_t)(sub.end - sub.p));
        } else if (!w.skip(wt)) {
            return false;
        }
    }
    return true;
}

// RateLimitDescriptor { repeated Entry entries = 1; RateLimitOverride limit = 2 } — descriptor number `desc` of the request:
// its entries are appended under their (descriptor, key) names; inside ONE descriptor a repeated key keeps its LAST value
// (HashMap::insert, envoy_rls/server.rs:121-128)
bool parse_descriptor(Wire w, uint32_t desc, std::vector<std::pair<std::string, std::string>>* entries) {
    const size_t first = entries->size();
    while (!w.done()) {
        uint64_t tag;
        if (!w.varint(&tag)) return false;
        const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
        Wire sub;
        if (field == 1) {
            if (wt != 2 || !w.bytes(&sub)) return false;
            std::string k, v;
            if (!parse_entry(sub, &k, &v)) return false;
            if (desc != 0 || (!k.empty() && k[0] == '\x1f')) k = desc_key_name(desc, k.data(), k.size());
            bool replaced = false;
            for (size_t q = first; q < entries->size(); ++q)
                if ((*entries)[q].first == k) {
                    (*entries)[q].second = v;
                    replaced = true;
                }
            if (!replaced) entries->emplace_back(std::move(k), std::move(v));
        } else if (!w.skip(wt)) {
            return false;
        }
    }
    return true;
}

}  // namespace

struct rli_ingest {
    Dictionary ns_ids, key_ids, val_ids;
    std::vector<LimitSpec> limits;  // index = limit id
    std::map<LimitSpec::Identity, uint32_t> by_identity;
    // compiled
    bool compiled = false;
    std::vector<rl_limit_row> rows;
    std::vector<rl_match_limit> table;
    std::vector<rl_match_cond> conds;
    int binding = RLI_BIND_DESCRIPTORS;  // rli_set_binding
    uint32_t n_ns_installed = 0;     // namespaces the installed match table knows (ids beyond it: no limits)
    uint32_t value_cap = 1u << 24;   // most distinct descriptor values the dictionary takes (rli_set_value_cap)
    // RLI_KEYS_HASHED (rli_set_key_mode): no request ever touches the dictionaries — the device decodes the messages,
    // compares the table's strings as bytes and keys every counter by a hash of its canonical key bytes (rl_keyhash.h)
    int key_mode = RLI_KEYS_EXACT;
    uint64_t desc_mask = 1;          // descriptor indices some limit reads (bit i = descriptors[i]; descriptors[0] always)
    // RLI_KEYS_EXACT, limits with more than two variables: the packed key has room for two value ids (rl_match_key), so such
    // a limit reads ONE synthetic descriptor key on the device, whose value is the id of the TUPLE of its variables' values —
    // interned here like every other string (exact: two tuples never share an id), added to the request when the request
    // carries all of the limit's variables.
    struct Composite {
        uint32_t limit, syn_key;          // the limit's id; key id of its synthetic entry
        std::vector<uint32_t> var_keys;   // key ids of its variables, in variable-name order
    };
    std::vector<std::vector<Composite>> composites;  // [namespace id]
    std::map<std::vector<uint32_t>, uint32_t> tuple_ids;  // (limit, value ids...) -> tuple id
    std::vector<rl_h128> prefix;     // [limit id]: hash of the limit's canonical prefix (compiled)
    std::vector<uint32_t> more_vars; // compiled: descriptor key ids of the limits with more than two variables (RLI_KEYS_HASHED)
    rl_hkey hash_key{0, 0};          // the secret every hash of this mode is keyed with (rl_keyhash.h): random at rli_create,
                                     // rli_set_hash_key to share it between front-ends / bring it back with a snapshot
    // batch
    std::vector<uint32_t> req_ns, req_delta, ent_off{0}, ent_key, ent_val;
    std::string err;
    // What each limit contributes to X-RateLimit-Limit lives on the device (rl_resp_table_set): re-sent to an engine when a
    // limit's max / name changed since the last time, or the engine is another one
    uint64_t resp_version = 1, resp_sent = 0;
    const rl_engine* resp_engine = nullptr;
    // The dictionaries are read by many decoding threads at once (rli_serve_batch splits a large batch over
    // threads) and written when a request brings a value never seen before: readers share, a writer excludes.
    mutable std::shared_mutex dict_mu;
    // Serving calls in flight: the hashed-key path with device-built responses takes ONE of the engine's RL_SERVE_SETS sets
    // (engine staging slots 4 s .. 4 s + 3, piece events, a helper pool), so several threads' calls overlap — one packs,
    // one copies in / decides, the others' responses cross PCIe and are handed on; every other form of the call takes ALL
    // of them (it uses this ingest's batch arrays and set 0's staging: one at a time, as before).
    std::mutex set_mu;
    std::condition_variable set_cv;
    bool set_busy[RL_SERVE_SETS] = {};
    std::mutex err_mu;   // `err` is written by whichever serving thread fails
    std::mutex frag_mu;  // send_fragments
};

namespace {
struct SetLease {
    rli_ingest* g;
    rl_engine* e;
    int set = 0;
    bool both = false;
    bool bytes_pending = false;  // a *_serve_batch call succeeded with RL_SERVE_ASYNC: its last byte must be waited for
    SetLease(rli_ingest* g_, rl_engine* e_, bool exclusive) : g(g_), e(e_), both(exclusive) {
        std::unique_lock<std::mutex> lk(g->set_mu);
        auto free_set = [&]() -> int {
            for (int q = 0; q < RL_SERVE_SETS; ++q)
                if (!g->set_busy[q]) return q;
            return -1;
        };
        auto any_busy = [&] {
            for (bool b : g->set_busy)
                if (b) return true;
            return false;
        };
        if (exclusive) {
            g->set_cv.wait(lk, [&] { return !any_busy(); });
            for (bool& b : g->set_busy) b = true;
        } else {
            g->set_cv.wait(lk, [&] { return free_set() >= 0; });
            set = free_set();
            g->set_busy[set] = true;
        }
    }
    ~SetLease() {
        // Whatever way the call ends — also an exception out of a scatter chunk, rethrown into the ABI barrier — the set is
        // only handed on once every byte of its responses has arrived (the next call would rewrite the staging
        // and the piece events under a scatter thread that still reads them).
        if (bytes_pending) (void)rl_serve_wait_set(e, (uint32_t)set, ~0ull);
        {
            std::lock_guard<std::mutex> lk(g->set_mu);
            if (both)
                for (bool& b : g->set_busy) b = false;
            else
                g->set_busy[set] = false;
        }
        g->set_cv.notify_all();
    }
};
}  // namespace

// One request, dictionary-encoded: what batch_add_sv appends to the batch arrays.
struct EncReq {
    uint32_t ns = 0, delta = 1;
    std::vector<std::pair<uint32_t, uint32_t>> kv;  // (key id, value id), a repeated key reduced to its last value
};

static int32_t gfail(rli_ingest* g, int32_t rc, const char* fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    std::lock_guard<std::mutex> lk(g->err_mu);
    g->err = buf;
    return rc;
}

// Threads for the host side of a batch (decode + dictionary encoding before the device call, response bytes after
// it): one per 1024 messages, at most half the hardware threads and 32; RLI_THREADS overrides (1 = serial).
static uint32_t serve_threads(uint32_t n) {
    uint32_t cap = std::max(1u, std::min(32u, std::thread::hardware_concurrency() / 2));
    if (const char* v = getenv("RLI_THREADS")) cap = (uint32_t)std::max(1, atoi(v));
    return std::max(1u, std::min(cap, n / 1024));
}

// The helper threads of the wire path's host side, started once (spawning and joining 31 threads twice per batch was
// ~0.9 ms of a 3.8 ms batch of 32 768 messages): chunk c of a job is run by whichever thread takes it.
class ChunkPool {
    struct Job {
        const std::function<void(uint32_t)>* f;
        uint32_t total;
        std::atomic<uint32_t> next{0};
        // An exception in a chunk (std::bad_alloc while a helper encodes its share) must not end a helper thread —
        // that is std::terminate for the whole host.  The first one is kept and rethrown by run() on the CALLER's
        // thread, where the entry point's barrier (rl_abi_guard.h) turns it into a status.
        std::mutex err_mu;
        std::exception_ptr err;
    };

public:
    // One pool per serving set (several serving calls may be in flight, rl_ingest.h: the pack of one runs beside the scatter
    // of another — one pool would queue them on its `job_mu`).
    static ChunkPool& get(uint32_t which = 0) {
        static ChunkPool p[RL_SERVE_SETS];
        return p[which % RL_SERVE_SETS];
    }
    // f(chunk) for chunk in [0, n_chunks), on the pool's threads and the caller's; returns when every chunk is done and
    // no helper is inside the job any more.  One job at a time (callers queue up on `job_mu`).
    void run(uint32_t n_chunks, const std::function<void(uint32_t)>& f) {
        if (n_chunks <= 1 || workers.empty()) {
            for (uint32_t c = 0; c < n_chunks; ++c) f(c);
            return;
        }
        std::lock_guard<std::mutex> one(job_mu);
        Job j;
        j.f = &f;
        j.total = n_chunks;
        {
            std::lock_guard<std::mutex> l(mu);
            job = &j;
            ++generation;
        }
        cv_work.notify_all();
        work(j);
        std::unique_lock<std::mutex> l(mu);
        job = nullptr;  // (a helper that wakes up late finds no job)
        cv_done.wait(l, [&] { return active == 0; });
        l.unlock();
        if (j.err) std::rethrow_exception(j.err);
    }

private:
    ChunkPool() {
        uint32_t nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency() / 2));
        if (const char* v = getenv("RLI_THREADS")) nt = (uint32_t)std::max(1, atoi(v));
        for (uint32_t t = 1; t < nt; ++t) workers.emplace_back([this] { loop(); });
    }
    ~ChunkPool() {
        {
            std::lock_guard<std::mutex> l(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto& w : workers) w.join();
    }
    static void work(Job& j) {
        for (;;) {
            const uint32_t c = j.next.fetch_add(1, std::memory_order_relaxed);
            if (c >= j.total) break;
            try {
                (*j.f)(c);
            } catch (...) {
                std::lock_guard<std::mutex> g(j.err_mu);
                if (!j.err) j.err = std::current_exception();
                j.next.store(j.total, std::memory_order_relaxed);  // (nobody starts another chunk of a failed job)
            }
        }
    }
    void loop() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> l(mu);
        for (;;) {
            cv_work.wait(l, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            Job* j = job;
            if (!j) continue;
            ++active;  // (under the lock: run() cannot return, and the job cannot die, while active != 0)
            l.unlock();
            work(*j);
            l.lock();
            if (--active == 0) cv_done.notify_all();
        }
    }
    std::vector<std::thread> workers;
    std::mutex mu, job_mu;
    std::condition_variable cv_work, cv_done;
    Job* job = nullptr;
    uint32_t active = 0;
    uint64_t generation = 0;
    bool stop = false;
};

template <class F>
static void parallel_chunks(uint32_t n, uint32_t threads, F f, uint32_t pool = 0) {  // f(lo, hi)
    if (threads <= 1 || n == 0) {
        f(0u, n);
        return;
    }
    const uint32_t per = (n + threads - 1) / threads;
    const std::function<void(uint32_t)> job = [&](uint32_t t) {
        const uint32_t lo = std::min(n, t * per), hi = std::min(n, (t + 1) * per);
        if (lo < hi) f(lo, hi);
    };
    ChunkPool::get(pool).run(threads, job);
}

extern "C" {

// behind the same barrier as every entry point of this library (../rl_abi_guard.h): proves THIS library was built
// with it (tests/test_abi_barrier.py, no GPU needed)
int32_t rli_abi_selftest(int32_t kind) try {
    std::vector<uint64_t> unwound(16, 1ull);
    switch (kind) {
        case 1: throw std::bad_alloc();
        case 2: throw std::length_error("rli_abi_selftest: std::length_error");
        case 3: throw 42;
        case 4: {
            std::vector<uint64_t> v;
            v.resize((size_t)1 << 58);
            return (int32_t)v.size();
        }
        default: return RL_OK;
    }
} RL_ABI_CATCH

int32_t rli_create(rli_ingest** out) try {
    if (!out) return RL_ERR_INVALID;
    rli_ingest* g = new (std::nothrow) rli_ingest();
    if (!g) return RL_ERR_NOMEM;
    g->ns_ids.intern("");  // namespace id 0: the namespace without limits
    {   // 128 bits from the OS: the key of RLI_KEYS_HASHED's SipHash (std::random_device reads /dev/urandom / getrandom here)
        std::random_device rd;
        g->hash_key.k0 = ((uint64_t)rd() << 32) | rd();
        g->hash_key.k1 = ((uint64_t)rd() << 32) | rd();
    }
    *out = g;
    return RL_OK;
} RL_ABI_CATCH

void rli_destroy(rli_ingest* g) { delete g; }

const char* rli_last_error(const rli_ingest* g) { return g ? g->err.c_str() : "null ingest"; }

int32_t rli_add_limit(rli_ingest* g, const char* ns, uint64_t max_value, uint64_t seconds,
                      const char* const* conditions, uint32_t n_conditions, const char* const* variables,
                      uint32_t n_variables) try {
    if (!g || !ns || (n_conditions && !conditions) || (n_variables && !variables)) return RL_ERR_INVALID;
    if (!*ns) return gfail(g, RL_ERR_INVALID, "empty namespace");
    LimitSpec L;
    L.ns = ns;
    L.max_value = max_value;
    L.seconds = seconds;
    std::set<std::string> csrc, vsrc;
    for (uint32_t i = 0; i < n_conditions; ++i) {
        if (!conditions[i]) return gfail(g, RL_ERR_INVALID, "null condition");
        csrc.insert(conditions[i]);
    }
    for (uint32_t i = 0; i < n_variables; ++i) {
        if (!variables[i]) return gfail(g, RL_ERR_INVALID, "null variable");
        vsrc.insert(variables[i]);
    }
    L.cond_src.assign(csrc.begin(), csrc.end());
    L.var_src.assign(vsrc.begin(), vsrc.end());
    for (const std::string& src : L.cond_src) {
        Cond c;
        if (!parse_condition(src, &c, g->binding))
            return gfail(g, RLI_HOST_ONLY, "condition `%s` is not descriptors[N]['key'] ==|!= 'value' (N <= %u): stays on the host",
                         src.c_str(), MAX_DESC_INDEX);
        L.conds.push_back(c);
    }
    for (const std::string& src : L.var_src) {
        std::string k;
        if (!parse_variable(src, &k, g->binding))
            return gfail(g, RLI_HOST_ONLY, "variable `%s` is not descriptors[N]['key'] (N <= %u): stays on the host", src.c_str(), MAX_DESC_INDEX);
        L.vars.push_back(k);
    }
    if (L.vars.size() > 8) return gfail(g, RLI_HOST_ONLY, "more than eight variables: stays on the host");
    auto reads = [&](const std::string& name) {  // the descriptors this limit reads are decoded from now on
        uint32_t desc;
        size_t off;
        desc_key_split(name, &desc, &off);
        g->desc_mask |= 1ull << desc;
    };
    for (const Cond& c : L.conds) reads(c.key);
    for (const std::string& v : L.vars) reads(v);
    auto it = g->by_identity.find(L.identity());
    if (it != g->by_identity.end()) {  // same limit (limit.rs:177-214): max_value is not identity
        g->limits[it->second].max_value = max_value;
        g->compiled = false;
        ++g->resp_version;
        return (int32_t)it->second;
    }
    if (g->limits.size() >= 4095) return gfail(g, RL_ERR_INVALID, "more than 4095 limits");
    const uint32_t id = (uint32_t)g->limits.size();
    g->by_identity.emplace(L.identity(), id);
    g->limits.push_back(std::move(L));
    g->compiled = false;
    ++g->resp_version;
    return (int32_t)id;
} RL_ABI_CATCH


// The canonical key bytes of a counter WITHOUT its variables' values (include/rl_keyhash.h; storage/keys.rs:220-248:
// version byte 1 + postcard of CounterKey { ns, seconds, conditions (sorted), variables (sorted by name) }).
static void pc_varint(std::string& o, uint64_t v) {
    while (v >= 0x80) {
        o.push_back((char)(v | 0x80));
        v >>= 7;
    }
    o.push_back((char)v);
}
static void pc_str(std::string& o, const std::string& s) {
    pc_varint(o, s.size());
    o += s;
}
static rl_h128 canonical_prefix_hash(const LimitSpec& L, rl_hkey key) {
    std::string o;
    o.push_back((char)1);
    pc_str(o, L.ns);
    pc_varint(o, L.seconds);
    pc_varint(o, L.cond_src.size());
    for (const std::string& c : L.cond_src) pc_str(o, c);  // (std::set order = Vec<String>::sort: bytewise)
    pc_varint(o, L.var_src.size());
    for (const std::string& v : L.var_src) pc_str(o, v);
    return rl_kh_bytes(reinterpret_cast<const uint8_t*>(o.data()), (uint32_t)o.size(), key);
}

int32_t rli_compile(rli_ingest* g) try {
    if (!g) return RL_ERR_INVALID;
    const uint32_t n = (uint32_t)g->limits.size();
    g->rows.assign(n, rl_limit_row{0, 0});
    g->prefix.assign(n, rl_h128{0, 0});
    std::vector<uint32_t> order(n), ns_of(n);
    for (uint32_t i = 0; i < n; ++i) {
        g->rows[i] = rl_limit_row{g->limits[i].max_value, g->limits[i].seconds};
        g->prefix[i] = canonical_prefix_hash(g->limits[i], g->hash_key);
        ns_of[i] = g->ns_ids.intern(g->limits[i].ns);
        order[i] = i;
    }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return ns_of[a] < ns_of[b]; });
    g->table.clear();
    g->conds.clear();
    g->more_vars.clear();
    g->composites.assign(g->ns_ids.ids.size(), {});
    auto key_id = [&](const std::string& name) {
        uint32_t desc;
        size_t off;
        desc_key_split(name, &desc, &off);
        g->desc_mask |= 1ull << desc;
        return g->key_ids.intern(name);
    };
    for (uint32_t id : order) {
        const LimitSpec& L = g->limits[id];
        rl_match_limit m{};
        m.limit = id | (L.vars.empty() ? RL_SIMPLE : 0u);
        m.ns = ns_of[id];
        m.cond_off = (uint32_t)g->conds.size();
        m.n_cond = (uint32_t)L.conds.size();
        m.n_vars = (uint32_t)L.vars.size();
        if (L.vars.size() <= 2) {
            for (size_t q = 0; q < L.vars.size(); ++q) m.var_key[q] = key_id(L.vars[q]);
        } else if (g->key_mode == RLI_KEYS_HASHED) {
            // the device hashes every variable's value itself: the key ids travel beside the table (rl_match_table_set_ex)
            m.var_key[0] = (uint32_t)g->more_vars.size();
            for (const std::string& v : L.vars) g->more_vars.push_back(key_id(v));
        } else {
            // exact keys: ONE synthetic variable whose value is the id of the tuple of the real ones (rli_ingest::Composite)
            rli_ingest::Composite c;
            c.limit = id;
            c.syn_key = g->key_ids.intern(std::string("\x1e") + std::to_string(id));
            for (const std::string& v : L.vars) c.var_keys.push_back(key_id(v));
            g->composites[m.ns].push_back(c);
            m.n_vars = 1;
            m.var_key[0] = c.syn_key;
        }
        for (const Cond& c : L.conds) {
            const uint32_t v = g->val_ids.intern(c.value);
            if (v >> 26) return gfail(g, RL_ERR_INVALID, "more than 2^26 distinct values");
            g->conds.push_back(rl_match_cond{key_id(c.key), c.op, v});
        }
        g->table.push_back(m);
    }
    g->compiled = true;
    return RL_OK;
} RL_ABI_CATCH

uint32_t rli_n_limits(const rli_ingest* g) { return g ? (uint32_t)g->limits.size() : 0; }
uint32_t rli_n_conds(const rli_ingest* g) { return g ? (uint32_t)g->conds.size() : 0; }
uint32_t rli_n_namespaces(const rli_ingest* g) { return g ? (uint32_t)g->ns_ids.ids.size() : 0; }
const rl_limit_row* rli_limit_rows(const rli_ingest* g) { return g && g->compiled ? g->rows.data() : nullptr; }
const rl_match_limit* rli_match_limits(const rli_ingest* g) { return g && g->compiled ? g->table.data() : nullptr; }
const rl_match_cond* rli_match_conds(const rli_ingest* g) { return g && g->compiled ? g->conds.data() : nullptr; }

int32_t rli_install(rli_ingest* g, rl_engine* e) try {
    if (!g || !e) return RL_ERR_INVALID;
    if (!g->compiled) {
        const int32_t rc = rli_compile(g);
        if (rc) return rc;
    }
    int32_t rc = rl_limits_set(e, 0, g->rows.data(), (uint32_t)g->rows.size());
    if (rc) return gfail(g, rc, "rl_limits_set: %s", rl_last_error(e));
    rc = rl_match_table_set_ex(e, g->table.data(), (uint32_t)g->table.size(), g->conds.data(), (uint32_t)g->conds.size(),
                               (uint32_t)g->ns_ids.ids.size(), g->more_vars.data(), (uint32_t)g->more_vars.size());
    if (rc) return gfail(g, rc, "rl_match_table_set: %s", rl_last_error(e));
    g->n_ns_installed = (uint32_t)g->ns_ids.ids.size();
    if (g->key_mode == RLI_KEYS_HASHED) {
        // the strings behind the table's ids, for the device to compare bytes against (rl_wire_table_set)
        std::vector<uint8_t> blob;
        auto strs_of = [&](const Dictionary& d) {
            std::vector<rl_wire_str> v(d.ids.size(), rl_wire_str{0, 0});
            for (const auto& kv : d.ids) {
                v[kv.second] = rl_wire_str{(uint32_t)blob.size(), (uint32_t)kv.first.size()};
                blob.insert(blob.end(), kv.first.begin(), kv.first.end());
            }
            return v;
        };
        const std::vector<rl_wire_str> ns = strs_of(g->ns_ids), vals = strs_of(g->val_ids);
        // descriptor keys: the key's own bytes, the descriptor's index in bits 24..31 of `len` (rl_engine.h: rl_wire_table_set)
        std::vector<rl_wire_str> keys(g->key_ids.ids.size(), rl_wire_str{0, 0});
        for (const auto& kv : g->key_ids.ids) {
            uint32_t desc;
            size_t off;
            desc_key_split(kv.first, &desc, &off);
            keys[kv.second] = rl_wire_str{(uint32_t)blob.size(), (uint32_t)(kv.first.size() - off) | (desc << 24)};
            blob.insert(blob.end(), kv.first.begin() + (std::ptrdiff_t)off, kv.first.end());
        }
        std::vector<uint64_t> pre(2 * g->prefix.size() + 2, 0);
        for (size_t i = 0; i < g->prefix.size(); ++i) {
            pre[2 * i] = g->prefix[i].h1;
            pre[2 * i + 1] = g->prefix[i].h2;
        }
        rc = rl_wire_table_set(e, blob.data(), (uint32_t)blob.size(), ns.data(), (uint32_t)ns.size(), keys.data(), (uint32_t)keys.size(),
                               vals.data(), (uint32_t)vals.size(), pre.data(), (uint32_t)g->prefix.size(), &g->hash_key.k0);
        if (rc) return gfail(g, rc == RL_ERR_INVALID ? RLI_HOST_ONLY : rc, "rl_wire_table_set: %s", rl_last_error(e));
    }
    for (uint32_t id = 0; id < g->limits.size(); ++id)
        if (g->limits[id].vars.empty()) {  // add_counter, in_memory.rs:38-44: limits without variables only
            uint64_t key = rl_match_key(id, 0, 0, 0);
            uint32_t chk = 0;
            if (g->key_mode == RLI_KEYS_HASHED) rl_counter_key(g->prefix[id], nullptr, 0, g->hash_key, &key, &chk);
            rc = rl_add_counter(e, id | RL_SIMPLE, key);
            if (rc) return gfail(g, rc, "rl_add_counter: %s", rl_last_error(e));
        }
    return RL_OK;
} RL_ABI_CATCH

int32_t rli_set_key_mode(rli_ingest* g, int32_t mode) try {
    if (!g || (mode != RLI_KEYS_EXACT && mode != RLI_KEYS_HASHED)) return RL_ERR_INVALID;
    if (!g->req_ns.empty()) return gfail(g, RL_ERR_INVALID, "the key mode is chosen before the first request is added");
    if (g->key_mode != mode) g->compiled = false;  // (limits with more than two variables compile differently: rli_compile)
    g->key_mode = mode;
    return RL_OK;
} RL_ABI_CATCH

int32_t rli_set_hash_key(rli_ingest* g, uint64_t k0, uint64_t k1) try {
    if (!g) return RL_ERR_INVALID;
    // (the prefix hashes of the compiled limits and everything installed from them carry the old key)
    if (g->compiled) return gfail(g, RL_ERR_INVALID, "the hash key is set before the limits are compiled / installed");
    g->hash_key = rl_hkey{k0, k1};
    return RL_OK;
} RL_ABI_CATCH

int32_t rli_hash_key(const rli_ingest* g, uint64_t out[2]) try {
    if (!g || !out) return RL_ERR_INVALID;
    out[0] = g->hash_key.k0;
    out[1] = g->hash_key.k1;
    return RL_OK;
} RL_ABI_CATCH

int32_t rli_counter_key(rli_ingest* g, uint32_t limit_id, const char* const* values, const uint32_t* value_lens,
                        uint32_t n_values, uint64_t* key, uint32_t* check) try {
    if (!g || !key || limit_id >= g->limits.size() || (n_values && (!values || !value_lens))) return RL_ERR_INVALID;
    if (!g->compiled) {
        const int32_t rc = rli_compile(g);
        if (rc) return rc;
    }
    if (n_values != g->limits[limit_id].vars.size() || n_values > 8)
        return gfail(g, RL_ERR_INVALID, "limit %u has %zu variables", limit_id, g->limits[limit_id].vars.size());
    rl_h128 v[8];
    for (uint32_t q = 0; q < n_values; ++q) v[q] = rl_kh_bytes(reinterpret_cast<const uint8_t*>(values[q]), value_lens[q], g->hash_key);
    uint32_t chk = 0;
    rl_counter_key(g->prefix[limit_id], v, n_values, g->hash_key, key, &chk);
    if (check) *check = chk;
    return RL_OK;
} RL_ABI_CATCH

void rli_batch_clear(rli_ingest* g) {
    if (!g) return;
    g->req_ns.clear();
    g->req_delta.clear();
    g->ent_off.assign(1, 0);
    g->ent_key.clear();
    g->ent_val.clear();
}

// One request, strings given with their lengths (protobuf strings may hold NULs) -> its dictionary-encoded form.
// Thread-safe: lookups under the shared lock; a request that carries a value never seen before takes the exclusive
// lock to intern it.  0, or RLI_HOST_ONLY when the value dictionary is at its cap (nothing is interned then).
static int32_t encode_request(const rli_ingest* g_c, const std::string& ns, const std::vector<std::pair<std::string, std::string>>& entries,
                              uint32_t delta, EncReq* out, std::shared_lock<std::shared_mutex>* held = nullptr) {
    rli_ingest* g = const_cast<rli_ingest*>(g_c);
    // (scratch that keeps its capacity from one request to the next: no allocation per request)
    static thread_local std::vector<std::pair<uint32_t, const std::string*>> kv;  // (key id, value)
    static thread_local std::vector<uint32_t> vids;
    kv.clear();
    vids.clear();
    size_t n_new = 0;
    {
        // (a thread that encodes many requests holds the shared lock across them — taking it per request makes
        // every thread write the lock's cache line 30 000 times per batch, and nothing scales)
        std::shared_lock<std::shared_mutex> own;
        if (!held) own = std::shared_lock<std::shared_mutex>(g->dict_mu);
        const int64_t nid = g->ns_ids.find(ns);
        // a namespace no limit names has no counters (lib.rs:434-440): the empty namespace 0; so does one interned
        // after the table was installed (its id is beyond the table's namespaces)
        out->ns = (nid < 0 || (g->n_ns_installed && (uint32_t)nid >= g->n_ns_installed)) ? 0u : (uint32_t)nid;
        out->delta = delta;
        // a repeated key keeps its LAST value (the reference collects the entries into a HashMap: server.rs:112-127)
        for (const auto& e : entries) {
            const int64_t kid = g->key_ids.find(e.first);
            if (kid < 0) continue;  // a key no limit reads: it cannot influence any condition or variable
            bool replaced = false;
            for (auto& q : kv)
                if (q.first == (uint32_t)kid) {
                    q.second = &e.second;
                    replaced = true;
                }
            if (!replaced) kv.emplace_back((uint32_t)kid, &e.second);
        }
        for (const auto& q : kv) {
            const int64_t v = g->val_ids.find(*q.second);
            if (v < 0) ++n_new;
            vids.push_back(v < 0 ? 0xFFFFFFFFu : (uint32_t)v);
        }
    }
    if (n_new) {
        if (held) held->unlock();
        struct Relock {
            std::shared_lock<std::shared_mutex>* h;
            ~Relock() {
                if (h) h->lock();
            }
        } relock{held};
        std::unique_lock<std::shared_mutex> wr(g->dict_mu);
        n_new = 0;  // (another thread may have interned some of them meanwhile)
        for (size_t q = 0; q < kv.size(); ++q)
            if (vids[q] == 0xFFFFFFFFu) {
                const int64_t v = g->val_ids.find(*kv[q].second);
                if (v < 0) ++n_new;
                else vids[q] = (uint32_t)v;
            }
        // The dictionary is at its cap (attacker-controlled values must not grow host state without bound; the
        // reference bounds its counters with moka's cache_size, in_memory.rs:205-212).  Only THIS request — it
        // carries a value never seen before — goes to the host path; requests made of known values go on.
        if (g->val_ids.ids.size() + n_new > (size_t)g->value_cap) return RLI_HOST_ONLY;
        for (size_t q = 0; q < kv.size(); ++q)
            if (vids[q] == 0xFFFFFFFFu) vids[q] = g->val_ids.intern(*kv[q].second);
    }
    out->kv.clear();
    for (size_t q = 0; q < kv.size(); ++q) out->kv.emplace_back(kv[q].first, vids[q]);
    // Limits of this namespace with more than two variables (exact keys): the request carries all of a limit's variables ->
    // one more entry, (the limit's synthetic key, id of the tuple of the values' ids).  A limit whose variables are not all
    // there adds nothing, and the device finds its synthetic key absent: no counter (limit/cel.rs:176-191).
    if (out->ns < g->composites.size() && !g->composites[out->ns].empty()) {
        static thread_local std::vector<uint32_t> tup;
        for (const rli_ingest::Composite& c : g->composites[out->ns]) {
            tup.assign(1, c.limit);
            for (const uint32_t vk : c.var_keys) {
                size_t q = 0;
                while (q < kv.size() && kv[q].first != vk) ++q;
                if (q == kv.size()) break;
                tup.push_back(vids[q]);
            }
            if (tup.size() != c.var_keys.size() + 1) continue;
            uint32_t tid = 0;
            bool have = false;
            {
                std::shared_lock<std::shared_mutex> own;
                if (!held) own = std::shared_lock<std::shared_mutex>(g->dict_mu);
                auto it = g->tuple_ids.find(tup);
                if (it != g->tuple_ids.end()) {
                    tid = it->second;
                    have = true;
                }
            }
            if (!have) {
                if (held) held->unlock();
                struct Relock {
                    std::shared_lock<std::shared_mutex>* h;
                    ~Relock() {
                        if (h) h->lock();
                    }
                } relock{held};
                std::unique_lock<std::shared_mutex> wr(g->dict_mu);
                auto it = g->tuple_ids.find(tup);
                if (it != g->tuple_ids.end()) tid = it->second;
                else {
                    // (tuples are made of caller-controlled values: the same cap as the value dictionary, the same answer)
                    if (g->tuple_ids.size() >= (size_t)g->value_cap) return RLI_HOST_ONLY;
                    tid = (uint32_t)g->tuple_ids.size();
                    g->tuple_ids.emplace(tup, tid);
                }
            }
            out->kv.emplace_back(c.syn_key, tid);
        }
    }
    return 0;
}

// Nothing is added unless the whole request is valid: the five batch arrays only ever grow together.
static int32_t batch_append(rli_ingest* g, const EncReq& r) {
    g->req_ns.push_back(r.ns);
    g->req_delta.push_back(r.delta);
    for (const auto& q : r.kv) {
        g->ent_key.push_back(q.first);
        g->ent_val.push_back(q.second);
    }
    g->ent_off.push_back((uint32_t)g->ent_key.size());
    return (int32_t)g->req_ns.size() - 1;
}

static int32_t batch_add_sv(rli_ingest* g, const std::string& ns, const std::vector<std::pair<std::string, std::string>>& entries,
                            uint32_t delta) {
    if (g->key_mode == RLI_KEYS_HASHED)
        return gfail(g, RL_ERR_INVALID, "rli_batch_add* builds dictionary-encoded requests: RLI_KEYS_EXACT only (RLI_KEYS_HASHED: rli_serve_batch)");
    EncReq r;
    const int32_t rc = encode_request(g, ns, entries, delta, &r);
    if (rc == RLI_HOST_ONLY)
        return gfail(g, RLI_HOST_ONLY, "value dictionary at its cap of %u: request with a new value stays on the host "
                                       "(sweep, then rli_create a fresh ingest to restart the dictionary)", g->value_cap);
    if (rc) return rc;
    return batch_append(g, r);
}

int32_t rli_batch_add(rli_ingest* g, const char* ns, const char* const* keys, const char* const* values,
                      uint32_t n_entries, uint32_t delta) try {
    if (!g || !ns || (n_entries && (!keys || !values))) return RL_ERR_INVALID;
    std::vector<std::pair<std::string, std::string>> entries;
    for (uint32_t q = 0; q < n_entries; ++q) {
        if (!keys[q] || !values[q]) return gfail(g, RL_ERR_INVALID, "null descriptor entry");
        entries.emplace_back(keys[q], values[q]);
    }
    return batch_add_sv(g, ns, entries, delta);
} RL_ABI_CATCH

int32_t rli_batch_add_descriptors(rli_ingest* g, const char* ns, uint32_t n_descriptors, const uint32_t* desc_off,
                                  const char* const* keys, const char* const* values, uint32_t delta) try {
    if (!g || !ns || (n_descriptors && !desc_off)) return RL_ERR_INVALID;
    const uint32_t n_entries = n_descriptors ? desc_off[n_descriptors] : 0;
    if (n_entries && (!keys || !values)) return RL_ERR_INVALID;
    std::vector<std::pair<std::string, std::string>> entries;
    for (uint32_t d = 0; d < n_descriptors; ++d) {
        if (desc_off[d] > desc_off[d + 1]) return gfail(g, RL_ERR_INVALID, "descriptor offsets must not decrease");
        if (d > MAX_DESC_INDEX) break;  // (no limit reads them)
        const size_t first = entries.size();
        for (uint32_t q = desc_off[d]; q < desc_off[d + 1]; ++q) {
            if (!keys[q] || !values[q]) return gfail(g, RL_ERR_INVALID, "null descriptor entry");
            std::string k = desc_key_name(d, keys[q], strlen(keys[q]));
            bool replaced = false;  // one map per descriptor: a repeated key keeps its LAST value
            for (size_t p = first; p < entries.size(); ++p)
                if (entries[p].first == k) {
                    entries[p].second = values[q];
                    replaced = true;
                }
            if (!replaced) entries.emplace_back(std::move(k), values[q]);
        }
    }
    return batch_add_sv(g, ns, entries, delta);
} RL_ABI_CATCH

int32_t rli_set_binding(rli_ingest* g, int32_t binding) try {
    if (!g || (binding != RLI_BIND_DESCRIPTORS && binding != RLI_BIND_ROOT)) return RL_ERR_INVALID;
    if (!g->limits.empty()) return gfail(g, RL_ERR_INVALID, "the binding is chosen before the first limit is added");
    g->binding = binding;
    return RL_OK;
} RL_ABI_CATCH

int32_t rli_set_value_cap(rli_ingest* g, uint32_t cap) try {
    if (!g || cap == 0 || cap > (1u << 26)) return RL_ERR_INVALID;  // value ids travel in 26 bits (rl_match_key)
    g->value_cap = cap;
    return RL_OK;
} RL_ABI_CATCH

// A serialized RateLimitRequest -> (domain, the entries of the descriptors some limit reads — bit i of desc_mask =
// descriptors[i] — under their (descriptor, key) names, delta).  Pure: no ingest state.
// 0, RLI_UNKNOWN_DOMAIN, or RL_ERR_INVALID with *what = the part that is malformed.
static int32_t decode_rls(const uint8_t* msg, uint32_t len, uint64_t desc_mask, std::string* domain,
                          std::vector<std::pair<std::string, std::string>>* entries, uint32_t* delta, const char** what) {
    Wire w{msg, msg + len};
    uint64_t hits_addend = 0;
    uint32_t n_descriptors = 0;
    *what = "";
    while (!w.done()) {
        uint64_t tag;
        if (!w.varint(&tag)) return *what = "tag", RL_ERR_INVALID;
        const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
        Wire sub;
        if (field == 1) {
            if (wt != 2 || !w.bytes(&sub) || !utf8_ok(sub.p, sub.end)) return *what = "domain", RL_ERR_INVALID;
            domain->assign(reinterpret_cast<const char*>(sub.p), (size_t)(sub.end - sub.p));
        } else if (field == 2) {
            if (wt != 2 || !w.bytes(&sub)) return *what = "descriptor", RL_ERR_INVALID;
            const uint32_t d = n_descriptors++;
            if (d <= MAX_DESC_INDEX && ((desc_mask >> d) & 1ull) && !parse_descriptor(sub, d, entries))
                return *what = "RateLimitDescriptor", RL_ERR_INVALID;
        } else if (field == 3) {
            if (wt != 0 || !w.varint(&hits_addend)) return *what = "hits_addend", RL_ERR_INVALID;
        } else if (!w.skip(wt)) {
            return *what = "unknown field", RL_ERR_INVALID;
        }
    }
    if (domain->empty()) return RLI_UNKNOWN_DOMAIN;
    // hits_addend is a uint32 on the wire (a longer varint is truncated by the protobuf runtime), and 0 means 1
    // (server.rs:131-137)
    *delta = (uint32_t)hits_addend;
    if (*delta == 0) *delta = 1;
    return 0;
}

int32_t rli_batch_add_rls(rli_ingest* g, const uint8_t* msg, uint32_t len) try {
    if (!g || (len && !msg)) return RL_ERR_INVALID;
    std::string domain;
    std::vector<std::pair<std::string, std::string>> entries;
    uint32_t delta = 1;
    const char* what = "";
    const int32_t rc = decode_rls(msg, len, g->desc_mask, &domain, &entries, &delta, &what);
    if (rc == RLI_UNKNOWN_DOMAIN) return rc;
    if (rc) return gfail(g, rc, "malformed RateLimitRequest (%s)", what);
    return batch_add_sv(g, domain, entries, delta);
} RL_ABI_CATCH

uint32_t rli_batch_n_requests(const rli_ingest* g) { return g ? (uint32_t)g->req_ns.size() : 0; }
uint32_t rli_batch_n_entries(const rli_ingest* g) { return g ? (uint32_t)g->ent_key.size() : 0; }
const uint32_t* rli_batch_req_ns(const rli_ingest* g) { return g ? g->req_ns.data() : nullptr; }
const uint32_t* rli_batch_req_delta(const rli_ingest* g) { return g ? g->req_delta.data() : nullptr; }
const uint32_t* rli_batch_ent_off(const rli_ingest* g) { return g ? g->ent_off.data() : nullptr; }
const uint32_t* rli_batch_ent_key(const rli_ingest* g) { return g ? g->ent_key.data() : nullptr; }
const uint32_t* rli_batch_ent_val(const rli_ingest* g) { return g ? g->ent_val.data() : nullptr; }

int32_t rli_check(rli_ingest* g, rl_engine* e, uint64_t now_us, uint8_t* verdict, int32_t* limited_limit) try {
    if (!g || !e || !verdict) return RL_ERR_INVALID;
    const uint32_t n = (uint32_t)g->req_ns.size();
    if (!n) return RL_OK;
    uint32_t n_hits = 0;
    const int32_t rc = rl_match_and_check_batch(e, g->req_ns.data(), g->ent_off.data(), g->ent_key.data(), g->ent_val.data(),
                                                g->req_delta.data(), n, now_us, 0, verdict, limited_limit, nullptr,
                                                nullptr, 0, &n_hits, nullptr, nullptr);
    if (rc) return gfail(g, rc, "rl_match_and_check_batch: %s", rl_last_error(e));
    return RL_OK;
} RL_ABI_CATCH

uint32_t rli_rls_response(int32_t verdict, uint8_t out[2]) {
    // RateLimitResponse { Code overall_code = 1 }: UNKNOWN = 0 is the default and is not put on the wire
    if (verdict == RLI_UNKNOWN_DOMAIN || verdict < 0 || !out) return 0;
    out[0] = (1u << 3) | 0u;
    out[1] = verdict ? 2u : 1u;  // OVER_LIMIT : OK
    return 2;
}

int32_t rli_set_limit_name(rli_ingest* g, uint32_t limit_id, const char* name) try {
    if (!g || limit_id >= g->limits.size()) return RL_ERR_INVALID;
    g->limits[limit_id].has_name = name != nullptr;
    g->limits[limit_id].name = name ? name : "";
    ++g->resp_version;
    return RL_OK;
} RL_ABI_CATCH

// ---- serving: RateLimitRequest bytes in, RateLimitResponse bytes out -----------------------------------------
namespace {

void put_varint(std::string& o, uint64_t v) {
    while (v >= 0x80) {
        o.push_back((char)(v | 0x80));
        v >>= 7;
    }
    o.push_back((char)v);
}
void put_string_field(std::string& o, uint32_t field, const std::string& v) {
    put_varint(o, (field << 3) | 2);
    put_varint(o, v.size());
    o += v;
}

}  // namespace

// `, {max};w={secs}[;name="{name}"]` — what a limit contributes to X-RateLimit-Limit (lib.rs:246-262), the same for every
// request: formatted once per limit.
static std::string limit_fragment(const LimitSpec* L) {
    std::string f = ", " + std::to_string(L ? L->max_value : 0) + ";w=" + std::to_string(L ? L->seconds : 0);
    if (L && L->has_name) {
        std::string nm = L->name;
        std::replace(nm.begin(), nm.end(), '"', '\'');
        f += ";name=\"" + nm + "\"";
    }
    return f;
}

// The fragments on the device (rl_resp_table_set), where the responses with headers are built (rl_resp.hpp).
static int32_t send_fragments(rli_ingest* g, rl_engine* e) {
    std::lock_guard<std::mutex> one(g->frag_mu);
    // (the engine itself says whether it HAS a fragment table: an engine destroyed and another one created at the same address
    // must not inherit "already sent" — ADVICE r05)
    if (g->resp_engine == e && g->resp_sent == g->resp_version && rl_resp_table_ready(e)) return RL_OK;
    std::vector<uint8_t> blob;
    std::vector<rl_wire_str> frag(g->limits.size());
    for (size_t l = 0; l < g->limits.size(); ++l) {
        const std::string f = limit_fragment(&g->limits[l]);
        frag[l] = rl_wire_str{(uint32_t)blob.size(), (uint32_t)f.size()};
        blob.insert(blob.end(), f.begin(), f.end());
    }
    const int32_t rc = rl_resp_table_set(e, blob.data(), (uint32_t)blob.size(), frag.data(), (uint32_t)frag.size());
    if (rc) return gfail(g, rc, "rl_resp_table_set: %s", rl_last_error(e));
    g->resp_engine = e;
    g->resp_sent = g->resp_version;
    return RL_OK;
}

// op: RL_OP_CHECK_AND_UPDATE = ShouldRateLimit (envoy_rls/server.rs:91-208); RL_OP_CHECK / RL_OP_UPDATE = the Kuadrant service's
// CheckRateLimit / Report (envoy_rls/kuadrant_service.rs:27-184), whose responses never carry rate-limit headers.
// (rli_frontend: two workers serve windows side by side, but the windows' DEVICE calls must enter the engine in arrival order —
// a Report behind its CheckRateLimit: `before` waits for the window's turn, `after` passes it on; either may be empty)
// The serving SETS are handed out in window order as well (`before_lease` / `after_lease(shared)`): a later window that took
// both sets and then waited for its device turn would starve the earlier window still waiting for a set.
struct ServeHooks {
    std::function<void()> before, after, before_lease;
    std::function<void(bool)> after_lease;
};

static int32_t serve_batch_op(rli_ingest* g, rl_engine* e, int32_t op, const uint8_t* const* msgs, const uint32_t* lens, uint32_t n,
                              uint64_t now_us, int32_t with_headers, uint8_t* out, uint32_t out_stride, uint32_t* out_len,
                              int32_t* status, const ServeHooks* hooks = nullptr) {
    if (!g || !e || (n && (!msgs || !lens || !out || !out_len || !status)) || out_stride < 2) return RL_ERR_INVALID;
    if (op != RL_OP_CHECK_AND_UPDATE && op != RL_OP_CHECK && op != RL_OP_UPDATE) return RL_ERR_INVALID;
    if (op != RL_OP_CHECK_AND_UPDATE) with_headers = 0;
    if (n == 0) return RL_OK;
    const uint32_t threads = serve_threads(n);
    const bool trace = RL_EXP_ENV("RLI_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (trace) {  // (since the call began; on the process's clock, so that calls in flight together can be laid side by side)
            const auto t = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[rli] %-10s at %8.1f us  (clock %12.1f us, thread %04x)\n", what,
                         std::chrono::duration<double, std::micro>(t - t_begin).count(),
                         std::chrono::duration<double, std::micro>(t.time_since_epoch()).count() - 1e8 * std::floor(std::chrono::duration<double, std::micro>(t.time_since_epoch()).count() / 1e8),
                         (unsigned)(std::hash<std::thread::id>{}(std::this_thread::get_id()) & 0xFFFFu));
        }
    };
    // With headers, the responses of a LARGE batch are built on the DEVICE (rl_*_serve_batch): what comes back is the bytes,
    // not the counters — 262 144 messages 5.9 -> 3.3 ms, 32 768 1.0 -> 0.8.  A small batch keeps the host assembly from the
    // counters' arrays: the device form adds five launches, a round trip for the total and an event per copy, 0.05 ms of a
    // 0.19 ms batch of 256.  RLI_RESP_HOST=1 / RLI_RESP_DEVICE=1 (experiment builds) force one or the other — the two are
    // compared byte for byte by tests/test_gpu_rls_e2e.py.
    const bool dev_resp = with_headers && !RL_EXP_ENV("RLI_RESP_HOST") && (RL_EXP_ENV("RLI_RESP_DEVICE") || n >= 4096u);
    // One of the serving sets for the form that overlaps (hashed keys, responses built on the device); all sets — i.e.
    // alone — for every other form (it uses this ingest's batch arrays and set 0).  Released, after the last byte has been
    // waited for, on every way out of this function.
    if (hooks && hooks->before_lease) hooks->before_lease();
    SetLease lease(g, e, !(dev_resp && g->key_mode == RLI_KEYS_HASHED));
    if (hooks && hooks->after_lease) hooks->after_lease(!lease.both);
    const uint32_t set = (uint32_t)lease.set;
    lap("leased");
    if (lease.both) rli_batch_clear(g);
    std::vector<int32_t> req_of(n, -1);
    // The results of the device call live in the ENGINE's pinned staging (rl_host_staging slot 1): fresh pageable arrays —
    // 60 MB of them for 262 144 requests with headers — cost the call 20 ms of page faults and staged copies.
    uint8_t* verdict = nullptr;
    int32_t *limited = nullptr, *dev_status = nullptr;
    uint32_t* req_off = nullptr;
    rl_hit* hits = nullptr;
    uint64_t *rem = nullptr, *exp = nullptr;
    auto result_arrays = [&](uint32_t n_r, size_t cap) -> int32_t {
        auto up = [](size_t b) { return (b + 63) & ~(size_t)63; };
        const size_t total = up(n_r + 1) + 2 * up(((size_t)n_r + 1) * 4) + up(((size_t)n_r + 2) * 4) + up(cap * sizeof(rl_hit)) + 2 * up(cap * 8) + 64;
        void* base = nullptr;
        const int32_t rc = rl_host_staging(e, 4u * set + 1u, total, &base);
        if (rc) return gfail(g, rc, "rl_host_staging: %s", rl_last_error(e));
        uint8_t* p = static_cast<uint8_t*>(base);
        verdict = p;
        p += up(n_r + 1);
        limited = reinterpret_cast<int32_t*>(p);
        p += up(((size_t)n_r + 1) * 4);
        dev_status = reinterpret_cast<int32_t*>(p);
        p += up(((size_t)n_r + 1) * 4);
        req_off = reinterpret_cast<uint32_t*>(p);
        p += up(((size_t)n_r + 2) * 4);
        hits = reinterpret_cast<rl_hit*>(p);
        p += up(cap * sizeof(rl_hit));
        rem = reinterpret_cast<uint64_t*>(p);
        p += up(cap * 8);
        exp = reinterpret_cast<uint64_t*>(p);
        return RL_OK;
    };
    const uint32_t* d_off = nullptr;  // device-built responses: request r's bytes are d_bytes[d_off[r] .. d_off[r + 1])
    const uint8_t* d_bytes = nullptr;
    if (dev_resp)
        if (const int32_t frc = send_fragments(g, e)) return frc;
    uint32_t n_req = 0;
    // every request derives at most one counter per limit of its namespace
    size_t per_ns = 1;
    {
        std::map<std::string, size_t> cnt;
        for (const auto& L : g->limits) per_ns = std::max(per_ns, ++cnt[L.ns]);
    }
    if (g->key_mode == RLI_KEYS_HASHED) {
        // ---- no decoding, no dictionaries: the messages are concatenated (into a buffer pinned in place once) and the
        //      device does the rest — protobuf walk, byte comparison with the table's strings, hashed counter keys
        //      (rl_wire.hpp).  Message i IS request i; a message the device finds malformed / without a domain derives no
        //      counter and carries its status.
        std::vector<uint8_t> skip(n, 0);  // messages taken out after a key collision: answered RLI_HOST_ONLY
        n_req = n;
        const size_t cap = with_headers && !dev_resp ? (size_t)n * per_ns : 0;
        if (const int32_t brc = result_arrays(n, cap)) return brc;
        uint64_t sum = 0;
        for (uint32_t i = 0; i < n; ++i) {
            if (lens[i] && !msgs[i]) return gfail(g, RL_ERR_INVALID, "message %u: null pointer with length %u", i, lens[i]);
            sum += lens[i];
        }
        if (sum > 0xFFFFFFFFull - 64) return gfail(g, RL_ERR_BATCH_TOO_LARGE, "the messages take %llu bytes", (unsigned long long)sum);
        // offsets and bytes live in the ENGINE's pinned staging (rl_host_staging slot 0: the copies to the device are plain DMA)
        const size_t off_bytes = (((size_t)n + 1) * sizeof(uint32_t) + 63) & ~(size_t)63;
        void* stage = nullptr;
        int32_t src = rl_host_staging(e, 4u * set + 0u, off_bytes + sum + 64, &stage);
        if (src) return gfail(g, src, "rl_host_staging: %s", rl_last_error(e));
        uint32_t* w_off = static_cast<uint32_t*>(stage);
        uint8_t* w_bytes = static_cast<uint8_t*>(stage) + off_bytes;
        for (int attempt = 0;; ++attempt) {
            uint64_t total = 0;
            for (uint32_t i = 0; i < n; ++i) {
                w_off[i] = (uint32_t)total;
                total += skip[i] ? 0u : lens[i];
            }
            w_off[n] = (uint32_t)total;
            parallel_chunks(n, threads, [&](uint32_t lo, uint32_t hi) {
                // (every message is an allocation of its own, cold in this thread's cache: asked for a few ahead, the
                // copies overlap their misses instead of taking them one by one)
                for (uint32_t i = lo; i < hi; ++i) {
                    if (i + 8 < hi) __builtin_prefetch(msgs[i + 8], 0, 0);
                    if (!skip[i] && lens[i]) memcpy(w_bytes + w_off[i], msgs[i], lens[i]);
                }
            }, set);
            if (attempt == 0) lap("packed");
            if (attempt == 0 && hooks && hooks->before) hooks->before();
            uint32_t n_hits = 0;
            int64_t collided = -1;
            const int32_t rc =
                dev_resp ? rl_wire_serve_batch_set(e, set, w_bytes, w_off, n, now_us, RL_SERVE_HEADERS | RL_SERVE_ASYNC, verdict,
                                                   dev_status, &d_off, &d_bytes, &collided)
                : op == RL_OP_CHECK_AND_UPDATE
                    ? rl_wire_match_and_check_batch(e, w_bytes, w_off, n, now_us, with_headers ? 1 : 0, verdict, limited, dev_status,
                                                    with_headers ? req_off : nullptr, with_headers ? hits : nullptr, (uint32_t)cap,
                                                    &n_hits, with_headers ? rem : nullptr, with_headers ? exp : nullptr, &collided)
                    : rl_wire_match_batch_op(e, op, w_bytes, w_off, n, now_us, verdict, limited, dev_status, &collided);
            if (rc == RL_ERR_KEY_COLLISION && attempt < 8) {
                // Counters that share a 64-bit key with another counter are never merged: their messages are taken out — ALL
                // of them at once, the device names every one in dev_status (-103) — and answered RLI_HOST_ONLY (the caller's
                // exact path); the rest of the batch is applied.  One re-run is the rule; a second only when a message that
                // was taken out had been hiding another collision behind it.
                uint32_t taken = 0;
                for (uint32_t i = 0; i < n; ++i)
                    if (dev_status[i] == -103 && !skip[i]) {
                        skip[i] = 1;
                        ++taken;
                    }
                if (!taken && collided >= 0 && (uint64_t)collided < n && !skip[collided]) {
                    skip[collided] = 1;
                    ++taken;
                }
                if (taken) continue;
            }
            if (rc) return gfail(g, rc, "rl_wire_match_and_check_batch: %s", rl_last_error(e));
            lease.bytes_pending = dev_resp;
            break;
        }
        for (uint32_t i = 0; i < n; ++i) {
            // (-103 on a call that succeeded: RL_OP_CHECK met a cell that is another counter's — nothing was written, the
            // message goes to the caller's exact path like the ones a counting call takes out)
            status[i] = skip[i] || dev_status[i] == -103 ? (int32_t)RLI_HOST_ONLY : dev_status[i];
            out_len[i] = 0;
            req_of[i] = status[i] == 0 ? (int32_t)i : -1;
        }
    } else {
        // ---- decode + dictionary encoding: every message on its own, many at a time.  A thread keeps the encoded entries
        //      of its share of the messages in ONE flat list (no allocation per request) ------------------------------
        struct EncSlot {
            uint32_t ns, delta, kv_off, kv_cnt;
        };
        std::vector<EncSlot> enc(n);
        const uint32_t per = threads > 1 ? (n + threads - 1) / threads : n;
        const uint32_t n_chunks = per ? (n + per - 1) / per : 0;
        std::vector<std::vector<std::pair<uint32_t, uint32_t>>> chunk_kv(n_chunks ? n_chunks : 1);
        std::vector<uint32_t> chunk_req(n_chunks + 1, 0), chunk_ent(n_chunks + 1, 0);
        parallel_chunks(n, threads, [&](uint32_t lo, uint32_t hi) {
            std::string domain;
            std::vector<std::pair<std::string, std::string>> entries;
            EncReq tmp;
            auto& flat = chunk_kv[lo / per];
            flat.reserve((size_t)(hi - lo) * 4);
            uint32_t n_ok = 0;
            std::shared_lock<std::shared_mutex> rd(g->dict_mu);
            for (uint32_t i = lo; i < hi; ++i) {
                if (((i - lo) & 255u) == 255u) {  // let a thread that has a new value to intern get its turn
                    rd.unlock();
                    rd.lock();
                }
                domain.clear();
                entries.clear();
                uint32_t delta = 1;
                const char* what = "";
                int32_t rc = (lens[i] && !msgs[i]) ? (int32_t)RL_ERR_INVALID : decode_rls(msgs[i], lens[i], g->desc_mask, &domain, &entries, &delta, &what);
                if (rc == 0) rc = encode_request(g, domain, entries, delta, &tmp, &rd);
                if (rc == 0) {
                    // (CheckRateLimit checks with 1 whatever hits_addend says: kuadrant_service.rs:62-64)
                    enc[i] = EncSlot{tmp.ns, op == RL_OP_CHECK ? 1u : tmp.delta, (uint32_t)flat.size(), (uint32_t)tmp.kv.size()};
                    flat.insert(flat.end(), tmp.kv.begin(), tmp.kv.end());
                    ++n_ok;
                }
                status[i] = rc;  // 0, RLI_UNKNOWN_DOMAIN, RLI_HOST_ONLY, < 0
                out_len[i] = 0;
            }
            chunk_req[lo / per + 1] = n_ok;
            chunk_ent[lo / per + 1] = (uint32_t)flat.size();
        });
        lap("decoded");
        // ---- the batch, in message order: every thread's share lands at its offset --------------------------------
        for (uint32_t c = 0; c < n_chunks; ++c) {
            chunk_req[c + 1] += chunk_req[c];
            chunk_ent[c + 1] += chunk_ent[c];
        }
        g->req_ns.resize(chunk_req[n_chunks]);
        g->req_delta.resize(chunk_req[n_chunks]);
        g->ent_off.resize((size_t)chunk_req[n_chunks] + 1);
        g->ent_key.resize(chunk_ent[n_chunks]);
        g->ent_val.resize(chunk_ent[n_chunks]);
        g->ent_off[0] = 0;
        parallel_chunks(n, threads, [&](uint32_t lo, uint32_t hi) {
            const uint32_t c = lo / per;
            const auto& flat = chunk_kv[c];
            uint32_t r = chunk_req[c];
            const uint32_t e0 = chunk_ent[c];
            for (uint32_t i = lo; i < hi; ++i) {
                if (status[i] != 0) continue;
                const EncSlot& q = enc[i];
                g->req_ns[r] = q.ns;
                g->req_delta[r] = q.delta;
                for (uint32_t k = 0; k < q.kv_cnt; ++k) {
                    g->ent_key[e0 + q.kv_off + k] = flat[q.kv_off + k].first;
                    g->ent_val[e0 + q.kv_off + k] = flat[q.kv_off + k].second;
                }
                g->ent_off[r + 1] = e0 + q.kv_off + q.kv_cnt;
                req_of[i] = (int32_t)r++;
            }
        });
        n_req = (uint32_t)g->req_ns.size();
        lap("appended");
        const size_t cap = with_headers && !dev_resp ? (size_t)n_req * per_ns : 0;
        if (const int32_t brc = result_arrays(n_req, cap)) return brc;
        if (hooks && hooks->before) hooks->before();
        if (n_req) {
            uint32_t n_hits = 0;
            const int32_t rc =
                dev_resp ? rl_match_serve_batch(e, g->req_ns.data(), g->ent_off.data(), g->ent_key.data(), g->ent_val.data(),
                                                g->req_delta.data(), n_req, now_us, RL_SERVE_HEADERS | RL_SERVE_ASYNC, verdict,
                                                &d_off, &d_bytes)
                : op == RL_OP_CHECK_AND_UPDATE
                    ? rl_match_and_check_batch(e, g->req_ns.data(), g->ent_off.data(), g->ent_key.data(), g->ent_val.data(),
                                               g->req_delta.data(), n_req, now_us, with_headers ? 1 : 0, verdict, limited,
                                               with_headers ? req_off : nullptr, with_headers ? hits : nullptr, (uint32_t)cap, &n_hits,
                                               with_headers ? rem : nullptr, with_headers ? exp : nullptr)
                    : rl_match_batch_op(e, op, g->req_ns.data(), g->ent_off.data(), g->ent_key.data(), g->ent_val.data(),
                                        g->req_delta.data(), n_req, now_us, verdict, limited);
            if (rc) return gfail(g, rc, "rl_match_and_check_batch: %s", rl_last_error(e));
            lease.bytes_pending = dev_resp;
        }
    }
    lap("device");
    if (hooks && hooks->after) hooks->after();
    // ---- the responses: independent of one another ----------------------------------------------------------
    std::atomic<uint32_t> too_long{0};
    // what a limit contributes to X-RateLimit-Limit — `, {max};w={secs}[;name="{name}"]` — is the same for every request:
    // built once per call, not once per counter (lib.rs:246-262)
    std::vector<std::string> frag(g->limits.size() + 1);
    if (with_headers && !dev_resp)
        for (size_t l = 0; l <= g->limits.size(); ++l)  // (the last one: a limit id this ingest does not know)
            frag[l] = limit_fragment(l < g->limits.size() ? &g->limits[l] : nullptr);
    auto put_u64 = [](std::string& o, uint64_t v) {
        char buf[24];
        int k = 24;
        do {
            buf[--k] = (char)('0' + v % 10);
            v /= 10;
        } while (v);
        o.append(buf + k, (size_t)(24 - k));
    };
    std::atomic<int32_t> wait_rc{0};
    parallel_chunks(n, threads, [&](uint32_t lo, uint32_t hi) {
        if (dev_resp && d_off) {
            // The bytes may still be travelling (RL_SERVE_ASYNC: they arrive in order): this share starts once ITS last
            // response is there — the shares in front of it are being handed on meanwhile.
            uint32_t i = hi;
            while (i > lo && req_of[i - 1] < 0) --i;
            if (i > lo)
                if (const int32_t wrc = rl_serve_wait_set(e, set, d_off[(uint32_t)req_of[i - 1] + 1])) wait_rc.store(wrc);
        }
        // (strings that keep their capacity from one request to the next: no allocation per request)
        std::string o, hv, val;
        std::vector<uint32_t> order;
        static const std::string k_limit = "X-RateLimit-Limit", k_rem = "X-RateLimit-Remaining", k_reset = "X-RateLimit-Reset";
        for (uint32_t i = lo; i < hi; ++i) {
            // (the caller's slots are out_stride apart — a fresh cache line and often a fresh page per response: the line
            // of a slot a few ahead is asked for now, for writing, so that the stores do not wait for it one at a time)
            if (i + 8 < hi) {
                __builtin_prefetch(out + (size_t)(i + 8) * out_stride, 1, 0);
                __builtin_prefetch(out + (size_t)(i + 8) * out_stride + 64, 1, 0);
            }
            if (dev_resp && d_off) {
                // the bytes exist already (rl_resp.hpp): request r's response goes to the caller's slot i
                if (status[i] != 0 && status[i] != RLI_UNKNOWN_DOMAIN) continue;  // malformed / RLI_HOST_ONLY: no response
                uint32_t len = 0;
                const uint8_t* src = nullptr;
                if (status[i] == 0) {
                    const uint32_t r = (uint32_t)req_of[i];
                    len = d_off[r + 1] - d_off[r];
                    src = d_bytes + d_off[r];
                    status[i] = verdict[r] ? 1 : 0;
                }  // (no domain: Code::Unknown, the empty message)
                if (len > out_stride) {
                    too_long.store(len);
                    status[i] = RLI_RESPONSE_TOO_LARGE;
                    out_len[i] = 0;
                    continue;
                }
                if (len) memcpy(out + (size_t)i * out_stride, src, len);
                out_len[i] = len;
                continue;
            }
            o.clear();
            if (status[i] == RLI_UNKNOWN_DOMAIN) {
                // Code::Unknown = 0, the proto3 default: an empty message (server.rs:105-115)
            } else if (status[i] == 0) {
                const uint32_t r = (uint32_t)req_of[i];
                put_varint(o, (1u << 3) | 0u);
                put_varint(o, verdict[r] ? 2u : 1u);  // OVER_LIMIT : OK
                if (with_headers && req_off[r + 1] > req_off[r]) {
                    // CheckResult::response_header (lib.rs:235-275): the request's counters sorted by remaining (stable: ties
                    // keep the storage's order), the most restrictive one first; RateLimitHeaders::headers sorts the three
                    // headers by key (envoy_rls/server.rs:44-57) — Limit < Remaining < Reset.
                    order.clear();
                    for (uint32_t q = req_off[r]; q < req_off[r + 1]; ++q) {
                        size_t at = order.size();
                        order.push_back(q);
                        while (at > 0 && rem[order[at - 1]] > rem[q]) {  // (insertion: a handful of counters)
                            order[at] = order[at - 1];
                            --at;
                        }
                        order[at] = q;
                    }
                    auto lid_of = [&](uint32_t q) {
                        const uint32_t l = RL_LIMIT_ID(hits[q].limit);
                        return l < g->limits.size() ? (size_t)l : g->limits.size();
                    };
                    const uint32_t f = order[0];
                    auto header = [&](const std::string& key) {  // response_headers_to_add = 3: HeaderValue { key = 1; value = 2 }
                        hv.clear();
                        put_string_field(hv, 1, key);
                        put_string_field(hv, 2, val);
                        put_string_field(o, 3, hv);
                    };
                    val.clear();
                    put_u64(val, lid_of(f) < g->limits.size() ? g->limits[lid_of(f)].max_value : 0);
                    for (uint32_t q : order) val += frag[lid_of(q)];
                    header(k_limit);
                    val.clear();
                    put_u64(val, rem[f]);
                    header(k_rem);
                    val.clear();
                    put_u64(val, exp[f] / 1000000ull);  // Duration::as_secs
                    header(k_reset);
                }
                status[i] = verdict[r] ? 1 : 0;
            } else {
                continue;  // malformed / RLI_HOST_ONLY: no response, the status says why
            }
            if (o.size() > out_stride) {
                // The request was decided and counted like the others; only ITS response does not fit the caller's
                // stride (the reference has no bound on X-RateLimit-Limit: one entry per counter).  It alone is told.
                too_long.store((uint32_t)o.size());
                status[i] = RLI_RESPONSE_TOO_LARGE;
                out_len[i] = 0;
                continue;
            }
            memcpy(out + (size_t)i * out_stride, o.data(), o.size());
            out_len[i] = (uint32_t)o.size();
        }
    }, set);
    lap("responses");
    if (dev_resp && d_off) {
        // (every byte has been waited for before the engine is called again — also the ones behind the last answered request)
        if (const int32_t wrc = rl_serve_wait_set(e, set, d_off[n_req])) wait_rc.store(wrc);
        if (wait_rc.load()) return gfail(g, wait_rc.load(), "rl_serve_wait: the responses' copy failed");
    }
    if (too_long.load())  // (not an error of the call: the message names the size a retry needs)
        (void)gfail(g, RL_OK, "a response of %u bytes does not fit the stride %u: status RLI_RESPONSE_TOO_LARGE for it", too_long.load(), out_stride);
    return RL_OK;
}

int32_t rli_serve_batch(rli_ingest* g, rl_engine* e, const uint8_t* const* msgs, const uint32_t* lens, uint32_t n,
                        uint64_t now_us, int32_t with_headers, uint8_t* out, uint32_t out_stride, uint32_t* out_len,
                        int32_t* status) try {
    return serve_batch_op(g, e, RL_OP_CHECK_AND_UPDATE, msgs, lens, n, now_us, with_headers, out, out_stride, out_len, status);
} RL_ABI_CATCH

int32_t rli_serve_batch_op(rli_ingest* g, rl_engine* e, int32_t op, const uint8_t* const* msgs, const uint32_t* lens, uint32_t n,
                           uint64_t now_us, uint8_t* out, uint32_t out_stride, uint32_t* out_len, int32_t* status) try {
    return serve_batch_op(g, e, op, msgs, lens, n, now_us, 0, out, out_stride, out_len, status);
} RL_ABI_CATCH

// The micro-batcher of the wire path: concurrent ShouldRateLimit callers are aggregated into one device batch,
// closed at max_batch requests or max_delay_us after its first request arrived, stamped with ONE clock value
// (what tonic's worker tasks would do around rli_serve_batch: envoy_rls/server.rs:91-208 is one call each).
struct rli_frontend {
    rli_ingest* g;
    rl_engine* e;
    uint32_t max_batch, max_delay_us, stride;
    int32_t with_headers;
    uint64_t fixed_now_us = 0;
    struct Slot {
        const uint8_t* msg;
        uint32_t len;
        uint8_t* resp;
        uint32_t resp_cap, resp_len = 0;
        int32_t status = 0;
        int32_t op = RL_OP_CHECK_AND_UPDATE;  // which RPC the caller is in: ShouldRateLimit, or Kuadrant's CheckRateLimit / Report
        bool done = false;
    };
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<Slot*> queue;
    bool stop = false;
    uint64_t n_batches = 0, n_requests = 0, n_windows = 0;  // device calls, requests, waits for a window to fill
    // Two workers (round 6): while one still waits for and hands on the responses of its window, the other collects, packs
    // and decides the next one (rli_serve_batch's two serving sets).  Windows are numbered as they are cut; their device
    // calls enter the engine in that order (`turn`), so arrival order holds across windows as it does inside one.
    std::thread worker[2];
    uint64_t next_window = 0;
    std::mutex turn_mu;
    std::condition_variable turn_cv;
    uint64_t turn = 0;        // the window whose device calls may enter the engine
    uint64_t lease_turn = 0;  // the window that may take a serving set

    void run() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [&] { return stop || !queue.empty(); });
            if (stop && queue.empty()) return;
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(max_delay_us);
            cv_work.wait_until(lk, deadline, [&] { return stop || queue.size() >= max_batch; });
            if (queue.empty()) continue;  // (the other worker took the window while this one waited)
            // The WINDOW is everything that arrived until the deadline (at most max_batch requests).  One device call is one
            // method, so the window is served as its consecutive same-method runs, back to back, under ONE clock value and
            // without another wait in between: arrival order is kept across methods (a Report that arrived behind a
            // CheckRateLimit is applied behind it), and interleaved Kuadrant traffic — C, R, C, R from concurrent clients, the
            // normal Check-then-Report pattern — costs one delay per window, not one per request (the batcher
            // used to cut at the first method change and wait a full max_delay_us again for the leftovers).
            std::vector<Slot*> window;
            const uint64_t my_window = next_window++;
            const size_t take = std::min<size_t>(queue.size(), max_batch);
            if (take < queue.size()) {
                window.assign(queue.begin(), queue.begin() + take);
                queue.erase(queue.begin(), queue.begin() + take);
            } else {
                window.swap(queue);
            }
            lk.unlock();
            uint64_t now = fixed_now_us;
            if (!now) {
                using namespace std::chrono;
                now = (uint64_t)duration_cast<microseconds>(system_clock::now().time_since_epoch()).count();
            }
            std::vector<int32_t> w_status(window.size(), 0);
            uint64_t calls = 0;
            bool turn_taken = false, turn_passed = false, lease_waited = false, lease_passed = false;
            size_t n_runs = window.empty() ? 0 : 1;
            for (size_t q = 1; q < window.size(); ++q) n_runs += window[q]->op != window[q - 1]->op;
            ServeHooks first_run, last_run, only_run;
            auto take_turn = [&] {
                std::unique_lock<std::mutex> tl(turn_mu);
                turn_cv.wait(tl, [&] { return turn == my_window; });
                turn_taken = true;
            };
            auto pass_turn = [&] {
                if (turn_passed) return;
                if (!turn_taken) take_turn();  // (a window that never reached its device call still has to pass the turn on)
                {
                    std::lock_guard<std::mutex> tl(turn_mu);
                    ++turn;
                }
                turn_passed = true;
                turn_cv.notify_all();
            };
            auto wait_lease = [&] {
                std::unique_lock<std::mutex> tl(turn_mu);
                turn_cv.wait(tl, [&] { return lease_turn == my_window; });
                lease_waited = true;
            };
            auto pass_lease = [&] {
                if (lease_passed) return;
                if (!lease_waited) wait_lease();
                {
                    std::lock_guard<std::mutex> tl(turn_mu);
                    ++lease_turn;
                }
                lease_passed = true;
                turn_cv.notify_all();
            };
            // A window of ONE run in the form that overlaps (it took one set) lets the next window take the other set at once;
            // any other window keeps the sets' gate until it is through (its later runs take sets again).
            auto leased = [&](bool shared) {
                if (shared && n_runs == 1) pass_lease();
            };
            first_run.before_lease = wait_lease;
            first_run.after_lease = leased;
            first_run.before = take_turn;
            last_run.after = pass_turn;
            only_run = first_run;
            only_run.after = pass_turn;
            for (size_t r0 = 0; r0 < window.size();) {
                size_t r1 = r0 + 1;
                while (r1 < window.size() && window[r1]->op == window[r0]->op) ++r1;
                const uint32_t n = (uint32_t)(r1 - r0);
                const int32_t op = window[r0]->op;
                std::vector<const uint8_t*> msgs;
                std::vector<uint32_t> lens, out_len;
                std::vector<int32_t> status;
                std::vector<uint8_t> out;
                int32_t rc;
                try {  // (this thread has no entry point's barrier above it: its callers get the status, the thread lives on)
                    msgs.resize(n);
                    lens.resize(n);
                    out_len.resize(n);
                    status.resize(n);
                    out.resize((size_t)n * stride);
                    for (uint32_t i = 0; i < n; ++i) {
                        msgs[i] = window[r0 + i]->msg;
                        lens[i] = window[r0 + i]->len;
                    }
                    const bool is_first = r0 == 0, is_last = r1 == window.size();
                    rc = serve_batch_op(g, e, op, msgs.data(), lens.data(), n, now, with_headers, out.data(), stride, out_len.data(),
                                        status.data(), is_first && is_last ? &only_run : is_first ? &first_run : is_last ? &last_run : nullptr);
                } catch (const std::bad_alloc&) {
                    rc = rl_abi_caught("rli_frontend worker", "std::bad_alloc (host memory exhausted)", RL_ERR_NOMEM);
                } catch (...) {
                    rc = rl_abi_caught("rli_frontend worker", "C++ exception", RL_ERR_INTERNAL);
                }
                ++calls;
                // (responses are handed over run by run, outside the lock: the slots belong to callers that are waiting)
                for (uint32_t i = 0; i < n; ++i) {
                    Slot* s = window[r0 + i];
                    w_status[r0 + i] = rc ? rc : status[i];
                    if (!rc && out_len[i] <= s->resp_cap) {
                        memcpy(s->resp, out.data() + (size_t)i * stride, out_len[i]);
                        s->resp_len = out_len[i];
                    } else if (!rc) {
                        w_status[r0 + i] = RL_ERR_INVALID;
                    }
                }
                r0 = r1;
            }
            pass_turn();
            pass_lease();
            lk.lock();
            n_batches += calls;
            ++n_windows;
            n_requests += window.size();
            for (size_t i = 0; i < window.size(); ++i) {
                window[i]->status = w_status[i];
                window[i]->done = true;
            }
            cv_done.notify_all();
        }
    }
};

int32_t rli_frontend_create(rli_ingest* g, rl_engine* e, uint32_t max_batch, uint32_t max_delay_us, int32_t with_headers,
                            rli_frontend** out) try {
    if (!g || !e || !out) return RL_ERR_INVALID;
    rli_frontend* f = new (std::nothrow) rli_frontend();
    if (!f) return RL_ERR_NOMEM;
    f->g = g;
    f->e = e;
    f->max_batch = max_batch ? max_batch : 1;
    f->max_delay_us = max_delay_us;
    f->with_headers = with_headers;
    f->stride = 1024;
    for (auto& w : f->worker) w = std::thread([f] { f->run(); });
    *out = f;
    return RL_OK;
} RL_ABI_CATCH

void rli_frontend_destroy(rli_frontend* f) {
    if (!f) return;
    {
        std::lock_guard<std::mutex> lk(f->mu);
        f->stop = true;
    }
    f->cv_work.notify_all();
    for (auto& w : f->worker) w.join();
    delete f;
}

void rli_frontend_set_clock(rli_frontend* f, uint64_t now_us) {
    if (f) f->fixed_now_us = now_us;
}

static int32_t frontend_call(rli_frontend* f, int32_t op, const uint8_t* msg, uint32_t len, uint8_t* resp, uint32_t resp_cap,
                             uint32_t* resp_len) {
    if (!f || (len && !msg) || !resp || !resp_len) return RL_ERR_INVALID;
    rli_frontend::Slot slot;
    slot.msg = msg;
    slot.len = len;
    slot.resp = resp;
    slot.resp_cap = resp_cap;
    slot.op = op;
    std::unique_lock<std::mutex> lk(f->mu);
    f->queue.push_back(&slot);
    if (f->queue.size() == 1 || f->queue.size() >= f->max_batch) f->cv_work.notify_one();
    f->cv_done.wait(lk, [&] { return slot.done; });
    *resp_len = slot.resp_len;
    return slot.status;
}

int32_t rli_frontend_should_rate_limit(rli_frontend* f, const uint8_t* msg, uint32_t len, uint8_t* resp, uint32_t resp_cap,
                                       uint32_t* resp_len) try {
    return frontend_call(f, RL_OP_CHECK_AND_UPDATE, msg, len, resp, resp_cap, resp_len);
} RL_ABI_CATCH

int32_t rli_frontend_check_rate_limit(rli_frontend* f, const uint8_t* msg, uint32_t len, uint8_t* resp, uint32_t resp_cap,
                                      uint32_t* resp_len) try {
    return frontend_call(f, RL_OP_CHECK, msg, len, resp, resp_cap, resp_len);
} RL_ABI_CATCH

int32_t rli_frontend_report(rli_frontend* f, const uint8_t* msg, uint32_t len, uint8_t* resp, uint32_t resp_cap,
                            uint32_t* resp_len) try {
    return frontend_call(f, RL_OP_UPDATE, msg, len, resp, resp_cap, resp_len);
} RL_ABI_CATCH

void rli_frontend_stats(rli_frontend* f, uint64_t* batches, uint64_t* requests) {
    if (!f) return;
    std::lock_guard<std::mutex> lk(f->mu);
    if (batches) *batches = f->n_batches;
    if (requests) *requests = f->n_requests;
}

uint64_t rli_frontend_windows(rli_frontend* f) {
    if (!f) return 0;
    std::lock_guard<std::mutex> lk(f->mu);
    return f->n_windows;
}

int64_t rli_key_id(const rli_ingest* g, const char* s) { return g && s ? g->key_ids.find(s) : -1; }
int64_t rli_value_id(const rli_ingest* g, const char* s) { return g && s ? g->val_ids.find(s) : -1; }
int64_t rli_namespace_id(const rli_ingest* g, const char* s) { return g && s ? g->ns_ids.find(s) : -1; }

}  // extern "C"
