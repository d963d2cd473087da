// rl_sharded.cpp — include/rl_sharded.h: the routed (multi-GPU) step over the engine's C ABI.
// Host code only: the device work is the engine's (k_route_*, the local batch, k_unpermute_u8) and the
// transport's (RCCL send/recv kernels, or device-to-device copies).  Nothing here has a counterpart in the
// reference — its in-memory storage is one process (limitador/src/storage/in_memory.rs) — the semantics it
// must preserve are the sequential ones of check_and_update (in_memory.rs:72-156) on the concatenated slices.
#include "rl_sharded.h"
#include "../rl_abi_guard.h"
#include <stdexcept>

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <link.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <deque>
#include <mutex>
#include <new>
#include <vector>

// experiment / diagnostics switches exist only in -DRL_EXPERIMENT builds (see rl_engine.hip)
#ifdef RL_EXPERIMENT
#define RL_EXP_ENV(name) getenv(name)
#else
#define RL_EXP_ENV(name) (static_cast<const char*>(nullptr))
#endif

namespace {

constexpr uint32_t RQ_GATHER = 16;    // rq.d_words[RQ_GATHER + p * n + r]: rank p's r-th word of the last gather (multi-counter step)
constexpr uint32_t RQ_BLIND_MAX = 4;  // rounds of the multi-counter step enqueued without a look at "did anything change" in between
constexpr uint32_t RQ_VETO = 191;     // rq.d_words[RQ_VETO]: this rank's word of the all-or-nothing commit (rl_gen_count_async_device); the changed words follow it
constexpr uint32_t RQ_CHANGED = 192;  // rq.d_words[RQ_CHANGED + r]: k_req_and of the group's round r raises it
constexpr uint32_t RQ_WORDS = 256;
constexpr uint32_t REQ_ID_LOCAL = 1u << 27;  // request id of the multi-counter step = rank * REQ_ID_LOCAL + the request's index on its rank
constexpr int SLOTS = 4;          // slices in flight: one routed + up to three on the engine (applied / returned).  Three until
                                  // round 5: with `submit(k); collect(k - 2)` the host's ~55 us of enqueues per slice only
                                  // started once the replay of slice k - 2 had ENDED (the collect waits for it) — a bubble of
                                  // 11-14 us per slice on the engine's stream (gpurun_out/r13g); one more slice in flight lets the
                                  // host run a slice ahead of the device
constexpr uint32_t MAX_WORLD = 16; // the router's limit (rl_route.hpp)
enum Stage { ROUTED = 1, APPLIED = 2, RETURNED = 3 };

struct Slice {
    int slot = 0;
    uint32_t n = 0;
    uint64_t now = 0;
    uint8_t* out = nullptr;
    int stage = ROUTED;
    uint32_t n_recv = 0;
    bool waits = false;  // a local batch was submitted to the engine for this slice
    // The slice failed ON THIS RANK (more routed hits than the engine takes in one batch, a full table, ...).  A failure
    // is an outcome of the collective step, not an exit from it: every exchange of the slice is still issued — a rank
    // that returned early would leave its peers' grouped send / recv unmatched for ever — the hits this rank owns are
    // answered 0xFF, and the error comes out of THIS slice's collect.
    int32_t err = RL_OK;
    char errmsg[200] = {0};
    bool applied_recorded = false;  // ev_applied[slot] has been recorded behind the slice's local batch
    uint64_t id = 0;                // the slice's sequence number
    bool sweep = false;             // not a slice: this rank's share of a routed sweep (rl_sharded_sweep_submit) — one engine
                                    // command, nothing to route or to return (stage RETURNED from the start)
};

// RCCL is bound at RUN time, to the copy the process has already mapped if there is one: a host that also uses
// torch.distributed has torch's own librccl.so loaded, and a second copy (the ROCm one this library used to link) in the
// same process means two sets of proxy threads and IPC state for one set of GPUs.  One process, one RCCL.
struct RcclApi {
    void* handle = nullptr;
    char path[512] = {0};
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
};

struct RcclScan {
    char first[512];
    int count;
};
int rccl_phdr_cb(struct dl_phdr_info* info, size_t, void* data) {
    auto* sc = static_cast<RcclScan*>(data);
    const char* nm = info->dlpi_name;
    if (!nm || !*nm) return 0;
    const char* base = std::strrchr(nm, '/');
    base = base ? base + 1 : nm;
    if (std::strncmp(base, "librccl.so", 10) != 0) return 0;
    if (sc->count == 0) std::snprintf(sc->first, sizeof(sc->first), "%s", nm);
    sc->count++;
    return 0;
}

RcclApi* rccl_api(char* err, size_t err_len) {
    static RcclApi api;
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    if (api.handle) return &api;
    RcclScan sc{};
    dl_iterate_phdr(rccl_phdr_cb, &sc);
    if (sc.count > 1) {
        std::snprintf(err, err_len, "%d copies of librccl.so are mapped into this process (first: %.300s): refusing to add a communicator", sc.count, sc.first);
        return nullptr;
    }
    const char* candidates[] = {sc.count ? sc.first : nullptr, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* c : candidates) {
        if (!c) continue;
        h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (h) {
            std::snprintf(api.path, sizeof(api.path), "%s", c);
            break;
        }
    }
    if (!h) {
        std::snprintf(err, err_len, "librccl.so not found (%s)", dlerror());
        return nullptr;
    }
#define RL_BIND(field, sym)                                                                  \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, sym));                        \
    if (!api.field) {                                                                        \
        std::snprintf(err, err_len, "%s has no %s", api.path, sym);                          \
        dlclose(h);                                                                          \
        return nullptr;                                                                      \
    }
    RL_BIND(GetUniqueId, "ncclGetUniqueId")
    RL_BIND(CommInitRank, "ncclCommInitRank")
    RL_BIND(CommDestroy, "ncclCommDestroy")
    RL_BIND(GroupStart, "ncclGroupStart")
    RL_BIND(GroupEnd, "ncclGroupEnd")
    RL_BIND(Send, "ncclSend")
    RL_BIND(Recv, "ncclRecv")
#undef RL_BIND
    api.handle = h;
    return &api;
}

struct RcclTransport {
    RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    uint32_t world = 0, rank = 0;
    rl_engine* e = nullptr;  // launches the copy of the rank's own segments
};

int32_t rccl_exchange(void* ctx, const rl_xfer* xs, uint32_t n, void* stream) {
    auto* t = static_cast<RcclTransport*>(ctx);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // What a rank sends to ITSELF never meets the communicator: a device-to-device copy on the same stream (an RCCL launch
    // costs 17-20 us even when all it moves is the local segment — at world 1 that was two such launches per slice, 36 of
    // the routed step's 98 us; at any world it is 1 / world of the traffic that stays off the links).
    rl_copy_seg own[4];
    uint32_t n_own = 0;
    for (uint32_t k = 0; k < n; ++k)
        if (xs[k].send_cnt[t->rank] && xs[k].send_cnt[t->rank] == xs[k].recv_cnt[t->rank]) {
            const void* src = static_cast<const char*>(xs[k].send) + xs[k].send_off[t->rank];
            void* dst = static_cast<char*>(xs[k].recv) + xs[k].recv_off[t->rank];
            if (src == dst) continue;
            if (n_own == 4) {  // (an exchange has at most 1 + SLOTS segments: flush)
                if (rl_copy_segments_stream(t->e, st, own, n_own) != RL_OK) return RL_ERR_DEVICE;
                n_own = 0;
            }
            own[n_own++] = rl_copy_seg{dst, src, xs[k].send_cnt[t->rank]};
        }
    if (n_own && rl_copy_segments_stream(t->e, st, own, n_own) != RL_OK) return RL_ERR_DEVICE;
    if (t->world == 1) return RL_OK;
    // one group = one launch: every segment's sends and receives to all peers travel concurrently over the
    // point-to-point xGMI links (RCCL's all-to-all-v as grouped ncclSend / ncclRecv)
    if (t->api->GroupStart() != ncclSuccess) return RL_ERR_DEVICE;
    bool ok = true;
    for (uint32_t k = 0; k < n && ok; ++k)
        for (uint32_t p = 0; p < t->world && ok; ++p) {
            if (p == t->rank && xs[k].send_cnt[p] == xs[k].recv_cnt[p]) continue;  // (copied above)
            if (xs[k].send_cnt[p])
                ok = t->api->Send(static_cast<const char*>(xs[k].send) + xs[k].send_off[p], xs[k].send_cnt[p], ncclUint8,
                                  (int)p, t->comm, st) == ncclSuccess;
            if (ok && xs[k].recv_cnt[p])
                ok = t->api->Recv(static_cast<char*>(xs[k].recv) + xs[k].recv_off[p], xs[k].recv_cnt[p], ncclUint8, (int)p,
                                  t->comm, st) == ncclSuccess;
        }
    if (t->api->GroupEnd() != ncclSuccess) ok = false;
    return ok ? RL_OK : RL_ERR_DEVICE;
}

}  // namespace

struct rl_sharded {
    rl_engine* e = nullptr;
    uint32_t world = 0, rank = 0;
    rl_transport t{};
    RcclTransport* rccl = nullptr;  // owned when created from a unique id
    int device = 0;
    hipStream_t cs = nullptr;  // routing + exchanges
    hipStream_t as = nullptr;  // the engine's batches, all on this one stream (default).  RL_SHARDED_ENGINE_STREAMS=own
                               // leaves the engine its own two (partition beside decisions), ordered by events: three
                               // streams then share the CUs and a routed step is slower (119 vs 105 us, world 1)
    uint32_t max_slice = 0, max_recv = 0;
    rl_hit* sorted[SLOTS] = {};
    uint32_t* perm[SLOTS] = {};
    rl_hit* recv_hits[SLOTS] = {};
    uint8_t* recv_verdict[SLOTS] = {};
    uint8_t* sorted_verdict[SLOTS] = {};
    uint32_t* d_counts[SLOTS] = {};  // [2][world]: row 0 hits this rank sends to each owner, row 1 hits it receives
    uint32_t* h_counts[SLOTS] = {};  // pinned copy
    hipEvent_t ev_counts[SLOTS] = {}, ev_exchanged[SLOTS] = {}, ev_applied[SLOTS] = {};
    // per slot, per peer, in HITS: what goes where in the sorted / received arrays
    std::vector<uint64_t> send_off[SLOTS], send_cnt[SLOTS], recv_off[SLOTS], recv_cnt[SLOTS];
    // byte-scaled copies handed to the transport (must outlive the call only)
    std::vector<uint64_t> b_so, b_sc, b_ro, b_rc, v_so[2], v_sc[2], v_ro[2], v_rc[2], c_off, c_cnt;
    // rl_sharded_check_requests_device (multi-counter requests): allocated at the first call
    struct ReqBufs {
        bool ready = false;
        uint32_t *req_of_hit = nullptr, *req_id_sorted = nullptr, *r_req = nullptr, *d_words = nullptr, *h_words = nullptr;
        uint8_t *pass_recv = nullptr, *pass_sorted = nullptr, *pass_home = nullptr, *adm = nullptr, *adm_sorted = nullptr,
                *adm_recv = nullptr, *reach_sorted = nullptr, *reach_recv = nullptr;
        int32_t* first = nullptr;
        uint64_t *rem_recv = nullptr, *exp_recv = nullptr, *rem_sorted = nullptr, *exp_sorted = nullptr;
        uint32_t rounds_hint = 2;  // rounds the last multi-counter step needed: the length of the next one's blind group
    } rq;
    std::deque<Slice> pending;
    uint64_t seq = 0;
    uint64_t last_engine_slice = ~0ull;  // the slice whose local batch was submitted to the engine last (its replay may be held back)
    mutable std::mutex mu;
    char err[320] = {0};
};

namespace {

// RL_SHARDED_TRACE=1 (experiment builds): host timestamps of the router's calls, one stderr line each
static bool g_trace = RL_EXP_ENV("RL_SHARDED_TRACE") != nullptr;
static double t_us() {
    static const auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}
#define SH_TRACE(...)                                  \
    do {                                               \
        if (g_trace) std::fprintf(stderr, __VA_ARGS__); \
    } while (0)

int32_t fail(rl_sharded* s, int32_t code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(s->err, sizeof(s->err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_S(s, call)                                                                                    \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) return fail((s), RL_ERR_DEVICE, "%s: %s", #call, hipGetErrorString(e_));    \
    } while (0)
#define ENG_S(s, call)                                                                                    \
    do {                                                                                                  \
        int32_t rc_ = (call);                                                                             \
        if (rc_ != RL_OK) return fail((s), rc_, "%s: %s", #call, rl_last_error((s)->e));                 \
    } while (0)

// Verdict bytes of an APPLIED slice as a transport segment: what this rank decided for peer p's hits goes back
// to p (the received layout), what comes in lands in the sorted layout of this rank's own slice.
void verdict_xfer(rl_sharded* s, const Slice& p, int which, rl_xfer* x) {
    const int slot = p.slot;
    s->v_so[which] = s->recv_off[slot];
    s->v_sc[which] = s->recv_cnt[slot];
    s->v_ro[which] = s->send_off[slot];
    s->v_rc[which] = s->send_cnt[slot];
    x->send = s->recv_verdict[slot];
    x->recv = s->sorted_verdict[slot];
    x->send_off = s->v_so[which].data();
    x->send_cnt = s->v_sc[which].data();
    x->recv_off = s->v_ro[which].data();
    x->recv_cnt = s->v_rc[which].data();
}

int32_t applied_event(rl_sharded* s, Slice& p);

// One submit on the exchange stream (what waits for what decides the order):
//   1. the hit records of the slice routed by the submit BEFORE (its sizes are on the host) to their owners, and that slice's
//      batch to the engine: nothing in front of them waits for anything, so the engine has the next batch queued while it
//      still replays the one before;
//   2. the owner partition of slice i and its SIZES, by themselves, then their copy to the host: the next submit needs them
//      and must not find them behind a wait for a batch (riding in the verdicts' group — one group fewer, the first cut —
//      they sat behind the replay of slice i - 2, the host's hipEventSynchronize on them blocked 16-35 us, and the engine got
//      the batch of slice i - 1 only once that replay had ended; riding in the hit records' group they put the route kernels
//      of slice i — 35 us beside a replay — in front of the records the engine was waiting for);
//   3. the verdicts of the slices whose batches an earlier submit enqueued (this one waits for their replays), un-permuted.
int32_t apply_prepare(rl_sharded* s, Slice& p, rl_xfer* x);
int32_t apply_finish(rl_sharded* s, Slice& p);

int32_t route(rl_sharded* s, const rl_hit* d_hits, uint32_t n, uint64_t now, uint8_t* out, const std::vector<Slice*>& returned,
              Slice* to_apply) {
    const int slot = (int)(s->seq % SLOTS);
    const uint32_t W = s->world;
    int32_t rc;
    if (to_apply) {
        rl_xfer xh;
        rc = apply_prepare(s, *to_apply, &xh);
        if (rc != RL_OK) return rc;
        rc = s->t.exchange(s->t.ctx, &xh, 1, s->cs);
        if (rc != RL_OK) return fail(s, rc, "exchange (hits) failed");  // (the transport itself is gone: nothing to keep in step with)
        rc = apply_finish(s, *to_apply);
        if (rc != RL_OK) return rc;
        SH_TRACE("[sh] %9.1f  hits exchanged, engine half handed over\n", t_us());
    }
    ENG_S(s, rl_route_partition_stream(s->e, s->cs, d_hits, n, W, s->sorted[slot], s->perm[slot], s->d_counts[slot]));
    SH_TRACE("[sh] %9.1f  route kernels enqueued\n", t_us());
    rl_xfer xc;
    xc.send = s->d_counts[slot];
    xc.recv = s->d_counts[slot] + W;
    xc.send_off = xc.recv_off = s->c_off.data();
    xc.send_cnt = xc.recv_cnt = s->c_cnt.data();
    rc = s->t.exchange(s->t.ctx, &xc, 1, s->cs);
    if (rc != RL_OK) return fail(s, rc, "exchange (counts) failed");
    HIP_S(s, hipMemcpyAsync(s->h_counts[slot], s->d_counts[slot], 2 * W * sizeof(uint32_t), hipMemcpyDeviceToHost, s->cs));
    HIP_S(s, hipEventRecord(s->ev_counts[slot], s->cs));
    SH_TRACE("[sh] %9.1f  counts exchanged + copy + event\n", t_us());
    rl_xfer xs[SLOTS];
    uint32_t nx = 0;
    int which = 0;
    for (Slice* p : returned) {
        const int32_t erc = applied_event(s, *p);
        if (erc != RL_OK) return erc;
        HIP_S(s, hipStreamWaitEvent(s->cs, s->ev_applied[p->slot], 0));
        verdict_xfer(s, *p, which++, &xs[nx++]);
    }
    if (nx) {
        rc = s->t.exchange(s->t.ctx, xs, nx, s->cs);
        if (rc != RL_OK) return fail(s, rc, "exchange (verdicts) failed");
    }
    SH_TRACE("[sh] %9.1f  group 2 enqueued\n", t_us());
    for (Slice* p : returned) {
        ENG_S(s, rl_unpermute_u8_stream(s->e, s->cs, s->sorted_verdict[p->slot], s->perm[p->slot], p->n, p->out));
        p->stage = RETURNED;
    }
    Slice sl;
    sl.slot = slot;
    sl.n = n;
    sl.now = now;
    sl.out = out;
    sl.id = s->seq;
    s->pending.push_back(sl);
    ++s->seq;
    return RL_OK;
}

// APPLIED, in two halves around the exchange that carries the slice's hit records to their owners (the sizes reached the
// host at least one submit ago): apply_prepare describes the transfer, apply_finish — behind the exchange on the stream —
// hands the received records to the engine.
int32_t apply_prepare(rl_sharded* s, Slice& p, rl_xfer* x) {
    const int slot = p.slot;
    const uint32_t W = s->world;
    SH_TRACE("[sh] %9.1f apply(%llu) begin\n", t_us(), (unsigned long long)p.id);
    HIP_S(s, hipEventSynchronize(s->ev_counts[slot]));
    SH_TRACE("[sh] %9.1f apply(%llu) counts seen\n", t_us(), (unsigned long long)p.id);
    uint64_t so = 0, ro = 0;
    for (uint32_t q = 0; q < W; ++q) {
        s->send_off[slot][q] = so;
        s->send_cnt[slot][q] = s->h_counts[slot][q];
        so += s->h_counts[slot][q];
        s->recv_off[slot][q] = ro;
        s->recv_cnt[slot][q] = s->h_counts[slot][W + q];
        ro += s->h_counts[slot][W + q];
    }
    if (so != p.n) return fail(s, RL_ERR_DEVICE, "router counted %llu of %u hits", (unsigned long long)so, p.n);  // (an engine bug, not an input)
    p.n_recv = (uint32_t)ro;  // (the receive buffers hold world x max_slice hits: whatever the skew, it can be received)
    for (uint32_t q = 0; q < W; ++q) {
        s->b_so[q] = s->send_off[slot][q] * sizeof(rl_hit);
        s->b_sc[q] = s->send_cnt[slot][q] * sizeof(rl_hit);
        s->b_ro[q] = s->recv_off[slot][q] * sizeof(rl_hit);
        s->b_rc[q] = s->recv_cnt[slot][q] * sizeof(rl_hit);
    }
    x->send = s->sorted[slot];
    x->recv = s->recv_hits[slot];
    x->send_off = s->b_so.data();
    x->send_cnt = s->b_sc.data();
    x->recv_off = s->b_ro.data();
    x->recv_cnt = s->b_rc.data();
    return RL_OK;
}

// (the second half of apply_finish: everything that goes to the APPLY stream / the engine)
int32_t apply_engine_half(rl_sharded* s, Slice& p) {
    const int slot = p.slot;
    const uint64_t ro = p.n_recv;
    if (s->as) {
        // a wait command costs its stream ~9 us even when its event completed long ago: where the HOST can see the exchange
        // complete (the usual case: a 6 us copy enqueued 20 us ago) the apply stream is not made to wait for it
        if (hipEventQuery(s->ev_exchanged[slot]) != hipSuccess) {
            (void)hipGetLastError();  // (hipErrorNotReady is an answer, not an error: do not leave it for the engine's next HIP_TRY)
            HIP_S(s, hipStreamWaitEvent(s->as, s->ev_exchanged[slot], 0));
        }
    } else {
        ENG_S(s, rl_engine_wait_event(s->e, s->ev_exchanged[slot]));
    }
    if (ro > s->max_recv) {
        p.err = RL_ERR_BATCH_TOO_LARGE;
        std::snprintf(p.errmsg, sizeof(p.errmsg), "rank %u: %llu routed hits exceed the engine's max_batch_hits (%u)", s->rank,
                      (unsigned long long)ro, s->max_recv);
    } else if (p.n_recv) {
        // (an engine that keeps its two streams records "applied" itself, behind the replay, whenever that goes out)
        const int32_t brc = rl_check_and_update_submit_device_ev(s->e, s->recv_hits[slot], p.n_recv, p.now, s->recv_verdict[slot],
                                                                 nullptr, s->as ? nullptr : s->ev_applied[slot]);
        if (brc == RL_OK) {
            p.waits = true;
            s->last_engine_slice = p.id;
            if (!s->as) p.applied_recorded = true;
        } else {
            p.err = brc;
            std::snprintf(p.errmsg, sizeof(p.errmsg), "rank %u: local batch refused: %s", s->rank, rl_last_error(s->e));
        }
    }
    if (p.err && p.n_recv)  // what this rank owed its peers: "failed" (on the exchange stream, in front of the verdict exchange)
        HIP_S(s, hipMemsetAsync(s->recv_verdict[slot], 0xFF, p.n_recv, s->cs));
    // "applied": on the communicator's own apply stream the event goes in right behind the batch.  An engine that keeps
    // its two streams enqueues the batch's replay one submit LATE (so that the wait for its partition is answered by the
    // host, rl_engine.h) and recording an event now would force it out behind a wait command: the event is recorded when
    // the verdict exchange needs it (applied_event), two submits from now, by when the replay has long been enqueued.
    if (s->as) {
        HIP_S(s, hipEventRecord(s->ev_applied[slot], s->as));
        p.applied_recorded = true;
    }
    p.stage = APPLIED;
    return RL_OK;
}

int32_t apply_finish(rl_sharded* s, Slice& p) {
    HIP_S(s, hipEventRecord(s->ev_exchanged[p.slot], s->cs));
    return apply_engine_half(s, p);
}

// the two halves around an exchange of its own (the pipeline drains: rl_sharded_collect of a slice that is still ROUTED)
int32_t apply(rl_sharded* s, Slice& p) {
    rl_xfer x;
    int32_t rc = apply_prepare(s, p, &x);
    if (rc != RL_OK) return rc;
    rc = s->t.exchange(s->t.ctx, &x, 1, s->cs);
    if (rc != RL_OK) return fail(s, rc, "exchange (hits) failed");  // (the transport itself is gone: nothing to keep in step with)
    return apply_finish(s, p);
}

// ev_applied[slot] of an APPLIED slice, recorded by now
int32_t applied_event(rl_sharded* s, Slice& p) {
    // (the replay of the batch the engine was given LAST may still be held back, and its event with it; every earlier one
    // went out when the next batch was submitted)
    if (!s->as && p.id == s->last_engine_slice) ENG_S(s, rl_engine_flush(s->e));
    if (!p.applied_recorded) {  // (no local batch for this slice: nothing received, or the slice failed here)
        ENG_S(s, rl_engine_record_event(s->e, s->ev_applied[p.slot]));
        p.applied_recorded = true;
    }
    return RL_OK;
}

// RETURNED, outside a submit (the pipeline drains): the verdict exchange alone
int32_t give_back(rl_sharded* s, Slice& p) {
    const int32_t erc = applied_event(s, p);
    if (erc != RL_OK) return erc;
    HIP_S(s, hipStreamWaitEvent(s->cs, s->ev_applied[p.slot], 0));
    rl_xfer x;
    verdict_xfer(s, p, 0, &x);
    const int32_t rc = s->t.exchange(s->t.ctx, &x, 1, s->cs);
    if (rc != RL_OK) return fail(s, rc, "exchange (verdicts) failed");
    ENG_S(s, rl_unpermute_u8_stream(s->e, s->cs, s->sorted_verdict[p.slot], s->perm[p.slot], p.n, p.out));
    p.stage = RETURNED;
    return RL_OK;
}

int32_t create_common(rl_engine* e, uint32_t world, uint32_t rank, uint32_t max_slice_hits, rl_sharded* s) {
    s->e = e;
    s->world = world;
    s->rank = rank;
    s->max_slice = max_slice_hits;
    int32_t dev = 0;
    ENG_S(s, rl_engine_info(e, &dev, &s->max_recv));
    s->device = dev;
    HIP_S(s, hipSetDevice(dev));
    {   // the exchange stream gets a priority of its own: streams of one priority share a few hardware queues
        // round-robin, and a queue shared with the engine's decision stream serialises the two (seen in the trace)
        int lo = 0, hi = 0;
        HIP_S(s, hipDeviceGetStreamPriorityRange(&lo, &hi));
        const char* pr = RL_EXP_ENV("RL_SHARDED_STREAM_PRIO");
        const int prio = (pr && pr[0] == '0') ? lo : hi;
        HIP_S(s, hipStreamCreateWithPriority(&s->cs, hipStreamNonBlocking, prio));
    }
    const char* es = RL_EXP_ENV("RL_SHARDED_ENGINE_STREAMS");
    if (!es || std::strcmp(es, "own") != 0) {
        HIP_S(s, hipStreamCreateWithFlags(&s->as, hipStreamNonBlocking));
        ENG_S(s, rl_engine_set_stream(e, s->as, 1));
    } else {
        ENG_S(s, rl_engine_set_stream(e, nullptr, 0));
    }
    // receive side: every rank's whole slice may hash to this one (the engine's max_batch_hits only bounds what it can
    // APPLY in one batch: a slice that exceeds it fails as a collective outcome, see Slice::err)
    const size_t ms = max_slice_hits ? max_slice_hits : 1;
    const size_t mr = std::max<size_t>(s->max_recv ? s->max_recv : 1, (size_t)world * ms);
    for (int q = 0; q < SLOTS; ++q) {
        HIP_S(s, hipMalloc(&s->sorted[q], ms * sizeof(rl_hit)));
        HIP_S(s, hipMalloc(&s->perm[q], ms * sizeof(uint32_t)));
        HIP_S(s, hipMalloc(&s->sorted_verdict[q], ms));
        HIP_S(s, hipMalloc(&s->recv_hits[q], mr * sizeof(rl_hit)));
        HIP_S(s, hipMalloc(&s->recv_verdict[q], mr));
        HIP_S(s, hipMalloc(&s->d_counts[q], 2 * world * sizeof(uint32_t)));
        HIP_S(s, hipHostMalloc(reinterpret_cast<void**>(&s->h_counts[q]), 2 * world * sizeof(uint32_t), hipHostMallocDefault));
        HIP_S(s, hipEventCreateWithFlags(&s->ev_counts[q], hipEventDisableTiming));
        HIP_S(s, hipEventCreateWithFlags(&s->ev_exchanged[q], hipEventDisableTiming));
        HIP_S(s, hipEventCreateWithFlags(&s->ev_applied[q], hipEventDisableTiming));
        s->send_off[q].assign(world, 0);
        s->send_cnt[q].assign(world, 0);
        s->recv_off[q].assign(world, 0);
        s->recv_cnt[q].assign(world, 0);
    }
    for (auto* v : {&s->b_so, &s->b_sc, &s->b_ro, &s->b_rc, &s->v_so[0], &s->v_sc[0], &s->v_ro[0], &s->v_rc[0], &s->v_so[1],
                    &s->v_sc[1], &s->v_ro[1], &s->v_rc[1]})
        v->assign(world, 0);
    HIP_S(s, hipMalloc(&s->rq.d_words, RQ_WORDS * 4));
    HIP_S(s, hipHostMalloc(reinterpret_cast<void**>(&s->rq.h_words), RQ_WORDS * 4, hipHostMallocDefault));
    s->c_off.resize(world);
    s->c_cnt.assign(world, sizeof(uint32_t));
    for (uint32_t q = 0; q < world; ++q) s->c_off[q] = q * sizeof(uint32_t);
    return RL_OK;
}

}  // namespace

extern "C" {

// behind the same barrier as every entry point of this library (../rl_abi_guard.h): proves THIS library was built
// with it (tests/test_abi_barrier.py, no GPU needed)
int32_t rl_sharded_abi_selftest(int32_t kind) try {
    std::vector<uint64_t> unwound(16, 1ull);
    switch (kind) {
        case 1: throw std::bad_alloc();
        case 2: throw std::length_error("rl_sharded_abi_selftest: std::length_error");
        case 3: throw 42;
        case 4: {
            std::vector<uint64_t> v;
            v.resize((size_t)1 << 58);
            return (int32_t)v.size();
        }
        default: return RL_OK;
    }
} RL_ABI_CATCH

int32_t rl_sharded_unique_id(uint8_t id[RL_UNIQUE_ID_BYTES]) try {
    static_assert(sizeof(ncclUniqueId) == RL_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!id) return RL_ERR_INVALID;
    char why[600];
    RcclApi* api = rccl_api(why, sizeof(why));
    if (!api) {
        std::fprintf(stderr, "rl_sharded_unique_id: %s\n", why);
        return RL_ERR_DEVICE;
    }
    ncclUniqueId u;
    if (api->GetUniqueId(&u) != ncclSuccess) return RL_ERR_DEVICE;
    std::memcpy(id, &u, sizeof(u));
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_sharded_create(rl_engine* e, uint32_t world, uint32_t rank, const rl_transport* t, uint32_t max_slice_hits,
                          rl_sharded** out) try {
    if (!e || !out || !t || !t->exchange || world == 0 || world > MAX_WORLD || rank >= world) return RL_ERR_INVALID;
    rl_sharded* s = new (std::nothrow) rl_sharded();
    if (!s) return RL_ERR_NOMEM;
    s->t = *t;
    const int32_t rc = create_common(e, world, rank, max_slice_hits, s);
    if (rc != RL_OK) {
        std::fprintf(stderr, "rl_sharded_create: %s\n", s->err);
        rl_sharded_destroy(s);
        return rc;
    }
    *out = s;
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_sharded_create_rccl(rl_engine* e, uint32_t world, uint32_t rank, const uint8_t id[RL_UNIQUE_ID_BYTES],
                               uint32_t max_slice_hits, rl_sharded** out) try {
    if (!e || !out || !id || world == 0 || world > MAX_WORLD || rank >= world) return RL_ERR_INVALID;
    int32_t dev = 0;
    if (rl_engine_info(e, &dev, nullptr) != RL_OK) return RL_ERR_INVALID;
    if (hipSetDevice(dev) != hipSuccess) return RL_ERR_DEVICE;
    char why[600];
    RcclApi* api = rccl_api(why, sizeof(why));
    if (!api) {
        std::fprintf(stderr, "rl_sharded_create_rccl: %s\n", why);
        return RL_ERR_DEVICE;
    }
    auto* r = new (std::nothrow) RcclTransport();
    if (!r) return RL_ERR_NOMEM;
    r->api = api;
    r->world = world;
    r->rank = rank;
    r->e = e;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    if (api->CommInitRank(&r->comm, (int)world, u, (int)rank) != ncclSuccess) {
        delete r;
        return RL_ERR_DEVICE;
    }
    rl_transport t;
    t.ctx = r;
    t.exchange = rccl_exchange;
    const int32_t rc = rl_sharded_create(e, world, rank, &t, max_slice_hits, out);
    if (rc != RL_OK) {
        api->CommDestroy(r->comm);
        delete r;
        return rc;
    }
    (*out)->rccl = r;
    return RL_OK;
} RL_ABI_CATCH

void rl_sharded_destroy(rl_sharded* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->cs) (void)hipStreamSynchronize(s->cs);
    if (s->as) (void)hipStreamSynchronize(s->as);
    // the local batches of slices nobody collected are still in flight on the engine: it would refuse to change streams
    // (RL_ERR_BUSY) and keep pointing at the stream destroyed below
    if (s->e)
        for (const Slice& p : s->pending) {
            if (p.sweep) (void)rl_sweep_expired_collect(s->e, nullptr);
            else if (p.waits) (void)rl_check_and_update_collect(s->e);
        }
    bool engine_released = true;
    if (s->e && s->as) engine_released = rl_engine_set_stream(s->e, nullptr, 0) == RL_OK;  // back to the engine's own streams
    if (s->rccl) {
        s->rccl->api->CommDestroy(s->rccl->comm);
        delete s->rccl;
    }
    for (int q = 0; q < SLOTS; ++q) {
        (void)hipFree(s->sorted[q]);
        (void)hipFree(s->perm[q]);
        (void)hipFree(s->sorted_verdict[q]);
        (void)hipFree(s->recv_hits[q]);
        (void)hipFree(s->recv_verdict[q]);
        (void)hipFree(s->d_counts[q]);
        if (s->h_counts[q]) (void)hipHostFree(s->h_counts[q]);
        if (s->ev_counts[q]) (void)hipEventDestroy(s->ev_counts[q]);
        if (s->ev_exchanged[q]) (void)hipEventDestroy(s->ev_exchanged[q]);
        if (s->ev_applied[q]) (void)hipEventDestroy(s->ev_applied[q]);
    }
    for (void* q : {(void*)s->rq.req_of_hit, (void*)s->rq.req_id_sorted, (void*)s->rq.r_req, (void*)s->rq.d_words,
                    (void*)s->rq.pass_recv, (void*)s->rq.pass_sorted, (void*)s->rq.pass_home, (void*)s->rq.adm,
                    (void*)s->rq.adm_sorted, (void*)s->rq.adm_recv, (void*)s->rq.first, (void*)s->rq.rem_recv,
                    (void*)s->rq.exp_recv, (void*)s->rq.rem_sorted, (void*)s->rq.exp_sorted, (void*)s->rq.reach_sorted,
                    (void*)s->rq.reach_recv})
        if (q) (void)hipFree(q);
    if (s->rq.h_words) (void)hipHostFree(s->rq.h_words);
    if (s->cs) (void)hipStreamDestroy(s->cs);
    if (s->as && engine_released) (void)hipStreamDestroy(s->as);  // (never destroy a stream the engine still references)
    delete s;
}

const char* rl_sharded_last_error(const rl_sharded* s) { return s ? s->err : "null communicator"; }

int32_t rl_sharded_submit_device(rl_sharded* s, const rl_hit* d_hits, uint32_t n_hits, uint64_t now_us, uint8_t* d_verdict) try {
    if (!s || (n_hits && (!d_hits || !d_verdict))) return RL_ERR_INVALID;
    std::lock_guard<std::mutex> g(s->mu);
    SH_TRACE("[sh] %9.1f submit(%llu) begin, %zu pending\n", t_us(), (unsigned long long)s->seq, s->pending.size());
    if (n_hits > s->max_slice) return fail(s, RL_ERR_BATCH_TOO_LARGE, "slice of %u hits, communicator sized for %u", n_hits, s->max_slice);
    if (s->pending.size() >= (size_t)SLOTS) return fail(s, RL_ERR_BUSY, "%d slices are in flight: collect first", SLOTS);
    HIP_S(s, hipSetDevice(s->device));
    // (the order of things on the exchange stream: see route())
    int32_t rc;
    std::vector<Slice*> to_return;  // the slices whose batch was enqueued by an EARLIER submit (at most two)
    for (auto& p : s->pending)
        if (p.stage == APPLIED && to_return.size() < 2) to_return.push_back(&p);
    Slice* to_apply = nullptr;      // the slice the submit before routed (at most one: every submit applies what it finds)
    for (auto& p : s->pending)
        if (p.stage == ROUTED) {
            if (!to_apply) to_apply = &p;
            else if ((rc = apply(s, p)) != RL_OK) return rc;
        }
    rc = route(s, d_hits, n_hits, now_us, d_verdict, to_return, to_apply);
    SH_TRACE("[sh] %9.1f submit end (returned %zu)\n", t_us(), to_return.size());
    return rc;
} RL_ABI_CATCH

int32_t rl_sharded_collect(rl_sharded* s, uint32_t* n_applied) try {
    if (!s) return RL_ERR_INVALID;
    std::lock_guard<std::mutex> g(s->mu);
    if (s->pending.empty()) return fail(s, RL_ERR_INVALID, "nothing in flight");
    HIP_S(s, hipSetDevice(s->device));
    Slice& p = s->pending.front();
    if (p.sweep) return fail(s, RL_ERR_INVALID, "the oldest command in flight is a sweep: rl_sharded_sweep_collect");
    int32_t rc;
    SH_TRACE("[sh] %9.1f collect(%llu) begin stage %d\n", t_us(), (unsigned long long)p.id, (int)p.stage);
    if (p.stage == ROUTED && (rc = apply(s, p)) != RL_OK) return rc;
    if (p.stage == APPLIED && (rc = give_back(s, p)) != RL_OK) return rc;
    const Slice done = p;
    s->pending.pop_front();
    if (n_applied) *n_applied = done.err ? 0u : done.n_recv;
    if (done.waits) {
        SH_TRACE("[sh] %9.1f collect(%llu) engine collect\n", t_us(), (unsigned long long)done.id);
        rc = rl_check_and_update_collect(s->e);  // the status of the local batch this rank applied for the slice
        SH_TRACE("[sh] %9.1f collect(%llu) done\n", t_us(), (unsigned long long)done.id);
        if (rc != RL_OK) return fail(s, rc, "local batch: %s", rl_last_error(s->e));
    }
    if (done.err) return fail(s, done.err, "%s (every exchange of the slice was issued; the hits this rank owns were answered 0xFF)", done.errmsg);
    return RL_OK;
} RL_ABI_CATCH

// A sweep as a COMMAND of the routed pipeline (BASELINE.json configs[4]: "mixed TTLs with concurrent expiry sweep"): every
// rank sweeps its own shard at the same point of the global sequence — behind every slice submitted so far, in front of every
// later one — which is the point a sequential storage would be swept at.  The slices this rank has routed but not yet handed
// to their owners go out first (that exchange is collective: every rank calls the sweep at the same point of its sequence of
// calls, like every other entry of this header); then the engine's own stream-ordered sweep (rl_sweep_expired_submit) takes
// one of the three in-flight places.  No drain: slices in front of it and behind it stay in flight.
int32_t rl_sharded_sweep_submit(rl_sharded* s, uint64_t now_us) try {
    if (!s) return RL_ERR_INVALID;
    std::lock_guard<std::mutex> g(s->mu);
    // (everything in flight is on the engine once the routed slice has been handed over below, plus the sweep itself: the
    // engine takes three commands — the same bound on every rank, whatever its share of the slices was)
    if (s->pending.size() >= (size_t)SLOTS - 1) return fail(s, RL_ERR_BUSY, "%d commands are in flight: collect first", SLOTS - 1);
    HIP_S(s, hipSetDevice(s->device));
    int32_t rc;
    for (auto& p : s->pending)
        if (!p.sweep && p.stage == ROUTED && (rc = apply(s, p)) != RL_OK) return rc;
    ENG_S(s, rl_sweep_expired_submit(s->e, now_us));
    Slice sl;
    sl.sweep = true;
    sl.stage = RETURNED;
    sl.now = now_us;
    sl.id = s->seq;
    sl.slot = (int)(s->seq % SLOTS);  // (its buffers stay unused: a slot is just a place in the window of three)
    s->pending.push_back(sl);
    ++s->seq;
    return RL_OK;
} RL_ABI_CATCH

// Finish the OLDEST command, which must be a sweep: *n_removed = cells THIS rank's shard dropped (the global figure is the
// sum over the ranks — the host's to add up if it wants it).
int32_t rl_sharded_sweep_collect(rl_sharded* s, uint64_t* n_removed) try {
    if (!s) return RL_ERR_INVALID;
    std::lock_guard<std::mutex> g(s->mu);
    if (s->pending.empty()) return fail(s, RL_ERR_INVALID, "nothing in flight");
    if (!s->pending.front().sweep) return fail(s, RL_ERR_INVALID, "the oldest command in flight is a slice: rl_sharded_collect");
    HIP_S(s, hipSetDevice(s->device));
    s->pending.pop_front();
    uint64_t n = 0;
    const int32_t rc = rl_sweep_expired_collect(s->e, &n);
    if (n_removed) *n_removed = n;
    if (rc != RL_OK) return fail(s, rc, "sweep: %s", rl_last_error(s->e));
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_sharded_check_and_update_device(rl_sharded* s, const rl_hit* d_hits, uint32_t n_hits, uint64_t now_us,
                                           uint8_t* d_verdict, uint32_t* n_applied) try {
    if (!s) return RL_ERR_INVALID;
    if (rl_sharded_in_flight(s)) return RL_ERR_BUSY;
    int32_t rc = rl_sharded_submit_device(s, d_hits, n_hits, now_us, d_verdict);
    if (rc != RL_OK) return rc;
    rc = rl_sharded_collect(s, n_applied);
    const int32_t rs = rl_sharded_sync(s);
    return rc != RL_OK ? rc : rs;
} RL_ABI_CATCH

// ---- multi-counter requests, counters sharded by key (SURVEY.md §8e "k > 1") --------------------------------------
namespace {

// The arrays of the multi-counter step, allocated at its first call.  (The words every collective decision travels in are
// allocated with the communicator: a rank that runs out of memory HERE can still tell its peers.)
int32_t req_bufs(rl_sharded* s) try {
    if (s->rq.ready) return RL_OK;
    const size_t ms = s->max_slice ? s->max_slice : 1;
    const size_t mr = std::max<size_t>(s->max_recv ? s->max_recv : 1, (size_t)s->world * ms);
    auto& q = s->rq;
    HIP_S(s, hipMalloc(&q.req_of_hit, ms * 4));
    HIP_S(s, hipMalloc(&q.req_id_sorted, ms * 4));
    HIP_S(s, hipMalloc(&q.r_req, mr * 4));
    HIP_S(s, hipMalloc(&q.pass_recv, mr));
    HIP_S(s, hipMalloc(&q.pass_sorted, ms));
    HIP_S(s, hipMalloc(&q.pass_home, ms));
    HIP_S(s, hipMalloc(&q.adm, ms));
    HIP_S(s, hipMalloc(&q.adm_sorted, ms));
    HIP_S(s, hipMalloc(&q.adm_recv, mr));
    HIP_S(s, hipMalloc(&q.reach_sorted, ms));  // (the walks' ends travel in arrays of their own: a step that is not at its fixpoint
    HIP_S(s, hipMalloc(&q.reach_recv, mr));    //  yet goes on with the admitted bits the owners hold)
    HIP_S(s, hipMalloc(&q.first, ms * 4));
    HIP_S(s, hipMalloc(&q.rem_recv, mr * 8));
    HIP_S(s, hipMalloc(&q.exp_recv, mr * 8));
    HIP_S(s, hipMalloc(&q.rem_sorted, ms * 8));
    HIP_S(s, hipMalloc(&q.exp_sorted, ms * 8));
    q.ready = true;
    return RL_OK;
} RL_ABI_CATCH

// n 32-bit words of every rank to every rank (d_send: this rank's, on the device): rq.d_words[RQ_GATHER + p * n + r] = rank p's
// r-th; gather_words also hands them to the host (out[p * n + r]).  A collective.
int32_t gather_words_device(rl_sharded* s, const uint32_t* d_send, uint32_t n) {
    const uint32_t W = s->world;
    std::vector<uint64_t> zero(W, 0), bytes(W, 4ull * n), ro(W);
    for (uint32_t p = 0; p < W; ++p) ro[p] = 4ull * n * p;
    rl_xfer x;
    x.send = d_send;
    x.recv = s->rq.d_words + RQ_GATHER;
    x.send_off = zero.data();
    x.send_cnt = bytes.data();
    x.recv_off = ro.data();
    x.recv_cnt = bytes.data();
    const int32_t rc = s->t.exchange(s->t.ctx, &x, 1, s->cs);
    if (rc != RL_OK) return fail(s, rc, "exchange (words per rank) failed");
    return RL_OK;
}

int32_t gather_words(rl_sharded* s, const uint32_t* d_send, uint32_t* out, uint32_t n = 1) {
    const uint32_t W = s->world;
    const int32_t rc = gather_words_device(s, d_send, n);
    if (rc != RL_OK) return rc;
    HIP_S(s, hipMemcpyAsync(s->rq.h_words, s->rq.d_words + RQ_GATHER, (size_t)W * n * 4, hipMemcpyDeviceToHost, s->cs));
    HIP_S(s, hipStreamSynchronize(s->cs));
    for (uint32_t q = 0; q < W * n; ++q) out[q] = s->rq.h_words[q];
    return RL_OK;
}
int32_t gather_host_word(rl_sharded* s, uint32_t mine, uint32_t* out) {
    if (s->world == 1) {  // (nobody to agree with: the word is the agreement — no device round trip)
        out[0] = mine;
        return RL_OK;
    }
    s->rq.h_words[RQ_WORDS - 1] = mine;
    HIP_S(s, hipMemcpyAsync(s->rq.d_words, s->rq.h_words + RQ_WORDS - 1, 4, hipMemcpyHostToDevice, s->cs));
    return gather_words(s, s->rq.d_words, out);
}
uint32_t max_of(const uint32_t* w, uint32_t n) {
    uint32_t m = 0;
    for (uint32_t p = 0; p < n; ++p) m = std::max(m, w[p]);
    return m;
}

// bytes of one element per routed hit, ingress -> owners (fwd) or owners -> ingress (!fwd), as ONE grouped exchange of
// up to two arrays
int32_t exchange_per_hit(rl_sharded* s, bool fwd, uint32_t elem, const void* a_send, void* a_recv, const void* b_send = nullptr,
                         void* b_recv = nullptr) {
    const uint32_t W = s->world;
    std::vector<uint64_t> so(W), sc(W), ro(W), rc(W);
    for (uint32_t p = 0; p < W; ++p) {
        const uint64_t ho = s->send_off[0][p], hc = s->send_cnt[0][p], oo = s->recv_off[0][p], oc = s->recv_cnt[0][p];
        so[p] = (fwd ? ho : oo) * elem;
        sc[p] = (fwd ? hc : oc) * elem;
        ro[p] = (fwd ? oo : ho) * elem;
        rc[p] = (fwd ? oc : hc) * elem;
    }
    rl_xfer xs[2];
    uint32_t n = 0;
    for (int k = 0; k < 2; ++k) {
        const void* sp = k ? b_send : a_send;
        void* rp = k ? b_recv : a_recv;
        if (!sp) continue;
        xs[n].send = sp;
        xs[n].recv = rp;
        xs[n].send_off = so.data();
        xs[n].send_cnt = sc.data();
        xs[n].recv_off = ro.data();
        xs[n].recv_cnt = rc.data();
        ++n;
    }
    const int32_t r = s->t.exchange(s->t.ctx, xs, n, s->cs);
    if (r != RL_OK) return fail(s, r, "exchange (%u bytes per hit, %s) failed", elem, fwd ? "to the owners" : "back to the ingress ranks");
    return RL_OK;
}

}  // namespace

int32_t rl_sharded_check_requests_device(rl_sharded* s, const rl_hit* d_hits, uint32_t n_hits, const uint32_t* d_req_off,
                                         uint32_t n_req, uint64_t now_us, int32_t load_counters, uint8_t* d_verdict,
                                         int32_t* d_first_limited, uint64_t* d_remaining, uint64_t* d_expires_in_us,
                                         uint32_t* rounds_out) try {
    if (!s) return RL_ERR_INVALID;
    std::lock_guard<std::mutex> g(s->mu);
    HIP_S(s, hipSetDevice(s->device));
    int32_t rc = RL_OK;
    const int32_t mem_rc = req_bufs(s);  // (a rank that cannot allocate says so in the first exchange, like any other refusal)
    const uint32_t W = s->world;
    auto& q = s->rq;
    uint32_t words[MAX_WORLD];
    // ---- 0. a rank whose arguments are unusable says so in the FIRST exchange (beside the counts), so that all ranks leave
    //         together; request ids are (rank, local index) — no rank needs the others' request counts to number its own
    int32_t bad = RL_OK;
    if (mem_rc != RL_OK) bad = RL_ERR_NOMEM;
    else if (!s->pending.empty()) bad = RL_ERR_BUSY;
    else if ((n_hits && !d_hits) || !d_req_off || (n_req && !d_verdict) || (load_counters && n_hits && (!d_remaining || !d_expires_in_us)))
        bad = RL_ERR_INVALID;
    else if (n_hits > s->max_slice || n_req > s->max_slice || n_req >= REQ_ID_LOCAL) bad = RL_ERR_BATCH_TOO_LARGE;
    const uint32_t base = s->rank * REQ_ID_LOCAL;  // (world <= 16, a slice < 2^27 requests: 31 bits)
    const uint32_t n_eff = bad ? 0u : n_hits, r_eff = bad ? 0u : n_req;
    int32_t* d_first = d_first_limited ? d_first_limited : q.first;
    // ---- 1. route: stable partition by owner, the request id of every routed hit, the counts (+ the word), the records -----
    if (mem_rc == RL_OK) {
        ENG_S(s, rl_route_partition_stream(s->e, s->cs, d_hits, n_eff, W, s->sorted[0], s->perm[0], s->d_counts[0]));
        ENG_S(s, rl_req_ids_stream(s->e, s->cs, d_req_off, r_eff, n_eff, base, s->perm[0], q.req_of_hit, q.req_id_sorted));
    } else {
        HIP_S(s, hipMemsetAsync(s->d_counts[0], 0, 2 * W * sizeof(uint32_t), s->cs));
    }
    s->rq.h_words[RQ_WORDS - 1] = bad ? 0xFFFFFFFFu : 0u;
    HIP_S(s, hipMemcpyAsync(s->rq.d_words, s->rq.h_words + RQ_WORDS - 1, 4, hipMemcpyHostToDevice, s->cs));
    {
        std::vector<uint64_t> zero(W, 0), four(W, 4), wro(W);
        for (uint32_t p = 0; p < W; ++p) wro[p] = 4ull * p;
        rl_xfer xs[2];
        xs[0].send = s->d_counts[0];
        xs[0].recv = s->d_counts[0] + W;
        xs[0].send_off = xs[0].recv_off = s->c_off.data();
        xs[0].send_cnt = xs[0].recv_cnt = s->c_cnt.data();
        xs[1].send = s->rq.d_words;  // the same word to every rank
        xs[1].recv = s->rq.d_words + RQ_GATHER;
        xs[1].send_off = zero.data();
        xs[1].send_cnt = four.data();
        xs[1].recv_off = wro.data();
        xs[1].recv_cnt = four.data();
        rc = s->t.exchange(s->t.ctx, xs, 2, s->cs);
        if (rc != RL_OK) return fail(s, rc, "exchange (counts) failed");
    }
    HIP_S(s, hipMemcpyAsync(s->h_counts[0], s->d_counts[0], 2 * W * sizeof(uint32_t), hipMemcpyDeviceToHost, s->cs));
    HIP_S(s, hipMemcpyAsync(s->rq.h_words, s->rq.d_words + RQ_GATHER, W * 4, hipMemcpyDeviceToHost, s->cs));
    HIP_S(s, hipStreamSynchronize(s->cs));
    for (uint32_t p = 0; p < W; ++p) words[p] = s->rq.h_words[p];
    if (max_of(words, W) == 0xFFFFFFFFu) {
        if (bad == RL_ERR_NOMEM) return fail(s, bad, "out of device memory for the step's arrays");
        if (bad == RL_ERR_BUSY) return fail(s, bad, "slices are in flight: collect them first");
        if (bad == RL_ERR_BATCH_TOO_LARGE) return fail(s, bad, "%u hits / %u requests, communicator sized for %u (and fewer than 2^27 requests per slice)", n_hits, n_req, s->max_slice);
        if (bad) return fail(s, bad, "null argument");
        return fail(s, RL_ERR_INVALID, "another rank refused the step (its arguments): nothing was applied anywhere");
    }
    uint64_t so = 0, ro = 0;
    for (uint32_t p = 0; p < W; ++p) {
        s->send_off[0][p] = so;
        s->send_cnt[0][p] = s->h_counts[0][p];
        so += s->h_counts[0][p];
        s->recv_off[0][p] = ro;
        s->recv_cnt[0][p] = s->h_counts[0][W + p];
        ro += s->h_counts[0][W + p];
    }
    if (so != n_hits) return fail(s, RL_ERR_DEVICE, "router counted %llu of %u hits", (unsigned long long)so, n_hits);
    const uint32_t n_recv = (uint32_t)ro;
    {   // records and ids in one group (16 + 4 bytes per hit)
        const uint32_t Wn = W;
        std::vector<uint64_t> hso(Wn), hsc(Wn), hro(Wn), hrc(Wn), iso(Wn), isc(Wn), iro(Wn), irc(Wn);
        for (uint32_t p = 0; p < Wn; ++p) {
            hso[p] = s->send_off[0][p] * sizeof(rl_hit);
            hsc[p] = s->send_cnt[0][p] * sizeof(rl_hit);
            hro[p] = s->recv_off[0][p] * sizeof(rl_hit);
            hrc[p] = s->recv_cnt[0][p] * sizeof(rl_hit);
            iso[p] = s->send_off[0][p] * 4;
            isc[p] = s->send_cnt[0][p] * 4;
            iro[p] = s->recv_off[0][p] * 4;
            irc[p] = s->recv_cnt[0][p] * 4;
        }
        rl_xfer xs[2];
        xs[0] = {s->sorted[0], s->recv_hits[0], hso.data(), hsc.data(), hro.data(), hrc.data()};
        xs[1] = {q.req_id_sorted, q.r_req, iso.data(), isc.data(), iro.data(), irc.data()};
        rc = s->t.exchange(s->t.ctx, xs, 2, s->cs);
        if (rc != RL_OK) return fail(s, rc, "exchange (hits + request ids) failed");
    }
    // The owners' side of THIS step runs on the exchange stream itself where the engine's stream is the communicator's (the
    // default): the step is one chain — exchange, owners' kernels, exchange, ingress kernels … — and a dependency that crosses
    // streams costs about 10 us of idle device on this part (barrier packet + signal; six of them per two rounds, measured
    // in a kernel trace), an in-order stream nothing.  The host only stops where it needs a value (round 5 synchronised the
    // exchange stream behind every exchange and the engine's behind every call: fourteen stops).  The engine is handed back
    // its own stream (and blocking rounds) however the step ends: the pipelined slices and other drivers of the phased calls
    // rely on both.
    const bool chained = s->as != nullptr;
    struct Restore {
        rl_engine* e;
        hipStream_t as;
        ~Restore() {
            if (!e) return;
            (void)rl_gen_set_async(e, 0);
            if (rl_engine_set_stream(e, as, 1) != RL_OK) {  // (a pass left open by an early return)
                (void)rl_gen_abort(e);
                (void)rl_engine_set_stream(e, as, 1);
            }
        }
    } restore{nullptr, s->as};
    if (chained) {
        ENG_S(s, rl_engine_set_stream(s->e, s->cs, 1));
        restore.e = s->e;
        ENG_S(s, rl_gen_set_async(s->e, 1));
    } else {
        HIP_S(s, hipStreamSynchronize(s->cs));
    }
    // ---- 2-4, at most three times (the owners' sort may ask for another go with its heavy keys promoted) -------------------
    uint32_t rounds = 0;
    for (int attempt = 0;; ++attempt) {
        // ---- 2. owners: sort by cell, read the cells.  Chained: the sort's outcome is looked at in step 4, where the host
        //         stops for the device anyway (rl_gen_set_async); a retry looks at it here, like round 5 did ------------
        if (chained && attempt > 0) ENG_S(s, rl_gen_set_async(s->e, 0));
        const int32_t brc = rl_gen_begin_device(s->e, s->recv_hits[0], q.r_req, n_recv, now_us, load_counters);
        if (chained && attempt > 0) ENG_S(s, rl_gen_set_async(s->e, 1));
        // A rank whose owner side cannot run (its begin, or later a round, returned a status) keeps taking part in every
        // exchange of the step — the others are in them — and says so in step 4's word, where all ranks drop the step together.
        int32_t dead_rc = RL_OK;
        char dead_msg[200] = {0};
        auto owner_died = [&](int32_t why) {
            if (dead_rc != RL_OK) return;
            dead_rc = why;
            std::snprintf(dead_msg, sizeof(dead_msg), "%s", rl_last_error(s->e));
        };
        if (brc != RL_OK) owner_died(brc);
        if (!chained || attempt > 0) {  // (a begin that waited for its sort: all ranks hear its outcome now)
            rc = gather_host_word(s, dead_rc != RL_OK ? 1u : 0u, words);
            if (rc != RL_OK) return rc;
            if (max_of(words, W)) {
                if (dead_rc == RL_OK) {
                    (void)rl_gen_abort(s->e);
                    return fail(s, RL_ERR_INVALID, "another rank refused the step: nothing was applied anywhere");
                }
                return fail(s, dead_rc, "rank %u: %s (refused on every rank, nothing applied)", s->rank, dead_msg);
            }
        }
        // ---- 3. Jacobi rounds: owners -> pass flags -> ingress AND per request -> admitted bits -> owners, until no rank
        //         saw the admitted set change: the unique fixpoint ----------------------------------------
        auto one_round = [&](bool first_round, uint32_t* d_changed) -> int32_t {
            if (dead_rc == RL_OK) {
                const int32_t rrc = rl_gen_round_device(s->e, first_round ? nullptr : q.adm_recv, q.pass_recv, q.rem_recv, q.exp_recv);
                if (rrc != RL_OK) owner_died(rrc);  // (misuse or a device error: not an outcome of the input)
            }
            int32_t r = exchange_per_hit(s, false, 1, q.pass_recv, q.pass_sorted);
            if (r != RL_OK) return r;
            r = rl_req_round_stream(s->e, s->cs, q.pass_sorted, s->perm[0], d_req_off, q.req_of_hit, n_req, n_hits, first_round ? 1 : 0, q.pass_home,
                                    q.adm, d_first, d_verdict, d_changed, q.adm_sorted);
            if (r != RL_OK) return fail(s, r, "engine: %s", rl_last_error(s->e));
            // the admitted bits go out behind the kernel that made them, before anybody has seen whether anything changed
            return exchange_per_hit(s, true, 1, q.adm_sorted, q.adm_recv);
        };
        rounds = 0;
        if (chained) {
            // The whole of it is ONE chain on the exchange stream, and the host stops once: a GROUP of rounds — as many as the
            // step before needed, at most RQ_BLIND_MAX — goes out without a look in between (a round behind the fixpoint
            // reproduces the fixpoint's flags, admitted bits and verdicts: its inputs are the same, so one too many changes
            // no answer); behind it the walks' ends, the owners' count, every rank's veto word (the cells do not fit / an
            // error / the sort overflowed / my last round still changed something) gathered ON THE DEVICE, and the commit,
            // which applies the pass only where every rank's word is zero.  All ranks then read the same words and take the
            // same turn: done, another group, once more from the sort, or refused everywhere.
            bool retry = false;
            for (;;) {
                const uint32_t group = std::min(std::max(s->rq.rounds_hint, 1u), RQ_BLIND_MAX), stride = 1 + group;
                HIP_S(s, hipMemsetAsync(q.d_words + RQ_VETO, 0, 4 * (1 + RQ_BLIND_MAX), s->cs));
                for (uint32_t gr = 0; gr < group; ++gr) {
                    rc = one_round(rounds + gr == 0, q.d_words + RQ_CHANGED + gr);
                    if (rc != RL_OK) return rc;
                }
                if (!load_counters) {  // ---- 4. the walks' ends -> owners
                    ENG_S(s, rl_req_reached_stream(s->e, s->cs, d_first, q.req_of_hit, s->perm[0], n_hits, q.reach_sorted));
                    rc = exchange_per_hit(s, true, 1, q.reach_sorted, q.reach_recv);
                    if (rc != RL_OK) return rc;
                }
                if (dead_rc == RL_OK) {
                    const int32_t crc = rl_gen_count_async_device(s->e, load_counters ? nullptr : q.reach_recv, q.d_words + RQ_CHANGED + group - 1,
                                                                  q.d_words + RQ_VETO);
                    if (crc != RL_OK) owner_died(crc);
                }
                if (dead_rc != RL_OK) {  // (no owner side here: the word comes from the host)
                    q.h_words[RQ_WORDS - 1] = 2u;
                    HIP_S(s, hipMemcpyAsync(q.d_words + RQ_VETO, q.h_words + RQ_WORDS - 1, 4, hipMemcpyHostToDevice, s->cs));
                }
                rc = gather_words_device(s, q.d_words + RQ_VETO, stride);  // [veto, changed of round 0 .. group-1] of every rank
                if (rc != RL_OK) return rc;
                uint32_t committed = 0;
                bool have_words = false;
                if (dead_rc == RL_OK) {
                    const int32_t crc = rl_gen_commit_gated_device(s->e, q.d_words + RQ_GATHER, q.h_words, W, stride, &committed);
                    if (crc != RL_OK) owner_died(crc);  // (its veto word — 2 or 4 — has told the others)
                    else have_words = true;
                }
                if (!have_words) {
                    HIP_S(s, hipMemcpyAsync(q.h_words, q.d_words + RQ_GATHER, (size_t)W * stride * 4, hipMemcpyDeviceToHost, s->cs));
                    HIP_S(s, hipStreamSynchronize(s->cs));
                }
                uint32_t any = 0, full_rank = 0;
                for (uint32_t p = 0; p < W; ++p) {
                    any |= q.h_words[p * stride];
                    if (q.h_words[p * stride] & 1u) full_rank = p;
                }
                if (any & 2u) {
                    (void)rl_gen_abort(s->e);
                    if (dead_rc != RL_OK && dead_rc != RL_ERR_BUSY) return fail(s, dead_rc, "rank %u: %s (refused on every rank, nothing applied)", s->rank, dead_msg);
                    return fail(s, RL_ERR_INVALID, "another rank refused the step: nothing was applied anywhere");
                }
                if (any & 4u) {  // some owner's sort overflowed and promoted its heavy keys: nothing was computed from it
                    (void)rl_gen_abort(s->e);
                    if (attempt >= 2) return fail(s, RL_ERR_BATCH_TOO_LARGE, "a hash bucket of some rank's share keeps overflowing: split the slice");
                    retry = true;
                    break;
                }
                if (dead_rc != RL_OK) {  // (a failure that was in nobody's word: a HIP error behind the gather)
                    (void)rl_gen_abort(s->e);
                    return fail(s, dead_rc, "rank %u: %s", s->rank, dead_msg);
                }
                if (any & 8u) {  // not there yet: the passes are open on every rank, another group
                    rounds += group;
                    if (rounds > (uint64_t)W * s->max_slice + 2 + RQ_BLIND_MAX) {  // (one more request of the trace prefix is settled per round)
                        (void)rl_gen_abort(s->e);
                        return fail(s, RL_ERR_DEVICE, "the rounds did not converge (bug)");
                    }
                    s->rq.rounds_hint = std::min(rounds + 1, RQ_BLIND_MAX);
                    continue;
                }
                if (any & 1u) {
                    (void)rl_gen_abort(s->e);
                    return fail(s, RL_ERR_TABLE_FULL, "refused, nothing applied on any rank: rank %u's shard cannot take the cells the step creates", full_rank);
                }
                if (!committed) {
                    (void)rl_gen_abort(s->e);
                    return fail(s, RL_ERR_DEVICE, "no rank objected and the pass was not applied (bug)");
                }
                uint32_t used = group;
                for (uint32_t gr = 0; gr < group; ++gr) {
                    uint32_t chg = 0;
                    for (uint32_t p = 0; p < W; ++p) chg |= q.h_words[p * stride + 1 + gr];
                    if (!chg) {  // the admitted set of this round is the one its flags were computed with
                        used = gr + 1;
                        break;
                    }
                }
                rounds += used;
                break;
            }
            if (retry) continue;
            s->rq.rounds_hint = rounds;
            break;
        }
        // (an engine with streams of its own, RL_SHARDED_ENGINE_STREAMS=own: round 5's blocking protocol, a stop per call)
        for (bool converged = false; !converged;) {
            HIP_S(s, hipMemsetAsync(q.d_words + RQ_CHANGED, 0, 4, s->cs));
            rc = one_round(rounds == 0, q.d_words + RQ_CHANGED);
            if (rc != RL_OK) return rc;
            uint32_t gw[MAX_WORLD];
            rc = gather_words(s, q.d_words + RQ_CHANGED, gw);  // did ANY rank see a change
            if (rc != RL_OK) return rc;
            converged = max_of(gw, W) == 0;
            if (++rounds > (uint64_t)W * s->max_slice + 2) {
                (void)rl_gen_abort(s->e);
                return fail(s, RL_ERR_DEVICE, "the rounds did not converge (bug)");
            }
        }
        // ---- 4. the walks' ends -> owners; cells to create, room: all ranks fit or none does -------------------------------
        uint32_t n_new = 0;
        uint64_t room = 0;
        int32_t crc;
        if (!load_counters) {
            ENG_S(s, rl_req_reached_stream(s->e, s->cs, d_first, q.req_of_hit, s->perm[0], n_hits, q.adm_sorted));
            rc = exchange_per_hit(s, true, 1, q.adm_sorted, q.adm_recv);
            if (rc != RL_OK) return rc;
            HIP_S(s, hipStreamSynchronize(s->cs));
            crc = dead_rc != RL_OK ? dead_rc : rl_gen_count_device(s->e, q.adm_recv, &n_new, &room);
        } else {
            crc = dead_rc != RL_OK ? dead_rc : rl_gen_count_device(s->e, nullptr, &n_new, &room);
        }
        char cmsg[200] = {0};
        if (dead_rc != RL_OK) std::snprintf(cmsg, sizeof(cmsg), "%s", dead_msg);
        else if (crc != RL_OK) std::snprintf(cmsg, sizeof(cmsg), "%s", rl_last_error(s->e));
        rc = gather_host_word(s, crc != RL_OK ? 2u : (n_new > room ? 1u : 0u), words);
        if (rc != RL_OK) return rc;
        if (const uint32_t worst = max_of(words, W)) {
            (void)rl_gen_abort(s->e);
            if (crc != RL_OK) return fail(s, crc, "rank %u: %s", s->rank, cmsg);
            if (worst == 2u) return fail(s, RL_ERR_DEVICE, "another rank failed while counting: nothing was applied anywhere");
            return fail(s, RL_ERR_TABLE_FULL, "refused, nothing applied on any rank: a shard cannot take the cells the step creates (here: %u new, room %llu)",
                        n_new, (unsigned long long)room);
        }
        {
            const int32_t mrc = rl_gen_commit_device(s->e);
            if (mrc != RL_OK) return fail(s, mrc, "engine: %s", rl_last_error(s->e));
        }
        break;
    }
    if (rounds_out) *rounds_out = rounds;
    // ---- 5. values read before the update, back to the ingress ranks (in_memory.rs:114-116,134-136) -------------------
    if (load_counters) {
        rc = exchange_per_hit(s, false, 8, q.rem_recv, q.rem_sorted, q.exp_recv, q.exp_sorted);
        if (rc != RL_OK) return rc;
        ENG_S(s, rl_unpermute_u64_stream(s->e, s->cs, q.rem_sorted, s->perm[0], n_hits, d_remaining));
        ENG_S(s, rl_unpermute_u64_stream(s->e, s->cs, q.exp_sorted, s->perm[0], n_hits, d_expires_in_us));
    }
    HIP_S(s, hipStreamSynchronize(s->cs));
    return RL_OK;
} RL_ABI_CATCH

void* rl_sharded_stream(rl_sharded* s) { return s ? s->cs : nullptr; }

int32_t rl_sharded_sync(rl_sharded* s) try {
    if (!s) return RL_ERR_INVALID;
    std::lock_guard<std::mutex> g(s->mu);
    HIP_S(s, hipSetDevice(s->device));
    HIP_S(s, hipStreamSynchronize(s->cs));
    return RL_OK;
} RL_ABI_CATCH

uint32_t rl_sharded_in_flight(const rl_sharded* s) {
    if (!s) return 0;
    std::lock_guard<std::mutex> g(s->mu);
    return (uint32_t)s->pending.size();
}

// ---- the in-process transport ---------------------------------------------------------------------------------
struct rl_local_group {
    uint32_t world = 0;
    std::mutex mu;
    std::condition_variable cv;
    uint32_t arrived = 0;
    uint64_t generation = 0;
    std::vector<const rl_xfer*> posted;
    std::vector<uint32_t> posted_n;
    struct Ctx {
        rl_local_group* g;
        uint32_t rank;
    };
    std::vector<Ctx> ctx;

    void barrier() {
        std::unique_lock<std::mutex> l(mu);
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(l, [&] { return generation != gen; });
        }
    }
};

static int32_t local_exchange(void* c, const rl_xfer* xs, uint32_t n, void* stream) {
    auto* ctx = static_cast<rl_local_group::Ctx*>(c);
    rl_local_group* g = ctx->g;
    const uint32_t me = ctx->rank;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int32_t rc = RL_OK;
    if (hipStreamSynchronize(st) != hipSuccess) rc = RL_ERR_DEVICE;  // my send buffers are complete
    g->posted[me] = xs;
    g->posted_n[me] = n;
    g->barrier();
    for (uint32_t k = 0; k < n && rc == RL_OK; ++k)
        for (uint32_t p = 0; p < g->world && rc == RL_OK; ++p) {
            if (g->posted_n[p] != n) {
                rc = RL_ERR_INVALID;  // the ranks disagree on the sequence of exchanges
                break;
            }
            const rl_xfer& theirs = g->posted[p][k];
            const uint64_t cnt = theirs.send_cnt[me];
            if (cnt != xs[k].recv_cnt[p]) {
                rc = RL_ERR_INVALID;
                break;
            }
            if (cnt && hipMemcpyAsync(static_cast<char*>(xs[k].recv) + xs[k].recv_off[p],
                                      static_cast<const char*>(theirs.send) + theirs.send_off[me], cnt, hipMemcpyDeviceToDevice,
                                      st) != hipSuccess)
                rc = RL_ERR_DEVICE;
        }
    if (hipStreamSynchronize(st) != hipSuccess && rc == RL_OK) rc = RL_ERR_DEVICE;
    g->barrier();  // every rank has read: the send buffers (and the posted descriptors) may change again
    return rc;
}

int32_t rl_local_group_create(uint32_t world, rl_local_group** out) try {
    if (!out || world == 0 || world > MAX_WORLD) return RL_ERR_INVALID;
    auto* g = new (std::nothrow) rl_local_group();
    if (!g) return RL_ERR_NOMEM;
    g->world = world;
    g->posted.assign(world, nullptr);
    g->posted_n.assign(world, 0);
    g->ctx.resize(world);
    for (uint32_t r = 0; r < world; ++r) g->ctx[r] = {g, r};
    *out = g;
    return RL_OK;
} RL_ABI_CATCH

void rl_local_group_destroy(rl_local_group* g) { delete g; }

int32_t rl_local_group_transport(rl_local_group* g, uint32_t rank, rl_transport* out) try {
    if (!g || !out || rank >= g->world) return RL_ERR_INVALID;
    out->ctx = &g->ctx[rank];
    out->exchange = local_exchange;
    return RL_OK;
} RL_ABI_CATCH

}  // extern "C"
