// rl_engine.hip — C ABI (include/rl_engine.h) over the gfx950 kernels.
//
// The engine owns: the counter table in HBM (32-byte cells), the device limit table, the
// per-batch scratch (hit -> slot map, ordered list, sort buffers, verdict staging) and one HIP
// stream.  There is no CPU implementation of any entry point: without a HIP device
// rl_engine_create fails with RL_ERR_NO_DEVICE.
#include "../../include/rl_engine.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "rl_abi_guard.h"
#include "rl_cell.hpp"
#include "rl_kernels.hpp"
#include "rl_bucket.hpp"
#include "rl_apply.hpp"
#include "rl_general.hpp"
#include "rl_route.hpp"
#include "rl_match.hpp"
#include "rl_wire.hpp"
#include "rl_resp.hpp"

using namespace rl;

// Switches.  A release build reads the documented handful (include/rl_engine.h: RL_FUSE, RL_SERVE, RL_TINY_MAX, RL_STREAM;
// include/rl_ingest.h: RLI_THREADS).  Everything else — shapes, budgets, priorities, diagnostics that copy from the device
// per batch — exists only in builds with -DRL_EXPERIMENT (scripts/exp/build_variant.sh, limitador_amd/build.py
// build_engine_exp: the library tests/test_gpu_variants.py loads); in a release build the names are not even in the binary.
#ifdef RL_EXPERIMENT
#define RL_EXP_ENV(name) getenv(name)
#else
#define RL_EXP_ENV(name) (static_cast<const char*>(nullptr))
#endif

// ---- red zones (experiment build, RL_REDZONE=1): every device allocation of this file gets RZ_BYTES of a known pattern on
//      either side; rl_debug_redzones() reads them back and names the first block whose pattern a kernel has written over.
//      GPU AddressSanitizer is not available on this pool: this bounds the "stray store next to an engine array" class of
//      device faults over whatever the test-suite drives through the engine (tests/test_gpu_redzones.py).
#ifdef RL_EXPERIMENT
namespace {
constexpr size_t RZ_BYTES = 4096;
constexpr unsigned char RZ_FILL = 0xA5;
struct RzBlock {
    void* base;
    size_t bytes;
};
std::mutex rz_mu;
std::vector<std::pair<void*, RzBlock>> rz_blocks;  // user pointer -> block
bool rz_on() {
    const char* v = getenv("RL_REDZONE");
    return v && atoi(v) != 0;
}
template <class T>
hipError_t rz_malloc(T** p, size_t bytes) {
    if (!rz_on()) return hipMalloc(p, bytes);
    const size_t user = (bytes + 255) & ~size_t(255);  // (hipMalloc's own alignment is kept for the user's pointer)
    void* base = nullptr;
    hipError_t r = hipMalloc(&base, user + 2 * RZ_BYTES);
    if (r != hipSuccess) return r;
    r = hipMemset(base, RZ_FILL, user + 2 * RZ_BYTES);
    if (r != hipSuccess) return r;
    *p = reinterpret_cast<T*>(static_cast<char*>(base) + RZ_BYTES);
    std::lock_guard<std::mutex> g(rz_mu);
    // RL_REDZONE=1: the block itself starts as zeros (what fresh device memory usually holds); =3: as the pattern too — an array
    // that is read before it is written shows (=2: only the arrays of rl_engine_create, by name: rz_poison below)
    if (atoi(getenv("RL_REDZONE")) < 3 && hipMemset(*p, 0, bytes) != hipSuccess) return hipErrorUnknown;
    // (the fills run on the null stream, which the engine's non-blocking streams are not ordered with: a kernel that initialises the
    // block must not be overtaken by them)
    if (hipDeviceSynchronize() != hipSuccess) return hipErrorUnknown;
    rz_blocks.push_back({static_cast<void*>(*p), RzBlock{base, bytes}});
    return hipSuccess;
}
// the two zones of one block: 0 = intact; else msg says where
int rz_check(void* user_ptr, const RzBlock& b, char* msg, size_t msg_len) {
    const size_t user = (b.bytes + 255) & ~size_t(255);
    std::vector<unsigned char> h;
    for (int side = 0; side < 2; ++side) {
        // (the zone behind the block starts at the byte after the user's last one: the alignment padding is part of it)
        const char* at = side == 0 ? static_cast<const char*>(b.base) : static_cast<const char*>(b.base) + RZ_BYTES + b.bytes;
        const size_t len = side == 0 ? RZ_BYTES : RZ_BYTES + (user - b.bytes);
        h.resize(len);
        if (hipMemcpy(h.data(), at, len, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        for (size_t i = 0; i < len; ++i)
            if (h[i] != RZ_FILL) {
                if (msg && msg_len)
                    std::snprintf(msg, msg_len, "block of %zu bytes at %p: the byte %zu %s it was overwritten with 0x%02x", b.bytes, user_ptr,
                                  side == 0 ? RZ_BYTES - i : i, side == 0 ? "before" : "behind", (unsigned)h[i]);
                return 1;
            }
    }
    return 0;
}
// (a block is checked when it is freed — every engine array is, at rl_engine_destroy at the latest — and a damaged zone ends the
// process: a test-suite run with RL_REDZONE=1 cannot pass over one)
// RL_REDZONE=2: the arrays rl_engine_create allocates start as the pattern — all of them, or the one RL_REDZONE_POISON names
bool rz_poison(const char* name) {
    const char* m = getenv("RL_REDZONE");
    if (!m || atoi(m) != 2) return false;
    const char* only = getenv("RL_REDZONE_POISON");
    return !only || std::strcmp(only, name) == 0;
}
hipError_t rz_free(void* p) {
    if (!p) return hipSuccess;
    void* base = p;
    RzBlock blk{};
    bool mine = false;
    {
        std::lock_guard<std::mutex> g(rz_mu);
        for (size_t i = 0; i < rz_blocks.size(); ++i)
            if (rz_blocks[i].first == p) {
                blk = rz_blocks[i].second;
                base = blk.base;
                mine = true;
                rz_blocks.erase(rz_blocks.begin() + i);
                break;
            }
    }
    if (mine) {
        (void)hipDeviceSynchronize();
        char msg[200] = {0};
        if (rz_check(p, blk, msg, sizeof msg) > 0) {
            std::fprintf(stderr, "RL_REDZONE: %s\n", msg);
            std::abort();
        }
    }
    return hipFree(base);
}
}  // namespace
// -> number of blocks with a damaged red zone (0 = none); msg: the first one.  Synchronises the device.
extern "C" __attribute__((visibility("default"))) int32_t rl_debug_redzones(uint32_t* n_blocks, char* msg, uint32_t msg_len) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    std::lock_guard<std::mutex> g(rz_mu);
    int32_t bad = 0;
    if (n_blocks) *n_blocks = (uint32_t)rz_blocks.size();
    for (const auto& kv : rz_blocks) {
        const int r = rz_check(kv.first, kv.second, bad ? nullptr : msg, msg_len);
        if (r < 0) return -1;
        bad += r;
    }
    return bad;
}
#define hipMalloc rz_malloc
#define hipFree rz_free
#endif

static_assert(sizeof(rl_hit) == sizeof(Hit), "rl_hit layout");
static_assert(sizeof(rl_cell_row) == sizeof(CellRow), "rl_cell_row layout");
static_assert(sizeof(rl_match_limit) == sizeof(MatchLimit), "rl_match_limit layout");
static_assert(sizeof(rl_match_cond) == sizeof(MatchCond), "rl_match_cond layout");

constexpr u32 GEN_SUB_MAX = 4u << 20;  // most hits of one pass of the general resolver (its scratch is ~80 B per hit)

// Rotation of the bucketed path's per-batch buffers.  Up to three batches are in flight and the partition of a
// batch may run as soon as the batch `pipe_depth` (2 or 3) before it has been applied:
//   PB_SETS  partitioned records / ranges / hot-bucket table / chunk table: batch p writes set p % 3, last read by
//            k_bkt_apply of batch p - 3
//   BS_ROT   scratch blocks: batch p uses [p % 6]; its replay zeroes [(p + 4) % 6] (= batch p - 2's, whose replay ENDED
//            before this one started — also when the replays of p - 1 and p overlap, RL_XOVER) for batch p + 4 — FOUR
//            ahead, so that the partition of a batch only ever depends on a replay kernel that has ENDED (see
//            apply_events); one more block ([BS_ROT]) belongs to k_bkt_tiny
//   HS_SETS  hot sets: batch p picks set p % 8 and is partitioned with the one batch p - pipe_depth picked; eight,
//            so that the set a partition rewrites is never one a batch still in flight reads (k_hot_state of batch
//            p - 1 and p - 2 read the sets of p - 1 - depth and p - 2 - depth)
constexpr u32 PB_SETS = 3;
constexpr u32 BS_ROT = 6;
constexpr u32 HS_SETS = 8;

struct rl_engine {
    std::mutex mu;
    std::string err;
    int device = 0;
    hipStream_t stream = nullptr;      // the stream every kernel is launched on
    hipStream_t own_stream = nullptr;  // created by the engine; `stream` may be replaced (rl_engine_set_stream)
    u64 seed = 0;

    Cell* table = nullptr;
    u64 cap = 0;
    u32 log2cap = 0;
    u64 live = 0;
    u64 tombs = 0;

    Cell* peer_tables[MERGE_MAX_ACTORS]{};  // what each remote actor is known to have contributed (rl_merge_cells)
    LimitDev* d_limits = nullptr;
    std::vector<LimitDev> h_limits;
    u32 max_limits = 0;
    bool any_zero_window = false;

    u32 max_batch = 0;
    Hit* d_hits = nullptr;        // staging for host-pointer calls
    u32* d_req_off = nullptr;     // staging
    uint8_t* d_verdict = nullptr; // staging
    int32_t* d_first = nullptr;   // staging
    u64* d_remaining = nullptr;   // staging
    u64* d_expires = nullptr;     // staging
    // general resolver (rl_general.hpp): scratch of one pass (gen_cap hits)
    u32 gen_cap = 0;
    u32 gen_sub_max = GEN_SUB_MAX;  // RL_GEN_SUB_MAX: hits per pass
    u32* d_hit_req = nullptr;       // [max_batch] hit -> request of the whole call
    u64* d_req_delta = nullptr;     // [max_batch] staging of per-request u64 deltas
    SHit* d_g_shits = nullptr;
    SegInfo* d_g_seginfo = nullptr;
    SegTot* d_g_segtot = nullptr;
    SegTot* d_g_piece = nullptr;
    u32* d_g_reqstop = nullptr;     // [max_batch] per request
    uint8_t* d_g_reached = nullptr;
    uint8_t* d_g_pass = nullptr;    // [2][gen_cap]
    uint8_t* d_g_admitted = nullptr;
    GenStatus* d_gst = nullptr;
    CellRow* d_row1 = nullptr;      // one staged row (rl_add_counter)
    Status* d_status = nullptr;
    Status* d_sweep_st = nullptr;   // [4] status blocks of the stream-ordered sweeps, by in-flight slot
    u64 last_sweep_removed = 0;
    Status* h_status = nullptr; // pinned
    BatchScratch* d_bs = nullptr;   // [BS_ROT + 1], rotating (see BS_ROT)
    u64 bs_seq = 0;                 // batches of the bucketed path (partitioned or tiny) submitted so far
    // batches of the bucketed path submitted but not yet collected (at most three)
    struct Inflight {
        hipEvent_t tev[10]{};  // start / stop of each kernel of a timed batch, see collect_k1_bucketed
        Status* h_st = nullptr;  // host-mapped: written by the batch's last workgroup
        u32 n = 0, n_wg = 0, ntiles = 0;
        int timed = 0;         // bit 0: k_bkt_part timed, bit 1: k_bkt_apply / k_bkt_tiny timed
        hipEvent_t ev_p_stop = nullptr, ev_a_stop = nullptr;  // the stop events of the timed launches (see submit_k1_bucketed)
        hipEvent_t ev_a_prev = nullptr;  // "applied" of the partitioned batch before this one (idle time in front of a timed k_bkt_apply)
        u32 seq = 0;  // value the batch's last workgroup stores into h_st->n_removed
        bool settled = false;  // completion seen and its new cells already added to `live` (settle_inflight)
        int kind = 0;          // 0: a batch; 1: a stream-ordered sweep (rl_sweep_expired_submit): h_st->n_ord = cells it removed
    } inflight[4];  // at most three in flight
    u64 sub_seq = 0, col_seq = 0;
    u64 inflight_hits = 0;
    unsigned long long* d_total = nullptr;
    unsigned long long* h_total = nullptr; // pinned
    // routing scratch
    u32* d_route_cnt = nullptr;
    // (the router's partition by owner as ONE launch with a ticket barrier, k_route_one: 33-50 us for the launch, the replay beside
    // it 45-58 us, 97 against 82 us per routed slice — scripts/exp/patches/route_one_launch.patch)
    // bucketed hot path (rl_bucket.hpp)
    // the phased form of the general resolver (rl_gen_begin_device .. rl_gen_commit_device / rl_gen_abort)
    bool ph_unchecked = false;  // an async rl_gen_begin_device left the sort's status for rl_gen_count_device to look at
    bool ph_async = false;  // rl_gen_set_async: rl_gen_round_device leaves its kernels on the (caller's) stream without waiting for them
    bool ph_open = false;
    GenArgs ph_A{};
    BatchScratch* ph_bs = nullptr;
    u32 ph_n = 0, ph_rounds = 0;
    bool ph_counted = false;
    u32 gen_bk_log2_big = 10;          // RL_GEN_BUCKET_LOG2_BIG: hash buckets of a pass of more than 2 M hits (2^10: r9g, r14t)
    u32 gen_bk_log2_max = BK_LOG2_MAX; // RL_GEN_BUCKET_LOG2: cap on the general resolver's hash buckets (tests)
    int gen_trace = 0;                 // RL_GEN_TRACE=1: one stderr line per pass of the general resolver; 2: + k_gen_sort phases
    unsigned long long* d_gen_trace = nullptr;
    u32 bk_log2_cfg = 10;              // RL_BUCKET_LOG2: most hash buckets for a batch of up to 2 M hits (larger: BK_LOG2_MAX).  1024, not
                                       // 2048: k_bkt_apply then has 1024 workgroups = four per CU, which leaves 16 wave slots and 80 KB of
                                       // LDS on every CU, and the partition kernels of the next batches (16-wave workgroups; k_bkt_scatter's
                                       // LDS is 70 KB with 1024 buckets) run BESIDE it instead of waiting for it to drain.  1 M-hit batches:
                                       // 2048 buckets 72 us per step (k_bkt_apply 33 us alone), 1024 buckets 66 us (39 us alone), 512: 79 us
    u32 bk_tiles_max = 0;
    u32* d_bk_hist = nullptr;
    u32* d_bk_total = nullptr;
    uint2* d_bk_ranges = nullptr;   // [PB_SETS][BK_MAX]
    // Hot sets, [HS_SETS], rotating: partitioned batch p is partitioned with the set batch p - pipe_depth picked and
    // picks set p % HS_SETS — interleaved lineages, so that the partition of batch p+1 never waits for k_bkt_apply
    // of batch p (any stale set is valid, see HotSet).
    HotSet* d_hot = nullptr;
    u64 part_seq = 0;               // partitioned batches submitted so far
    // Two streams: the partition of batch k+1 (k_bkt_part_c / k_bkt_part: ONE launch, rl_part.hpp, on `pstream`) overlaps the
    // replay of batch k (k_bkt_step, rl_apply.hpp, on `stream`).  With a caller's stream (rl_engine_set_stream) or
    // RL_OVERLAP=0 both are the same stream.  (k_bkt_hist / scan / scatter — rl_bucket.hpp — partition the GENERAL resolver's
    // passes only, on `stream`.)
    hipStream_t pstream = nullptr, own_pstream = nullptr;
    // RL_XOVER=2 (experiment builds; a MEASUREMENT device, the results are wrong): the replays of odd partitioned batches go to
    // `stream2` and nothing orders two consecutive replays — the ceiling of what a per-bucket hand-over between overlapped
    // replays could reach.  Measured: the step does not move (profiles/r04b_overlap_bound.md), so the hand-over is not built.
    hipStream_t stream2 = nullptr;
    int xover = 0;
    bool overlap = true;
    bool ext_events = true;         // RL_EXT_EVENTS=0: hipEventRecord markers behind k_bkt_part_c / k_bkt_step instead of the
                                    // launches' own stop events (two marker commands fewer per batch on the two streams)
    u32* d_hot_arrive = nullptr;    // [2][HOT_MAX] (apply2_hot_item), by the parity of the partitioned batch
    u64* d_cmark = nullptr;         // do_compact in place: the segments' bounds (k_compact_bounds), allocated at the first compaction
    u64 cmark_words = 0;
    // the single-pass partition (rl_part.hpp): per set, the tiles' runs and the hot buckets' work items
    u32* d_runs = nullptr;          // [PB_SETS][BKT_MAX * run_tt_max]
    u32 run_tt_max = TT_SMALL;      // TT_SMALL, or TT_LARGE for engines whose largest batch has more than TT_SMALL tiles
    HotItems* d_items = nullptr;    // [PB_SETS]
    u32 hot_wgs = 256;              // workgroups of k_bkt_apply that walk the hot work items (RL_HOT_WGS)
    size_t bk_stride = 0;           // records per set of d_bk_hits: max_batch rounded up to the largest tile
    int apply_trace = 0;            // RL_APPLY_TRACE=1: phase stamps of k_bkt_apply, one stderr line per collected batch (diagnostics)
    unsigned long long* d_apply_trace = nullptr;
    u32 part_steps_cfg = 0;         // RL_PART_STEPS (4, 8, 16): at least this many 64-hit steps per wave of k_bkt_part (experiments)
    u64 n_wait_parted = 0, n_wait_applied = 0, n_part_batches = 0;  // wait commands that had to be enqueued (RL_APPLY_TRACE: printed at destroy)
    bool last_k1_was_part = false;  // the batch submitted last went through k_bkt_part / k_bkt_apply (not k_bkt_tiny)
    // Events on the replay launches ("applied", for the partition stream).  RL_APPLY_EVENTS=0: the replay is a plain launch
    // and the partition stream takes its ordering from what the HOST has seen instead — the partition of batch p takes
    // over buffers last read by the replay of batch p - 3, whose completion word the caller has collected (its workgroups
    // are done reading), and everything it needs a replay kernel to have WRITTEN (its hot set, its zeroed scratch) comes
    // from batch p - 4, whose kernel had ended before the replay of batch p - 3 could start.  Measured: no difference
    // (50 us per step either way) — the ~11 us the replay's stream idles between two launches while the partition's queue
    // is busy (4.9 us when it is not, 1.7 us between plain launches of one queue) are not the events'.
    bool apply_events = true;
    int part_compact = 1;           // RL_PART_COMPACT=0: k_bkt_part (1024 threads, ~78 KB of LDS) for 4096-hit tiles too instead of
                                    // k_bkt_part_c (512 threads, ~41 KB: resident beside the replay's workgroups; 1.7 us per step).
                                    // (k_bkt_part_l, the compact shape with the per-hit work cut down — 26.7 us alone / 33.2 in the
                                    // pipeline, the STEP unchanged: profiles/r05b_lean_partition.md — is scripts/exp/patches/lean_partition.patch)
    bool fuse = false;              // one stream, the partition of batch j + 1 as a role of the launch that replays batch j: the default of
                                    // engines with max_batch_hits <= 256 k (RL_FUSE=0 / 1 overrides)
                                    // (k_bkt_step; parity-green, but the role's 4-wave workgroups walk a tile in 2 x 16 dependent steps:
                                    // 57 us alone against 24 us for k_bkt_part's 16 waves — measured slower, kept for the record)
    bool defer_apply = true;        // RL_DEFER_APPLY=0: enqueue k_bkt_apply at submit (see PendingApply)
    // k_bkt_apply of the batch submitted last, not yet enqueued.  Its partition was enqueued at submit; the apply waits
    // for the NEXT submit (or its own collect), by when the host usually sees the partition's event complete and the
    // apply stream is spared a wait command — a satisfied hipStreamWaitEvent still idles the stream ~9 us, the gap that
    // used to sit in front of every k_bkt_apply.
    struct PendingApply {
        bool valid = false;
        u64 p = 0;       // partitioned-batch number
        u32 slot = 0;    // inflight slot
        const Hit* d_hits = nullptr;
        u32 n = 0, nb = 0, ntiles = 0, tile_shift = 0, run_tt = 0, n_wg = 0;
        u64 now = 0;
        uint8_t* d_verdict = nullptr;
        int32_t* d_first = nullptr;
        bool t_apply = false;
        u32 hot_long = 0;
        hipEvent_t done_event = nullptr;  // the caller's, recorded behind the replay when it goes out (rl_check_and_update_submit_device_ev)
    } pend;
    // The replay of the batch BEFORE `pend`, held back across one more submit (the default since round 5: parity-green on the
    // pipeline suites and at full size, 49.1 -> 46.0 us per step over the driver's 20 steps, 45.5 -> 42.9 over 200, the replay
    // stream's idle time 9.8 -> 4.4 us, 111 -> 14 wait commands per 117 batches: gpurun_out/r13a, profiles/r05a_defer2.md;
    // RL_DEFER2=0 in experiment builds is the form before).  With today's kernel times the partition of batch p ends ~25 us after the submit of batch
    // p + 1 — while the replay of batch p - 1 is running and the host is spinning in its collect — so four replays out of five
    // go out behind a wait command (n_wait_parted), and a wait on an event that is not complete when it is enqueued costs the
    // stream 5.4 us at the boundary however long ago the event completed by then (scripts/microbench/kernel_gap2.hip:
    // 3.6 -> 9.0 us between two kernels).  Held back, the replay goes out from the collect's spin the moment the host sees its
    // partition's event complete — no wait command, still 25 us before the stream needs it.  Order among replays, the
    // partition -> replay dependency (the host has SEEN the event, or the wait command goes in as before) and everything
    // the in-flight limit guarantees (batch p - 3 collected before batch p is submitted) are unchanged.
    PendingApply pend_old;
    bool defer2 = true;
    hipEvent_t submit_done_event = nullptr;  // (argument of the submit being processed)
    hipEvent_t input_event = nullptr;        // rl_engine_wait_event on a two-stream engine: gates the next batch's inputs
    u32 pipe_depth = 3;             // RL_PIPE_DEPTH (2 or 3): the partition of batch p waits for k_bkt_apply of batch p - depth
    hipEvent_t ev_parted[4]{}, ev_applied[4]{};
    hipEvent_t ev_match = nullptr;  // behind the copies of the matcher's count pass (match_and_check_locked)
    u32 hot_threshold = HOT_PROMOTE;  // doubled while more keys qualify than there are hot buckets
    u32 hot_floor = HOT_PROMOTE;      // RL_HOT_PROMOTE: the floor of hot_threshold
    u32 hot_long_cfg = HOT_LONG_BUCKET;  // RL_HOT_LONG: a hash bucket of this many hits has its middling keys promoted too
    u32 hot_seen = 0;                 // keys that qualified for a hot bucket in the batch collected last
    bool hot_report = false;          // RL_HOT_REPORT=1: one stderr line per eighth batch (qualified keys, threshold)
    HotParam* d_hot_param = nullptr;  // [PB_SETS][HOT_MAX + 1]
    bool external_stream = false;  // the caller orders its own work on `stream`: routing helpers do not block
    u32 n_cus = 256;
    bool auto_grow = false;  // RL_CFG_AUTO_GROW
    bool gen_carry_req = true;     // RL_GEN_CARRY_REQ=0: k_gen_sort gathers every record's request through the record's index
    bool gen_load_deferred = true; // RL_GEN_LOAD_DEFERRED=0: k_gen_round stores remaining / expires_in in every round
    bool gen_pass_prefill = true;  // RL_GEN_PASS_PREFILL=0: k_gen_round stores every pass flag (the form before round 5)
    u32 gen_rounds_hint = 3; // fixpoint rounds the last general pass ran + 1: the length of the next pass's first blind group
    u32 gen_tiny_max = 64;   // general form: calls of up to this many hits take k_gen_tiny (RL_GEN_TINY_MAX=0 disables)
    u32 gen_seq = 0;
    bool h_tiny_coherent = false;  // h_status / h_tiny are fine-grained (hipHostMallocCoherent): device stores reach the host in order
    uint8_t* h_tiny = nullptr;  // host-mapped staging of a tiny host-buffer call: the kernel reads and writes it directly
    u32 tiny_max = TINY_MAX;  // batches up to this many hits take the one-launch path (RL_TINY_MAX=0 disables)
    BHit* d_bk_hits = nullptr;              // [PB_SETS][max_batch]
    u32* d_bk_req = nullptr;                // [bk_stride]: the request of every partitioned record (general resolver; passes are blocking: one set)
    BHit* d_tiny_hits = nullptr;            // k_bkt_tiny's record buffer
    unsigned short* d_chunk_tab = nullptr;  // [PB_SETS][...] hot chunk -> hot bucket (k_bkt_scatter -> k_bkt_apply)
    size_t chunk_tab_len = 0;
    int apply2_cfg = 0;     // RL_APPLY2_CFG: which instantiation of k_bkt_apply (see launch_apply)

    // on-device limit matching (rl_match.hpp)
    MatchLimit* d_match_limits = nullptr;
    MatchCond* d_match_conds = nullptr;
    u32* d_match_ns_off = nullptr;
    u32 n_match_limits = 0, n_match_ns = 0, n_match_conds = 0;
    // the slot form of the table (k_match_fast): few distinct keys, <= 64 limits per namespace, small enough for LDS
    bool match_fast = false;
    MatchLimitF* d_match_flimits = nullptr;
    MatchCondF* d_match_fconds = nullptr;
    MatchSlots match_slots{};
    u32 match_var_slots = 0;        // slots some limit of the fast table reads as a variable
    u64 wire_key_fp = 0;            // fingerprint of the hash key the table's hashed cells were named under (rl_wire_table_set; 0: none yet)
    u32 match_max_vars = 0;         // most variables of one limit of the table (> 2: only the hashed-key wire path derives its counters)
    u32 match_max_limit_id = 0;
    // the wire path without host dictionaries (rl_wire.hpp): tables of rl_wire_table_set, staging of a batch of messages
    bool wire_ready = false;
    WireTables wire_t{};
    uint8_t* d_w_blob = nullptr;
    WireStr* d_w_ns = nullptr;
    WireLit* d_w_lit = nullptr;
    u64* d_w_prefix = nullptr;
    uint8_t* d_w_bytes = nullptr;   // the messages of one batch, concatenated
    u64 w_bytes_cap = 0;
    // The OTHER serving sets' copies of the two (rl_wire_serve_batch_set, set s >= 1: [s - 1]), and what lets a serving call copy
    // its messages in BEFORE it takes the engine's mutex: a copy stream of its own (the transfer runs beside the other sets'
    // kernels instead of in front of them) and an event per set that the decide phase picks up.
    uint8_t* d_w_bytes_more[RL_SERVE_SETS - 1] = {};
    u64 w_bytes_cap_more[RL_SERVE_SETS - 1] = {};
    u32* d_w_off_more[RL_SERVE_SETS - 1] = {};
    hipStream_t in_stream = nullptr;
    hipEvent_t in_ev[RL_SERVE_SETS] = {};
    // RateLimitResponse bytes built on the device (rl_resp.hpp): what each limit contributes to X-RateLimit-Limit
    // (rl_resp_table_set), the responses' offsets and bytes of one batch
    uint8_t* d_resp_blob = nullptr;
    WireStr* d_resp_frag = nullptr;
    u32 n_resp_frag = 0;
    u32 resp_blob_len = 0;
    bool resp_ready = false;
    u32* d_resp_off = nullptr;      // [max_batch + 2]
    uint8_t* d_resp_bytes = nullptr;
    u64 resp_bytes_cap = 0;
    // RL_SERVE_ASYNC: the responses travel to the host in up to RESP_CHUNKS copies, an event behind each; rl_serve_wait(upto)
    // waits for the one that covers byte upto - 1 (the caller scatters the first responses while the last ones still travel)
    static constexpr u32 RESP_CHUNKS = 32;
    // Two SETS of everything a serving call leaves behind for the host to pick up (round 6): while the responses of the call
    // on set s are still being written to the host and handed on, the next call — on the other set — packs, copies in and
    // decides (include/rl_engine.h: rl_wire_serve_batch_set).
    static constexpr u32 SERVE_SETS = RL_SERVE_SETS;
    hipEvent_t resp_ev[SERVE_SETS][RESP_CHUNKS] = {};
    u64 resp_chunk_end[SERVE_SETS][RESP_CHUNKS] = {};  // the set's last serving call's pieces: piece c is complete when resp_ev[c] is,
    u32 resp_n_chunks[SERVE_SETS] = {};                // and ends at this byte of the responses (0 pieces: the call was synchronous)
    // The bytes' kernels (k_resp<true>: PCIe-bound, ~1 ms per 262 144 responses) run on a stream of their own, from a SNAPSHOT of
    // what they read (statuses, verdicts, the derived counters, remaining / expires_in, offsets): the engine's stream — and
    // the arrays the next call's matcher and resolver overwrite — are free the moment the snapshot is taken.
    hipStream_t resp_stream = nullptr;
    hipEvent_t snap_ev[SERVE_SETS] = {};
    void* resp_snap[SERVE_SETS] = {};
    u64 resp_snap_cap[SERVE_SETS] = {};
    // How the blind path's response bytes reach the pinned staging.  The tree decides per call (resp_via_copy = AUTO):
    //   one caller at a time: k_resp<true> writes the staging itself (0) — the shortest single call;
    //   callers in flight together (another set served a call within the last 20 ms): k_resp writes a device buffer and the pieces leave as copy commands issued LAZILY by
    //   whoever waits for them, one in flight per set (4) — a kernel that streams stores to the host stretches every short kernel of
    //   the next call's decide phase, and copy commands issued all at once make each of that phase's host round trips wait behind
    //   the whole transfer (profiles/r06_wire_two_in_flight.md): four calls in flight 1.7-2.1 ms per batch against 2.3-2.4.
    // RL_RESP_VIA_COPY (experiment) forces one form: 0, 4, or 1 = copy commands issued at once.  (Measured and parked,
    // scripts/exp/patches/resp_thin_copy_nocu.patch: copies of kind hipMemcpyDeviceToDeviceNoCU, a thin streaming copy kernel, the
    // staging's coherence flags.)
    static constexpr u32 RESP_AUTO = 0xFFu;
    u32 resp_via_copy = RESP_AUTO;
    u32 resp_lazy_depth = 1;        // RL_RESP_LAZY_DEPTH: pieces of a set in flight at once in form 4
    u64 serve_last_us[SERVE_SETS] = {};  // when the set last served a call through the blind path (steady clock)
    // RL_RESP_VIA_COPY=4: the copy commands of a set's pieces are issued LAZILY, by whoever waits for them (rl_serve_wait_set), one
    // piece in flight per set: what the next call's host round trips wait behind is then one piece, not the whole transfer
    struct LazyCopy {
        std::mutex mu;
        u32 n = 0;
        std::atomic<u32> issued{0};
        u64 lo[RESP_CHUNKS] = {}, hi[RESP_CHUNKS] = {};
        uint8_t* dst = nullptr;
        hipStream_t stream = nullptr;
        hipEvent_t built = nullptr;  // k_resp has written the set's device buffer
    } lazy[SERVE_SETS];
    uint8_t* d_resp_set[SERVE_SETS] = {};
    u64 d_resp_set_cap[SERVE_SETS] = {};
    u32 resp_pieces = 8;            // RL_RESP_PIECES
    u32 resp_writers = 128;         // RL_RESP_WRITERS: workgroups of k_resp<true> that write host memory at once
    bool resp_blind = true;         // RL_RESP_BLIND=0: the host reads the responses' total before their kernels go out
    u32 resp_max_frag = 0;          // longest fragment of rl_resp_table_set
    hipEvent_t resp_off_ev = nullptr;
    bool results_direct = true;     // RL_RESULTS_DIRECT=0: small result sets travel as copy commands too (ResultsOut)
    bool resp_direct = true;        // RL_RESP_DIRECT=0: k_resp writes a device buffer and copy commands carry it to the host
    u32* d_w_off = nullptr;         // [max_batch + 1]
    int32_t* d_w_status = nullptr;  // [max_batch]
    uint4* d_w_slot_h = nullptr;    // [max_batch][MATCH_SLOTS]: hashes of the values the variables read
    u32* d_hit_check = nullptr;     // [max_batch]: the check word of every derived counter (rl_keyhash.h)
    u32 collide_hit = 0;            // RL_ERR_KEY_COLLISION: a hit (index in the call) of the colliding pair
    void* h_stage[4 * RL_SERVE_SETS] = {};  // rl_host_staging: pinned buffers the engine lends to a host layer, by slot (4 per serving set)
    u64 h_stage_cap[4 * RL_SERVE_SETS] = {};
    unsigned long long* d_m_mask = nullptr;  // [max_batch] limits of its namespace that apply to a request
    u32* d_m_ns = nullptr;      // staging for host-pointer calls: per request namespace, delta
    u32* d_m_delta = nullptr;
    u32* d_m_ent_off = nullptr;
    u32* d_m_ent_key = nullptr;
    u32* d_m_ent_val = nullptr;
    u32* d_m_count = nullptr;
    int32_t* d_m_limited = nullptr;
    u32* h_m_total = nullptr;   // pinned: {n_hits, not_all_single}
    u32* d_m_flags = nullptr;
    void* d_m_scan_tmp = nullptr;
    size_t m_scan_tmp_bytes = 0;
    // the slot form without library scan / copies / marker (k_match_count2 -> k_match_scan2 -> k_match_fill2): the
    // workgroups' totals, the call counter, and the host-mapped 16 bytes {total, error bits, 0, call} the scan stores
    MatchScan* d_m_scan1 = nullptr;
    u32* h_m_word = nullptr;
    u32 m_call = 0;
    bool match_one = true;      // RL_MATCH_ONE=0: count pass + library scan + two copies + an event + fill pass
    // end of a general pass: {error bits, cells created, flags, sequence number} as one 16-byte store (k_gen_post)
    // k_gen_serve (rl_general.hpp): per-request calls without a launch per call
    ServeBox* h_serve = nullptr;  // host-mapped mailbox
    bool serve_enabled = true;    // RL_SERVE=0: every per-request call is a launch of its own
    bool serve_live = false;      // a server may be running (it leaves by itself after `serve_linger_us` without a request)
    u32 serve_linger_us = 200;    // RL_SERVE_LINGER_US
    u32 serve_timeout_ms = 60000; // RL_SERVE_TIMEOUT_MS
    u64 n_serve_calls = 0, n_serve_launches = 0;
    bool gen_clean = false;     // d_bs[0..BS_ROT) and d_gst are zero (the last stream command was a general pass's clean-up)
    u32* h_gen_word = nullptr;
    u32 gen_post_seq = 0;
    bool gen_post = true;       // RL_GEN_POST=0: two copy commands + a stream synchronise

    rl_stats_t stats{};

    int timing = 0;  // 0 off, 1 every kernel of the hot path, 2 k_bkt_apply only, 3 k_bkt_apply of every 4th batch
    // (reading a timed batch's events at the start of the NEXT collect instead of in its own, RL_TIMING_LAZY: moved nothing, +-0.3 us:
    // profiles/r05a_defer2.md 2 — scripts/exp/patches/timing_lazy.patch)
    hipEvent_t ev[8]{};
    double ms_slot[RL_TIMING_SLOTS]{};
    u64 timed_launches = 0;
};

namespace {

inline bool engine_busy(const rl_engine* e) { return e->sub_seq != e->col_seq || e->ph_open; }
// The read-only entry points (is_within_limits, get_counters: the reference serves them under the read lock
// check_and_update holds too, in_memory.rs:20-35,78,159-187) only refuse while a phased pass is open: with batches in
// flight they are enqueued behind every batch submitted so far and wait for THAT, the batches stay in flight.
inline bool engine_busy_for_reads(const rl_engine* e) { return e->ph_open; }


// layout of rl_engine::h_tiny (host-mapped staging of a one-launch host-buffer call)
constexpr size_t TIO_HITS = 1024;
constexpr size_t TIO_OFF_HITS = 0, TIO_OFF_REQ = 16384, TIO_OFF_VERDICT = 24576, TIO_OFF_FIRST = 25600,
                 TIO_OFF_REM = 29696, TIO_OFF_EXP = 37888, TIO_OFF_DELTA = 46080, TIO_BYTES = 54272;

int fail(rl_engine* e, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (e) e->err = buf;
    return code;
}

#define HIP_TRY(e, call)                                                                  \
    do {                                                                                  \
        hipError_t _r = (call);                                                           \
        if (_r != hipSuccess)                                                             \
            return fail((e), RL_ERR_DEVICE, "%s failed: %s (%s:%d)", #call,               \
                        hipGetErrorString(_r), __FILE__, __LINE__);                       \
    } while (0)

inline u32 cdiv(u64 a, u64 b) { return (u32)((a + b - 1) / b); }

// An engine that holds PEER state (rl_merge_cells has created a per-actor table): its cells are CrCounterValues, not plain
// AtomicExpiringValues — a window restarted by a local update keeps the peers' part of the window that ended
// (cr_counter_value.rs:53-59) — and every check_and_update / update_counter of it goes through the general resolver, whose
// per-cell resolve and commit know about the peer tables (rl_general.hpp).  The single-counter hot path (k_bkt_*), the
// one-launch kernels and the lingering server do not, and are not taken: the CRDT rule is exact, off the hot path.
inline bool engine_has_peers(const rl_engine* e) {
    for (const Cell* t : e->peer_tables)
        if (t) return true;
    return false;
}

u32 ceil_log2(u64 x) {
    u32 l = 0;
    while ((1ull << l) < x) ++l;
    return l;
}

int status_to_error(rl_engine* e, u32 bits) {
    if (bits & ERRBIT_BAD_LIMIT) return fail(e, RL_ERR_INVALID, "hit references a limit id outside the limit table");
    if (bits & ERRBIT_RESERVED_KEY) return fail(e, RL_ERR_INVALID, "key 0xFFFFFFFFFFFFFFFE/F is reserved");
    if (bits & ERRBIT_MISSING_SIMPLE)
        return fail(e, RL_ERR_MISSING_SIMPLE,
                    "simple counter without a pre-created cell (reference: in_memory.rs:107 unwrap panics)");
    if (bits & ERRBIT_TABLE_FULL) return fail(e, RL_ERR_TABLE_FULL, "counter table is full");
    // (hashed keys carry the limit IN the key: two limit ids under one key are two counters that share it, and the check
    // words — which k_gen_check_keys compares whatever k_gen_sort said about the limit ids — name the hits)
    if (bits & ERRBIT_KEY_COLLISION)
        return fail(e, RL_ERR_KEY_COLLISION, "two counters share a 64-bit key (their check words differ): nothing was applied");
    if (bits & ERRBIT_KEY_LIMIT) return fail(e, RL_ERR_KEY_LIMIT, "a key was used with two different limit ids");
    return fail(e, RL_ERR_DEVICE, "unknown device status 0x%x", bits);
}

int read_status(rl_engine* e) {
    HIP_TRY(e, hipMemcpyAsync(e->h_status, e->d_status, sizeof(Status), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return RL_OK;
}

int alloc_table(rl_engine* e, u64 cap, Cell** out) {
    Cell* t = nullptr;
    hipError_t r = hipMalloc((void**)&t, cap * sizeof(Cell));
    if (r != hipSuccess) return fail(e, RL_ERR_NOMEM, "hipMalloc of %llu-cell table failed: %s",
                                     (unsigned long long)cap, hipGetErrorString(r));
    k_table_init<<<2048, 256, 0, e->stream>>>(t, cap);
    HIP_TRY(e, hipGetLastError());
    *out = t;
    return RL_OK;
}

int do_compact(rl_engine* e, u32 new_log2cap);

// With RL_CFG_AUTO_GROW: double the table (rehash) instead of refusing — possible only while no batch
// is in flight.  -> true if the table was grown and the caller should look again.
bool grow_instead(rl_engine* e, int* rc) {
    if (!e->auto_grow || e->sub_seq != e->col_seq || e->log2cap >= 31) return false;
    *rc = do_compact(e, e->log2cap + 1);
    return *rc == RL_OK;
}

int check_room(rl_engine* e, u64 incoming, bool* need_count = nullptr) {
    // All-or-nothing for every mutating call: the kernels commit as they go, so a call must never run out
    // of table half-way.  A call is accepted only if
    //   (live + tombstones) <= 3/4 capacity                  (linear probing stays short), and
    //   (live + tombstones) + new keys <= 15/16 capacity     (every probe ends at an empty slot),
    // so RL_ERR_TABLE_FULL is only ever answered HERE or by the caller's exact count, before anything is
    // applied.  "new keys" is first bounded by `incoming` (every hit / row could bring a new key); when that
    // bound does not fit, a caller that can count exactly passes `need_count` and runs k_bkt_count_new on the
    // partitioned batch (the single-counter path), everything else is refused on the bound.  With
    // RL_CFG_AUTO_GROW the table is doubled as soon as used + incoming would pass 3/4 (only possible while
    // no batch is in flight).
    if (need_count) *need_count = false;
    for (int rc = RL_OK;;) {
        const u64 used = e->live + e->tombs;
        const bool low = used <= e->cap - e->cap / 4;
        const bool fits = low && used + incoming <= e->cap - e->cap / 16;
        const bool grow_now = e->auto_grow && used + incoming > e->cap - e->cap / 4;
        if (fits && !grow_now) return RL_OK;
        if (grow_instead(e, &rc)) continue;
        if (rc) return rc;
        if (fits) return RL_OK;  // could not grow right now (a batch is in flight): the plain bound still holds
        if (low && need_count) {
            *need_count = true;
            return RL_OK;
        }
        return fail(e, RL_ERR_TABLE_FULL,
                    "refused, nothing applied: %llu incoming hits could push the table past its occupancy bound "
                    "(live=%llu tombstones=%llu capacity=%llu): rl_resize, sweep, compact or create a larger engine",
                    (unsigned long long)incoming, (unsigned long long)e->live, (unsigned long long)e->tombs,
                    (unsigned long long)e->cap);
    }
}

// Rehash the live cells into a fresh table of 2^new_log2cap cells (0: same size — a compaction).  The peer tables
// (rl_merge_cells) share the main table's geometry: EVERY new table is allocated and filled first and the pointers, the
// capacity and the counters are committed together — a failure half-way frees the new ones and leaves the engine as it was.
int do_compact(rl_engine* e, u32 new_log2cap) {
    if (!new_log2cap) new_log2cap = e->log2cap;
    if (new_log2cap == e->log2cap && e->log2cap >= 6) {
        // Same geometry: IN PLACE (k_compact_bounds / k_compact_seg, rl_kernels.hpp) — no second table, nothing that can
        // fail half-way; the peer tables (same geometry) the same way, behind the main table.
        const u32 n_seg = (u32)std::max<u64>(e->cap >> CSEG_LOG2, 1);
        if (e->cmark_words < n_seg) {
            if (e->d_cmark) (void)hipFree(e->d_cmark);
            e->d_cmark = nullptr;
            e->cmark_words = 0;
            if (hipMalloc((void**)&e->d_cmark, (size_t)n_seg * sizeof(u64)) != hipSuccess)
                return fail(e, RL_ERR_NOMEM, "hipMalloc of the compaction's %u segment bounds failed", n_seg);
            e->cmark_words = n_seg;
        }
        HIP_TRY(e, hipMemsetAsync(e->d_status, 0, sizeof(Status), e->stream));
        k_compact_bounds<<<(n_seg + 255) / 256, 256, 0, e->stream>>>(e->table, e->cap, e->d_cmark, n_seg);
        k_compact_seg<<<(n_seg + 3) / 4, 256, 0, e->stream>>>(e->table, e->log2cap, e->seed, e->d_cmark, n_seg, e->d_status, 1u);
        for (u32 a = 0; a < (u32)MERGE_MAX_ACTORS; ++a)
            if (e->peer_tables[a]) {
                k_compact_bounds<<<(n_seg + 255) / 256, 256, 0, e->stream>>>(e->peer_tables[a], e->cap, e->d_cmark, n_seg);
                k_compact_seg<<<(n_seg + 3) / 4, 256, 0, e->stream>>>(e->peer_tables[a], e->log2cap, e->seed, e->d_cmark, n_seg, e->d_status, 0u);
            }
        HIP_TRY(e, hipGetLastError());
        int rc = read_status(e);
        if (rc) return rc;
        e->live = e->h_status->n_inserted;
        e->tombs = 0;
        e->stats.rebuilds++;
        return RL_OK;
    }
    Cell* fresh = nullptr;
    Cell* fresh_peer[MERGE_MAX_ACTORS] = {};
    auto undo = [&]() {
        if (fresh) (void)hipFree(fresh);
        for (auto& pf : fresh_peer)
            if (pf) (void)hipFree(pf);
    };
    int rc = alloc_table(e, 1ull << new_log2cap, &fresh);
    for (u32 a = 0; a < (u32)MERGE_MAX_ACTORS && !rc; ++a)
        if (e->peer_tables[a]) rc = alloc_table(e, 1ull << new_log2cap, &fresh_peer[a]);
    if (rc) {
        undo();
        return rc;
    }
    hipError_t r = hipMemsetAsync(e->d_status, 0, sizeof(Status), e->stream);
    if (r == hipSuccess) {
        // (peer entries of windows that are over are not carried along: k_rehash copies live cells; a peer cell is "live"
        // like any other, its expiry is checked where it is read)
        for (u32 a = 0; a < (u32)MERGE_MAX_ACTORS; ++a)
            if (e->peer_tables[a])
                k_rehash<<<2048, 256, 0, e->stream>>>(e->peer_tables[a], e->cap, fresh_peer[a], new_log2cap, e->seed, e->d_status, 0u);
        // (one status block for all of them: a peer table's cell that found no room raises the same error bit as the main
        // table's and refuses the whole resize; n_inserted counts the MAIN table's cells only)
    }
    if (r == hipSuccess) {
        k_rehash<<<2048, 256, 0, e->stream>>>(e->table, e->cap, fresh, new_log2cap, e->seed, e->d_status, 1u);
        r = hipGetLastError();
    }
    if (r != hipSuccess) {
        (void)hipStreamSynchronize(e->stream);
        undo();
        return fail(e, RL_ERR_DEVICE, "rehash failed: %s", hipGetErrorString(r));
    }
    rc = read_status(e);
    if (!rc && e->h_status->err) rc = status_to_error(e, e->h_status->err);  // (a live cell found no room in a smaller table)
    if (rc) {
        undo();
        return rc;
    }
    (void)hipFree(e->table);
    e->table = fresh;
    for (u32 a = 0; a < (u32)MERGE_MAX_ACTORS; ++a)
        if (e->peer_tables[a]) {
            (void)hipFree(e->peer_tables[a]);
            e->peer_tables[a] = fresh_peer[a];
        }
    e->log2cap = new_log2cap;
    e->cap = 1ull << new_log2cap;
    e->stats.capacity_cells = e->cap;
    e->live = e->h_status->n_inserted;
    e->tombs = 0;
    e->stats.rebuilds++;
    return RL_OK;
}


// Spin until the batch's last workgroup has stored its sequence number (see apply_finish).
int poll_pending_apply(rl_engine* e);
int read_timing(rl_engine* e, rl_engine::Inflight& f);

// `poll` (RL_DEFER2): while the host waits, replays that are held back go out as soon as their partitions are seen complete.
int wait_done(rl_engine* e, rl_engine::Inflight& f, bool poll = false) {
    const volatile u32* done = &f.h_st->n_removed;
    const auto t_start = std::chrono::steady_clock::now();
    for (u64 spins = 0; __atomic_load_n(done, __ATOMIC_ACQUIRE) != f.seq; ++spins) {
        __builtin_ia32_pause();
        if (poll && (spins & 63u) == 63u && (e->pend_old.valid || e->pend.valid)) {
            const int prc = poll_pending_apply(e);
            if (prc) return prc;
        }
        if ((spins & 0xFFFFu) == 0xFFFFu) {
            if (std::chrono::steady_clock::now() - t_start > std::chrono::seconds(60))
                return fail(e, RL_ERR_DEVICE, "batch %u did not complete within 60 s", f.seq);
            std::this_thread::yield();
        }
    }
    return RL_OK;
}

// Poll a host-mapped word a kernel stores last (the 16-byte block it is part of is ONE store).
int wait_word(rl_engine* e, const u32* word, u32 seq, const char* what) {
    const auto t_start = std::chrono::steady_clock::now();
    for (u64 spins = 0; __atomic_load_n(word, __ATOMIC_ACQUIRE) != seq; ++spins) {
        __builtin_ia32_pause();
        if ((spins & 0xFFFFu) == 0xFFFFu) {
            if (std::chrono::steady_clock::now() - t_start > std::chrono::seconds(60))
                return fail(e, RL_ERR_DEVICE, "%s did not complete within 60 s", what);
            std::this_thread::yield();
        }
    }
    return RL_OK;
}

// Tell a lingering k_gen_serve to leave and wait until it has: from here on the stream and the table are the caller's.
// (The command carries the next sequence number, so the answer is "gone, waiting for exactly that one" whether the server
// saw the command or had just left by itself.)
void serve_stop(rl_engine* e) {
    if (!e->serve_live) return;
    ServeBox* b = e->h_serve;
    const u32 seq = ++e->gen_seq ? e->gen_seq : ++e->gen_seq;
    b->cmd[1] = 0;
    b->cmd[2] = SRV_QUIT;
    __atomic_store_n(&b->cmd[0], seq, __ATOMIC_RELEASE);
    (void)wait_word(e, &b->gone[3], seq, "k_gen_serve (asked to leave)");
    e->serve_live = false;
}

// The engine's mutex, and the device to ourselves: every entry point but the per-request host-buffer call sends a
// lingering server away before it touches the stream or the table.
struct EngineLock {
    std::lock_guard<std::mutex> g;
    explicit EngineLock(rl_engine* e, bool keep_server = false) : g(e->mu) {
        if (!keep_server) serve_stop(e);
    }
};

// the partition kernels' wave-private counters (dynamic LDS): PT_WAVES x (hash buckets + hot buckets) x 2 bytes
inline u32 scatter_lds_bytes(u32 nbt) { return (u32)PT_WAVES * nbt * (u32)sizeof(unsigned short); }

// A launch that is timed carries its own start / stop events (hipExtLaunchKernelGGL: the events get the dispatch's
// begin and end timestamps), so a timed kernel has no marker commands around it — two hipEventRecord markers added
// 4-6 us to the interval and ~5 us of idle device each (the figure then disagreed with rocprofv3's by that much).
#define RL_LAUNCH_TS(timed, ev0, ev1, kern, grid, block, lds, stream, ...)                                         \
    do {                                                                                                           \
        if (timed) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, stream, ev0, ev1, 0, __VA_ARGS__);    \
        else kern<<<dim3(grid), dim3(block), lds, stream>>>(__VA_ARGS__);                                          \
    } while (0)
#define RL_LAUNCH_T(timed, ev0, ev1, kern, grid, block, stream, ...) \
    RL_LAUNCH_TS(timed, ev0, ev1, kern, grid, block, 0, stream, __VA_ARGS__)

// k_bkt_step in the instantiation RL_APPLY2_CFG selects (see RL_DEF_STEP).
void launch_step(rl_engine* e, hipStream_t rs, const StepParams& S, bool timed, hipEvent_t ev0, hipEvent_t ev1) {
    const u32 n_wg = S.n_apply_wgs + S.n_part_wgs;
#define RL_ST(KERN)                                                                                                  \
    RL_LAUNCH_TS(timed, ev0, ev1, KERN, n_wg, AP_BLOCK, std::max(sizeof(KERN##_lds), sizeof(PartLds)), rs, S)
    if (S.A.run_tt == (u32)TT_LARGE) {  // the largest batches (more than TT_SMALL tiles): a bucket view of 1024 tiles
        RL_ST(k_bkt_step_large);
        return;
    }
    // 0 (default): 80 VGPRs (no scratch), limit ids in 16 bits — engines with more than 32768 limit rows take 1: the
    // same kernel with 32-bit limit ids.  2: 64 VGPRs (eleven of them in scratch: 1.5-2 us slower per 1 M-hit step,
    // measured in both halves of this round), 3: 96.
    switch (e->apply2_cfg == 0 && e->max_limits > 32768u ? 1 : e->apply2_cfg) {
        default:
        case 0: RL_ST(k_bkt_step); break;
        case 1: RL_ST(k_bkt_step_wide); break;
        case 2: RL_ST(k_bkt_step_v64); break;
        case 3: RL_ST(k_bkt_step_v96); break;
    }
#undef RL_ST
}

// The partition half of a k_bkt_step launch (submit_k1_bucketed).
struct PartLaunch {
    PartParams Q;
    u32 slot;    // inflight slot of the batch being partitioned
    bool timed;  // events around a launch that only partitions
};

// One k_bkt_step launch: the replay that submit_k1_bucketed left pending (rl_engine::PendingApply), if any, and —
// `part`, engines in the fused mode — the partition of the batch being submitted.
int flush_one(rl_engine* e, rl_engine::PendingApply& q, const PartLaunch* part) {
    if (!q.valid && !part) return RL_OK;
    StepParams S{};
    bool timed = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const bool two_streams = e->pstream != e->stream;
    rl_engine::Inflight* fa = nullptr;
    u64 p = 0;
    hipStream_t rs = e->stream;
    if (q.valid) {
        q.valid = false;
        rl_engine::Inflight& f = e->inflight[q.slot];
        fa = &f;
        p = q.p;
        if (e->xover && (p & 1u)) rs = e->stream2;
        const u32 par = (u32)(p % PB_SETS);
        if (two_streams && hipEventQuery(e->ev_parted[p & 3u]) != hipSuccess) {
            (void)hipGetLastError();  // (hipErrorNotReady is not an error here)
            HIP_TRY(e, hipStreamWaitEvent(rs, e->ev_parted[p & 3u], 0));
            e->n_wait_parted++;
        }
        ApplyParams& P = S.A;
        P.table = e->table;
        P.log2cap = e->log2cap;
        P.seed = e->seed;
        P.b_hits = e->d_bk_hits + (size_t)par * e->bk_stride;
        P.hits = q.d_hits;
        P.runs = e->d_runs + (size_t)par * BKT_MAX * e->run_tt_max;
        P.run_tt = q.run_tt;
        P.ntiles = q.ntiles;
        P.tile_shift = q.tile_shift;
        P.nb = q.nb;
        P.items = e->d_items + par;
        P.limits = e->d_limits;
        P.now = q.now;
        P.verdict = q.d_verdict;
        P.first_limited = q.d_first;
        P.bs = e->d_bs + p % BS_ROT;
        P.bs_zero = e->d_bs + (p + 4) % BS_ROT;
        P.host_status = f.h_st;
        P.done_seq = f.seq;
        P.hot_next = e->d_hot + p % HS_SETS;
        P.hot_threshold = e->hot_threshold;
        P.hot_arrive = e->d_hot_arrive + (size_t)(p & 1u) * HOT_MAX;
        P.sparse_out = 1u;
        P.hot_long = q.hot_long;
        if (e->apply_trace) {
            // one stamp buffer per batch in flight (indexed like e->inflight): the stamps a collect reads are its own batch's
            constexpr size_t TR = (size_t)(BK_MAX + 1024) * 8;
            if (!e->d_apply_trace) HIP_TRY(e, hipMalloc((void**)&e->d_apply_trace, 4 * TR * sizeof(unsigned long long)));
            HIP_TRY(e, hipMemsetAsync(e->d_apply_trace + (size_t)q.slot * TR, 0, TR * sizeof(unsigned long long), e->stream));
            P.trace = e->d_apply_trace + (size_t)q.slot * TR;
        }
        S.n_apply_wgs = q.n_wg;
        // "applied" for the partition stream (two-stream engines): the stop event of the launch itself where possible
        // (no marker command); a timed launch only adds a start event
        const bool chain_a = two_streams && e->ext_events && e->apply_events;
        f.ev_a_stop = chain_a ? e->ev_applied[p & 3u] : (q.t_apply ? f.tev[5] : nullptr);
        timed = q.t_apply || chain_a;
        ev0 = q.t_apply ? f.tev[4] : nullptr;
        ev1 = f.ev_a_stop;
    }
    if (part) {
        S.Q = part->Q;
        S.n_part_wgs = part->Q.ntiles + 1;
        if (!fa && part->timed) {  // a launch that only partitions, timed as such
            rl_engine::Inflight& fp = e->inflight[part->slot];
            timed = true;
            ev0 = fp.tev[0];
            ev1 = fp.ev_p_stop = fp.tev[1];
        }
    }
    const hipEvent_t done_event = fa ? q.done_event : nullptr;
    q.done_event = nullptr;
    launch_step(e, rs, S, timed, ev0, ev1);
    HIP_TRY(e, hipGetLastError());
    if (fa && two_streams && e->apply_events && !e->ext_events) HIP_TRY(e, hipEventRecord(e->ev_applied[p & 3u], rs));
    if (done_event) HIP_TRY(e, hipEventRecord(done_event, rs));
    return RL_OK;
}

// Everything held back goes out now, oldest first (and, fused engines, the partition of the batch being submitted).
int flush_pending_apply(rl_engine* e, const PartLaunch* part = nullptr) {
    if (e->pend_old.valid) {
        const int rc = flush_one(e, e->pend_old, nullptr);
        if (rc) return rc;
    }
    return flush_one(e, e->pend, part);
}

// Replays that are held back go out, oldest first, as far as the host sees their partitions complete: no wait command.
int poll_pending_apply(rl_engine* e) {
    if (e->pstream == e->stream) return RL_OK;
    if (e->pend_old.valid) {
        if (hipEventQuery(e->ev_parted[e->pend_old.p & 3u]) != hipSuccess) {
            (void)hipGetLastError();  // (hipErrorNotReady is not an error here)
            return RL_OK;
        }
        const int rc = flush_one(e, e->pend_old, nullptr);
        if (rc) return rc;
    }
    if (e->pend.valid) {
        if (hipEventQuery(e->ev_parted[e->pend.p & 3u]) != hipSuccess) {
            (void)hipGetLastError();
            return RL_OK;
        }
        return flush_one(e, e->pend, nullptr);
    }
    return RL_OK;
}

// Wait for every batch in flight and add the cells they created to `live` now (their status is still
// handed out by rl_check_and_update_collect): used when a new batch needs the exact occupancy.
int settle_inflight(rl_engine* e) {
    int rc = flush_pending_apply(e);
    if (rc) return rc;
    for (u64 q = e->col_seq; q < e->sub_seq; ++q) {
        rl_engine::Inflight& f = e->inflight[q & 3u];
        if (f.settled) continue;
        rc = wait_done(e, f);
        if (rc) return rc;
        if (f.kind == 1) {
            e->live -= f.h_st->n_ord;
            e->tombs += f.h_st->n_ord;
        } else {
            e->live += f.h_st->n_inserted;
        }
        e->inflight_hits -= f.n;
        f.settled = true;
    }
    return RL_OK;
}

// Enqueue one batch of the bucketed path (no host synchronisation): the partition on `pstream`, k_bkt_apply on
// `stream`, which therefore overlaps the partition of the NEXT batches.
//   partitioned batch p:  buffers [p % PB_SETS] (records, runs, hot work items, chunk table); scratch [p % BS_ROT],
//   zeroes scratch [(p + 3) % BS_ROT]; partitioned with the hot set of batch p - depth, picks hot set [p % HS_SETS].
//   stream order:  pstream: (wait applied(p - depth)) | k_bkt_part | parted(p)
//                  stream:  (wait parted(p)) | k_bkt_apply | applied(p)
// depth = pipe_depth.  With 3 the batch the partition waits for has been collected by the caller already (at most
// three batches are in flight), so the wait never holds the partition stream back and the partitions run ahead of the
// apply stream.  Both waits are answered by the HOST where it can see the event complete (hipEventQuery): a wait command
// costs the stream ~9 us even when its event completed long ago.  For "parted" that works because k_bkt_apply of batch p
// is only enqueued when batch p + 1 is submitted (or batch p collected): PendingApply.
int submit_k1_bucketed(rl_engine* e, const Hit* d_hits, u32 n, u64 now, uint8_t* d_verdict, int32_t* d_first) {
    if (e->sub_seq - e->col_seq >= 3) return fail(e, RL_ERR_BUSY, "three batches are already in flight: collect one first");
    bool need_count = false;
    int rc = check_room(e, n + e->inflight_hits, &need_count);
    if (rc) return rc;
    if (need_count && e->sub_seq != e->col_seq) {
        // the exact count needs the exact occupancy: let the batches in flight finish first
        rc = settle_inflight(e);
        if (!rc) rc = check_room(e, n + e->inflight_hits, &need_count);
        if (rc) return rc;
    }
    e->gen_clean = false;  // (this batch's scratch block is not the general resolver's to find clean)
    rl_engine::Inflight& f = e->inflight[e->sub_seq & 3u];
    // timing: 1 both kernels of every batch, 2 k_bkt_apply of every batch, 3 both kernels of every fourth batch
    const bool t_apply = e->timing == 1 || e->timing == 2 || (e->timing == 3 && (e->sub_seq & 3u) == 0);
    const bool t = e->timing == 1 || (e->timing == 3 && (e->sub_seq & 3u) == 0);
    const bool two_streams = e->pstream != e->stream;
    if (n <= e->tiny_max && !need_count) {
        // one launch: the batch is one bucket (k_bkt_tiny), on the apply stream (behind the k_bkt_apply of every batch
        // submitted before it), with a scratch and a record buffer of its own (a partitioned batch's partition may be
        // running beside it); the hot sets are left untouched
        rc = flush_pending_apply(e);
        if (rc) return rc;
        if (e->input_event) {
            HIP_TRY(e, hipStreamWaitEvent(e->stream, e->input_event, 0));
            e->input_event = nullptr;
        }
        {
            const LimitDev* limits = e->d_limits;
            RL_LAUNCH_T(t_apply, f.tev[4], f.tev[5], k_bkt_tiny, 1, AP_BLOCK, e->stream, e->table, e->log2cap, e->seed, d_hits, n,
                        e->d_tiny_hits, limits, (u32)e->h_limits.size(), now, d_verdict, d_first, e->d_bs + BS_ROT,
                        e->d_bs + BS_ROT, f.h_st, (u32)(e->sub_seq + 1), (u32)HOT_MAX / 2);
        }
        HIP_TRY(e, hipGetLastError());
        if (e->submit_done_event) HIP_TRY(e, hipEventRecord(e->submit_done_event, e->stream));
        f.kind = 0;
        f.n = n;
        f.n_wg = 1;
        f.ntiles = 0;
        f.timed = t_apply ? 2 : 0;
        f.ev_a_stop = f.tev[5];
        f.ev_a_prev = nullptr;
        e->last_k1_was_part = false;
        f.seq = (u32)(e->sub_seq + 1);
        f.settled = false;
        e->inflight_hits += n;
        e->sub_seq++;
        return RL_OK;
    }
    u32 bk_log2 = ceil_log2(cdiv(n, 384));
    const u32 bk_cap = n > (2u << 20) ? (u32)BK_LOG2_MAX : e->bk_log2_cfg;
    if (bk_log2 > bk_cap) bk_log2 = bk_cap;
    const u32 nb = 1u << bk_log2;
    // Hits per tile: 1024 for small batches (so that the partition still spreads over the CUs), else the smallest of
    // 4096 / 8192 / 16384 that keeps the batch within TT_SMALL tiles (a bucket view of 256 tiles); the largest batches
    // take 16384-hit tiles and the TT_LARGE view.
    u32 steps = 1;
    if (!(cdiv(n, PT_TILE_SMALL) <= PT_SMALL_MAX_TILES))
        for (steps = PT_STEPS; steps < PT_STEPS_MAX && cdiv(n, PT_BLOCK * steps) > (u32)TT_SMALL; steps *= 2) {}
    if (e->part_steps_cfg > steps && n > (u32)PT_BLOCK * e->part_steps_cfg) steps = e->part_steps_cfg;  // (experiments)
    const u32 tile_shift = 10u + (steps == 1 ? 0u : steps == 4 ? 2u : steps == 8 ? 3u : 4u);
    const u32 ntiles = cdiv(n, 1u << tile_shift);
    const u32 run_tt = ntiles <= (u32)TT_SMALL ? (u32)TT_SMALL : (u32)TT_LARGE;
    const u32 nbt = nb + HOT_MAX;
    const u64 p = e->part_seq;
    const u32 par = (u32)(p % PB_SETS);
    const u32 depth = e->pipe_depth;
    BatchScratch* bs = e->d_bs + p % BS_ROT;
    // (one-stream engines partition batch p inside the launch that replays batch p - 1: the newest complete hot set is
    // the one batch p - 2 picked)
    const HotSet* hot_use = e->d_hot + (p + HS_SETS - (two_streams ? (e->apply_events ? depth : 4u) : 2u)) % HS_SETS;
    HotSet* hot_prod = e->d_hot + p % HS_SETS;
    BHit* b_hits = e->d_bk_hits + (size_t)par * e->bk_stride;
    u32* runs = e->d_runs + (size_t)par * BKT_MAX * e->run_tt_max;
    HotItems* items = e->d_items + par;
    hipStream_t ps = e->pstream;
    if (e->input_event) {  // (rl_engine_wait_event: this batch's inputs)
        HIP_TRY(e, hipStreamWaitEvent(ps, e->input_event, 0));
        e->input_event = nullptr;
    }
    const Cell* ctable = e->table;
    const LimitDev* climits = e->d_limits;
    // ---- one stream: the partition of this batch rides in the launch that replays the batch before it (k_bkt_step) ----
    const bool fused = e->fuse && !two_streams && nbt <= (u32)PART_ROLE_BINS && !need_count;
    if (fused) {
        PartLaunch PL{};
        PartParams& Q = PL.Q;
        Q.table = ctable;
        Q.log2cap = e->log2cap;
        Q.seed = e->seed;
        Q.hits = d_hits;
        Q.n = n;
        Q.limits = climits;
        Q.n_limits = (u32)e->h_limits.size();
        Q.bk_log2 = bk_log2;
        Q.ntiles = ntiles;
        Q.run_tt = run_tt;
        Q.tile_shift = tile_shift;
        Q.check_simple = 1u;
        Q.bs = bs;
        Q.hot = hot_use;
        Q.verdict_fill = d_verdict;
        Q.first_fill = d_first;
        Q.b_hits = b_hits;
        Q.runs = runs;
        Q.items = items;
        Q.hot_next = hot_prod;
        PL.slot = (u32)(e->sub_seq & 3u);
        PL.timed = t;
        f.ev_p_stop = nullptr;
        rc = flush_pending_apply(e, &PL);
        if (rc) return rc;
    } else {
    // ---- partition ----------------------------------------------------------------------------------
    // The batch whose buffers and hot set this partition takes over (p - depth) has, with depth 3, been collected by
    // the caller: the host asks the event itself — once it has seen it complete, everything enqueued from here on is
    // ordered behind that batch — and the partition stream is spared a wait command.  Only a batch that is really
    // still running gets the device-side wait.  (A batch whose k_bkt_apply is still pending is p - 1: never p - depth.)
    if (two_streams && e->apply_events && p >= depth && hipEventQuery(e->ev_applied[(p - depth) & 3u]) != hipSuccess) {
        (void)hipGetLastError();  // (hipErrorNotReady is not an error here)
        HIP_TRY(e, hipStreamWaitEvent(ps, e->ev_applied[(p - depth) & 3u], 0));
        e->n_wait_applied++;
    }
    // "partitioned" for the other stream: the stop event of the launch itself where possible (no marker command); a
    // timed launch only adds a start event
    const bool chain_p = two_streams && e->ext_events && !need_count;
    f.ev_p_stop = chain_p ? e->ev_parted[p & 3u] : (t ? f.tev[1] : nullptr);
#define RL_PART(STEPS)                                                                                                         \
    RL_LAUNCH_TS(t || chain_p, t ? f.tev[0] : nullptr, f.ev_p_stop, k_bkt_part<STEPS>, ntiles + 1,                            \
                 PT_BLOCK, scatter_lds_bytes(nbt), ps, ctable, e->log2cap, e->seed, d_hits, n, climits,                        \
                 (u32)e->h_limits.size(), bk_log2, ntiles, run_tt, bs, hot_use, 1u, d_verdict, d_first, b_hits, runs, items,   \
                 hot_prod)
#define RL_PART_C(STEPS)                                                                                                       \
    RL_LAUNCH_TS(t || chain_p, t ? f.tev[0] : nullptr, f.ev_p_stop, k_bkt_part_c<STEPS>, ntiles + 1, PC_BLOCK,                 \
                 (u32)PC_WAVES * nbt * (u32)sizeof(unsigned short) + 4u, ps, ctable, e->log2cap, e->seed, d_hits, n, climits,  \
                 (u32)e->h_limits.size(), bk_log2, ntiles, run_tt, bs, hot_use, 1u, d_verdict, d_first, b_hits, runs, items,   \
                 hot_prod)
    if (e->part_compact && steps == 4) RL_PART_C(8);  // (1024-hit tiles of small batches: the 1024-thread kernel is 1 us faster)
    else
    switch (steps) {
        case 1: RL_PART(1); break;
        case 4: RL_PART(4); break;
        case 8: RL_PART(8); break;
        default: RL_PART(16); break;
    }
#undef RL_PART_C
#undef RL_PART
    HIP_TRY(e, hipGetLastError());
    if (need_count) {
        // the cheap bound (every hit a new key) does not fit: count the batch's new keys exactly, before
        // anything is applied, and refuse the whole batch if they do not fit (nothing is in flight here)
        HIP_TRY(e, hipMemsetAsync(e->d_m_flags, 0, sizeof(u32), ps));
        if (run_tt == (u32)TT_SMALL)
            k_bkt_count_new<10, TT_SMALL><<<nb < 64u ? 64u : nb, AP_BLOCK, 0, ps>>>(e->table, e->log2cap, e->seed, b_hits, runs, run_tt,
                                                                                    ntiles, tile_shift, nb, e->d_m_flags);
        else
            k_bkt_count_new<10, TT_LARGE><<<nb < 64u ? 64u : nb, AP_BLOCK, 0, ps>>>(e->table, e->log2cap, e->seed, b_hits, runs, run_tt,
                                                                                    ntiles, tile_shift, nb, e->d_m_flags);
        HIP_TRY(e, hipGetLastError());
        HIP_TRY(e, hipMemcpyAsync(e->h_m_total, e->d_m_flags, sizeof(u32), hipMemcpyDeviceToHost, ps));
        HIP_TRY(e, hipStreamSynchronize(ps));
        const u32 n_new = e->h_m_total[0];
        if (e->live + e->tombs + n_new > e->cap - e->cap / 16) {
            // leave the engine as it was: this batch's scratch goes back, clean, to the next batch
            HIP_TRY(e, hipMemsetAsync(bs, 0, sizeof(BatchScratch), ps));
            HIP_TRY(e, hipStreamSynchronize(ps));
            return fail(e, RL_ERR_TABLE_FULL,
                        "refused, nothing applied: the batch brings %u new keys into a table with live=%llu "
                        "tombstones=%llu capacity=%llu (bound 15/16): rl_resize, sweep, compact or create a larger engine",
                        n_new, (unsigned long long)e->live, (unsigned long long)e->tombs, (unsigned long long)e->cap);
        }
    }
    if (two_streams && !chain_p) HIP_TRY(e, hipEventRecord(e->ev_parted[p & 3u], ps));
    // ---- apply --------------------------------------------------------------------------------------
    // the batch before this one first (if its partition is complete by now: no wait command), then this one —
    // now, or when the next batch is submitted / this one collected
    if (e->pend_old.valid) {
        rc = flush_one(e, e->pend_old, nullptr);
        if (rc) return rc;
    }
    // (not when the held-back replay would be the only thing on its stream — the first batch after an idle stretch, the
    // start of a run: a wait command in front of a kernel costs nothing on a stream that has nothing else to do, and the
    // replay then starts when its partition ends instead of when the host next looks: 27 us instead of 39 into a run)
    if (e->pend.valid && e->defer2 && e->defer_apply && two_streams && e->pipe_depth >= 3u && !e->external_stream && !e->pend.done_event &&
        !e->submit_done_event && e->sub_seq - e->col_seq >= 2 && hipEventQuery(e->ev_parted[e->pend.p & 3u]) != hipSuccess) {
        (void)hipGetLastError();  // its partition is still running: held back once more (see pend_old)
        e->pend_old = e->pend;
        e->pend.valid = false;
    } else {
        rc = flush_one(e, e->pend, nullptr);
        if (rc) return rc;
    }
    }
    rl_engine::PendingApply& q = e->pend;
    q.valid = true;
    q.p = p;
    q.slot = (u32)(e->sub_seq & 3u);
    q.d_hits = d_hits;
    q.n = n;
    q.nb = nb;
    q.ntiles = ntiles;
    q.tile_shift = tile_shift;
    q.run_tt = run_tt;
    // one workgroup per hash bucket + the ones that walk the hot work items; the last one out writes the status block
    // straight into f.h_st (host-mapped)
    q.n_wg = nb + e->hot_wgs;
    q.now = now;
    q.d_verdict = d_verdict;
    q.d_first = d_first;
    q.t_apply = t_apply;
    q.done_event = e->submit_done_event;
    // The long-bucket rule (keys with a quarter of the threshold are promoted out of buckets of >= 1024 hits)
    // evens the buckets out — with 1024 buckets it fills all 512 hot buckets and is worth ~4 us per step — but on a
    // cold start every bucket is long and the first set would be 512 keys in arrival order, the Zipf head not
    // necessarily among them (six slow batches instead of three): it only applies once a collected batch has
    // reported a populated set.
    q.hot_long = e->hot_seen >= 64u ? e->hot_long_cfg : 0xFFFFFFFFu;
    e->part_seq++;
    e->n_part_batches++;
    f.kind = 0;
    f.n = n;
    f.n_wg = q.n_wg;
    f.ntiles = ntiles;
    f.timed = (t ? 1 : 0) | (t_apply ? 2 : 0);
    f.ev_a_prev = (two_streams && e->ext_events && e->apply_events && p >= 1 && e->last_k1_was_part) ? e->ev_applied[(p - 1) & 3u] : nullptr;
    e->last_k1_was_part = true;
    f.seq = (u32)(e->sub_seq + 1);
    f.settled = false;
    e->inflight_hits += n;
    e->sub_seq++;
    // The replay of this batch goes out with the next submit (one-stream engines: in one launch with that batch's
    // partition; two-stream engines: by then the host sees the partition's event complete) or with its own collect — unless
    // the caller orders its own work on the engine's stream, which must then hold everything submitted so far.
    if (!e->defer_apply || e->external_stream || !(two_streams || e->fuse)) return flush_pending_apply(e);
    // A batch submitted into an EMPTY pipeline has no replay in front of it to hide its partition behind: held back, its
    // replay would only go out with the second submit behind it or the first collect — 12 us after the partition had ended, at
    // the start of every burst (the kernel trace of bench.py's 20 timed steps: the first replay started 39 us into the region,
    // 27 of them the partition).  It goes out now, behind a wait for its partition's event, on a stream that is idle anyway.
    if (two_streams && e->sub_seq - e->col_seq == 1) return flush_pending_apply(e);
    return RL_OK;
}

// Wait for the oldest batch in flight and account for it.
int collect_k1_bucketed(rl_engine* e) {
    if (e->sub_seq == e->col_seq) return fail(e, RL_ERR_INVALID, "no batch in flight");
    rl_engine::Inflight& f = e->inflight[e->col_seq & 3u];
    if (e->pend_old.valid && e->pend_old.slot == (u32)(e->col_seq & 3u)) {  // the batch being collected is itself held back
        const int prc = flush_one(e, e->pend_old, nullptr);
        if (prc) return prc;
    }
    if (e->pend.valid && e->pend.slot == (u32)(e->col_seq & 3u)) {  // nothing was submitted behind it: its k_bkt_apply goes out now
        const int prc = flush_pending_apply(e);
        if (prc) return prc;
    } else if (e->pend.valid || e->pend_old.valid) {
        // The replay of a LATER batch is still held back (the pipeline is draining: no submit came to send it out — or
        // RL_DEFER2).  If its partition has finished, it goes out now, behind the replay this collect waits for — not after
        // it, when the stream would sit idle for a launch latency.  (Not finished yet: it stays held back, no wait command.)
        const int prc = poll_pending_apply(e);
        if (prc) return prc;
    }
    if (!f.settled) {
        const int wrc = wait_done(e, f, e->defer2);
        if (wrc) {
            // give the slot up (a stuck batch must not leave the engine RL_ERR_BUSY for ever); the table's
            // state is unknown from here on, which is what the device error tells the caller
            e->col_seq++;
            e->inflight_hits -= f.n;
            return wrc;
        }
        e->inflight_hits -= f.n;
        if (f.kind == 1) {
            e->live -= f.h_st->n_ord;
            e->tombs += f.h_st->n_ord;
        } else {
            e->live += f.h_st->n_inserted;
        }
    }
    e->col_seq++;
    if (f.kind == 1) {  // a stream-ordered sweep: nothing else to account for
        e->last_sweep_removed = f.h_st->n_ord;
        if (f.h_st->err) return status_to_error(e, f.h_st->err);
        return RL_OK;
    }
    // RL_APPLY_TRACE_AT=<batch>: only that batch is looked at (a copy per collect holds the host back: with it the pipeline
    // of a three-deep feeder runs 20 % slower; one copy in a run does not show)
    const char* trace_at = e->apply_trace ? RL_EXP_ENV("RL_APPLY_TRACE_AT") : nullptr;
    if (e->apply_trace && f.n_wg > 1 && e->d_apply_trace && (!trace_at || e->stats.batches == strtoull(trace_at, nullptr, 10))) {
        // diagnostics: where the workgroups of k_bkt_apply spent their time (wall clock, 100 MHz)
        constexpr size_t TR = (size_t)(BK_MAX + 1024) * 8;
        std::vector<unsigned long long> t(TR);  // (the rows behind the replay's: the partition role's)
        // (the batch has completed — its status word was seen — so its own buffer is final; the streams are non-blocking,
        // the copy does not wait for the batches behind it)
        HIP_TRY(e, hipMemcpy(t.data(), e->d_apply_trace + (size_t)((e->col_seq - 1) & 3u) * TR, t.size() * sizeof(unsigned long long),
                             hipMemcpyDeviceToHost));
        bool has_part = false;  // (a launch that only replays — the last batch of a burst — does not overwrite a full one)
        for (size_t r = f.n_wg; r < (size_t)(BK_MAX + 1024) && !has_part; ++r) has_part = t[r * 8] != 0;
        // raw stamps, for offline analysis: of the first batches, or of batch RL_APPLY_TRACE_AT (a steady-state one)
        if (const char* path = RL_EXP_ENV("RL_APPLY_TRACE_FILE"); path && (trace_at || has_part || e->stats.batches < 2)) {
            if (FILE* fp = std::fopen(path, "wb")) {
                const unsigned long long hdr[8] = {f.n_wg, e->hot_wgs, 0, 0, 0, 0, 0, 0};
                std::fwrite(hdr, sizeof(unsigned long long), 8, fp);
                std::fwrite(t.data(), sizeof(unsigned long long), t.size(), fp);
                std::fclose(fp);
            }
        }
        unsigned long long t_min = ~0ull, t_max = 0, first_end = ~0ull;
        double ph[5] = {0, 0, 0, 0, 0}, hits = 0, longest = 0;
        u32 live = 0;
        for (u32 b = 0; b < f.n_wg; ++b) {
            const unsigned long long* q = &t[(size_t)b * 8];
            if (!q[5]) continue;
            ++live;
            t_min = std::min(t_min, q[0]);
            t_max = std::max(t_max, q[5]);
            first_end = std::min(first_end, q[5]);
            const unsigned long long q1 = q[1] ? q[1] : q[0];
            ph[0] += (double)(q1 - q[0]) / 100.0;
            ph[1] += (double)(q[2] - q1) / 100.0;
            ph[2] += (double)(q[3] - q[2]) / 100.0;
            ph[3] += (double)(q[4] - q[3]) / 100.0;
            ph[4] += (double)(q[5] - q[4]) / 100.0;
            hits += (double)q[6];
            longest = std::max(longest, (double)(q[5] - q[0]) / 100.0);
        }
        {   // the partition role's workgroups of the same launch (the next batch), if any
            double pp[6] = {0, 0, 0, 0, 0, 0}, p_end = 0;
            u32 np = 0;
            for (size_t r = f.n_wg; r < (size_t)(BK_MAX + 1024); ++r) {
                const unsigned long long* q = &t[r * 8];
                if (!q[0] || !q[6]) continue;
                ++np;
                for (int k = 0; k < 6; ++k) pp[k] += (double)(q[k + 1] - q[k]) / 100.0;
                p_end = std::max(p_end, (double)(q[6] - t_min) / 100.0);
            }
            if (np)
                std::fprintf(stderr, "[part]  %u workgroups beside it, last one out after %.1f us; mean per workgroup: init %.2f, hot table %.2f, "
                             "walk 1 %.2f, scan %.2f, walk 2 %.2f, flags %.2f us\n", np, p_end, pp[0] / np, pp[1] / np, pp[2] / np, pp[3] / np,
                             pp[4] / np, pp[5] / np);
        }
        if (live)
            std::fprintf(stderr, "[apply] %u workgroups, span %.1f us (first one out after %.1f); mean per workgroup: view %.2f, bucket %.2f "
                         "(%.0f hits), hot items %.2f, drain %.2f, finish %.2f us; longest workgroup %.1f us; %u keys qualified as hot at "
                         "threshold %u\n", live,
                         (double)(t_max - t_min) / 100.0, (double)(first_end - t_min) / 100.0, ph[0] / live, ph[1] / live, hits / live,
                         ph[2] / live, ph[3] / live, ph[4] / live, longest, f.h_st->pad[2], e->hot_threshold);
    }
    // keep the hot set selective: the hottest keys are the ones that stay when more qualify than fit
    if (f.n_wg > 1) e->hot_seen = f.h_st->pad[2];  // (a partitioned batch: k_bkt_tiny leaves the hot sets alone)
    // (A threshold that moves in steps of 1/16 towards a nearly full set — 430-520 hot keys instead of ~375 — and long
    // buckets handed to the hot set from 896 / 768 / 640 hits on were measured, scripts/exp/r5a.sh, r5b.sh: the step
    // stays within 0.3 us or gets longer.  Coarse steps with hysteresis never overflow the set, and an overflow drops
    // whichever keys come last — possibly the hottest.)
    if (f.h_st->pad[2] > (u32)HOT_MAX && e->hot_threshold < (1u << 30)) e->hot_threshold *= 2;
    else if (f.h_st->pad[2] < (u32)HOT_MAX / 4 && e->hot_threshold > e->hot_floor) e->hot_threshold /= 2;
    if (e->hot_report && f.n_wg > 1 && (e->stats.batches & 7u) == 0)  // RL_HOT_REPORT=1 (diagnostics)
        std::fprintf(stderr, "[hot] batch %llu: %u keys qualified, threshold now %u\n", (unsigned long long)e->stats.batches,
                     f.h_st->pad[2], e->hot_threshold);
    e->stats.batches++;
    e->stats.hits += f.n;
    if (f.h_st->err) return status_to_error(e, f.h_st->err);
    return read_timing(e, f);
}

// The timed launches' events of one collected batch -> the engine's timing sums (rl_kernel_timing_read).
int read_timing(rl_engine* e, rl_engine::Inflight& f) {
    if (f.timed) {
        // start events: k_bkt_part tev[0], k_bkt_apply / k_bkt_tiny tev[4]; the stop events are the ones the streams
        // hand to each other anyway ("partitioned" / "applied") or tev[1] / tev[5]
        float ms[2] = {0, 0};
        if ((f.timed & 1) && f.ev_p_stop) {
            HIP_TRY(e, hipEventSynchronize(f.ev_p_stop));
            HIP_TRY(e, hipEventElapsedTime(&ms[0], f.tev[0], f.ev_p_stop));
        }
        if ((f.timed & 2) && f.ev_a_stop) {
            HIP_TRY(e, hipEventSynchronize(f.ev_a_stop));
            HIP_TRY(e, hipEventElapsedTime(&ms[1], f.tev[4], f.ev_a_stop));
        }
        e->ms_slot[RL_T_PART] += ms[0];
        e->ms_slot[RL_T_APPLY] += ms[1];
        if ((f.timed & 2) && f.n_wg > 1 && f.ev_a_prev && f.ev_a_stop != f.tev[5]) {
            // how long the apply stream sat idle in front of this k_bkt_apply, and how long its partition had been ready
            float gap = 0, slack = 0;
            if (hipEventElapsedTime(&gap, f.ev_a_prev, f.tev[4]) == hipSuccess) e->ms_slot[RL_T_APPLY_GAP] += gap;
            if ((f.timed & 1) && f.ev_p_stop && hipEventElapsedTime(&slack, f.ev_p_stop, f.tev[4]) == hipSuccess)
                e->ms_slot[RL_T_PART_SLACK] += slack;
            (void)hipGetLastError();
        }
        if (f.timed & 2) e->timed_launches++;
    }
    return RL_OK;
}

// check_and_update for single-counter requests, all pointers on the device: the bucketed
// single-pass path (rl_bucket.hpp).  Four launches, one host synchronisation at the end.
int run_check_k1_bucketed(rl_engine* e, const Hit* d_hits, u32 n, u64 now, uint8_t* d_verdict, int32_t* d_first) {
    int rc = submit_k1_bucketed(e, d_hits, n, now, d_verdict, d_first);
    if (rc) return rc;
    return collect_k1_bucketed(e);
}

int run_check_k1(rl_engine* e, const Hit* d_hits, u32 n, u64 now, uint8_t* d_verdict, int32_t* d_first) {
    return run_check_k1_bucketed(e, d_hits, n, now, d_verdict, d_first);
}

// ---- the general form (rl_general.hpp): multi-counter requests, load_counters, u64 deltas, update_counter ----
constexpr u32 GEN_ROUNDS_FIRST_MAX = 12;    // longest first group the hint may ask for (GenStatus::changed has room for 30)
constexpr u32 GEN_ROUNDS_ENQ = 3;           // fixpoint rounds enqueued blind before the first look at the status block (enough when
                                            // the second round's admitted set is already the fixpoint: the usual large batch) ...
constexpr u32 GEN_ROUNDS_ENQ_MORE = 6;      // ... and between two looks after that (long chains of dependent requests)

// The hot set holds HOT_MAX keys, appended in whatever order the workgroups get there: while more keys qualify than fit, the
// heaviest may be among the ones left out, and a pass that overflowed because of them overflows again.  An ordinary pass
// moves the threshold one step; a pass that has to be REPEATED moves it as far as its own count says is needed (counts of
// a Zipf head roughly halve when the threshold doubles).
static void adapt_hot_threshold(rl_engine* e, u32 hot_n, bool repeat) {
    if (hot_n > (u32)HOT_MAX) {
        u32 q = std::min(hot_n, 0xFFFFu);  // (k_gen_post's word saturates there)
        do {
            if (e->hot_threshold >= (1u << 30)) break;
            e->hot_threshold *= 2;
            q /= 2;
        } while (repeat && q > (u32)HOT_MAX);
    } else if (hot_n < (u32)HOT_MAX / 4 && e->hot_threshold > e->hot_floor && !repeat) {
        e->hot_threshold /= 2;
    }
}

struct GenCall {
    const Hit* d_hits;
    u32 n_hits;
    const u32* d_req_off;  // null: every hit its own request
    u32 n_req;
    const u64* d_req_delta;  // null: the wire field
    u64 now;
    bool load, update_mode;
    uint8_t* d_verdict;
    int32_t* d_first;
    u64* d_rem;
    u64* d_exp;
    int32_t* d_limited = nullptr;  // per request: id of the limit that limited it, -1 (rl_match_and_check_batch)
    const u32* d_hit_req_ext = nullptr;  // phased form: the caller's request id of every hit (any u32, equal = same request)
    bool hit_req_filled = false;         // e->d_hit_req already holds the request of every hit (k_match_fast wrote it)
    bool host_mapped_results = false;    // every result pointer is fine-grained host-mapped memory (rl_engine::h_tiny)
    const u32* d_hit_check = nullptr;    // hashed keys: the check word of every hit (rl_keyhash.h), verified before the commit
    int32_t* d_msg_status = nullptr;     // hashed keys: per-request status words; a colliding request is marked there (k_gen_check_keys)
};

// Partition of the pass's hits + k_gen_sort: everything up to the first fixpoint round.  A is filled for the kernels
// that follow; *bs_out = the pass's scratch block.  Shared by the blocking resolver (run_general_pass) and the phased
// one (rl_gen_begin_device: admission decided by the host, for requests whose counters live on several GPUs).
static int gen_setup_and_sort(rl_engine* e, const GenCall& c, u32 req0, u32 n_req, u32 hit0, u32 n, bool mark, GenArgs& A_out,
                              BatchScratch** bs_out) {
    hipStream_t st = e->stream;
    const u64 p = e->part_seq;
    const u32 par = (u32)(p % PB_SETS);
    BatchScratch* bs = e->d_bs + p % BS_ROT;
    // Passes are blocking (nothing else is in flight), so a pass partitions with the hot set the pass right before
    // it produced — not the one from two back that the pipelined single-counter path has to use.  (With the
    // pipelined rotation every call overflowed once: the retry's promotions were never the set of the next call.)
    const HotSet* hot_use = e->d_hot + (p + HS_SETS - 1) % HS_SETS;
    HotSet* hot_prod = e->d_hot + p % HS_SETS;
    BHit* b_hits = e->d_bk_hits + (size_t)par * e->bk_stride;
    uint2* ranges = e->d_bk_ranges + (size_t)par * BK_MAX;
    HotParam* hot_param = e->d_hot_param + (size_t)par * (HOT_MAX + 1);
    unsigned short* chunk_tab = e->d_chunk_tab + (size_t)par * e->chunk_tab_len;
    const Hit* hits = c.d_hits + hit0;
    // ---- partition ----------------------------------------------------------------------------------
    u32 bk_log2 = ceil_log2(cdiv(n, 512));
    if (bk_log2 > (u32)BK_LOG2_MAX) bk_log2 = BK_LOG2_MAX;
    // passes of more than 2 M hits: 1024 buckets, not 2048 — twice the hits per (tile, bucket) run of k_bkt_scatter's stores
    // (3.1 M counters: 0.474 -> 0.461 ms per call, scripts/exp/r9g.sh; 512 buckets: 0.467, 256: 0.873)
    if (n > (2u << 20) && bk_log2 > e->gen_bk_log2_big) bk_log2 = e->gen_bk_log2_big;
    if (e->gen_bk_log2_max < bk_log2) bk_log2 = e->gen_bk_log2_max;  // (tests: long buckets on purpose)
    const u32 nb = 1u << bk_log2, nbt = nb + HOT_MAX;
    const bool small = cdiv(n, PT_TILE_SMALL) <= PT_SMALL_MAX_TILES && cdiv(n, PT_TILE_SMALL) < e->bk_tiles_max;
    const u32 ntiles = cdiv(n, small ? PT_TILE_SMALL : PT_TILE);
    // (a pass that follows a pass finds both blocks as its predecessor's clean-up left them: two fill commands less
    // in front of the first kernel)
    if (!e->gen_clean) {
        HIP_TRY(e, hipMemsetAsync(bs, 0, sizeof(BatchScratch), st));
        HIP_TRY(e, hipMemsetAsync(e->d_gst, 0, sizeof(GenStatus), st));
    }
    e->gen_clean = false;
    auto hist_k = small ? k_bkt_hist<1> : k_bkt_hist<PT_STEPS>;
    hist_k<<<ntiles, PT_BLOCK, 0, st>>>(e->table, e->log2cap, e->seed, hits, n, e->d_limits, (u32)e->h_limits.size(), bk_log2,
                                        ntiles, e->d_bk_hist, bs, hot_use, c.update_mode ? 0u : 1u, nullptr, nullptr, nullptr);
    k_bkt_scan<<<cdiv(nbt + HOT_COLS, 32), 1024, 0, st>>>(e->d_bk_hist, ntiles, nbt, e->d_bk_total);
    auto scatter_k = small ? k_bkt_scatter<1> : k_bkt_scatter<PT_STEPS>;
    // (the request of every hit travels with its record: k_gen_sort reads it beside the record instead of gathering it)
    const u32* hit_req_all = c.d_hit_req_ext ? c.d_hit_req_ext : (c.d_req_off ? e->d_hit_req : nullptr);
    u32* b_req = (hit_req_all && e->gen_carry_req) ? e->d_bk_req : nullptr;
    scatter_k<<<ntiles + 1, PT_BLOCK, scatter_lds_bytes(nbt), st>>>(hits, n, e->seed, bk_log2, e->d_bk_hist, e->d_bk_total, hot_use, b_hits, ranges,
                                               &bs->st, ntiles, hot_param, hot_prod, e->hot_threshold, chunk_tab, 1u, nullptr,
                                               hit_req_all ? hit_req_all + hit0 : nullptr, b_req);
    // ---- sort by cell, resolve the cells ---------------------------------------------------------------
    GenArgs A{};
    A.table = e->table;
    A.log2cap = e->log2cap;
    A.seed = e->seed;
    A.limits = e->d_limits;
    A.now = c.now;
    A.hits = hits;
    A.hit_req = hit_req_all;
    A.b_req = b_req;
    A.req_off = c.d_req_off;
    A.req_delta = c.d_req_delta;
    A.hit_check = c.d_hit_check ? c.d_hit_check + hit0 : nullptr;
    A.msg_status = c.d_hit_check ? c.d_msg_status : nullptr;
    A.has_peers = engine_has_peers(e) ? 1u : 0u;
    for (int a = 0; a < MERGE_MAX_ACTORS; ++a) A.peers.t[a] = e->peer_tables[a];
    A.peers.log2cap = e->log2cap;
    A.hit0 = hit0;
    A.req0 = req0;
    A.n_hits = n;
    A.n_req = n_req;
    A.b_hits = b_hits;
    A.ranges = ranges;
    A.nb = nb;
    A.hot_param = hot_param;
    A.chunk_tab = chunk_tab;
    A.hot_next = hot_prod;
    A.hot_threshold = e->hot_threshold;
    A.s_hits = e->d_g_shits;
    A.seg_info = e->d_g_seginfo;
    A.seg_tot = e->d_g_segtot;
    A.piece_sum = e->d_g_piece;
    A.req_stop = e->d_g_reqstop;
    A.reached = e->d_g_reached;
    A.pass[0] = e->d_g_pass;
    A.pass[1] = e->d_g_pass + e->gen_cap;
    A.admitted = e->d_g_admitted;
    A.verdict = c.d_verdict + req0;
    A.first_limited = c.d_first ? c.d_first + req0 : nullptr;
    A.limited_limit = c.d_limited ? c.d_limited + req0 : nullptr;
    A.remaining = c.load ? c.d_rem + hit0 : nullptr;
    A.expires_in = c.load ? c.d_exp + hit0 : nullptr;
    A.gst = e->d_gst;
    A.pst = &bs->st;
    A.load = c.load ? 1u : 0u;
    A.update_mode = c.update_mode ? 1u : 0u;
    A.mark_reached = mark ? 1u : 0u;
    if (e->gen_trace >= 2) {
        if (!e->d_gen_trace) HIP_TRY(e, hipMalloc((void**)&e->d_gen_trace, (size_t)(BK_MAX + HOT_MAX) * 8 * sizeof(unsigned long long)));
        HIP_TRY(e, hipMemsetAsync(e->d_gen_trace, 0, (size_t)(BK_MAX + HOT_MAX) * 8 * sizeof(unsigned long long), st));
        A.trace = e->d_gen_trace;
    }
    k_gen_sort<<<nb + GS_HOT_BLOCKS, GS_BLOCK, 0, st>>>(A);
    HIP_TRY(e, hipGetLastError());
    if (e->gen_trace >= 2) {
        std::vector<unsigned long long> t((size_t)nb * 8);
        HIP_TRY(e, hipMemcpyAsync(t.data(), e->d_gen_trace, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIP_TRY(e, hipStreamSynchronize(st));
        unsigned long long t_min = ~0ull, t_max = 0;
        double ph[4] = {0, 0, 0, 0}, p2_issue = 0;
        u32 live = 0, longest = 0, longest_b = 0;
        double longest_us = 0;
        for (u32 b = 0; b < nb; ++b) {
            const unsigned long long* q = &t[(size_t)b * 8];
            if (!q[4]) continue;
            ++live;
            t_min = std::min(t_min, q[0]);
            t_max = std::max(t_max, q[4]);
            for (int k = 0; k < 4; ++k) ph[k] += (double)(q[k + 1] - q[k]) / 100.0;  // wall clock: 100 MHz
            p2_issue += (double)(q[5] - q[2]) / 100.0;
            const double us = (double)(q[4] - q[0]) / 100.0;
            if (us > longest_us) {
                longest_us = us;
                longest = (u32)q[7];
                longest_b = b;
            }
        }
        if (live)
            std::fprintf(stderr, "[gen] k_gen_sort: %u buckets, span %.1f us; mean per workgroup: init+pass1 %.1f, offsets %.1f, pass2 %.1f (wave 0 done issuing after %.1f), resolve %.1f us; "
                         "slowest: bucket %u with %u hits, %.1f us\n", live, (double)(t_max - t_min) / 100.0, ph[0] / live, ph[1] / live,
                         ph[2] / live, p2_issue / live, ph[3] / live, longest_b, longest, longest_us);
        A.trace = nullptr;
    }
    A_out = A;
    *bs_out = bs;
    return RL_OK;
}

// One pass: requests [req0, req0 + n_req) = hits [hit0, hit0 + n) of the call.  *overflow: a hash bucket was
// too long for k_gen_sort, nothing was applied, the caller retries with smaller passes.
int run_general_pass(rl_engine* e, const GenCall& c, u32 req0, u32 n_req, u32 hit0, u32 n, bool* overflow) {
    *overflow = false;
    hipStream_t st = e->stream;
    if (n == 0) {  // only empty requests: lib.rs:434-440, not limited
        HIP_TRY(e, hipMemsetAsync(c.d_verdict + req0, 0, n_req, st));
        if (c.d_first) HIP_TRY(e, hipMemsetAsync(c.d_first + req0, 0xFF, (size_t)n_req * sizeof(int32_t), st));
        if (c.d_limited) HIP_TRY(e, hipMemsetAsync(c.d_limited + req0, 0xFF, (size_t)n_req * sizeof(int32_t), st));
        HIP_TRY(e, hipStreamSynchronize(st));
        return RL_OK;
    }
    int rc = check_room(e, 0);  // (the cells the pass creates are counted exactly below, before the commit)
    if (rc) return rc;
    const bool mark = !c.load && !c.update_mode && c.d_req_off != nullptr;
    GenArgs A{};
    BatchScratch* bs = nullptr;
    rc = gen_setup_and_sort(e, c, req0, n_req, hit0, n, mark, A, &bs);
    if (rc) return rc;
    A.pass_prefilled = e->gen_pass_prefill ? 1u : 0u;
    A.load_deferred = (c.load && e->gen_load_deferred) ? 1u : 0u;
    if (A.hit_check) k_gen_check_keys<<<cdiv(n, 256), 256, 0, st>>>(A);  // (a collision refuses the pass: k_gen_commit reads gst->err)
    const u64 p = e->part_seq;
    // ---- fixpoint rounds: a few at a time, each returning at once if the one before changed nothing; then
    //      k_gen_commit, which applies the pass only if the status block says it is final and fits -------------
    u32 round = 0;
    bool tail_open = false;  // the group before ended with round `round`'s k_gen_admit alone
    Status h_bst;
    GenStatus h_gst;
    auto cleanup = [&]() -> int {  // leave the rotating scratches clean for whatever batch comes next
        HIP_TRY(e, hipMemsetAsync(e->d_bs, 0, BS_ROT * sizeof(BatchScratch), st));
        HIP_TRY(e, hipMemsetAsync(e->d_gst, 0, sizeof(GenStatus), st));
        e->gen_clean = true;
        return RL_OK;
    };
    u32 rounds_total = 0;  // rounds of this pass that really ran (a converged round returns at once and does not count)
    for (;;) {
        // The first group is as long as the pass BEFORE needed (+ the round that sees the fixpoint): traffic whose requests
        // depend on each other four deep needed a second group every call — a host round trip (40 us) and five rounds that
        // returned at once (18 us each) per 262 144-message batch of the wire path.  A blind round too many costs 18 us, a
        // group too short 40 + what the next group over-enqueues: the hint follows the last pass, never below GEN_ROUNDS_ENQ.
        const u32 first = std::min(std::max(e->gen_rounds_hint, GEN_ROUNDS_ENQ), std::min<u32>(GEN_ROUNDS_FIRST_MAX, n_req + 2u));
        const u32 n_enq = c.update_mode ? 1u : (round == 0 ? first : GEN_ROUNDS_ENQ_MORE);
        for (u32 q = 0; q < n_enq; ++q) {
            // changed[] slots of one group: round q's k_gen_admit checks slot q (did the round before still change
            // the admitted set?) and writes slot q + 1, which is what the round's own kernels check.  Round 0 has
            // no k_gen_admit and runs unconditionally.
            // The LAST round of a group is only its k_gen_admit — the look at whether the admitted set still moves, which is all
            // a converged pass needs of it (its k_gen_piece_sum / k_gen_round would return at once: ~4.7 us apiece for 6 000
            // workgroups that start and leave).  If it does still move, the next group opens with that round's scan
            // (`tail_open`: admitted[] is already this round's), unconditionally, like round 0.
            const bool admit_done = tail_open && q == 0;
            const u32 check = q == 0 ? 0u : q;
            if (round > 0 && !admit_done) k_gen_admit<<<cdiv(n_req, 256), 256, 0, st>>>(A, round, check, q + 1);
            if (round > 0 && q + 1 == n_enq) {
                tail_open = true;
                break;
            }
            tail_open = false;
            const u32 run_if = (round == 0 || admit_done) ? 0u : q + 1;
            k_gen_piece_sum<<<cdiv(n, (u32)GS_MAX), GS_BLOCK, 0, st>>>(A, round, run_if);
            k_gen_round<<<cdiv(n, (u32)GS_MAX), GS_BLOCK, 0, st>>>(A, round, run_if, q + 1);
            ++round;
        }
        if (A.load_deferred) k_gen_load<<<cdiv(n, (u32)GS_MAX), GS_BLOCK, 0, st>>>(A);
        k_gen_final<<<cdiv(n_req, 256), 256, 0, st>>>(A);
        if (mark) k_gen_reach<<<cdiv(n, (u32)GS_MAX), GS_BLOCK, 0, st>>>(A);
        const u64 used = e->live + e->tombs, bound = e->cap - e->cap / 16;
        // the exact count of the cells the pass creates is only needed when they might not fit: a pass of n hits
        // creates at most n (k_gen_commit reports how many it created)
        const bool count_first = used + n > bound;
        if (count_first) k_gen_count<<<std::min(cdiv(n, 256), 1024u), 256, 0, st>>>(A);
        k_gen_commit<<<std::min(cdiv(n, 256), 1024u), 256, 0, st>>>(A, used < bound ? (u32)std::min<u64>(bound - used, 0xFFFFFFFFull) : 0u, nullptr, 0u, 0u);
        HIP_TRY(e, hipGetLastError());
        if (e->gen_post) {
            // what the host decides on, as ONE 16-byte store into host-mapped memory behind the last kernel: no copy
            // command, no stream synchronise (the pass's results are complete when the word is there: k_gen_post runs
            // behind the kernels that wrote them)
            const u32 seq = ++e->gen_post_seq ? e->gen_post_seq : ++e->gen_post_seq;
            k_gen_post<<<1, 256, 0, st>>>(e->d_gst, &bs->st, e->h_gen_word, seq, reinterpret_cast<u32*>(e->d_bs),
                                          (u32)(BS_ROT * sizeof(BatchScratch) / sizeof(u32)));
            HIP_TRY(e, hipGetLastError());
            const int wrc = wait_word(e, e->h_gen_word + 3, seq, "the general resolver's pass");
            if (wrc) return wrc;
            const u32 w0 = e->h_gen_word[0], w1 = e->h_gen_word[1], w2 = e->h_gen_word[2];
            h_gst = GenStatus{};
            h_bst = Status{};
            h_gst.err = w0;
            h_gst.n_inserted = w1;
            h_gst.overflow = w2 & 1u;
            h_gst.committed = (w2 >> 1) & 1u;
            h_gst.last_slot = 0;
            h_gst.changed[0][0] = (w2 >> 2) & 1u;
            h_gst.rounds_run = (w2 >> 8) & 0xFFu;
            h_gst.hot_n = w2 >> 16 == 0xFFFFu ? 0xFFFFFFFFu : w2 >> 16;
            if (!h_gst.err && !h_gst.overflow && !h_gst.committed && !h_gst.changed[0][0]) {
                // converged but refused for room (rare): the message wants the exact count
                HIP_TRY(e, hipMemcpy(&h_gst.n_new, &e->d_gst->n_new, sizeof(u32), hipMemcpyDeviceToHost));
            }
        } else {
        HIP_TRY(e, hipMemcpyAsync(&h_gst, e->d_gst, sizeof(GenStatus), hipMemcpyDeviceToHost, st));
        HIP_TRY(e, hipMemcpyAsync(&h_bst, &bs->st, sizeof(Status), hipMemcpyDeviceToHost, st));
        HIP_TRY(e, hipStreamSynchronize(st));
        }
        e->stats.probe_steps += h_gst.rounds_run;
        rounds_total += h_gst.rounds_run;
        // keep the hot set selective: the hottest keys are the ones that stay when more qualify than fit
        adapt_hot_threshold(e, h_gst.hot_n, h_gst.overflow != 0);
        const u32 err = h_bst.err | h_gst.err;
        if (e->gen_trace)
            std::fprintf(stderr, "[gen] pass seq=%llu n=%u req=%u round=%u rounds_run=%u overflow=%u committed=%u n_new=%u hot_n=%u thr=%u err=%u\n",
                         (unsigned long long)p, n, n_req, round, h_gst.rounds_run, h_gst.overflow, h_gst.committed, h_gst.n_new,
                         h_gst.hot_n, e->hot_threshold, err);
        if (err || h_gst.overflow || h_gst.committed) break;
        bool still_changing = false;
        for (u32 wd = 0; wd < GEN_CHG_W; ++wd) still_changing |= h_gst.changed[h_gst.last_slot][wd * GEN_CHG_STRIDE] != 0u;
        if (c.update_mode || !still_changing) break;  // converged, yet not committed: no room
        if (round > n_req + 2 + GEN_ROUNDS_ENQ_MORE) return fail(e, RL_ERR_DEVICE, "general resolver did not converge (bug)");
        // not yet: forget this group's flags, counts and reached marks, go on from the last round's pass flags
        HIP_TRY(e, hipMemsetAsync(e->d_gst, 0, sizeof(GenStatus), st));
        if (mark) HIP_TRY(e, hipMemsetAsync(e->d_g_reached, 0, n, st));
    }
    if (h_bst.err | h_gst.err) {
        if (h_gst.err & ERRBIT_KEY_COLLISION) {  // which hit: the caller takes its request out and calls again
            u32 cw = 0;
            HIP_TRY(e, hipMemcpy(&cw, &e->d_gst->collide, sizeof(u32), hipMemcpyDeviceToHost));
            e->collide_hit = hit0 + ~cw;
        }
        rc = cleanup();
        return rc ? rc : status_to_error(e, h_bst.err | h_gst.err);  // nothing was applied
    }
    if (h_gst.overflow) {
        // nothing was applied.  k_gen_sort promoted the heavy keys of the long buckets into this attempt's hot
        // set, which is the one the next pass — the retry — partitions with.
        *overflow = true;
        e->part_seq += 1;
        return cleanup();
    }
    if (!h_gst.committed) {
        // ---- converged, but the cells the pass creates do not fit: grow (and redo the pass against the new
        //      table: the cells' slots are stale) or refuse; nothing was applied ---------------------------------
        rc = RL_OK;
        const u64 cap_before = e->cap;
        if (grow_instead(e, &rc) && e->cap > cap_before) {
            rc = cleanup();
            if (rc) return rc;
            return run_general_pass(e, c, req0, n_req, hit0, n, overflow);
        }
        const int crc = cleanup();
        if (rc) return rc;
        if (crc) return crc;
        return fail(e, RL_ERR_TABLE_FULL,
                    "refused, nothing applied: the batch creates %u cells in a table with live=%llu tombstones=%llu "
                    "capacity=%llu (bound 15/16): rl_resize, sweep, compact or create a larger engine",
                    h_gst.n_new, (unsigned long long)e->live, (unsigned long long)e->tombs, (unsigned long long)e->cap);
    }
    e->live += h_gst.n_inserted;  // (== n_new when k_gen_count ran)
    if (!c.update_mode) e->gen_rounds_hint = rounds_total + 1u;
    e->part_seq++;
    if (e->gen_post) e->gen_clean = true;  // (k_gen_post zeroed the status block and the scratch blocks behind the commit)
    else if ((rc = cleanup()) != RL_OK) return rc;
    e->stats.batches++;
    e->stats.hits += n;
    e->stats.ordered_hits += n;
    e->stats.ordered_batches++;
    return RL_OK;
}

int run_check_general(rl_engine* e, const GenCall& c) {
    if (c.n_hits && c.n_hits <= e->gen_tiny_max && c.n_req <= GT_MAX_REQ && !c.update_mode && !c.d_hit_check && !engine_has_peers(e)) {
        // A few requests (the per-request calls of the trait): one workgroup, one launch (k_gen_tiny),
        // completion through a sequence word in the host-mapped status block.
        int rc = check_room(e, c.n_hits);
        if (rc) return rc;
        const u32 seq = ++e->gen_seq ? e->gen_seq : ++e->gen_seq;
        __atomic_store_n(&e->h_status->n_removed, 0u, __ATOMIC_RELEASE);
        k_gen_tiny<<<1, 256, 0, e->stream>>>(e->table, e->log2cap, e->seed, c.d_hits, c.n_hits, c.d_req_off, c.n_req,
                                             e->d_limits, (u32)e->h_limits.size(), c.now, c.load ? 1 : 0, c.d_verdict, c.d_first,
                                             c.load ? c.d_rem : nullptr, c.load ? c.d_exp : nullptr, c.d_req_delta,
                                             e->h_status, seq);
        HIP_TRY(e, hipGetLastError());
        const volatile u32* done = &e->h_status->n_removed;
        const auto t_start = std::chrono::steady_clock::now();
        for (u64 spins = 0; __atomic_load_n(done, __ATOMIC_ACQUIRE) != seq; ++spins) {
            __builtin_ia32_pause();
            if ((spins & 0xFFFFu) == 0xFFFFu) {
                if (std::chrono::steady_clock::now() - t_start > std::chrono::seconds(60))
                    return fail(e, RL_ERR_DEVICE, "k_gen_tiny did not complete within 60 s");
                std::this_thread::yield();
            }
        }
        // return only once the kernel has ended: the results (device or host-mapped memory) are then visible to any
        // reader, not just to this stream
        // (Tried and withdrawn: skipping the synchronise when the results live in fine-grained host memory — the stores were
        // acknowledged before the completion word went out, but acknowledged is not "arrived in host memory in that
        // order": tests/test_gpu_parity.py::test_u64_deltas_and_per_request_clocks failed one run in three.)
        HIP_TRY(e, hipStreamSynchronize(e->stream));
        const u32 err = e->h_status->err, dropped = e->h_status->n_ord, created = e->h_status->n_inserted;
        e->live += created;
        if (err) return status_to_error(e, err);
        e->live -= dropped;
        e->tombs += dropped;
        e->stats.batches++;
        e->stats.hits += c.n_hits;
        e->stats.ordered_hits += c.n_hits;
        e->stats.ordered_batches++;
        if (c.d_limited && c.d_first) {
            k_match_limited_limit<<<cdiv(c.n_req, 256), 256, 0, e->stream>>>(c.d_first, c.d_hits, c.n_req, c.d_limited);
            HIP_TRY(e, hipGetLastError());
            HIP_TRY(e, hipStreamSynchronize(e->stream));
        }
        return RL_OK;
    }
    if (e->tombs > e->cap / 8) {
        const int crc = do_compact(e, 0);
        if (crc) return crc;
    }
    if (c.d_req_off && !c.hit_req_filled) k_gen_hit_req<<<cdiv(c.n_req, 256), 256, 0, e->stream>>>(c.d_req_off, c.n_req, e->d_hit_req);
    // ---- passes of at most sub_max hits (consecutive requests: the semantics are sequential anyway) -----
    u32 sub_max = std::min(e->gen_sub_max, e->gen_cap);
    u32 req_cur = 0, hit_cur = 0, attempts = 0;
    while (req_cur < c.n_req) {
        u32 req_end = c.n_req, hit_end = c.n_hits;
        if (c.n_hits - hit_cur > sub_max) {
            if (!c.d_req_off) {
                req_end = req_cur + sub_max;
                hit_end = hit_cur + sub_max;
            } else {
                // the last request boundary at or below hit_cur + sub_max
                k_gen_cuts<<<1, 64, 0, e->stream>>>(c.d_req_off, c.n_req, hit_cur, sub_max, 1u, e->d_m_flags);
                HIP_TRY(e, hipGetLastError());
                HIP_TRY(e, hipMemcpyAsync(e->h_m_total, e->d_m_flags, 2 * sizeof(u32), hipMemcpyDeviceToHost, e->stream));
                HIP_TRY(e, hipStreamSynchronize(e->stream));
                req_end = e->h_m_total[0];
                hit_end = e->h_m_total[1];
                if (req_end <= req_cur) {  // one request alone is larger than a pass
                    req_end = req_cur + 1;
                    HIP_TRY(e, hipMemcpy(&hit_end, c.d_req_off + req_end, sizeof(u32), hipMemcpyDeviceToHost));
                    if (hit_end - hit_cur > e->gen_cap)
                        return fail(e, RL_ERR_BATCH_TOO_LARGE, "one request carries %u counters (> %u)", hit_end - hit_cur, e->gen_cap);
                }
            }
        }
        bool overflow = false;
        const int rc = run_general_pass(e, c, req_cur, req_end - req_cur, hit_cur, hit_end - hit_cur, &overflow);
        if (rc) return rc;
        if (overflow) {
            // nothing was applied.  First retry: the same pass with the heavy keys promoted (see run_general_pass);
            // after that: the same requests in smaller passes.
            if (++attempts > 1) {
                const u32 had = hit_end - hit_cur;
                if (req_end - req_cur <= 1 || had <= 1)
                    return fail(e, RL_ERR_BATCH_TOO_LARGE, "one request puts more than %d counters into one hash bucket", GS_MAX);
                sub_max = had / 2;
            }
            continue;
        }
        attempts = 0;
        sub_max = std::min(e->gen_sub_max, e->gen_cap);
        req_cur = req_end;
        hit_cur = hit_end;
    }
    // (every pass ends with a stream synchronisation after its last kernel: the results are complete; only the
    // memset that leaves the scratch blocks clean may still be queued, and whatever comes next is ordered behind it)
    return RL_OK;
}

int validate_batch(rl_engine* e, const void* hits, u32 n_hits, const u32* req_off, u32 n_req,
                   const void* verdict) {
    if (!e) return RL_ERR_INVALID;
    if (n_hits > e->max_batch) return fail(e, RL_ERR_BATCH_TOO_LARGE, "n_hits %u > max_batch_hits %u", n_hits, e->max_batch);
    if (n_hits && !hits) return fail(e, RL_ERR_INVALID, "hits is null");
    if (n_req && !verdict) return fail(e, RL_ERR_INVALID, "verdict is null");
    if (!req_off && n_req != n_hits) return fail(e, RL_ERR_INVALID, "req_off is null but n_req != n_hits");
    return RL_OK;
}

}  // namespace

static int32_t insert_rows_locked(rl_engine* e, const CellRow* d_rows, u64 n, int overwrite) {
    int rc = check_room(e, n);
    if (rc) return rc;
    HIP_TRY(e, hipMemsetAsync(e->d_status, 0, sizeof(Status), e->stream));
    k_insert_rows<<<cdiv(n, 256), 256, 0, e->stream>>>(e->table, e->log2cap, e->seed, d_rows, n, overwrite,
                                                       e->d_status);
    HIP_TRY(e, hipGetLastError());
    rc = read_status(e);
    if (rc) return rc;
    e->live += e->h_status->n_inserted;
    if (e->h_status->err) return status_to_error(e, e->h_status->err);
    return RL_OK;
}

template <int MODE>
static int32_t scan_locked(rl_engine* e, u32 limit, u64 now, rl_cell_row* out, u64 cap, uint64_t* n_out) {
    HIP_TRY(e, hipSetDevice(e->device));
    CellRow* d_out = nullptr;
    if (cap) {
        hipError_t r = hipMalloc((void**)&d_out, cap * sizeof(CellRow));
        if (r != hipSuccess) return fail(e, RL_ERR_NOMEM, "hipMalloc for scan output failed");
    }
    auto cleanup = [&]() {
        if (d_out) (void)hipFree(d_out);
    };
    hipError_t r = hipMemsetAsync(e->d_status, 0, sizeof(Status), e->stream);
    if (r == hipSuccess) r = hipMemsetAsync(e->d_total, 0, sizeof(unsigned long long), e->stream);
    if (r == hipSuccess) {
        k_scan<MODE><<<2048, 256, 0, e->stream>>>(e->table, e->cap, limit, now, d_out, cap, e->d_status, e->d_total);
        r = hipGetLastError();
    }
    if (r == hipSuccess)
        r = hipMemcpyAsync(e->h_total, e->d_total, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream);
    if (r == hipSuccess)
        r = hipMemcpyAsync(e->h_status, e->d_status, sizeof(Status), hipMemcpyDeviceToHost, e->stream);
    if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
    if (r == hipSuccess && out && cap) {
        const u64 n = *e->h_total < cap ? *e->h_total : cap;
        if (n) r = hipMemcpy(out, d_out, n * sizeof(CellRow), hipMemcpyDeviceToHost);
    }
    cleanup();
    if (r != hipSuccess) return fail(e, RL_ERR_DEVICE, "table scan failed: %s", hipGetErrorString(r));
    if (n_out) *n_out = *e->h_total;
    const u32 removed = e->h_status->n_removed;
    e->live -= removed;
    e->tombs += removed;
    // The peer tables (rl_merge_cells) follow: what a remote actor contributed to a counter that is deleted or cleared,
    // or to a window that is over, is dropped with it — they would otherwise only ever grow.  Their
    // tombstones go with the main table's at the next compaction (do_compact rehashes all of them together).
    if (MODE == SCAN_DELETE_LIMIT || MODE == SCAN_CLEAR_SIMPLE || MODE == SCAN_SWEEP) {
        const u32 arg = MODE == SCAN_DELETE_LIMIT ? limit : (MODE == SCAN_CLEAR_SIMPLE ? 0xFFFFFFFEu : 0xFFFFFFFFu);
        const u64 t = MODE == SCAN_SWEEP ? now : 0ull;  // (expiry <= 0 never holds: only the sweep drops by time)
        bool any = false;
        for (Cell* pt : e->peer_tables)
            if (pt) {
                k_scan<SCAN_PEER><<<2048, 256, 0, e->stream>>>(pt, e->cap, arg, t, (CellRow*)nullptr, 0ull, e->d_status, e->d_total);
                any = true;
            }
        if (any) {
            HIP_TRY(e, hipGetLastError());
            HIP_TRY(e, hipStreamSynchronize(e->stream));
        }
    }
    return RL_OK;
}

extern "C" {

// ---- the exception barrier of the C ABI (rl_abi_guard.h) ------------------------------------------------------------
namespace {
thread_local char g_abi_msg[320] = "";
}
int32_t rl_abi_caught(const char* fn, const char* what, int32_t status) {
    std::snprintf(g_abi_msg, sizeof g_abi_msg, "%s: %s", fn ? fn : "?", what ? what : "?");
    return status;
}
const char* rl_last_internal_error(void) { return g_abi_msg; }
namespace {
struct NotAStdException {
    int why;
};
}
int32_t rl_abi_selftest(int32_t kind) try {
    std::vector<u64> guard_is_unwound(16, 1ull);  // (RAII behind the barrier unwinds like anywhere else)
    switch (kind) {
        case 1: throw std::bad_alloc();
        case 2: throw std::length_error("rl_abi_selftest: std::length_error");
        case 3: throw NotAStdException{3};
        case 4: {  // a REAL failed allocation, the way a bogus size reaches operator new behind an entry point
            std::vector<u64> v;
            v.resize((size_t)1 << 58);
            return (int32_t)v.size();
        }
        case 5: {
            std::vector<u64> v;
            v.reserve(v.max_size() + 1);  // std::length_error from the library itself
            return (int32_t)v.capacity();
        }
        default: return RL_OK;
    }
} RL_ABI_CATCH

int32_t rl_engine_create(const rl_config* cfg, rl_engine** out) try {
    if (!cfg || !out) return RL_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev)
        return RL_ERR_NO_DEVICE;
    rl_engine* e = new (std::nothrow) rl_engine();
    if (!e) return RL_ERR_NOMEM;
    e->device = cfg->device;
    e->seed = cfg->hash_seed;
    e->auto_grow = (cfg->flags & RL_CFG_AUTO_GROW) != 0;
    e->max_batch = cfg->max_batch_hits ? cfg->max_batch_hits : (1u << 20);
    if (e->max_batch > MAX_BATCH_HITS) e->max_batch = MAX_BATCH_HITS;
    e->max_limits = cfg->max_limits ? cfg->max_limits : 1024;
    if (const char* v = RL_EXP_ENV("RL_OVERLAP")) e->overlap = atoi(v) != 0;
    if (const char* v = RL_EXP_ENV("RL_PIPE_DEPTH")) e->pipe_depth = atoi(v) == 2 ? 2u : 3u;
    if (const char* v = RL_EXP_ENV("RL_DEFER_APPLY")) e->defer_apply = atoi(v) != 0;
    if (const char* v = RL_EXP_ENV("RL_APPLY_EVENTS")) e->apply_events = atoi(v) != 0;
    // Engines for small batches run the fused form (one stream, one launch per step): a step of <= 256 k hits is bound by
    // launches, not by kernels (64 k hits: 16.4 us per batch fused, 19.4 on two streams; 1 M hits: 65-75 against 47).
    e->fuse = e->max_batch <= (1u << 18);
    if (const char* v = getenv("RL_FUSE")) e->fuse = atoi(v) != 0;
    if (const char* v = RL_EXP_ENV("RL_PART_COMPACT")) e->part_compact = atoi(v) != 0 ? 1 : 0;
    if (const char* v = RL_EXP_ENV("RL_DEFER2")) e->defer2 = atoi(v) != 0;
    if (const char* v = RL_EXP_ENV("RL_RESP_DIRECT")) e->resp_direct = atoi(v) != 0;
    if (const char* v = RL_EXP_ENV("RL_RESULTS_DIRECT")) e->results_direct = atoi(v) != 0;
    if (const char* v = RL_EXP_ENV("RL_RESP_BLIND")) e->resp_blind = atoi(v) != 0;
    if (const char* v = RL_EXP_ENV("RL_RESP_VIA_COPY")) e->resp_via_copy = (u32)std::max(0, atoi(v));
    if (const char* v = RL_EXP_ENV("RL_RESP_LAZY_DEPTH")) e->resp_lazy_depth = (u32)std::min(std::max(atoi(v), 1), 32);
    if (const char* v = RL_EXP_ENV("RL_GEN_PASS_PREFILL")) e->gen_pass_prefill = atoi(v) != 0;
    if (const char* v = RL_EXP_ENV("RL_GEN_LOAD_DEFERRED")) e->gen_load_deferred = atoi(v) != 0;
    if (const char* v = RL_EXP_ENV("RL_GEN_CARRY_REQ")) e->gen_carry_req = atoi(v) != 0;
    if (const char* v = RL_EXP_ENV("RL_RESP_WRITERS")) e->resp_writers = (u32)std::max(1, atoi(v));
    if (const char* v = RL_EXP_ENV("RL_RESP_PIECES")) e->resp_pieces = std::min<u32>(rl_engine::RESP_CHUNKS, std::max(1, atoi(v)));
    if (e->fuse) e->overlap = false;  // one stream: the partition rides in the replay's launch
    if (const char* v = RL_EXP_ENV("RL_APPLY_TRACE")) e->apply_trace = atoi(v);
    if (const char* v = RL_EXP_ENV("RL_HOT_WGS")) e->hot_wgs = (u32)std::min(std::max(atoi(v), 8), 1024);
    if (const char* v = RL_EXP_ENV("RL_PART_STEPS")) {
        const int b = atoi(v);
        if (b == 4 || b == 8 || b == 16) e->part_steps_cfg = (u32)b;
    }
    if (const char* v = RL_EXP_ENV("RL_EXT_EVENTS")) e->ext_events = atoi(v) != 0;
    if (const char* v = RL_EXP_ENV("RL_HOT_PROMOTE")) {
        const long b = strtol(v, nullptr, 10);
        if (b >= 16 && b <= (1 << 20)) e->hot_floor = e->hot_threshold = (u32)b;
    }
    if (const char* v = RL_EXP_ENV("RL_HOT_REPORT")) e->hot_report = atoi(v) != 0;
    if (const char* v = RL_EXP_ENV("RL_HOT_LONG")) e->hot_long_cfg = (u32)std::max<unsigned long>(1ul, strtoul(v, nullptr, 10));
    if (const char* v = RL_EXP_ENV("RL_GEN_TRACE")) e->gen_trace = atoi(v);
    if (const char* v = RL_EXP_ENV("RL_GEN_BUCKET_LOG2_BIG")) e->gen_bk_log2_big = (u32)std::min(std::max(atoi(v), 6), (int)BK_LOG2_MAX);
    if (const char* v = RL_EXP_ENV("RL_GEN_BUCKET_LOG2")) e->gen_bk_log2_max = (u32)std::min(std::max(atoi(v), 0), (int)BK_LOG2_MAX);
    if (const char* v = RL_EXP_ENV("RL_GEN_SUB_MAX")) {
        const long b = strtol(v, nullptr, 10);
        if (b >= 1024) e->gen_sub_max = (u32)std::min<long>(b, GEN_SUB_MAX);
    }
    if (const char* v = RL_EXP_ENV("RL_APPLY2_CFG")) e->apply2_cfg = atoi(v);
    if (const char* v = getenv("RL_TINY_MAX")) {
        const long b = strtol(v, nullptr, 10);
        if (b >= 0 && b <= (long)TINY_MAX) e->tiny_max = (u32)b;
        if (b == 0) e->gen_tiny_max = 0;  // RL_TINY_MAX=0 switches both one-launch kernels off
    }
    if (const char* v = RL_EXP_ENV("RL_GEN_TINY_MAX")) {
        const long b = strtol(v, nullptr, 10);
        if (b >= 0 && b <= (long)GT_MAX) e->gen_tiny_max = (u32)b;
    }
    if (const char* v = RL_EXP_ENV("RL_BUCKET_LOG2")) {
        const long b = strtol(v, nullptr, 10);
        if (b >= 0 && b <= BK_LOG2_MAX) e->bk_log2_cfg = (u32)b;
    }
    e->log2cap = ceil_log2(cfg->capacity_cells < 1024 ? 1024 : cfg->capacity_cells);
    if (e->log2cap > 31) {
        delete e;
        return RL_ERR_INVALID;
    }
    e->cap = 1ull << e->log2cap;
    auto bail = [&](int rc) {
        rl_engine_destroy(e);
        return rc;
    };
    if (hipSetDevice(e->device) != hipSuccess) return bail(RL_ERR_NO_DEVICE);
    if (hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) != hipSuccess) return bail(RL_ERR_DEVICE);
    e->stream = e->own_stream;
    if (e->overlap) {
        // the partition stream at a higher priority than the replay stream: its few, fat workgroups (1024 threads,
        // 100 KB of LDS) then get CU slots ahead of k_bkt_apply's many small ones instead of queueing behind them
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);  // (lo = least urgent, hi = most: numerically lower)
        int prio = prio_hi;
        if (const char* v = RL_EXP_ENV("RL_PSTREAM_PRIO")) prio = atoi(v) == 0 ? prio_lo : (atoi(v) == 2 ? 0 : prio_hi);
        if (hipStreamCreateWithPriority(&e->own_pstream, hipStreamNonBlocking, prio) != hipSuccess) return bail(RL_ERR_DEVICE);
        e->pstream = e->own_pstream;
        if (const char* v = RL_EXP_ENV("RL_XOVER")) e->xover = atoi(v);
        if (e->xover && hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking) != hipSuccess) return bail(RL_ERR_DEVICE);
    } else {
        e->pstream = e->stream;
    }
    // "partitioned" / "applied": timing-capable (they are also the stop events of timed launches, see ext_events), and
    // WITHOUT the system-scope fence an event carries by default: they order two streams of one device, nothing the
    // host reads depends on them (the status block is fine-grained memory written with explicit stores).
    unsigned ev_flags = hipEventDisableSystemFence;
    if (const char* v = RL_EXP_ENV("RL_EVENT_FLAGS")) ev_flags = (unsigned)strtoul(v, nullptr, 0);
    for (auto& ev : e->ev_parted)
        if (hipEventCreateWithFlags(&ev, ev_flags) != hipSuccess) return bail(RL_ERR_DEVICE);
    for (auto& ev : e->ev_applied)
        if (hipEventCreateWithFlags(&ev, ev_flags) != hipSuccess) return bail(RL_ERR_DEVICE);
    if (hipEventCreateWithFlags(&e->ev_match, hipEventDisableTiming) != hipSuccess) return bail(RL_ERR_DEVICE);
    // the partition kernels' dynamic LDS goes up to 80 KB (beside their static LDS)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_bkt_part<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)scatter_lds_bytes(BKT_MAX)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_bkt_part<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)scatter_lds_bytes(BKT_MAX)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_bkt_part<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)scatter_lds_bytes(BKT_MAX)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_bkt_part<16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)scatter_lds_bytes(BKT_MAX)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_bkt_scatter<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)scatter_lds_bytes(BKT_MAX)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_bkt_scatter<PT_STEPS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)scatter_lds_bytes(BKT_MAX)) != hipSuccess)
        return bail(RL_ERR_DEVICE);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) == hipSuccess && cus > 0)
            e->n_cus = (u32)cus;
        if (const char* v = RL_EXP_ENV("RL_APPLY_WG_PER_CU")) {  // tuning knob
            const long m = strtol(v, nullptr, 10);
            if (m >= 1 && m <= 8) e->n_cus = e->n_cus * (u32)m / 2;
        }
    }
    int rc = alloc_table(e, e->cap, &e->table);
    if (rc) return bail(rc);
    const size_t mb = e->max_batch;
    // RL_LOG_ALLOCS=1 (experiment builds): every device buffer's address range on stderr — a "Memory access fault by GPU ... on
    // address X" of a later kernel can then be attributed to the buffer it ran off (scripts/exp/r13_abort_hunt.sh)
    const bool log_allocs = RL_EXP_ENV("RL_LOG_ALLOCS") != nullptr;
    if (log_allocs) std::fprintf(stderr, "[alloc] %-18s %p %zu\n", "e->table", (void*)e->table, (size_t)e->cap * sizeof(Cell));
#ifdef RL_EXPERIMENT
#define RZ_POISON(ptr, bytes) (rz_poison(#ptr) ? (void)(hipMemset((ptr), RZ_FILL, (bytes)), hipDeviceSynchronize()) : (void)0)
#else
#define RZ_POISON(ptr, bytes) ((void)0)
#endif
#define ALLOC(ptr, bytes)                                                                                       \
    if (hipMalloc((void**)&(ptr), (bytes)) != hipSuccess) return bail(RL_ERR_NOMEM);                               \
    else if (RZ_POISON(ptr, bytes), log_allocs) std::fprintf(stderr, "[alloc] %-18s %p %zu\n", #ptr, (void*)(ptr), (size_t)(bytes))
    ALLOC(e->d_limits, e->max_limits * sizeof(LimitDev));
    ALLOC(e->d_hits, mb * sizeof(Hit));
    ALLOC(e->d_req_off, (mb + 1) * sizeof(u32));
    ALLOC(e->d_verdict, mb);
    ALLOC(e->d_first, mb * sizeof(int32_t));
    ALLOC(e->d_remaining, mb * sizeof(u64));
    ALLOC(e->d_expires, mb * sizeof(u64));
    e->gen_cap = (u32)std::min<size_t>(mb, GEN_SUB_MAX);
    ALLOC(e->d_hit_req, mb * sizeof(u32));
    ALLOC(e->d_req_delta, mb * sizeof(u64));
    ALLOC(e->d_g_shits, (size_t)e->gen_cap * sizeof(SHit));
    ALLOC(e->d_g_seginfo, (size_t)e->gen_cap * sizeof(SegInfo));
    ALLOC(e->d_g_segtot, (size_t)e->gen_cap * sizeof(SegTot));
    ALLOC(e->d_g_piece, ((size_t)e->gen_cap / GS_MAX + 8) * sizeof(SegTot));
    ALLOC(e->d_g_reqstop, mb * sizeof(u32));
    ALLOC(e->d_g_reached, (size_t)e->gen_cap);
    ALLOC(e->d_g_pass, 2 * (size_t)e->gen_cap);
    ALLOC(e->d_g_admitted, mb);
    ALLOC(e->d_gst, sizeof(GenStatus));
    ALLOC(e->d_row1, sizeof(CellRow));
    ALLOC(e->d_status, sizeof(Status));
    ALLOC(e->d_sweep_st, 4 * sizeof(Status));
    ALLOC(e->d_bs, (BS_ROT + 1) * sizeof(BatchScratch));  // the rotating ones + k_bkt_tiny's own
    if (hipMemset(e->d_bs, 0, (BS_ROT + 1) * sizeof(BatchScratch)) != hipSuccess) return bail(RL_ERR_DEVICE);
    ALLOC(e->d_total, sizeof(unsigned long long));
    {
        const u32 big = cdiv(mb, PT_TILE), sm = cdiv(mb, PT_TILE_SMALL) < PT_SMALL_MAX_TILES ? cdiv(mb, PT_TILE_SMALL) : PT_SMALL_MAX_TILES;
        e->bk_tiles_max = (big > sm ? big : sm) + 1;
    }
    ALLOC(e->d_bk_hist, (size_t)ROW_MAX * e->bk_tiles_max * sizeof(u32));
    ALLOC(e->d_bk_total, (size_t)ROW_MAX * sizeof(u32));
    ALLOC(e->d_bk_ranges, PB_SETS * (size_t)BK_MAX * sizeof(uint2));
    ALLOC(e->d_hot, HS_SETS * sizeof(HotSet));
    if (hipMemset(e->d_hot, 0, HS_SETS * sizeof(HotSet)) != hipSuccess) return bail(RL_ERR_DEVICE);
    ALLOC(e->d_hot_param, PB_SETS * (size_t)(HOT_MAX + 1) * sizeof(HotParam));
    if (hipMemset(e->d_hot_param, 0, PB_SETS * (size_t)(HOT_MAX + 1) * sizeof(HotParam)) != hipSuccess) return bail(RL_ERR_DEVICE);
    e->bk_stride = (mb + (PT_BLOCK * PT_STEPS_MAX - 1)) / (PT_BLOCK * PT_STEPS_MAX) * (PT_BLOCK * PT_STEPS_MAX);
    ALLOC(e->d_bk_hits, PB_SETS * e->bk_stride * sizeof(BHit));
    ALLOC(e->d_bk_req, e->bk_stride * sizeof(u32));
    e->run_tt_max = cdiv(mb, PT_BLOCK * PT_STEPS_MAX) > (u32)TT_SMALL ? (u32)TT_LARGE : (u32)TT_SMALL;
    ALLOC(e->d_runs, PB_SETS * (size_t)BKT_MAX * e->run_tt_max * sizeof(u32));
    ALLOC(e->d_items, PB_SETS * sizeof(HotItems));
    if (hipMemset(e->d_items, 0, PB_SETS * sizeof(HotItems)) != hipSuccess) return bail(RL_ERR_DEVICE);
    ALLOC(e->d_hot_arrive, (size_t)2 * HOT_MAX * sizeof(u32));  // by the parity of the partitioned batch
    if (hipMemset(e->d_hot_arrive, 0, (size_t)2 * HOT_MAX * sizeof(u32)) != hipSuccess) return bail(RL_ERR_DEVICE);
    ALLOC(e->d_tiny_hits, (size_t)TINY_MAX * sizeof(BHit));
    e->chunk_tab_len = std::max<size_t>((size_t)mb / HOT_CHUNK + HOT_MAX + 8, (size_t)HOT_MAX * HOT_NK_MAX);
    ALLOC(e->d_chunk_tab, PB_SETS * e->chunk_tab_len * sizeof(unsigned short));
    ALLOC(e->d_route_cnt, (size_t)ROUTE_MAX_BLOCKS * ROUTE_MAX_WORLD * sizeof(u32) + 64 * sizeof(u32));
    if (hipMemset(e->d_route_cnt, 0, (size_t)ROUTE_MAX_BLOCKS * ROUTE_MAX_WORLD * sizeof(u32) + 64 * sizeof(u32)) != hipSuccess) return bail(RL_ERR_DEVICE);
    ALLOC(e->d_m_ns, mb * sizeof(u32));
    ALLOC(e->d_m_delta, mb * sizeof(u32));
    ALLOC(e->d_m_ent_off, (mb + 1) * sizeof(u32));
    ALLOC(e->d_m_ent_key, mb * sizeof(u32));
    ALLOC(e->d_m_ent_val, mb * sizeof(u32));
    ALLOC(e->d_m_count, (mb + 1) * sizeof(u32));
    ALLOC(e->d_m_limited, mb * sizeof(int32_t));
    ALLOC(e->d_m_mask, mb * sizeof(unsigned long long));
    ALLOC(e->d_m_flags, 16);
    e->m_scan_tmp_bytes = ((size_t)cdiv(mb + 1, XSCAN_PER_WG) + 4) * sizeof(u32);  // the generic matcher's scan: workgroup totals
    ALLOC(e->d_m_scan_tmp, e->m_scan_tmp_bytes);
    if (hipHostMalloc((void**)&e->h_m_total, 16) != hipSuccess) return bail(RL_ERR_NOMEM);
    {
        const size_t bytes = sizeof(MatchScan) + (size_t)cdiv(mb, 256) * sizeof(u32);
        ALLOC(e->d_m_scan1, bytes);
        if (hipMemsetAsync(e->d_m_scan1, 0, bytes, e->stream) != hipSuccess) return bail(RL_ERR_DEVICE);
    }
    if (const char* v = RL_EXP_ENV("RL_MATCH_ONE")) e->match_one = atoi(v) != 0;
    if (const char* v = RL_EXP_ENV("RL_GEN_POST")) e->gen_post = atoi(v) != 0;
#undef ALLOC
    // Host-mapped blocks the device writes while the host polls them: fine-grained (coherent) memory, so a
    // device store is on its way to the host when the wave's vmcnt says so, not when the kernel ends.
    bool all_coherent = true;
    auto host_block = [&all_coherent](void** p, size_t bytes) {
        if (hipHostMalloc(p, bytes, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) return true;
        (void)hipGetLastError();
        all_coherent = false;
        return hipHostMalloc(p, bytes, hipHostMallocMapped) == hipSuccess;
    };
    if (!host_block((void**)&e->h_status, sizeof(Status))) return bail(RL_ERR_NOMEM);
    if (!host_block((void**)&e->h_tiny, TIO_BYTES)) return bail(RL_ERR_NOMEM);
    if (!host_block((void**)&e->h_serve, sizeof(ServeBox))) return bail(RL_ERR_NOMEM);
    memset(e->h_serve, 0, sizeof(ServeBox));
    if (const char* v = getenv("RL_SERVE")) e->serve_enabled = atoi(v) != 0;
    if (const char* v = RL_EXP_ENV("RL_SERVE_TIMEOUT_MS")) e->serve_timeout_ms = (u32)std::max(1, atoi(v));
    if (const char* v = RL_EXP_ENV("RL_SERVE_LINGER_US")) e->serve_linger_us = (u32)std::min(std::max(1, atoi(v)), 1000000);
    if (!host_block((void**)&e->h_m_word, 64)) return bail(RL_ERR_NOMEM);
    if (!host_block((void**)&e->h_gen_word, 64)) return bail(RL_ERR_NOMEM);
    memset(e->h_m_word, 0, 64);
    memset(e->h_gen_word, 0, 64);
    e->h_tiny_coherent = all_coherent;
    if (hipHostMalloc((void**)&e->h_total, sizeof(unsigned long long)) != hipSuccess) return bail(RL_ERR_NOMEM);
    for (auto& ev : e->ev)
        if (hipEventCreate(&ev) != hipSuccess) return bail(RL_ERR_DEVICE);
    for (auto& f : e->inflight) {
        for (auto& ev : f.tev)
            if (hipEventCreateWithFlags(&ev, ev_flags) != hipSuccess) return bail(RL_ERR_DEVICE);
        if (!host_block((void**)&f.h_st, sizeof(Status))) return bail(RL_ERR_NOMEM);
        memset(f.h_st, 0, sizeof(Status));
    }
    if (hipStreamSynchronize(e->stream) != hipSuccess) return bail(RL_ERR_DEVICE);
    // (the fills above that went out on the null stream: the engine's streams are non-blocking, i.e. not ordered with it — an
    // allocator that filled its blocks there was overtaken by k_table_init, found with the red zones of the experiment build)
    if (hipDeviceSynchronize() != hipSuccess) return bail(RL_ERR_DEVICE);
    e->stats.capacity_cells = e->cap;
    *out = e;
    return RL_OK;
} RL_ABI_CATCH

void rl_engine_destroy(rl_engine* e) {
    if (!e) return;
    if (e->apply_trace)
        std::fprintf(stderr, "[engine] %llu partitioned batches; wait commands enqueued: %llu for a partition, %llu for an apply\n",
                     (unsigned long long)e->n_part_batches, (unsigned long long)e->n_wait_parted, (unsigned long long)e->n_wait_applied);
    (void)hipSetDevice(e->device);
    if (e->h_serve) serve_stop(e);
    if (e->apply_trace && e->n_serve_calls)
        std::fprintf(stderr, "[engine] %llu per-request calls served by %llu launches of k_gen_serve\n",
                     (unsigned long long)e->n_serve_calls, (unsigned long long)e->n_serve_launches);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->resp_stream) (void)hipStreamSynchronize(e->resp_stream);  // (its kernels write the pinned staging freed below)
    if (e->own_pstream) (void)hipStreamSynchronize(e->own_pstream);
    for (auto& pt : e->peer_tables)
        if (pt) (void)hipFree(pt);
    void* ptrs[] = {e->table,      e->d_limits,   e->d_hits,     e->d_req_off, e->d_verdict, e->d_first,
                    e->d_remaining, e->d_expires, e->d_status,   e->d_total,    e->d_route_cnt,
                    e->d_hit_req,  e->d_req_delta, e->d_g_shits,  e->d_g_seginfo, e->d_g_segtot, e->d_g_piece, e->d_g_reqstop,
                    e->d_g_reached, e->d_g_pass,  e->d_g_admitted, e->d_gst,      e->d_row1,
                    e->d_bk_hist,  e->d_bk_total, e->d_bk_ranges, e->d_bk_hits,  e->d_bk_req, e->d_tiny_hits, e->d_chunk_tab,
                    e->d_hot,     e->d_hot_param, e->d_bs,       e->d_hot_arrive, e->d_cmark, e->d_w_blob, e->d_w_ns, e->d_w_lit, e->d_w_prefix, e->d_w_bytes, e->d_w_off, e->d_w_status, e->d_w_slot_h, e->d_hit_check, e->d_resp_blob, e->d_resp_frag, e->d_resp_off, e->d_resp_bytes, e->d_runs,    e->d_items,  e->d_apply_trace, e->d_sweep_st,
                    e->d_match_limits, e->d_match_conds, e->d_match_ns_off, e->d_m_ns, e->d_m_delta, e->d_m_ent_off,
                    e->d_m_ent_key, e->d_m_ent_val, e->d_m_count, e->d_m_limited, e->d_m_flags, e->d_m_scan_tmp, e->d_m_mask, e->d_match_flimits, e->d_match_fconds, e->d_gen_trace};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (e->h_status) (void)hipHostFree(e->h_status);
    if (e->h_tiny) (void)hipHostFree(e->h_tiny);
    if (e->h_total) (void)hipHostFree(e->h_total);
    if (e->h_m_total) (void)hipHostFree(e->h_m_total);
    if (e->h_m_word) (void)hipHostFree(e->h_m_word);
    for (void* hs : e->h_stage)
        if (hs) (void)hipHostFree(hs);
    if (e->resp_stream) {
        (void)hipStreamSynchronize(e->resp_stream);
        (void)hipStreamDestroy(e->resp_stream);
    }
    if (e->in_stream) {
        (void)hipStreamSynchronize(e->in_stream);
        (void)hipStreamDestroy(e->in_stream);
    }
    for (hipEvent_t ev : e->in_ev)
        if (ev) (void)hipEventDestroy(ev);
    for (uint8_t* q : e->d_w_bytes_more)
        if (q) (void)hipFree(q);
    for (u32* q : e->d_w_off_more)
        if (q) (void)hipFree(q);
    for (u32 q = 0; q < rl_engine::SERVE_SETS; ++q) {
        if (e->snap_ev[q]) (void)hipEventDestroy(e->snap_ev[q]);
        if (e->resp_snap[q]) (void)hipFree(e->resp_snap[q]);
        if (e->d_resp_set[q]) (void)hipFree(e->d_resp_set[q]);
        if (e->lazy[q].stream) {
            (void)hipStreamSynchronize(e->lazy[q].stream);
            (void)hipStreamDestroy(e->lazy[q].stream);
        }
        if (e->lazy[q].built) (void)hipEventDestroy(e->lazy[q].built);
    }
    for (auto& evs : e->resp_ev)
      for (hipEvent_t ev : evs)
        if (ev) (void)hipEventDestroy(ev);
    if (e->resp_off_ev) (void)hipEventDestroy(e->resp_off_ev);
    if (e->h_serve) (void)hipHostFree(e->h_serve);
    if (e->h_gen_word) (void)hipHostFree(e->h_gen_word);
    if (e->d_m_scan1) (void)hipFree(e->d_m_scan1);
    for (auto& ev : e->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& f : e->inflight) {
        for (auto& ev : f.tev)
            if (ev) (void)hipEventDestroy(ev);
        if (f.h_st) (void)hipHostFree(f.h_st);
    }
    for (auto& ev : e->ev_parted)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : e->ev_applied)
        if (ev) (void)hipEventDestroy(ev);
    if (e->ev_match) (void)hipEventDestroy(e->ev_match);
    if (e->own_pstream) (void)hipStreamDestroy(e->own_pstream);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
}

const char* rl_last_error(const rl_engine* e) { return e ? e->err.c_str() : "null engine"; }

int32_t rl_status_is_transient(int32_t status) { return (status == RL_ERR_DEVICE || status == RL_ERR_BUSY) ? 1 : 0; }

int32_t rl_stats(rl_engine* e, rl_stats_t* out) try {
    if (!e || !out) return RL_ERR_INVALID;
    EngineLock g(e);
    e->stats.capacity_cells = e->cap;
    e->stats.live_cells = e->live;
    e->stats.tombstones = e->tombs;
    *out = e->stats;
    return RL_OK;
} RL_ABI_CATCH

void* rl_engine_stream(rl_engine* e) { return e ? (void*)e->stream : nullptr; }

int32_t rl_engine_set_stream(rl_engine* e, void* stream, int32_t external) try {
    if (!e) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (e->own_pstream) HIP_TRY(e, hipStreamSynchronize(e->own_pstream));
    e->stream = external ? (hipStream_t)stream : e->own_stream;  // NULL is a stream too: the default one
    e->external_stream = external != 0;
    // with a caller's stream everything is enqueued there, in order (no overlap of consecutive batches)
    e->pstream = (external || !e->own_pstream) ? e->stream : e->own_pstream;
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_engine_wait_event(rl_engine* e, void* event) try {
    if (!e || !event) return RL_ERR_INVALID;
    EngineLock g(e);
    HIP_TRY(e, hipSetDevice(e->device));
    if (e->pstream != e->stream) {
        // Two streams: the event gates the NEXT batch's inputs, and the first kernel that reads them is its partition
        // (the replay waits for the partition anyway).  Making the replay stream wait as well would hold back the
        // replay of the batch BEFORE it, which goes out with the next submit and has nothing to do with these inputs.
        if (e->input_event) {  // (one already waiting: it gates everything, the old way)
            HIP_TRY(e, hipStreamWaitEvent(e->stream, e->input_event, 0));
            HIP_TRY(e, hipStreamWaitEvent(e->pstream, e->input_event, 0));
        }
        e->input_event = (hipEvent_t)event;
        return RL_OK;
    }
    HIP_TRY(e, hipStreamWaitEvent(e->stream, (hipEvent_t)event, 0));
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_engine_record_event(rl_engine* e, void* event) try {
    if (!e || !event) return RL_ERR_INVALID;
    EngineLock g(e);
    HIP_TRY(e, hipSetDevice(e->device));
    const int rc = flush_pending_apply(e);  // "everything submitted so far" includes a replay that was waiting for the next submit
    if (rc) return rc;
    HIP_TRY(e, hipEventRecord((hipEvent_t)event, e->stream));
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_limits_set(rl_engine* e, uint32_t first, const rl_limit_row* rows, uint32_t n) try {
    if (!e) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    if (n && !rows) return fail(e, RL_ERR_INVALID, "rows is null");
    if ((u64)first + n > e->max_limits) return fail(e, RL_ERR_INVALID, "limit rows [%u,%u) exceed max_limits %u", first, first + n, e->max_limits);
    HIP_TRY(e, hipSetDevice(e->device));
    if (e->h_limits.size() < (size_t)first + n) e->h_limits.resize((size_t)first + n, LimitDev{0, 0});
    for (u32 i = 0; i < n; ++i) {
        e->h_limits[first + i].max_value = rows[i].max_value;
        e->h_limits[first + i].window_us = rows[i].seconds * 1000000ull;
    }
    e->any_zero_window = false;
    for (auto& l : e->h_limits) e->any_zero_window |= (l.window_us == 0);
    if (!e->h_limits.empty()) {
        if (e->resp_stream) HIP_TRY(e, hipStreamSynchronize(e->resp_stream));  // (response kernels of a call still in flight read the rows)
        HIP_TRY(e, hipMemcpyAsync(e->d_limits, e->h_limits.data(), e->h_limits.size() * sizeof(LimitDev),
                                  hipMemcpyHostToDevice, e->stream));
        HIP_TRY(e, hipStreamSynchronize(e->stream));
    }
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_add_counter(rl_engine* e, uint32_t limit, uint64_t key) try {
    if (!e) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    if (!(limit & RL_SIMPLE)) return RL_OK;  // in_memory.rs:39: only limits without variables
    if (RL_LIMIT_ID(limit) >= e->h_limits.size()) return fail(e, RL_ERR_INVALID, "unknown limit id %u", RL_LIMIT_ID(limit));
    HIP_TRY(e, hipSetDevice(e->device));
    CellRow row{key, limit, 0, 0, 0};  // Default: (0, UNIX_EPOCH)
    CellRow* d_row = e->d_row1;
    HIP_TRY(e, hipMemcpyAsync(d_row, &row, sizeof(row), hipMemcpyHostToDevice, e->stream));
    return insert_rows_locked(e, d_row, 1, 0);
} RL_ABI_CATCH

int32_t rl_check_and_update_batch_device(rl_engine* e, const rl_hit* d_hits, uint32_t n_hits,
                                         const uint32_t* d_req_off, uint32_t n_req, uint64_t now_us,
                                         int32_t load_counters, uint8_t* d_verdict,
                                         int32_t* d_first_limited, uint64_t* d_remaining,
                                         uint64_t* d_expires_in_us) try {
    int rc = validate_batch(e, d_hits, n_hits, d_req_off, n_req, d_verdict);
    if (rc) return rc;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    if (n_hits == 0) return RL_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    if (load_counters && (!d_remaining || !d_expires_in_us))
        return fail(e, RL_ERR_INVALID, "load_counters needs remaining and expires_in_us buffers");
    if (d_req_off || load_counters || engine_has_peers(e))
        return run_check_general(e, GenCall{reinterpret_cast<const Hit*>(d_hits), n_hits, d_req_off, n_req, nullptr, now_us,
                                            load_counters != 0, false, d_verdict, d_first_limited,
                                            reinterpret_cast<u64*>(d_remaining), reinterpret_cast<u64*>(d_expires_in_us)});
    rc = run_check_k1(e, reinterpret_cast<const Hit*>(d_hits), n_hits, now_us, d_verdict, d_first_limited);
    // The batch is known to be applied as soon as its completion word is seen; a BLOCKING call also
    // promises that the verdicts can be read from any stream, i.e. that the kernel has ended (with the
    // caller's own stream set, stream order is the caller's synchronisation, as documented).
    if (!rc && !e->external_stream) HIP_TRY(e, hipStreamSynchronize(e->stream));
    return rc;
} RL_ABI_CATCH

// Host-buffer form of check_and_update for one clock value (the caller holds the engine's mutex).
static int32_t check_batch_locked(rl_engine* e, const rl_hit* hits, uint32_t n_hits, const uint32_t* req_off,
                                  uint32_t n_req, const uint64_t* req_delta, uint64_t now_us, int32_t load_counters,
                                  uint8_t* verdict, int32_t* first_limited, uint64_t* remaining, uint64_t* expires_in_us) {
    int rc = RL_OK;
    if (n_req == 0) return RL_OK;
    {
        // A call the one-launch kernels take (k_bkt_tiny / k_gen_tiny): no copy commands at all.  The
        // request is staged in host-mapped memory the kernel reads directly, the results land there too
        // and are complete when the kernel's completion word is (both kernels are waited for by polling).
        const bool cr = engine_has_peers(e);  // (cells with peer parts: the general resolver only, see engine_has_peers)
        const bool general = req_off || load_counters || req_delta || cr;
        const bool one_launch = !cr && (general ? (n_hits && n_hits <= e->gen_tiny_max && n_req <= GT_MAX_REQ)
                                                : (n_hits && n_hits <= e->tiny_max));
        // ONE request of a few counters — the trait's per-request call: no launch at all when a server is lingering
        // (k_gen_serve).  Only while the table has room to spare: growing or refusing is the ordinary path's business.
        // (a single-counter request is a request of one counter: the general body is exact for it too)
        // Micro-batches of up to 64 hits / 64 requests ride the same way (verdict and first_limited then come back in one
        // tagged 8-byte slot per request).
        if (!cr && e->serve_enabled && e->h_tiny_coherent && !e->external_stream && n_req >= 1 && n_req <= SRV_MAX_HITS && n_hits >= 1 &&
            n_hits <= SRV_MAX_HITS && n_hits <= e->gen_tiny_max && (n_req == 1 || general) &&
            e->live + e->tombs + n_hits <= e->cap - e->cap / 4) {
            ServeBox* b = e->h_serve;
            Hit* t_hits = reinterpret_cast<Hit*>(e->h_tiny + TIO_OFF_HITS);
            u32* t_off = reinterpret_cast<u32*>(e->h_tiny + TIO_OFF_REQ);
            u64* t_delta = reinterpret_cast<u64*>(e->h_tiny + TIO_OFF_DELTA);
            memcpy(t_hits, hits, (size_t)n_hits * sizeof(Hit));
            if (n_req > 1) {
                if (req_off) memcpy(t_off, req_off, ((size_t)n_req + 1) * sizeof(u32));
                else
                    for (u32 r = 0; r <= n_req; ++r) t_off[r] = r;
                if (req_delta) memcpy(t_delta, req_delta, (size_t)n_req * sizeof(u64));
            }
            const u32 seq = ++e->gen_seq ? e->gen_seq : ++e->gen_seq;
            b->now = now_us;
            b->delta = req_delta ? req_delta[0] : 0ull;
            b->cmd[1] = n_hits;
            b->cmd[2] = (load_counters ? SRV_LOAD : 0u) | (req_delta ? SRV_DELTA : 0u);
            b->cmd[3] = n_req;
            __atomic_store_n(&b->cmd[0], seq, __ATOMIC_RELEASE);
            const volatile u32* done = &e->h_status->n_removed;
            const auto t_start = std::chrono::steady_clock::now();
            bool launched = false;  // (at most one launch per request: a server started FOR this command takes it at once,
                                    // and the `gone` word that sent us here still says this request's number)
            for (u64 spins = 0;; ++spins) {
                if (!e->serve_live) {  // nobody is there (yet, or any more): a server that starts at this very command
                    k_gen_serve<<<1, 256, 0, e->stream>>>(e->table, e->log2cap, e->seed, e->d_limits, (u32)e->h_limits.size(), t_hits,
                                                          t_off, t_delta, b, e->h_status, seq,
                                                          e->serve_linger_us * 100u);  // (100 MHz clock)
                    HIP_TRY(e, hipGetLastError());
                    e->serve_live = true;
                    launched = true;
                    e->n_serve_launches++;
                }
                if (__atomic_load_n(done, __ATOMIC_ACQUIRE) == seq) break;
                if (!launched && __atomic_load_n(&b->gone[3], __ATOMIC_ACQUIRE) == seq) {  // it left without taking this one
                    e->serve_live = false;
                    continue;
                }
                __builtin_ia32_pause();
                if ((spins & 0xFFFFu) == 0xFFFFu) {
                    if (std::chrono::steady_clock::now() - t_start > std::chrono::milliseconds(e->serve_timeout_ms)) {
                        e->serve_live = false;  // (whatever is there is no server of ours any more)
                        return fail(e, RL_ERR_DEVICE, "k_gen_serve did not answer request %u within %u ms (cmd %u/%u/%u, done %u, gone %u, %llu calls, %llu launches, stream %s)",
                                    seq, e->serve_timeout_ms, b->cmd[0], b->cmd[1], b->cmd[2], e->h_status->n_removed, b->gone[3],
                                    (unsigned long long)e->n_serve_calls, (unsigned long long)e->n_serve_launches,
                                    hipStreamQuery(e->stream) == hipSuccess ? "idle" : "busy");
                    }
                    std::this_thread::yield();
                }
            }
            e->n_serve_calls++;
            const u32 w0 = e->h_status->err, dropped = e->h_status->n_ord, created = e->h_status->n_inserted;
            e->live += created;
            if (w0 & 0xFFu) return status_to_error(e, w0 & 0xFFu);
            e->live -= dropped;
            e->tombs += dropped;
            e->stats.batches++;
            e->stats.hits += n_hits;
            e->stats.ordered_hits += n_hits;
            e->stats.ordered_batches++;
            if (n_req == 1) {
                verdict[0] = (uint8_t)((w0 >> 8) & 1u);
                if (first_limited) first_limited[0] = (int32_t)((w0 >> 16) & 0xFFFFu) - 1;
            } else {
                for (u32 r = 0; r < n_req; ++r) {  // every slot says itself which command it answers
                    const volatile u32* sl = b->rslot[r];
                    for (u64 spins = 0; __atomic_load_n(&sl[1], __ATOMIC_ACQUIRE) != seq; ++spins) {
                        __builtin_ia32_pause();
                        if ((spins & 0xFFFFFu) == 0xFFFFFu && std::chrono::steady_clock::now() - t_start > std::chrono::seconds(60))
                            return fail(e, RL_ERR_DEVICE, "k_gen_serve: a request's slot did not arrive within 60 s");
                    }
                    const u32 w = sl[0];
                    verdict[r] = (uint8_t)(w & 1u);
                    if (first_limited) first_limited[r] = (int32_t)(w >> 1) - 1;
                }
            }
            if (load_counters) {
                for (u32 j = 0; j < 2 * n_hits; ++j) {  // every slot says itself which request it answers
                    const volatile u32* sl = b->slot[j];
                    for (u64 spins = 0; __atomic_load_n(&sl[2], __ATOMIC_ACQUIRE) != seq || sl[3] != j; ++spins) {
                        __builtin_ia32_pause();
                        if ((spins & 0xFFFFFu) == 0xFFFFFu && std::chrono::steady_clock::now() - t_start > std::chrono::seconds(60))
                            return fail(e, RL_ERR_DEVICE, "k_gen_serve: a result slot did not arrive within 60 s");
                    }
                    const u64 val = ((u64)sl[1] << 32) | sl[0];
                    if (j & 1u) expires_in_us[j >> 1] = val;
                    else remaining[j >> 1] = val;
                }
            }
            return RL_OK;
        }
        serve_stop(e);
        if (one_launch && n_hits <= TIO_HITS && n_req <= TIO_HITS) {
            Hit* t_hits = reinterpret_cast<Hit*>(e->h_tiny + TIO_OFF_HITS);
            u32* t_off = reinterpret_cast<u32*>(e->h_tiny + TIO_OFF_REQ);
            uint8_t* t_verdict = e->h_tiny + TIO_OFF_VERDICT;
            int32_t* t_first = reinterpret_cast<int32_t*>(e->h_tiny + TIO_OFF_FIRST);
            u64* t_rem = reinterpret_cast<u64*>(e->h_tiny + TIO_OFF_REM);
            u64* t_exp = reinterpret_cast<u64*>(e->h_tiny + TIO_OFF_EXP);
            memcpy(t_hits, hits, (size_t)n_hits * sizeof(Hit));
            if (req_off) memcpy(t_off, req_off, ((size_t)n_req + 1) * sizeof(u32));
            u64* t_delta = reinterpret_cast<u64*>(e->h_tiny + TIO_OFF_DELTA);
            if (req_delta) memcpy(t_delta, req_delta, (size_t)n_req * sizeof(u64));
            if (general) {
                GenCall gc{t_hits, n_hits, req_off ? t_off : nullptr, n_req, req_delta ? t_delta : nullptr, now_us, load_counters != 0,
                           false, t_verdict, t_first, t_rem, t_exp};
                gc.host_mapped_results = true;
                rc = run_check_general(e, gc);
            } else {
                rc = run_check_k1(e, t_hits, n_hits, now_us, t_verdict, first_limited ? t_first : nullptr);
            }
            if (rc) return rc;
            // The completion word has been seen; the RESULTS are read only after the kernel has ended
            // (its end-of-kernel release makes every store host-visible whatever the memory's caching).
            HIP_TRY(e, hipStreamSynchronize(e->stream));
            memcpy(verdict, t_verdict, n_req);
            if (first_limited) memcpy(first_limited, t_first, (size_t)n_req * sizeof(int32_t));
            if (load_counters) {
                memcpy(remaining, t_rem, (size_t)n_hits * sizeof(u64));
                memcpy(expires_in_us, t_exp, (size_t)n_hits * sizeof(u64));
            }
            return RL_OK;
        }
    }
    const bool general_path = req_off || load_counters || req_delta || engine_has_peers(e);
    if (n_hits)  // (the single-counter path reads the batch on the partition stream first)
        HIP_TRY(e, hipMemcpyAsync(e->d_hits, hits, (size_t)n_hits * sizeof(Hit), hipMemcpyHostToDevice,
                                  general_path ? e->stream : e->pstream));
    if (general_path) {
        if (req_delta)
            HIP_TRY(e, hipMemcpyAsync(e->d_req_delta, req_delta, (size_t)n_req * sizeof(u64), hipMemcpyHostToDevice, e->stream));
        if (req_off) {
            HIP_TRY(e, hipMemcpyAsync(e->d_req_off, req_off, ((size_t)n_req + 1) * sizeof(u32), hipMemcpyHostToDevice,
                                      e->stream));
        }
        rc = run_check_general(e, GenCall{e->d_hits, n_hits, req_off ? e->d_req_off : nullptr, n_req,
                                          req_delta ? e->d_req_delta : nullptr, now_us, load_counters != 0, false, e->d_verdict,
                                          e->d_first, e->d_remaining, e->d_expires});
    } else {
        rc = run_check_k1(e, e->d_hits, n_hits, now_us, e->d_verdict, first_limited ? e->d_first : nullptr);
    }
    if (rc) return rc;
    HIP_TRY(e, hipMemcpyAsync(verdict, e->d_verdict, n_req, hipMemcpyDeviceToHost, e->stream));
    if (first_limited)
        HIP_TRY(e, hipMemcpyAsync(first_limited, e->d_first, (size_t)n_req * sizeof(int32_t), hipMemcpyDeviceToHost,
                                  e->stream));
    if (load_counters && n_hits) {
        HIP_TRY(e, hipMemcpyAsync(remaining, e->d_remaining, (size_t)n_hits * sizeof(u64), hipMemcpyDeviceToHost,
                                  e->stream));
        HIP_TRY(e, hipMemcpyAsync(expires_in_us, e->d_expires, (size_t)n_hits * sizeof(u64), hipMemcpyDeviceToHost,
                                  e->stream));
    }
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return RL_OK;
}

int32_t rl_check_and_update_batch_ex(rl_engine* e, const rl_hit* hits, uint32_t n_hits, const uint32_t* req_off,
                                     uint32_t n_req, const uint64_t* req_delta, const uint64_t* req_now_us, uint64_t now_us,
                                     int32_t load_counters, uint8_t* verdict, int32_t* first_limited, uint64_t* remaining,
                                     uint64_t* expires_in_us) try {
    int rc = validate_batch(e, hits, n_hits, req_off, n_req, verdict);
    if (rc) return rc;
    EngineLock g(e, /*keep_server=*/true);  // (check_batch_locked sends the server away unless the call is one for it)
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    if (n_req == 0) return RL_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    if (load_counters && (!remaining || !expires_in_us))
        return fail(e, RL_ERR_INVALID, "load_counters needs remaining and expires_in_us buffers");
    if (n_req > e->max_batch) return fail(e, RL_ERR_BATCH_TOO_LARGE, "n_req %u > max_batch_hits %u", n_req, e->max_batch);
    if (req_off) {
        if (req_off[0] != 0 || req_off[n_req] != n_hits) return fail(e, RL_ERR_INVALID, "req_off must start at 0 and end at n_hits");
        for (u32 r = 0; r < n_req; ++r)
            if (req_off[r] > req_off[r + 1]) return fail(e, RL_ERR_INVALID, "req_off must be non-decreasing");
    }
    if (!req_now_us)
        return check_batch_locked(e, hits, n_hits, req_off, n_req, req_delta, now_us, load_counters, verdict, first_limited,
                                  remaining, expires_in_us);
    // One clock value per request (in_memory.rs:83 reads the clock once per call): the batch is applied as
    // consecutive runs of requests that share a clock value, each run one device batch.  (A run that fails
    // leaves the runs before it applied: the requests are sequential calls of the reference.)
    std::vector<u32> off;
    for (u32 r0 = 0; r0 < n_req;) {
        u32 r1 = r0 + 1;
        while (r1 < n_req && req_now_us[r1] == req_now_us[r0]) ++r1;
        const u32 h0 = req_off ? req_off[r0] : r0, h1 = req_off ? req_off[r1] : r1;
        const u32* run_off = nullptr;
        if (req_off) {
            off.resize(r1 - r0 + 1);
            for (u32 r = r0; r <= r1; ++r) off[r - r0] = req_off[r] - h0;
            run_off = off.data();
        }
        rc = check_batch_locked(e, hits + h0, h1 - h0, run_off, r1 - r0, req_delta ? req_delta + r0 : nullptr, req_now_us[r0],
                                load_counters, verdict + r0, first_limited ? first_limited + r0 : nullptr,
                                remaining ? remaining + h0 : nullptr, expires_in_us ? expires_in_us + h0 : nullptr);
        if (rc) return rc;
        if (first_limited)
            for (u32 r = r0; r < r1; ++r)
                if (first_limited[r] >= 0) first_limited[r] += (int32_t)h0;  // index in the caller's batch
        r0 = r1;
    }
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_check_and_update_batch(rl_engine* e, const rl_hit* hits, uint32_t n_hits, const uint32_t* req_off,
                                  uint32_t n_req, uint64_t now_us, int32_t load_counters, uint8_t* verdict,
                                  int32_t* first_limited, uint64_t* remaining, uint64_t* expires_in_us) try {
    return rl_check_and_update_batch_ex(e, hits, n_hits, req_off, n_req, nullptr, nullptr, now_us, load_counters, verdict,
                                        first_limited, remaining, expires_in_us);
} RL_ABI_CATCH

int32_t rl_check_and_update_submit_device_ev(rl_engine* e, const rl_hit* d_hits, uint32_t n_hits, uint64_t now_us,
                                             uint8_t* d_verdict, int32_t* d_first_limited, void* done_event) try {
    int rc = validate_batch(e, d_hits, n_hits, nullptr, n_hits, d_verdict);
    if (rc) return rc;
    EngineLock g(e);
    if (n_hits == 0) return fail(e, RL_ERR_INVALID, "empty batch");
    if (e->ph_open) return fail(e, RL_ERR_BUSY, "a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    if (engine_has_peers(e))
        return fail(e, RL_ERR_INVALID, "this engine holds peer state (rl_merge_cells): its counters follow CrCounterValue and are "
                                       "served by the blocking entries (rl_check_and_update_batch*), not by the pipelined hot path");
    HIP_TRY(e, hipSetDevice(e->device));
    e->submit_done_event = (hipEvent_t)done_event;
    rc = submit_k1_bucketed(e, reinterpret_cast<const Hit*>(d_hits), n_hits, now_us, d_verdict, d_first_limited);
    e->submit_done_event = nullptr;
    return rc;
} RL_ABI_CATCH

int32_t rl_check_and_update_submit_device(rl_engine* e, const rl_hit* d_hits, uint32_t n_hits, uint64_t now_us,
                                          uint8_t* d_verdict, int32_t* d_first_limited) try {
    return rl_check_and_update_submit_device_ev(e, d_hits, n_hits, now_us, d_verdict, d_first_limited, nullptr);
} RL_ABI_CATCH

int32_t rl_engine_flush(rl_engine* e) try {
    if (!e) return RL_ERR_INVALID;
    EngineLock g(e);
    HIP_TRY(e, hipSetDevice(e->device));
    return flush_pending_apply(e);
} RL_ABI_CATCH

int32_t rl_check_and_update_collect(rl_engine* e) try {
    if (!e) return RL_ERR_INVALID;
    EngineLock g(e);
    HIP_TRY(e, hipSetDevice(e->device));
    return collect_k1_bucketed(e);
} RL_ABI_CATCH

int32_t rl_is_within_limits_batch_ex(rl_engine* e, const rl_hit* hits, uint32_t n_hits, const uint64_t* delta,
                                     uint64_t now_us, uint8_t* within) try {
    int rc = validate_batch(e, hits, n_hits, nullptr, n_hits, within);
    if (rc) return rc;
    EngineLock g(e);
    if (engine_busy_for_reads(e)) return fail(e, RL_ERR_BUSY, "a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    if (n_hits == 0) return RL_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    rc = flush_pending_apply(e);  // behind every batch submitted so far
    if (rc) return rc;
    HIP_TRY(e, hipMemcpyAsync(e->d_hits, hits, (size_t)n_hits * sizeof(Hit), hipMemcpyHostToDevice, e->stream));
    if (delta) HIP_TRY(e, hipMemcpyAsync(e->d_req_delta, delta, (size_t)n_hits * sizeof(u64), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipMemsetAsync(e->d_status, 0, sizeof(Status), e->stream));
    k_within<<<cdiv(n_hits, 256), 256, 0, e->stream>>>(e->table, e->log2cap, e->seed, e->d_hits, n_hits,
                                                        delta ? e->d_req_delta : nullptr, e->d_limits,
                                                        (u32)e->h_limits.size(), now_us, e->d_verdict, e->d_status);
    HIP_TRY(e, hipGetLastError());
    HIP_TRY(e, hipMemcpyAsync(within, e->d_verdict, n_hits, hipMemcpyDeviceToHost, e->stream));
    rc = read_status(e);
    if (rc) return rc;
    if (e->h_status->err) return status_to_error(e, e->h_status->err);
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_is_within_limits_batch(rl_engine* e, const rl_hit* hits, uint32_t n_hits, uint64_t now_us, uint8_t* within) try {
    return rl_is_within_limits_batch_ex(e, hits, n_hits, nullptr, now_us, within);
} RL_ABI_CATCH

int32_t rl_update_counter_batch_ex(rl_engine* e, const rl_hit* hits, uint32_t n_hits, const uint64_t* delta,
                                   uint64_t now_us) try {
    uint8_t dummy = 0;
    int rc = validate_batch(e, hits, n_hits, nullptr, n_hits, &dummy);
    if (rc) return rc;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    if (n_hits == 0) return RL_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    // the general resolver without the limit test: every hit is admitted (in_memory.rs:47-69 never checks)
    HIP_TRY(e, hipMemcpyAsync(e->d_hits, hits, (size_t)n_hits * sizeof(Hit), hipMemcpyHostToDevice, e->stream));
    if (delta) HIP_TRY(e, hipMemcpyAsync(e->d_req_delta, delta, (size_t)n_hits * sizeof(u64), hipMemcpyHostToDevice, e->stream));
    return run_check_general(e, GenCall{e->d_hits, n_hits, nullptr, n_hits, delta ? e->d_req_delta : nullptr, now_us, false, true,
                                        e->d_verdict, nullptr, nullptr, nullptr});
} RL_ABI_CATCH

int32_t rl_update_counter_batch(rl_engine* e, const rl_hit* hits, uint32_t n_hits, uint64_t now_us) try {
    return rl_update_counter_batch_ex(e, hits, n_hits, nullptr, now_us);
} RL_ABI_CATCH

int32_t rl_get_counters(rl_engine* e, uint32_t limit, uint64_t now_us, rl_cell_row* out, uint64_t cap,
                        uint64_t* n_out) try {
    if (!e || (cap && !out)) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy_for_reads(e)) return fail(e, RL_ERR_BUSY, "a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    const int rc = flush_pending_apply(e);  // behind every batch submitted so far
    if (rc) return rc;
    return scan_locked<SCAN_GET>(e, limit, now_us, out, cap, n_out);
} RL_ABI_CATCH

int32_t rl_dump_cells(rl_engine* e, rl_cell_row* out, uint64_t cap, uint64_t* n_out) try {
    if (!e || (cap && !out)) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    return scan_locked<SCAN_DUMP>(e, 0, 0, out, cap, n_out);
} RL_ABI_CATCH

int32_t rl_delete_counters(rl_engine* e, uint32_t limit) try {
    if (!e) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    return scan_locked<SCAN_DELETE_LIMIT>(e, limit, 0, nullptr, 0, nullptr);
} RL_ABI_CATCH

int32_t rl_clear(rl_engine* e) try {
    if (!e) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    return scan_locked<SCAN_CLEAR_SIMPLE>(e, 0, 0, nullptr, 0, nullptr);
} RL_ABI_CATCH

int32_t rl_sweep_expired_rows(rl_engine* e, uint64_t now_us, rl_cell_row* out, uint64_t cap, uint64_t* n_removed) try {
    if (!e || (cap && !out)) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    const u64 before = e->live;
    int rc = scan_locked<SCAN_SWEEP>(e, 0, now_us, out, cap, nullptr);
    if (rc) return rc;
    if (n_removed) *n_removed = before - e->live;
    if (e->tombs > e->cap / 8) return do_compact(e, 0);
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_sweep_expired(rl_engine* e, uint64_t now_us, uint64_t* n_removed) try {
    return rl_sweep_expired_rows(e, now_us, nullptr, 0, n_removed);
} RL_ABI_CATCH

int32_t rl_sweep_expired_submit(rl_engine* e, uint64_t now_us) try {
    if (!e) return RL_ERR_INVALID;
    EngineLock g(e);
    if (e->ph_open) return fail(e, RL_ERR_BUSY, "a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    if (e->sub_seq - e->col_seq >= 3) return fail(e, RL_ERR_BUSY, "three commands are already in flight: collect one first");
    HIP_TRY(e, hipSetDevice(e->device));
    int rc = flush_pending_apply(e);  // the sweep runs behind every batch submitted so far, in front of every later one
    if (rc) return rc;
    rl_engine::Inflight& f = e->inflight[e->sub_seq & 3u];
    Status* ds = e->d_sweep_st + (e->sub_seq & 3u);
    HIP_TRY(e, hipMemsetAsync(ds, 0, sizeof(Status), e->stream));
    k_scan<SCAN_SWEEP><<<2048, 256, 0, e->stream>>>(e->table, e->cap, 0u, now_us, (CellRow*)nullptr, 0ull, ds, e->d_total);
    k_post_status<<<1, 64, 0, e->stream>>>(ds, f.h_st, (u32)(e->sub_seq + 1));
    HIP_TRY(e, hipGetLastError());
    f.kind = 1;
    f.n = 0;
    f.n_wg = 0;
    f.ntiles = 0;
    f.timed = 0;
    f.seq = (u32)(e->sub_seq + 1);
    f.settled = false;
    e->last_k1_was_part = false;
    e->sub_seq++;
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_sweep_expired_collect(rl_engine* e, uint64_t* n_removed) try {
    if (!e) return RL_ERR_INVALID;
    EngineLock g(e);
    if (e->sub_seq == e->col_seq) return fail(e, RL_ERR_INVALID, "nothing in flight");
    if (e->inflight[e->col_seq & 3u].kind != 1) return fail(e, RL_ERR_INVALID, "the oldest command in flight is a batch: rl_check_and_update_collect");
    HIP_TRY(e, hipSetDevice(e->device));
    const int rc = collect_k1_bucketed(e);
    if (n_removed) *n_removed = e->last_sweep_removed;
    return rc;
} RL_ABI_CATCH

int32_t rl_compact(rl_engine* e) try {
    if (!e) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    HIP_TRY(e, hipSetDevice(e->device));
    return do_compact(e, 0);
} RL_ABI_CATCH

int32_t rl_resize(rl_engine* e, uint64_t capacity_cells) try {
    if (!e) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    HIP_TRY(e, hipSetDevice(e->device));
    const u32 lg = ceil_log2(capacity_cells < 1024 ? 1024 : capacity_cells);
    if (lg > 31) return fail(e, RL_ERR_INVALID, "capacity %llu beyond 2^31 cells", (unsigned long long)capacity_cells);
    if (e->live > (1ull << lg) / 2)
        return fail(e, RL_ERR_INVALID, "%llu live cells would fill a %llu-cell table beyond one half",
                    (unsigned long long)e->live, (unsigned long long)(1ull << lg));
    return do_compact(e, lg);
} RL_ABI_CATCH

int32_t rl_load_cells_device(rl_engine* e, const rl_cell_row* d_rows, uint64_t n) try {
    if (!e || (n && !d_rows)) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    HIP_TRY(e, hipSetDevice(e->device));
    if (n == 0) return RL_OK;
    return insert_rows_locked(e, reinterpret_cast<const CellRow*>(d_rows), n, 1);
} RL_ABI_CATCH

int32_t rl_load_cells(rl_engine* e, const rl_cell_row* rows, uint64_t n) try {
    if (!e || (n && !rows)) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    HIP_TRY(e, hipSetDevice(e->device));
    if (n == 0) return RL_OK;
    CellRow* d_rows = nullptr;
    if (hipMalloc((void**)&d_rows, n * sizeof(CellRow)) != hipSuccess) return fail(e, RL_ERR_NOMEM, "hipMalloc failed");
    hipError_t r = hipMemcpy(d_rows, rows, n * sizeof(CellRow), hipMemcpyHostToDevice);
    int rc = r == hipSuccess ? insert_rows_locked(e, d_rows, n, 1)
                             : fail(e, RL_ERR_DEVICE, "hipMemcpy failed: %s", hipGetErrorString(r));
    (void)hipFree(d_rows);
    return rc;
} RL_ABI_CATCH

// ---- snapshot files, cross-node merge -----------------------------------------
namespace {
struct SnapshotHeader {
    char magic[8];  // "RLSNAP02"
    u64 n_cells, n_limits, hash_seed, reserved;  // reserved: fingerprint of the hash key of the hashed cells (0: none)
};
}  // namespace

int32_t rl_snapshot_save(rl_engine* e, const char* path) try {
    if (!e || !path) return RL_ERR_INVALID;
    uint64_t n = 0;
    int rc = rl_dump_cells(e, nullptr, 0, &n);
    if (rc) return rc;
    std::vector<rl_cell_row> rows(n);
    if (n) {
        rc = rl_dump_cells(e, rows.data(), n, &n);
        if (rc) return rc;
        rows.resize(n);
    }
    EngineLock g(e);
    FILE* f = fopen(path, "wb");
    if (!f) return fail(e, RL_ERR_INVALID, "cannot open %s for writing", path);
    SnapshotHeader h{};
    memcpy(h.magic, "RLSNAP02", 8);
    h.n_cells = rows.size();
    h.n_limits = e->h_limits.size();
    h.hash_seed = e->seed;
    h.reserved = e->wire_key_fp;  // (fingerprint of the hash key the hashed cells were named under; 0: none)
    std::vector<rl_limit_row> lim(e->h_limits.size());
    for (size_t i = 0; i < lim.size(); ++i) lim[i] = rl_limit_row{e->h_limits[i].max_value, e->h_limits[i].window_us / 1000000ull};
    bool ok = fwrite(&h, sizeof h, 1, f) == 1 && (lim.empty() || fwrite(lim.data(), sizeof(rl_limit_row), lim.size(), f) == lim.size()) &&
              (rows.empty() || fwrite(rows.data(), sizeof(rl_cell_row), rows.size(), f) == rows.size());
    ok = fclose(f) == 0 && ok;
    return ok ? RL_OK : fail(e, RL_ERR_INVALID, "short write to %s", path);
} RL_ABI_CATCH

int32_t rl_snapshot_load(rl_engine* e, const char* path) try {
    if (!e || !path) return RL_ERR_INVALID;
    FILE* f = fopen(path, "rb");
    if (!f) return fail(e, RL_ERR_INVALID, "cannot open %s", path);
    SnapshotHeader h{};
    std::vector<rl_limit_row> lim;
    std::vector<rl_cell_row> rows;
    bool ok = fread(&h, sizeof h, 1, f) == 1 && memcmp(h.magic, "RLSNAP02", 8) == 0 && h.n_limits <= (1u << 24) &&
              h.n_cells <= (1ull << 31);
    if (ok) {
        lim.resize(h.n_limits);
        rows.resize(h.n_cells);
        ok = (lim.empty() || fread(lim.data(), sizeof(rl_limit_row), lim.size(), f) == lim.size()) &&
             (rows.empty() || fread(rows.data(), sizeof(rl_cell_row), rows.size(), f) == rows.size());
    }
    fclose(f);
    if (!ok) return fail(e, RL_ERR_INVALID, "%s is not a snapshot of this engine's format", path);
    int rc = lim.empty() ? RL_OK : rl_limits_set(e, 0, lim.data(), (uint32_t)lim.size());
    if (rc) return rc;
    // in chunks: rl_load_cells answers TABLE_FULL (or grows) before anything of a chunk is inserted
    for (size_t lo = 0; lo < rows.size(); lo += (1u << 20)) {
        const size_t m = std::min<size_t>(1u << 20, rows.size() - lo);
        rc = rl_load_cells(e, rows.data() + lo, m);
        if (rc) return rc;
    }
    if (h.reserved) {
        EngineLock g(e);
        e->wire_key_fp = h.reserved;  // (rl_wire_table_set refuses another key while these cells are here)
    }
    return RL_OK;
} RL_ABI_CATCH

static PeerTables peer_tables_of(rl_engine* e) {
    PeerTables p{};
    for (int a = 0; a < MERGE_MAX_ACTORS; ++a) p.t[a] = e->peer_tables[a];
    p.log2cap = e->log2cap;
    return p;
}

int32_t rl_merge_cells(rl_engine* e, uint32_t self_actor, uint32_t actor, const rl_cell_row* rows, uint64_t n, uint64_t now_us) try {
    if (!e || (n && !rows)) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    if (actor >= (u32)MERGE_MAX_ACTORS || self_actor >= (u32)MERGE_MAX_ACTORS)
        return fail(e, RL_ERR_INVALID, "actor ids are 0..%d", MERGE_MAX_ACTORS - 1);
    if (n == 0) return RL_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    int rc = check_room(e, n);
    if (rc) return rc;
    if (actor != self_actor && !e->peer_tables[actor]) {
        rc = alloc_table(e, e->cap, &e->peer_tables[actor]);
        if (rc) return rc;
    }
    CellRow* d_rows = nullptr;
    if (hipMalloc((void**)&d_rows, n * sizeof(CellRow)) != hipSuccess) return fail(e, RL_ERR_NOMEM, "hipMalloc failed");
    hipError_t r = hipMemcpyAsync(d_rows, rows, n * sizeof(CellRow), hipMemcpyHostToDevice, e->stream);
    if (r == hipSuccess) r = hipMemsetAsync(e->d_status, 0, sizeof(Status), e->stream);
    if (r == hipSuccess) {
        k_merge_rows<<<cdiv(n, 256), 256, 0, e->stream>>>(e->table, e->log2cap, e->seed, peer_tables_of(e), actor, self_actor,
                                                          d_rows, n, now_us, e->d_status);
        r = hipGetLastError();
    }
    if (r == hipSuccess) rc = read_status(e);
    (void)hipFree(d_rows);
    if (r != hipSuccess) return fail(e, RL_ERR_DEVICE, "merge failed: %s", hipGetErrorString(r));
    if (rc) return rc;
    e->live += e->h_status->n_inserted;
    if (e->h_status->err) return status_to_error(e, e->h_status->err);
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_export_local(rl_engine* e, uint64_t now_us, rl_cell_row* out, uint64_t cap, uint64_t* n_out) try {
    if (!e || (cap && !out)) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    HIP_TRY(e, hipSetDevice(e->device));
    CellRow* d_out = nullptr;
    if (cap && hipMalloc((void**)&d_out, cap * sizeof(CellRow)) != hipSuccess) return fail(e, RL_ERR_NOMEM, "hipMalloc failed");
    hipError_t r = hipMemsetAsync(e->d_total, 0, sizeof(unsigned long long), e->stream);
    if (r == hipSuccess) {
        k_export_local<<<2048, 256, 0, e->stream>>>(e->table, e->cap, e->seed, peer_tables_of(e), now_us, d_out, cap, e->d_total);
        r = hipGetLastError();
    }
    if (r == hipSuccess) r = hipMemcpyAsync(e->h_total, e->d_total, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream);
    if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
    if (r == hipSuccess && cap) {
        const u64 m = *e->h_total < cap ? *e->h_total : cap;
        if (m) r = hipMemcpy(out, d_out, m * sizeof(CellRow), hipMemcpyDeviceToHost);
    }
    if (d_out) (void)hipFree(d_out);
    if (r != hipSuccess) return fail(e, RL_ERR_DEVICE, "export failed: %s", hipGetErrorString(r));
    if (n_out) *n_out = *e->h_total;
    return RL_OK;
} RL_ABI_CATCH

// ---- on-device limit matching (rl_match.hpp) -------------------------------------------------------
uint64_t rl_match_key(uint32_t limit_id, uint32_t n_vars, uint32_t v0, uint32_t v1) {
    return match_key(limit_id, n_vars, v0, v1);
}

// ---- the phased form of the general resolver ----------------------------------------------------------------
// For hosts that decide admission themselves: requests whose counters live on several GPUs (key-sharded
// multi-counter requests, limitador_amd/sharded.py ShardedMultiCounterEngine) — the per-request AND of
// in_memory.rs:141-153 then spans engines, so every fixpoint round goes through the host.
static int32_t gen_phase_close(rl_engine* e, bool cleared = false) {
    e->ph_open = false;
    e->ph_counted = false;
    if (cleared) return RL_OK;  // (the caller enqueued the clear ahead of the stop it made anyway)
    HIP_TRY(e, hipMemsetAsync(e->d_bs, 0, BS_ROT * sizeof(BatchScratch), e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return RL_OK;
}

int32_t rl_gen_begin_device(rl_engine* e, const rl_hit* d_hits, const uint32_t* d_req_id, uint32_t n_hits, uint64_t now_us,
                            int32_t load_counters) try {
    if (!e || (n_hits && (!d_hits || !d_req_id))) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight or a phased pass is open");
    if (n_hits > e->gen_cap) return fail(e, RL_ERR_BATCH_TOO_LARGE, "n_hits %u > %u: split the slice", n_hits, e->gen_cap);
    HIP_TRY(e, hipSetDevice(e->device));
    e->ph_n = n_hits;
    e->ph_rounds = 0;
    e->ph_counted = false;
    e->ph_unchecked = false;
    if (n_hits == 0) {
        e->ph_open = true;
        return RL_OK;
    }
    if (e->tombs > e->cap / 8) {
        const int crc = do_compact(e, 0);
        if (crc) return crc;
    }
    int rc = check_room(e, 0);
    if (rc) return rc;
    GenCall c{reinterpret_cast<const Hit*>(d_hits), n_hits, nullptr, n_hits, nullptr, now_us, load_counters != 0, false,
              nullptr, nullptr, e->d_remaining, e->d_expires};
    c.d_hit_req_ext = d_req_id;
    for (int attempt = 0;; ++attempt) {
        GenArgs A{};
        BatchScratch* bs = nullptr;
        rc = gen_setup_and_sort(e, c, 0, n_hits, 0, n_hits, !c.load, A, &bs);
        if (rc) return rc;
        if (e->ph_async && e->external_stream) {
            // rl_gen_set_async: the sort's outcome — an error bit, buckets that overflowed — is looked at by rl_gen_count_device,
            // which stops for the device anyway; every kernel of the rounds in between returns at once on either (they
            // read the same status words), so nothing is computed from an unusable sort and nothing is ever applied.
            e->ph_A = A;
            e->ph_bs = bs;
            e->ph_unchecked = true;
            break;
        }
        Status h_bst;
        GenStatus h_gst;
        HIP_TRY(e, hipMemcpyAsync(&h_gst, e->d_gst, offsetof(GenStatus, changed), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(e, hipMemcpyAsync(&h_bst, &bs->st, sizeof(Status), hipMemcpyDeviceToHost, e->stream));
        // (keys that qualified for the set this sort picked: no kernel has put the count into the status block yet)
        if (A.hot_next) HIP_TRY(e, hipMemcpyAsync(&h_gst.hot_n, &A.hot_next->n, sizeof(u32), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(e, hipStreamSynchronize(e->stream));
        if (h_bst.err | h_gst.err) {
            HIP_TRY(e, hipMemsetAsync(e->d_bs, 0, BS_ROT * sizeof(BatchScratch), e->stream));
            return status_to_error(e, h_bst.err | h_gst.err);
        }
        if (h_gst.overflow) {  // the heavy keys were promoted: partition again with that set
            e->part_seq += 1;
            adapt_hot_threshold(e, h_gst.hot_n, true);
            HIP_TRY(e, hipMemsetAsync(e->d_bs, 0, BS_ROT * sizeof(BatchScratch), e->stream));
            if (attempt >= 2)
                return fail(e, RL_ERR_BATCH_TOO_LARGE, "a hash bucket of this slice holds more than %d hits or %d cells: split the slice", GS_LONG_MAX, GS_E);
            continue;
        }
        e->ph_A = A;
        e->ph_bs = bs;
        break;
    }
    e->ph_open = true;
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_gen_round_device(rl_engine* e, const uint8_t* d_admitted, uint8_t* d_pass, uint64_t* d_remaining,
                            uint64_t* d_expires_in_us) try {
    if (!e) return RL_ERR_INVALID;
    EngineLock g(e);
    if (!e->ph_open) return fail(e, RL_ERR_INVALID, "no phased pass is open");
    if (e->ph_n == 0) return RL_OK;
    if (!d_pass) return fail(e, RL_ERR_INVALID, "d_pass is null");
    HIP_TRY(e, hipSetDevice(e->device));
    GenArgs A = e->ph_A;
    A.admitted_hit = d_admitted;  // null: every hit admitted (the first round)
    A.pass[0] = A.pass[1] = d_pass;
    // (round number 0 below: k_gen_piece_sum fills the caller's flags with 1, k_gen_round stores only the failures — the
    // random byte store per hit is what bounds a round)
    A.pass_prefilled = e->gen_pass_prefill ? 1u : 0u;
    A.remaining = A.load ? reinterpret_cast<u64*>(d_remaining) : nullptr;
    A.expires_in = A.load ? reinterpret_cast<u64*>(d_expires_in_us) : nullptr;
    if (A.load && (!d_remaining || !d_expires_in_us)) return fail(e, RL_ERR_INVALID, "load_counters: remaining / expires_in are null");
    const u32 n = e->ph_n;
    // (round number 0: no pass flags of a previous round are read; the admission comes with the call)
    k_gen_piece_sum<<<cdiv(n, (u32)GS_MAX), GS_BLOCK, 0, e->stream>>>(A, 0u, 0u);
    k_gen_round<<<cdiv(n, (u32)GS_MAX), GS_BLOCK, 0, e->stream>>>(A, 0u, 0u, 1u);
    HIP_TRY(e, hipGetLastError());
    // (a caller that owns the engine's stream — rl_engine_set_stream(…, external) — orders what reads d_pass by that stream or
    // by events recorded on it: the router's rounds no longer stop here, rl_gen_set_async)
    if (!(e->ph_async && e->external_stream)) HIP_TRY(e, hipStreamSynchronize(e->stream));
    e->ph_rounds++;
    e->ph_counted = false;
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_gen_set_async(rl_engine* e, int32_t on) try {
    if (!e) return RL_ERR_INVALID;
    EngineLock g(e);
    e->ph_async = on != 0;
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_gen_count_device(rl_engine* e, const uint8_t* d_reached, uint32_t* n_new, uint64_t* room) try {
    if (!e) return RL_ERR_INVALID;
    EngineLock g(e);
    if (!e->ph_open) return fail(e, RL_ERR_INVALID, "no phased pass is open");
    const u64 used = e->live + e->tombs, bound = e->cap - e->cap / 16;
    if (room) *room = used < bound ? bound - used : 0;
    if (n_new) *n_new = 0;
    e->ph_counted = true;
    if (e->ph_n == 0) return RL_OK;
    if (e->ph_rounds == 0) return fail(e, RL_ERR_INVALID, "rl_gen_round_device has not run");
    HIP_TRY(e, hipSetDevice(e->device));
    GenArgs A = e->ph_A;
    const u32 n = e->ph_n;
    A.reached_hit = d_reached;
    A.mark_reached = d_reached ? 1u : 0u;  // null: every hit was reached (load_counters walks all counters)
    e->ph_A.mark_reached = A.mark_reached;
    if (d_reached) {
        HIP_TRY(e, hipMemsetAsync(e->d_g_reached, 0, n, e->stream));
        k_gen_reach<<<cdiv(n, (u32)GS_MAX), GS_BLOCK, 0, e->stream>>>(A);
    }
    HIP_TRY(e, hipMemsetAsync(&e->d_gst->n_new, 0, sizeof(u32), e->stream));
    k_gen_count<<<std::min(cdiv(n, 256), 1024u), 256, 0, e->stream>>>(A);
    GenStatus h_gst;
    Status h_bst{};
    HIP_TRY(e, hipMemcpyAsync(&h_gst, e->d_gst, offsetof(GenStatus, changed), hipMemcpyDeviceToHost, e->stream));
    if (e->ph_unchecked) HIP_TRY(e, hipMemcpyAsync(&h_bst, &e->ph_bs->st, sizeof(Status), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (e->ph_unchecked) {  // what an async rl_gen_begin_device did not wait for
        e->ph_unchecked = false;
        if (h_bst.err | h_gst.err) {
            (void)gen_phase_close(e);
            return status_to_error(e, h_bst.err | h_gst.err);
        }
        if (h_gst.overflow) {  // the heavy keys were promoted: begin again, with that set
            e->part_seq += 1;
            adapt_hot_threshold(e, h_gst.hot_n, true);
            (void)gen_phase_close(e);
            return fail(e, RL_ERR_BUSY, "hash buckets of this slice overflowed; their heavy keys were promoted: begin the pass again "
                                        "(nothing was applied)");
        }
    }
    adapt_hot_threshold(e, h_gst.hot_n, false);
    if (n_new) *n_new = h_gst.n_new;
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_gen_commit_device(rl_engine* e) try {
    if (!e) return RL_ERR_INVALID;
    EngineLock g(e);
    if (!e->ph_open) return fail(e, RL_ERR_INVALID, "no phased pass is open");
    if (e->ph_n == 0) return gen_phase_close(e);
    if (!e->ph_counted) return fail(e, RL_ERR_INVALID, "rl_gen_count_device must follow the last round");
    HIP_TRY(e, hipSetDevice(e->device));
    GenArgs A = e->ph_A;
    A.update_mode = 1u;  // (k_gen_commit: the caller has seen the fixpoint; the device-side convergence test does not apply)
    const u32 n = e->ph_n;
    const u64 used = e->live + e->tombs, bound = e->cap - e->cap / 16;
    k_gen_commit<<<std::min(cdiv(n, 256), 1024u), 256, 0, e->stream>>>(A, used < bound ? (u32)std::min<u64>(bound - used, 0xFFFFFFFFull) : 0u, nullptr, 0u, 0u);
    GenStatus h_gst;
    HIP_TRY(e, hipMemcpyAsync(&h_gst, e->d_gst, offsetof(GenStatus, changed), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipMemsetAsync(e->d_bs, 0, BS_ROT * sizeof(BatchScratch), e->stream));  // (what closing the pass clears: one stop, not two)
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (!h_gst.committed) {
        (void)gen_phase_close(e, true);
        return fail(e, RL_ERR_TABLE_FULL, "refused, nothing applied: the slice creates %u cells in a table with live=%llu tombstones=%llu capacity=%llu",
                    h_gst.n_new, (unsigned long long)e->live, (unsigned long long)e->tombs, (unsigned long long)e->cap);
    }
    e->live += h_gst.n_new;
    e->part_seq++;
    e->stats.batches++;
    e->stats.hits += n;
    e->stats.ordered_hits += n;
    e->stats.ordered_batches++;
    return gen_phase_close(e, true);
} RL_ABI_CATCH

// The count and the commit of a phased pass WITHOUT the host looking in between (the key-sharded step in async mode): the
// count leaves this rank's veto word on the device, the caller gathers all ranks' words there, the commit is gated on them.
int32_t rl_gen_count_async_device(rl_engine* e, const uint8_t* d_reached, const uint32_t* d_also, uint32_t* d_veto) try {
    if (!e || !d_veto) return RL_ERR_INVALID;
    EngineLock g(e);
    if (!e->ph_open) return fail(e, RL_ERR_INVALID, "no phased pass is open");
    if (!(e->ph_async && e->external_stream)) return fail(e, RL_ERR_INVALID, "rl_gen_set_async on a caller's stream first");
    HIP_TRY(e, hipSetDevice(e->device));
    if (e->ph_n == 0) {
        k_gen_veto<<<1, 64, 0, e->stream>>>(nullptr, nullptr, 0u, d_also, d_veto);
        HIP_TRY(e, hipGetLastError());
        e->ph_counted = true;
        return RL_OK;
    }
    if (e->ph_rounds == 0) return fail(e, RL_ERR_INVALID, "rl_gen_round_device has not run");
    GenArgs A = e->ph_A;
    const u32 n = e->ph_n;
    A.reached_hit = d_reached;
    A.mark_reached = d_reached ? 1u : 0u;
    e->ph_A.mark_reached = A.mark_reached;
    if (d_reached) {
        HIP_TRY(e, hipMemsetAsync(e->d_g_reached, 0, n, e->stream));
        k_gen_reach<<<cdiv(n, (u32)GS_MAX), GS_BLOCK, 0, e->stream>>>(A);
    }
    HIP_TRY(e, hipMemsetAsync(&e->d_gst->n_new, 0, sizeof(u32), e->stream));
    k_gen_count<<<std::min(cdiv(n, 256), 1024u), 256, 0, e->stream>>>(A);
    const u64 used = e->live + e->tombs, bound = e->cap - e->cap / 16;
    k_gen_veto<<<1, 64, 0, e->stream>>>(e->d_gst, &e->ph_bs->st, used < bound ? (u32)std::min<u64>(bound - used, 0xFFFFFFFFull) : 0u, d_also, d_veto);
    HIP_TRY(e, hipGetLastError());
    e->ph_counted = true;
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_gen_commit_gated_device(rl_engine* e, const uint32_t* d_veto, uint32_t* h_veto, uint32_t n_veto, uint32_t veto_stride,
                                   uint32_t* committed) try {
    if (!e || !d_veto || !h_veto || !n_veto || !veto_stride || !committed) return RL_ERR_INVALID;
    EngineLock g(e);
    *committed = 0;
    if (!e->ph_open) return fail(e, RL_ERR_INVALID, "no phased pass is open");
    if (!(e->ph_async && e->external_stream)) return fail(e, RL_ERR_INVALID, "rl_gen_set_async on a caller's stream first");
    if (!e->ph_counted) return fail(e, RL_ERR_INVALID, "rl_gen_count_async_device must follow the last round");
    HIP_TRY(e, hipSetDevice(e->device));
    // every rank's word, for the host — ahead of the kernels whose last store is what the host waits for
    HIP_TRY(e, hipMemcpyAsync(h_veto, d_veto, (size_t)n_veto * veto_stride * sizeof(u32), hipMemcpyDeviceToHost, e->stream));
    auto vetoed = [&]() {
        u32 v = 0;
        for (u32 q = 0; q < n_veto; ++q) v |= h_veto[q * veto_stride];
        return v != 0;
    };
    if (e->ph_n == 0) {
        HIP_TRY(e, hipStreamSynchronize(e->stream));
        e->ph_counted = false;
        if (vetoed()) return RL_OK;  // the pass stays open: more rounds, or rl_gen_abort
        *committed = 1;
        return gen_phase_close(e, true);
    }
    GenArgs A = e->ph_A;
    A.update_mode = 1u;
    const u32 n = e->ph_n;
    const u64 used = e->live + e->tombs, bound = e->cap - e->cap / 16;
    k_gen_commit<<<std::min(cdiv(n, 256), 1024u), 256, 0, e->stream>>>(A, used < bound ? (u32)std::min<u64>(bound - used, 0xFFFFFFFFull) : 0u,
                                                                      d_veto, n_veto, veto_stride);
    // (k_gen_post: the outcome as one 16-byte store into host-mapped memory, and — applied passes only — the status block and
    // the scratch blocks zeroed for whatever comes next: no copy command, no stream synchronise, no fill commands)
    const u32 seq = ++e->gen_post_seq ? e->gen_post_seq : ++e->gen_post_seq;
    k_gen_post<<<1, 256, 0, e->stream>>>(e->d_gst, &e->ph_bs->st, e->h_gen_word, seq, reinterpret_cast<u32*>(e->d_bs),
                                         (u32)(BS_ROT * sizeof(BatchScratch) / sizeof(u32)));
    HIP_TRY(e, hipGetLastError());
    const int wrc = wait_word(e, e->h_gen_word + 3, seq, "the gated commit of a phased pass");
    if (wrc) return wrc;
    const u32 err = e->h_gen_word[0], created = e->h_gen_word[1], flags = e->h_gen_word[2];
    e->ph_counted = false;
    e->ph_unchecked = false;  // (whatever an async rl_gen_begin_device did not wait for is in this word)
    if (err) {
        (void)gen_phase_close(e);
        return status_to_error(e, err);
    }
    const u32 hot_n = flags >> 16 == 0xFFFFu ? 0xFFFFFFFFu : flags >> 16;
    if (flags & 1u) {  // the heavy keys were promoted: begin again, with that set
        e->part_seq += 1;
        adapt_hot_threshold(e, hot_n, true);
        (void)gen_phase_close(e);
        return fail(e, RL_ERR_BUSY, "hash buckets of this slice overflowed; their heavy keys were promoted: begin the pass again "
                                    "(nothing was applied)");
    }
    if (!(flags & 2u)) {
        if (!vetoed()) {  // (no rank objected, no error, no overflow: the cells did not fit after all — cannot happen, k_gen_veto said they do)
            (void)gen_phase_close(e);
            return fail(e, RL_ERR_DEVICE, "the gated commit did not apply a pass nobody vetoed (bug)");
        }
        return RL_OK;  // the pass stays open
    }
    adapt_hot_threshold(e, hot_n, false);
    e->live += created;
    e->part_seq++;
    e->stats.batches++;
    e->stats.hits += n;
    e->stats.ordered_hits += n;
    e->stats.ordered_batches++;
    e->gen_clean = true;  // (k_gen_post zeroed the status block and the scratch blocks behind the commit)
    *committed = 1;
    return gen_phase_close(e, true);
} RL_ABI_CATCH

int32_t rl_gen_abort(rl_engine* e) try {
    if (!e) return RL_ERR_INVALID;
    EngineLock g(e);
    if (!e->ph_open) return RL_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    return gen_phase_close(e);
} RL_ABI_CATCH

int32_t rl_match_table_set(rl_engine* e, const rl_match_limit* limits, uint32_t n_limits, const rl_match_cond* conds,
                           uint32_t n_conds, uint32_t n_namespaces) try {
    return rl_match_table_set_ex(e, limits, n_limits, conds, n_conds, n_namespaces, nullptr, 0);
} RL_ABI_CATCH

int32_t rl_match_table_set_ex(rl_engine* e, const rl_match_limit* limits, uint32_t n_limits, const rl_match_cond* conds,
                              uint32_t n_conds, uint32_t n_namespaces, const uint32_t* more_vars, uint32_t n_more_vars) try {
    if (!e || (n_limits && !limits) || (n_conds && !conds) || (n_more_vars && !more_vars)) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    HIP_TRY(e, hipSetDevice(e->device));
    std::vector<u32> ns_off(n_namespaces + 1, 0);
    u32 max_id = 0, max_vars = 0;
    for (u32 i = 0; i < n_limits; ++i) {
        const rl_match_limit& L = limits[i];
        if (L.ns >= n_namespaces) return fail(e, RL_ERR_INVALID, "match limit %u: namespace id %u out of range", i, L.ns);
        if (i && limits[i - 1].ns > L.ns) return fail(e, RL_ERR_INVALID, "match limits must be sorted by namespace id");
        if (L.n_vars > MATCH_MAX_VARS_F) return fail(e, RL_ERR_INVALID, "match limit %u: more than %u variables stay on the host path", i, MATCH_MAX_VARS_F);
        if (L.n_vars > MATCH_MAX_VARS && (u64)L.var_key[0] + L.n_vars > n_more_vars)
            return fail(e, RL_ERR_INVALID, "match limit %u: %u variables need their key ids in more_vars (rl_match_table_set_ex)", i, L.n_vars);
        max_vars = std::max<u32>(max_vars, L.n_vars);
        if (((L.limit & RL_SIMPLE) != 0) != (L.n_vars == 0)) return fail(e, RL_ERR_INVALID, "match limit %u: RL_SIMPLE must be set iff the limit has no variables", i);
        if (RL_LIMIT_ID(L.limit) >= e->h_limits.size() || RL_LIMIT_ID(L.limit) >= 4095u) return fail(e, RL_ERR_INVALID, "match limit %u: unknown limit id", i);
        if ((u64)L.cond_off + L.n_cond > n_conds) return fail(e, RL_ERR_INVALID, "match limit %u: conditions out of range", i);
        ns_off[L.ns + 1]++;
        max_id = std::max<u32>(max_id, RL_LIMIT_ID(L.limit));
    }
    e->match_max_limit_id = max_id;
    e->match_max_vars = max_vars;
    for (u32 n = 0; n < n_namespaces; ++n) ns_off[n + 1] += ns_off[n];
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (e->d_match_limits) (void)hipFree(e->d_match_limits);
    if (e->d_match_conds) (void)hipFree(e->d_match_conds);
    if (e->d_match_ns_off) (void)hipFree(e->d_match_ns_off);
    e->d_match_limits = nullptr;
    e->d_match_conds = nullptr;
    e->d_match_ns_off = nullptr;
    if (hipMalloc((void**)&e->d_match_limits, (n_limits ? n_limits : 1) * sizeof(MatchLimit)) != hipSuccess ||
        hipMalloc((void**)&e->d_match_conds, (n_conds ? n_conds : 1) * sizeof(MatchCond)) != hipSuccess ||
        hipMalloc((void**)&e->d_match_ns_off, ns_off.size() * sizeof(u32)) != hipSuccess)
        return fail(e, RL_ERR_NOMEM, "hipMalloc of the match table failed");
    if (n_limits) HIP_TRY(e, hipMemcpy(e->d_match_limits, limits, n_limits * sizeof(MatchLimit), hipMemcpyHostToDevice));
    if (n_conds) HIP_TRY(e, hipMemcpy(e->d_match_conds, conds, n_conds * sizeof(MatchCond), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->d_match_ns_off, ns_off.data(), ns_off.size() * sizeof(u32), hipMemcpyHostToDevice));
    e->n_match_limits = n_limits;
    e->n_match_ns = n_namespaces;
    e->n_match_conds = n_conds;
    // ---- the slot form, when the table has it (k_match_fast) ---------------------------------------------
    e->match_fast = false;
    e->wire_ready = false;  // (the strings of rl_wire_table_set belong to the table that is being replaced)
    if (e->d_match_flimits) (void)hipFree(e->d_match_flimits);
    if (e->d_match_fconds) (void)hipFree(e->d_match_fconds);
    e->d_match_flimits = nullptr;
    e->d_match_fconds = nullptr;
    bool fast = n_limits && n_limits <= MATCH_LDS_LIMITS && n_conds <= MATCH_LDS_CONDS && n_namespaces <= MATCH_LDS_NS;
    for (u32 n = 0; fast && n < n_namespaces; ++n) fast = ns_off[n + 1] - ns_off[n] <= 64u;
    MatchSlots slots{};
    u32 var_slots = 0;
    auto slot_of = [&](u32 key) -> int {
        for (u32 q = 0; q < slots.n; ++q)
            if (slots.key[q] == key) return (int)q;
        if (slots.n == MATCH_SLOTS) return -1;
        slots.key[slots.n] = key;
        return (int)slots.n++;
    };
    std::vector<MatchLimitF> fl(n_limits);
    std::vector<MatchCondF> fc(n_conds ? n_conds : 1);
    for (u32 i = 0; fast && i < n_limits; ++i) {
        const rl_match_limit& L = limits[i];
        if (L.n_cond > 255u) fast = false;
        u32 vs[MATCH_MAX_VARS_F] = {0, 0, 0, 0, 0, 0, 0, 0}, vslots = 0;
        for (u32 q = 0; fast && q < L.n_vars; ++q) {
            const int sl = slot_of(L.n_vars > MATCH_MAX_VARS ? more_vars[L.var_key[0] + q] : L.var_key[q]);
            if (sl < 0) fast = false;
            else {
                vs[q] = (u32)sl;
                vslots |= (u32)sl << (4u * q);
                var_slots |= 1u << sl;
            }
        }
        for (u32 c = 0; fast && c < L.n_cond; ++c) {
            const rl_match_cond& cd = conds[L.cond_off + c];
            const int sl = slot_of(cd.key);
            if (sl < 0 || cd.op > 1u) fast = false;
            else fc[L.cond_off + c] = MatchCondF{(u32)sl | (cd.op << 8), cd.value};
        }
        fl[i] = MatchLimitF{L.limit, L.cond_off, L.n_cond | (L.n_vars << 8) | (vs[0] << 16) | (vs[1] << 24), vslots};
    }
    if (fast) {
        if (hipMalloc((void**)&e->d_match_flimits, n_limits * sizeof(MatchLimitF)) != hipSuccess ||
            hipMalloc((void**)&e->d_match_fconds, fc.size() * sizeof(MatchCondF)) != hipSuccess)
            return fail(e, RL_ERR_NOMEM, "hipMalloc of the match table failed");
        HIP_TRY(e, hipMemcpy(e->d_match_flimits, fl.data(), n_limits * sizeof(MatchLimitF), hipMemcpyHostToDevice));
        HIP_TRY(e, hipMemcpy(e->d_match_fconds, fc.data(), fc.size() * sizeof(MatchCondF), hipMemcpyHostToDevice));
        e->match_slots = slots;
        e->match_var_slots = var_slots;
        e->match_fast = RL_EXP_ENV("RL_MATCH_GENERIC") == nullptr;
    }
    return RL_OK;
} RL_ABI_CATCH

// device pointers for the request arrays and for verdict / limited_limit; derived hits stay in the
// engine's staging buffers (e->d_hits, e->d_req_off)
// What follows the matcher, by `op` (rl_engine.h RL_OP_*): the derived counters lie in e->d_hits / e->d_req_off / e->d_hit_req.
//   RL_OP_CHECK_AND_UPDATE  check_rate_limited_and_update (lib.rs:425-464): the general resolver
//   RL_OP_CHECK             is_rate_limited (lib.rs:362-409): k_req_within, nothing is written
//   RL_OP_UPDATE            update_counters (lib.rs:411-423): the general resolver with every hit admitted (update_counter)
static int32_t matched_op_locked(rl_engine* e, int op, u32 n_hits, u32 n_req, u64 now, bool load, uint8_t* d_verdict,
                                 int32_t* d_limited, bool hit_req_filled, const u32* d_hit_check, int32_t* d_msg_status,
                                 u32 force_delta) {
    if (op == RL_OP_CHECK) {
        HIP_TRY(e, hipMemsetAsync(e->d_status, 0, sizeof(Status), e->stream));
        k_req_within<<<cdiv(n_req, 256), 256, 0, e->stream>>>(e->table, e->log2cap, e->seed, e->d_hits, e->d_req_off, n_req,
                                                              d_hit_check, d_msg_status, e->d_limits, (u32)e->h_limits.size(),
                                                              now, force_delta, d_verdict, d_limited, e->d_status);
        HIP_TRY(e, hipGetLastError());
        const int rc = read_status(e);
        if (rc) return rc;
        if (e->h_status->err) return status_to_error(e, e->h_status->err);
        return RL_OK;
    }
    GenCall gc{e->d_hits, n_hits, e->d_req_off, n_req, nullptr, now, load && op == RL_OP_CHECK_AND_UPDATE, op == RL_OP_UPDATE,
               d_verdict, e->d_first, e->d_remaining, e->d_expires};
    gc.d_limited = d_limited;
    gc.hit_req_filled = hit_req_filled && n_hits > 0;
    gc.d_hit_check = d_hit_check;
    gc.d_msg_status = d_msg_status;
    return run_check_general(e, gc);
}

// Where a serving call (rl_match_serve_batch / rl_wire_serve_batch) wants its answer: host pointers.
struct ServeOut {
    int32_t with_headers;
    bool async;                 // RL_SERVE_ASYNC: return with the response bytes still travelling (rl_serve_wait)
    const uint32_t** resp_off;  // -> [n + 1], in the engine's pinned staging (slot 4 * set + 2)
    const uint8_t** resp;       // -> the bytes, in the engine's pinned staging (slot 4 * set + 3)
    uint32_t set = 0;           // which of the two sets of staging / piece events the call uses (rl_wire_serve_batch_set)
};
static int32_t host_staging_locked(rl_engine* e, uint32_t slot, uint64_t bytes, void** out);

// The serialized RateLimitResponse of every request of the batch the resolver just decided (rl_resp.hpp): lengths ->
// exclusive scan -> bytes, then the offsets and the bytes to the host.  d_status: per request, null = every request is
// answered.  The caller synchronises the stream.
static int32_t responses_locked(rl_engine* e, u32 n, u32 n_hits, const int32_t* d_status, const uint8_t* d_verdict, const ServeOut& so) {
    if (so.with_headers && !e->resp_ready) return fail(e, RL_ERR_INVALID, "rl_resp_table_set was not called for the installed limits");
    if (!e->d_resp_off && hipMalloc((void**)&e->d_resp_off, ((size_t)e->max_batch + 2) * sizeof(u32)) != hipSuccess)
        return fail(e, RL_ERR_NOMEM, "hipMalloc of the responses' offsets failed");
    RespArgs R{};
    R.blob = e->d_resp_blob;
    R.blob_len = e->resp_blob_len;
    R.frag = e->d_resp_frag;
    R.n_frag = e->n_resp_frag;
    R.limits = e->d_limits;
    R.n_limits = (u32)e->h_limits.size();
    R.status = d_status;
    R.verdict = d_verdict;
    R.req_off = e->d_req_off;
    R.hits = e->d_hits;
    R.remaining = e->d_remaining;
    R.expires_in = e->d_expires;
    R.n = n;
    R.with_headers = so.with_headers ? 1u : 0u;
    u32* len = e->d_m_count;  // (the matcher's per-request counts are done with: [max_batch + 1])
    k_resp<false><<<cdiv(n + 1, 256), 256, 0, e->stream>>>(R, len, nullptr, nullptr, 0u, cdiv(n + 1, 256));
    {
        const u32 ns = n + 1, gs = cdiv(ns, XSCAN_PER_WG);
        u32* tot = static_cast<u32*>(e->d_m_scan_tmp);
        k_xscan_sums<<<gs, 256, 0, e->stream>>>(len, ns, tot);
        k_xscan_tot<<<1, 1024, 0, e->stream>>>(tot, gs);
        k_xscan_apply<<<gs, 256, 0, e->stream>>>(len, ns, tot, e->d_resp_off);
    }
    HIP_TRY(e, hipGetLastError());
    const u32 set = so.set < rl_engine::SERVE_SETS ? so.set : 0u;
    const u32 slot_off = 4u * set + 2u, slot_bytes = 4u * set + 3u;
    void* h_off = nullptr;
    int32_t rc = host_staging_locked(e, slot_off, ((u64)n + 1) * sizeof(u32), &h_off);
    if (rc) return rc;
    e->resp_n_chunks[set] = 0;
    e->lazy[set].n = 0;  // (a set is only reused once its last byte has been waited for: nobody is inside its LazyCopy)
    const u32 n_blocks = cdiv(n, 256);
    // What the responses can take at most: 2 bytes of overall_code; with headers 157 more of tags, lengths, keys and three
    // numbers of up to 20 digits, + per derived counter its limit's fragment (rl_resp.hpp).  When the pinned staging holds
    // that much, the bytes' kernels go out BEHIND the offsets' copy without the host having seen the total: the round trip
    // in the middle of the phase (a synchronise, the total, then the launches: ~40 us of idle stream) is gone; the host
    // waits for the offsets while the responses are being written.
    const u64 bound = so.with_headers ? (u64)n * 160u + (u64)n_hits * std::max<u32>(e->resp_max_frag, 7u) : (u64)n * 2u;
    if (e->resp_direct && so.async && e->resp_blind && bound <= (1ull << 30)) {
        void* h_bytes = nullptr;
        rc = host_staging_locked(e, slot_bytes, bound ? bound : 1, &h_bytes);  // (grows — and synchronises — only until the largest batch has been seen)
        if (rc) return rc;
        HIP_TRY(e, hipMemcpyAsync(h_off, e->d_resp_off, ((size_t)n + 1) * sizeof(u32), hipMemcpyDeviceToHost, e->stream));
        if (!e->resp_off_ev) HIP_TRY(e, hipEventCreateWithFlags(&e->resp_off_ev, hipEventDisableTiming));
        HIP_TRY(e, hipEventRecord(e->resp_off_ev, e->stream));
        // ---- the snapshot: everything the bytes' kernels read, copied beside the engine's arrays (27 MB for 262 144 requests
        //      with three counters each: ~10 us), so that those kernels — an eighth of the batch each, bound by the link to
        //      the host — run on their own stream while THIS stream takes the next call ----------------------------------
        auto up = [](u64 b) { return (b + 255ull) & ~255ull; };
        const u64 b_status = d_status ? up((u64)n * 4) : 0, b_verdict = up(n), b_reqoff = so.with_headers ? up(((u64)n + 1) * 4) : 0,
                  b_hits = so.with_headers ? up((u64)n_hits * sizeof(Hit)) : 0, b_u64 = so.with_headers ? up((u64)n_hits * 8) : 0,
                  b_off = up(((u64)n + 1) * 4);
        const u64 need = b_status + b_verdict + b_reqoff + b_hits + 2 * b_u64 + b_off;
        if (need > e->resp_snap_cap[set]) {
            if (e->resp_stream) HIP_TRY(e, hipStreamSynchronize(e->resp_stream));
            if (e->resp_snap[set]) (void)hipFree(e->resp_snap[set]);
            e->resp_snap[set] = nullptr;
            e->resp_snap_cap[set] = 0;
            u64 cap = 1u << 20;
            while (cap < need) cap <<= 1;
            if (hipMalloc(&e->resp_snap[set], cap) != hipSuccess) return fail(e, RL_ERR_NOMEM, "hipMalloc of %llu bytes for the responses' snapshot failed", (unsigned long long)cap);
            e->resp_snap_cap[set] = cap;
        }
        if (!e->resp_stream) HIP_TRY(e, hipStreamCreateWithFlags(&e->resp_stream, hipStreamNonBlocking));
        if (!e->snap_ev[set]) HIP_TRY(e, hipEventCreateWithFlags(&e->snap_ev[set], hipEventDisableTiming));
        uint8_t* sp = static_cast<uint8_t*>(e->resp_snap[set]);
        CopySegs S1{}, S2{};
        auto seg = [&](CopySegs& S, const void* src, u64 bytes, u64 room) -> const void* {
            uint8_t* dst = sp;
            sp += room;
            if (!bytes) return dst;
            S.dst[S.n] = dst;
            S.src[S.n] = src;
            S.bytes[S.n] = bytes;
            ++S.n;
            return dst;
        };
        RespArgs RS = R;
        RS.status = d_status ? static_cast<const int32_t*>(seg(S1, d_status, (u64)n * 4, b_status)) : nullptr;
        RS.verdict = static_cast<const uint8_t*>(seg(S1, d_verdict, n, b_verdict));
        if (so.with_headers) {
            RS.req_off = static_cast<const u32*>(seg(S1, e->d_req_off, ((u64)n + 1) * 4, b_reqoff));
            RS.hits = static_cast<const Hit*>(seg(S1, e->d_hits, (u64)n_hits * sizeof(Hit), b_hits));
            RS.remaining = static_cast<const u64*>(seg(S2, e->d_remaining, (u64)n_hits * 8, b_u64));
            RS.expires_in = static_cast<const u64*>(seg(S2, e->d_expires, (u64)n_hits * 8, b_u64));
        }
        const u32* snap_off = static_cast<const u32*>(seg(S2, e->d_resp_off, ((u64)n + 1) * 4, b_off));
        for (CopySegs* S : {&S1, &S2}) {
            if (!S->n) continue;
            u64 most = 0;
            for (u32 k = 0; k < S->n; ++k) most = std::max(most, S->bytes[k]);
            k_copy_segs<<<(u32)std::min<u64>(std::max<u64>(most >> 14, 1), 2048), 256, 0, e->stream>>>(*S);
        }
        HIP_TRY(e, hipEventRecord(e->snap_ev[set], e->stream));
        HIP_TRY(e, hipStreamWaitEvent(e->resp_stream, e->snap_ev[set], 0));
        RS.out_cap = e->h_stage_cap[slot_bytes];
        const u32 pieces = std::min<u32>(e->resp_pieces, std::max<u32>(1u, (u32)(bound >> 22)));  // (the bound is ~2 x the bytes)
        const u32 per = cdiv(n_blocks, pieces);
        u32 nc = 0, b_end[rl_engine::RESP_CHUNKS];
        u32 resp_mode = e->resp_via_copy;
        const u64 t_serve = (u64)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
        if (resp_mode == rl_engine::RESP_AUTO) {
            // another set served a call within the last 20 ms: callers are in flight together (one caller alone is always given set 0)
            resp_mode = 0;
            for (u32 q = 0; q < rl_engine::SERVE_SETS; ++q)
                if (q != set && e->serve_last_us[q] && t_serve - e->serve_last_us[q] < 20000u) resp_mode = 4;
        }
        e->serve_last_us[set] = t_serve;
        if (resp_mode) {
            if (bound > e->d_resp_set_cap[set]) {
                HIP_TRY(e, hipStreamSynchronize(e->resp_stream));
                if (e->d_resp_set[set]) (void)hipFree(e->d_resp_set[set]);
                e->d_resp_set[set] = nullptr;
                e->d_resp_set_cap[set] = 0;
                u64 cap = 1u << 20;
                while (cap < bound) cap <<= 1;
                if (hipMalloc((void**)&e->d_resp_set[set], cap) != hipSuccess) return fail(e, RL_ERR_NOMEM, "hipMalloc of %llu bytes of responses failed", (unsigned long long)cap);
                e->d_resp_set_cap[set] = cap;
            }
            RS.out_cap = e->d_resp_set_cap[set];
            k_resp<true><<<n_blocks, 256, 0, e->resp_stream>>>(RS, nullptr, snap_off, e->d_resp_set[set], 0u, n_blocks);
            HIP_TRY(e, hipGetLastError());
            HIP_TRY(e, hipEventSynchronize(e->resp_off_ev));
            const u32* off = static_cast<const u32*>(h_off);
            if (off[n] > e->h_stage_cap[slot_bytes])
                return fail(e, RL_ERR_INTERNAL, "the responses take %u bytes, the bound said %llu (the batch was applied)", off[n], (unsigned long long)bound);
            if (resp_mode == 4) {
                rl_engine::LazyCopy& L = e->lazy[set];
                std::lock_guard<std::mutex> lg(L.mu);
                if (!L.stream) HIP_TRY(e, hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
                if (!L.built) HIP_TRY(e, hipEventCreateWithFlags(&L.built, hipEventDisableTiming));
                HIP_TRY(e, hipEventRecord(L.built, e->resp_stream));
                HIP_TRY(e, hipStreamWaitEvent(L.stream, L.built, 0));
                for (u32 b0 = 0; b0 < n_blocks; b0 += per, ++nc) {
                    const u32 nb = std::min(per, n_blocks - b0);
                    L.lo[nc] = off[std::min<u64>((u64)b0 * 256u, n)];
                    L.hi[nc] = off[std::min<u64>((u64)(b0 + nb) * 256u, n)];
                    if (!e->resp_ev[set][nc]) HIP_TRY(e, hipEventCreateWithFlags(&e->resp_ev[set][nc], hipEventDisableTiming));
                    e->resp_chunk_end[set][nc] = L.hi[nc];
                }
                L.n = nc;
                L.issued.store(0, std::memory_order_release);
                L.dst = static_cast<uint8_t*>(h_bytes);
                e->resp_n_chunks[set] = nc;
                *so.resp_off = off;
                *so.resp = static_cast<const uint8_t*>(h_bytes);
                return RL_OK;
            }
            for (u32 b0 = 0; b0 < n_blocks; b0 += per, ++nc) {
                const u32 nb = std::min(per, n_blocks - b0);
                const u64 lo = off[std::min<u64>((u64)b0 * 256u, n)], hi = off[std::min<u64>((u64)(b0 + nb) * 256u, n)];
                if (hi > lo)
                    HIP_TRY(e, hipMemcpyAsync(static_cast<uint8_t*>(h_bytes) + lo, e->d_resp_set[set] + lo, hi - lo, hipMemcpyDeviceToHost, e->resp_stream));
                if (!e->resp_ev[set][nc]) HIP_TRY(e, hipEventCreateWithFlags(&e->resp_ev[set][nc], hipEventDisableTiming));
                HIP_TRY(e, hipEventRecord(e->resp_ev[set][nc], e->resp_stream));
                e->resp_chunk_end[set][nc] = (u32)hi;
            }
            e->resp_n_chunks[set] = nc;
            *so.resp_off = off;
            *so.resp = static_cast<const uint8_t*>(h_bytes);
            return RL_OK;
        }
        for (u32 b0 = 0; b0 < n_blocks; b0 += per, ++nc) {
            const u32 nb = std::min(per, n_blocks - b0);
            k_resp<true><<<std::min(nb, e->resp_writers), 256, 0, e->resp_stream>>>(RS, nullptr, snap_off, static_cast<uint8_t*>(h_bytes), b0, nb);
            if (!e->resp_ev[set][nc]) HIP_TRY(e, hipEventCreateWithFlags(&e->resp_ev[set][nc], hipEventDisableTiming));
            HIP_TRY(e, hipEventRecord(e->resp_ev[set][nc], e->resp_stream));
            b_end[nc] = b0 + nb;
        }
        HIP_TRY(e, hipGetLastError());
        HIP_TRY(e, hipEventSynchronize(e->resp_off_ev));
        const u32* off = static_cast<const u32*>(h_off);
        for (u32 c = 0; c < nc; ++c) e->resp_chunk_end[set][c] = off[std::min<u64>((u64)b_end[c] * 256u, n)];
        e->resp_n_chunks[set] = nc;
        *so.resp_off = off;
        *so.resp = static_cast<const uint8_t*>(h_bytes);
        if (off[n] > e->h_stage_cap[slot_bytes])  // (cannot happen while the bound above is one; the kernel wrote nothing beyond the buffer)
            return fail(e, RL_ERR_INTERNAL, "the responses take %u bytes, the bound said %llu (the batch was applied)", off[n], (unsigned long long)bound);
        return RL_OK;
    }
    // (offsets are 32 bits — k_xscan over 32-bit lengths: a batch whose responses COULD pass 4 GiB is refused here, like the
    // path above does with its bound, instead of letting the scan wrap and the responses overlap — ADVICE r05)
    if (bound > 0xFFFFFFFFull)
        return fail(e, RL_ERR_BATCH_TOO_LARGE, "the batch's responses may take %llu bytes (> 4 GiB of 32-bit offsets): serve it in smaller batches (the batch WAS applied)",
                    (unsigned long long)bound);
    HIP_TRY(e, hipMemcpyAsync(h_off, e->d_resp_off, ((size_t)n + 1) * sizeof(u32), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    const u64 total = static_cast<const u32*>(h_off)[n];
    void* h_bytes = nullptr;
    rc = host_staging_locked(e, slot_bytes, total ? total : 1, &h_bytes);
    if (rc) return rc;
    *so.resp_off = static_cast<const u32*>(h_off);
    *so.resp = static_cast<const uint8_t*>(h_bytes);
    if (!total) return RL_OK;
    R.out_cap = ~0ull;
    if (e->resp_direct) {
        // k_resp writes the responses INTO the host's pinned staging: its coalesced copy-out is the transfer (no device
        // buffer, no copy command — those were 0.23 ms of kernel and then 1.05 ms of copies for 262 144 responses).  With
        // RL_SERVE_ASYNC the launch is cut into pieces of workgroups with an event behind each.
        uint8_t* const d_out = static_cast<uint8_t*>(h_bytes);
        u32 pieces = so.async ? std::min<u32>(e->resp_pieces, std::max<u32>(1u, (u32)(total >> 21))) : 1u;  // >= 2 MB each
        const u32 per = cdiv(n_blocks, pieces);
        u32 nc = 0;
        for (u32 b0 = 0; b0 < n_blocks; b0 += per, ++nc) {
            const u32 nb = std::min(per, n_blocks - b0);
            k_resp<true><<<std::min(nb, e->resp_writers), 256, 0, e->stream>>>(R, nullptr, e->d_resp_off, d_out, b0, nb);
            if (so.async) {
                if (!e->resp_ev[set][nc]) HIP_TRY(e, hipEventCreateWithFlags(&e->resp_ev[set][nc], hipEventDisableTiming));
                HIP_TRY(e, hipEventRecord(e->resp_ev[set][nc], e->stream));
                e->resp_chunk_end[set][nc] = static_cast<const u32*>(h_off)[std::min<u64>((u64)(b0 + nb) * 256u, n)];
            }
        }
        HIP_TRY(e, hipGetLastError());
        if (so.async) e->resp_n_chunks[set] = nc;
        return RL_OK;
    }
    if (total > e->resp_bytes_cap) {
        if (e->d_resp_bytes) (void)hipFree(e->d_resp_bytes);
        e->d_resp_bytes = nullptr;
        e->resp_bytes_cap = 0;
        u64 cap = 1u << 16;
        while (cap < total) cap <<= 1;
        if (hipMalloc((void**)&e->d_resp_bytes, cap) != hipSuccess) return fail(e, RL_ERR_NOMEM, "hipMalloc of %llu bytes of responses failed", (unsigned long long)cap);
        e->resp_bytes_cap = cap;
    }
    k_resp<true><<<n_blocks, 256, 0, e->stream>>>(R, nullptr, e->d_resp_off, e->d_resp_bytes, 0u, n_blocks);
    HIP_TRY(e, hipGetLastError());
    if (!so.async) {
        HIP_TRY(e, hipMemcpyAsync(h_bytes, e->d_resp_bytes, total, hipMemcpyDeviceToHost, e->stream));
        return RL_OK;
    }
    u64 chunk = (total + rl_engine::RESP_CHUNKS - 1) / rl_engine::RESP_CHUNKS;
    chunk = std::max<u64>((chunk + 65535ull) & ~65535ull, 2ull << 20);  // (a copy command costs ~11 us: at least 2 MB each)
    u32 nc = 0;
    for (u64 at = 0; at < total; at += chunk, ++nc) {
        if (!e->resp_ev[set][nc]) HIP_TRY(e, hipEventCreateWithFlags(&e->resp_ev[set][nc], hipEventDisableTiming));
        HIP_TRY(e, hipMemcpyAsync(static_cast<uint8_t*>(h_bytes) + at, e->d_resp_bytes + at, std::min(chunk, total - at),
                                  hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(e, hipEventRecord(e->resp_ev[set][nc], e->stream));
        e->resp_chunk_end[set][nc] = std::min(at + chunk, total);
    }
    e->resp_n_chunks[set] = nc;
    return RL_OK;
}

static int32_t match_and_check_locked(rl_engine* e, int op, const u32* d_ns, const u32* d_ent_off, const u32* d_ent_key,
                                      const u32* d_ent_val, const u32* d_delta, u32 n_req, u64 now, bool load,
                                      uint8_t* d_verdict, int32_t* d_limited, u32* n_hits_out) {
    if (!e->d_match_limits) return fail(e, RL_ERR_INVALID, "rl_match_table_set was not called");
    if (e->match_max_vars > MATCH_MAX_VARS)
        return fail(e, RL_ERR_INVALID, "a limit of the match table has %u variables: the packed exact key takes %u (rl_match_key) — "
                                       "the hashed-key wire path derives such counters (rl_wire_match_and_check_batch), or the "
                                       "caller folds the variables into one (what rli_compile does for exact keys)",
                    e->match_max_vars, MATCH_MAX_VARS);
    const u32 g = cdiv(n_req, 256);
    if (e->match_fast && e->match_one) {
        // count pass -> one-workgroup scan, which hands {total, error bits} to the host through a host-mapped word ->
        // fill pass, enqueued right away: the host's poll and the launches of the resolver run under it
        const u32 call = ++e->m_call ? e->m_call : ++e->m_call;
        const MatchTables T{e->d_match_flimits, e->n_match_limits, e->d_match_ns_off, e->n_match_ns, e->d_match_fconds,
                            e->n_match_conds, e->match_slots};
        k_match_count2<<<g, 256, 0, e->stream>>>(d_ns, d_ent_off, d_ent_key, d_ent_val, n_req, T, e->d_m_mask, e->d_m_scan1);
        k_match_scan2<<<1, 1024, 0, e->stream>>>(e->d_m_scan1, g, e->d_req_off + n_req, e->h_m_word, call);
        k_match_fill2<<<g, 256, 0, e->stream>>>(d_ns, d_ent_off, d_ent_key, d_ent_val, d_delta, n_req, T, e->d_m_mask,
                                                e->d_m_scan1, e->d_req_off, e->d_hits, e->d_hit_req, e->max_batch);
        HIP_TRY(e, hipGetLastError());
        int wrc = wait_word(e, e->h_m_word + 3, call, "the matcher's count pass");
        if (wrc) return wrc;
        const u32 n_hits = e->h_m_word[0], m_err = e->h_m_word[1];
        if (n_hits_out) *n_hits_out = n_hits;
        if (m_err || n_hits > e->max_batch) HIP_TRY(e, hipStreamSynchronize(e->stream));  // refused
        if (m_err & ERRBIT_RESERVED_KEY)
            return fail(e, RL_ERR_INVALID, "a value id does not fit %u bits: such dictionaries keep the host path", MATCH_VAL_BITS);
        if (m_err) return status_to_error(e, m_err);
        if (n_hits > e->max_batch) return fail(e, RL_ERR_BATCH_TOO_LARGE, "the requests expand to %u counters > max_batch_hits %u", n_hits, e->max_batch);
        return matched_op_locked(e, op, n_hits, n_req, now, load, d_verdict, d_limited, true, nullptr, nullptr, 0u);
    }
    HIP_TRY(e, hipMemsetAsync(e->d_status, 0, sizeof(Status), e->stream));
    HIP_TRY(e, hipMemsetAsync(e->d_m_count + n_req, 0, sizeof(u32), e->stream));
    const bool in_lds = e->n_match_limits <= MATCH_LDS_LIMITS && e->n_match_conds <= MATCH_LDS_CONDS &&
                        e->n_match_ns <= MATCH_LDS_NS;
    auto k_count = in_lds ? k_match<false, true> : k_match<false, false>;
    auto k_fill = in_lds ? k_match<true, true> : k_match<true, false>;
    if (e->match_fast)
        k_match_fast<false><<<g, 256, 0, e->stream>>>(d_ns, d_ent_off, d_ent_key, d_ent_val, d_delta, n_req, e->d_match_flimits,
                                                      e->n_match_limits, e->d_match_ns_off, e->n_match_ns, e->d_match_fconds,
                                                      e->n_match_conds, e->match_slots, e->d_m_count, e->d_m_mask, nullptr,
                                                      nullptr, e->d_status, nullptr, 0u);
    else
        k_count<<<g, 256, 0, e->stream>>>(d_ns, d_ent_off, d_ent_key, d_ent_val, d_delta, n_req, e->d_match_limits,
                                          e->n_match_limits, e->d_match_ns_off, e->n_match_ns, e->d_match_conds,
                                          e->n_match_conds, e->d_m_count, nullptr, nullptr, e->d_status);
    {
        const u32 ns = n_req + 1, gs = cdiv(ns, XSCAN_PER_WG);
        u32* tot = static_cast<u32*>(e->d_m_scan_tmp);
        k_xscan_sums<<<gs, 256, 0, e->stream>>>(e->d_m_count, ns, tot);
        k_xscan_tot<<<1, 1024, 0, e->stream>>>(tot, gs);
        k_xscan_apply<<<gs, 256, 0, e->stream>>>(e->d_m_count, ns, tot, e->d_req_off);
    }
    HIP_TRY(e, hipMemcpyAsync(e->h_m_total, e->d_req_off + n_req, sizeof(u32), hipMemcpyDeviceToHost, e->stream));
    int rc = RL_OK;
    bool filled = false;
    if (e->match_fast && e->ev_match) {
        // the slot form: the fill pass goes into the queue right away (it reads the total on the device) and the host
        // waits for the two copies only — its round trip and the launches of the resolver run under the fill pass
        HIP_TRY(e, hipMemcpyAsync(e->h_status, e->d_status, sizeof(Status), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(e, hipEventRecord(e->ev_match, e->stream));
        k_match_fast<true><<<g, 256, 0, e->stream>>>(d_ns, d_ent_off, d_ent_key, d_ent_val, d_delta, n_req,
                                                     e->d_match_flimits, e->n_match_limits, e->d_match_ns_off,
                                                     e->n_match_ns, e->d_match_fconds, e->n_match_conds, e->match_slots,
                                                     nullptr, e->d_m_mask, e->d_req_off, e->d_hits, e->d_status, e->d_hit_req,
                                                     e->max_batch);
        HIP_TRY(e, hipGetLastError());
        HIP_TRY(e, hipEventSynchronize(e->ev_match));
        filled = true;
        if (e->h_status->err || e->h_m_total[0] > e->max_batch) HIP_TRY(e, hipStreamSynchronize(e->stream));  // refused below
    } else {
        rc = read_status(e);
        if (rc) return rc;
    }
    if (e->h_status->err & ERRBIT_RESERVED_KEY)
        return fail(e, RL_ERR_INVALID, "a value id does not fit %u bits: such dictionaries keep the host path", MATCH_VAL_BITS);
    if (e->h_status->err) return status_to_error(e, e->h_status->err);
    const u32 n_hits = e->h_m_total[0];
    if (n_hits_out) *n_hits_out = n_hits;
    if (n_hits > e->max_batch) return fail(e, RL_ERR_BATCH_TOO_LARGE, "the requests expand to %u counters > max_batch_hits %u", n_hits, e->max_batch);
    if (n_hits && !filled) {
        HIP_TRY(e, hipMemsetAsync(e->d_status, 0, sizeof(Status), e->stream));
        if (e->match_fast)
            k_match_fast<true><<<g, 256, 0, e->stream>>>(d_ns, d_ent_off, d_ent_key, d_ent_val, d_delta, n_req,
                                                         e->d_match_flimits, e->n_match_limits, e->d_match_ns_off,
                                                         e->n_match_ns, e->d_match_fconds, e->n_match_conds, e->match_slots,
                                                         nullptr, e->d_m_mask, e->d_req_off, e->d_hits, e->d_status, e->d_hit_req,
                                                         e->max_batch);
        else
            k_fill<<<g, 256, 0, e->stream>>>(d_ns, d_ent_off, d_ent_key, d_ent_val, d_delta, n_req, e->d_match_limits,
                                             e->n_match_limits, e->d_match_ns_off, e->n_match_ns, e->d_match_conds,
                                             e->n_match_conds, nullptr, e->d_req_off, e->d_hits, e->d_status);
        HIP_TRY(e, hipGetLastError());  // (the fill pass cannot fail: the count pass saw every value it writes)
    }
    return matched_op_locked(e, op, n_hits, n_req, now, load, d_verdict, d_limited, e->match_fast, nullptr, nullptr, 0u);
}

int32_t rl_match_and_check_batch_device(rl_engine* e, const uint32_t* d_req_ns, const uint32_t* d_ent_off,
                                        const uint32_t* d_ent_key, const uint32_t* d_ent_val,
                                        const uint32_t* d_req_delta, uint32_t n_req, uint64_t now_us,
                                        int32_t load_counters, uint8_t* d_verdict, int32_t* d_limited_limit,
                                        uint32_t* n_hits_out) try {
    if (!e || !n_req || !d_req_ns || !d_ent_off || !d_req_delta || !d_verdict) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    if (n_req > e->max_batch) return fail(e, RL_ERR_BATCH_TOO_LARGE, "n_req %u > max_batch_hits %u", n_req, e->max_batch);
    HIP_TRY(e, hipSetDevice(e->device));
    return match_and_check_locked(e, RL_OP_CHECK_AND_UPDATE, d_req_ns, d_ent_off, d_ent_key, d_ent_val, d_req_delta, n_req, now_us,
                                  load_counters != 0, d_verdict, d_limited_limit, n_hits_out);
} RL_ABI_CATCH

// The results of a call on their way to the host.  A SMALL call whose result arrays all live in the engine's own pinned staging
// (rl_host_staging: what the ingest hands in) gets them by ONE kernel that stores into that memory, instead of a copy command
// per array — up to seven of them, ~8 us apiece back to back, were a quarter of a 256-message serving call; larger results,
// or a caller's own arrays, travel as copy commands like before.  The same for a small call's INPUT arrays (the kernel then reads
// the staging).  Everything is on e->stream; the caller synchronises.
struct ResultsOut {
    rl_engine* e;
    CopySegs S{};
    u64 total = 0;
    bool direct = true;
    explicit ResultsOut(rl_engine* e_) : e(e_) {}
    // (`host`: the pointer of the pair that is the host's — the destination of a result, the source of an input)
    void add(void* dst, const void* src, u64 bytes, const void* host = nullptr) {
        if (!dst || !src || !bytes) return;
        if (!host) host = dst;
        if (S.n == COPY_SEGS_MAX) {
            direct = false;
            return;
        }
        S.dst[S.n] = dst;
        S.src[S.n] = src;
        S.bytes[S.n] = bytes;
        ++S.n;
        total += bytes;
        bool inside = false;
        for (u32 q = 0; q < 4u * RL_SERVE_SETS && !inside; ++q) {
            const char* lo = static_cast<const char*>(e->h_stage[q]);
            inside = lo && static_cast<const char*>(host) >= lo && static_cast<const char*>(host) + bytes <= lo + e->h_stage_cap[q];
        }
        direct = direct && inside;
    }
    int32_t go() {
        if (!S.n) return RL_OK;
        if (direct && total <= (256u << 10) && e->results_direct) {
            k_copy_segs<<<(u32)std::min<u64>(std::max<u64>(total >> 12, 1), 64), 256, 0, e->stream>>>(S);
            HIP_TRY(e, hipGetLastError());
            return RL_OK;
        }
        for (u32 k = 0; k < S.n; ++k) HIP_TRY(e, hipMemcpyAsync(S.dst[k], S.src[k], S.bytes[k], hipMemcpyDefault, e->stream));
        return RL_OK;
    }
};

static int32_t match_batch_host(rl_engine* e, int op, const uint32_t* req_ns, const uint32_t* ent_off, const uint32_t* ent_key,
                                const uint32_t* ent_val, const uint32_t* req_delta, uint32_t n_req, uint64_t now_us,
                                int32_t load_counters, uint8_t* verdict, int32_t* limited_limit, uint32_t* req_off_out,
                                rl_hit* hits_out, uint32_t hits_cap, uint32_t* n_hits_out, uint64_t* remaining,
                                uint64_t* expires_in_us, const ServeOut* so = nullptr) {
    if (!e || !n_req || !req_ns || !ent_off || !req_delta || !verdict) return RL_ERR_INVALID;
    if (so && (!so->resp_off || !so->resp)) return RL_ERR_INVALID;
    if (op != RL_OP_CHECK_AND_UPDATE && op != RL_OP_CHECK && op != RL_OP_UPDATE) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    if (n_req > e->max_batch) return fail(e, RL_ERR_BATCH_TOO_LARGE, "n_req %u > max_batch_hits %u", n_req, e->max_batch);
    const u32 n_ent = ent_off[n_req];
    if (n_ent > e->max_batch) return fail(e, RL_ERR_BATCH_TOO_LARGE, "descriptor entries %u > max_batch_hits %u", n_ent, e->max_batch);
    if (n_ent && (!ent_key || !ent_val)) return fail(e, RL_ERR_INVALID, "ent_key / ent_val are null");
    HIP_TRY(e, hipSetDevice(e->device));
    {
        ResultsOut In(e);
        In.add(e->d_m_ns, req_ns, (u64)n_req * 4, req_ns);
        In.add(e->d_m_delta, req_delta, (u64)n_req * 4, req_delta);
        In.add(e->d_m_ent_off, ent_off, ((u64)n_req + 1) * 4, ent_off);
        if (n_ent) {
            In.add(e->d_m_ent_key, ent_key, (u64)n_ent * 4, ent_key);
            In.add(e->d_m_ent_val, ent_val, (u64)n_ent * 4, ent_val);
        }
        const int32_t irc = In.go();
        if (irc) return irc;
    }
    u32 n_hits = 0;
    int rc = match_and_check_locked(e, op, e->d_m_ns, e->d_m_ent_off, e->d_m_ent_key, e->d_m_ent_val, e->d_m_delta, n_req,
                                    now_us, load_counters != 0, e->d_verdict, limited_limit ? e->d_m_limited : nullptr,
                                    &n_hits);
    if (n_hits_out) *n_hits_out = n_hits;
    if (rc) return rc;
    ResultsOut R(e);
    R.add(verdict, e->d_verdict, n_req);
    if (so) {  // the answer as RateLimitResponse bytes instead of the counters' arrays
        rc = R.go();
        if (rc) return rc;
        rc = responses_locked(e, n_req, n_hits, nullptr, e->d_verdict, *so);  // (synchronises behind the verdicts' copy)
        if (rc) return rc;
        if (!e->resp_n_chunks[so->set < rl_engine::SERVE_SETS ? so->set : 0u]) HIP_TRY(e, hipStreamSynchronize(e->stream));
        return RL_OK;
    }
    if (limited_limit) R.add(limited_limit, e->d_m_limited, (u64)n_req * 4);
    if (req_off_out) R.add(req_off_out, e->d_req_off, ((u64)n_req + 1) * 4);
    const u32 n_copy = n_hits < hits_cap ? n_hits : hits_cap;
    if (hits_out && n_copy) R.add(hits_out, e->d_hits, (u64)n_copy * sizeof(Hit));
    if (load_counters && n_copy && remaining && expires_in_us) {
        R.add(remaining, e->d_remaining, (u64)n_copy * 8);
        R.add(expires_in_us, e->d_expires, (u64)n_copy * 8);
    }
    rc = R.go();
    if (rc) return rc;
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return RL_OK;
}

int32_t rl_match_and_check_batch(rl_engine* e, const uint32_t* req_ns, const uint32_t* ent_off, const uint32_t* ent_key,
                                 const uint32_t* ent_val, const uint32_t* req_delta, uint32_t n_req, uint64_t now_us,
                                 int32_t load_counters, uint8_t* verdict, int32_t* limited_limit, uint32_t* req_off_out,
                                 rl_hit* hits_out, uint32_t hits_cap, uint32_t* n_hits_out, uint64_t* remaining,
                                 uint64_t* expires_in_us) try {
    return match_batch_host(e, RL_OP_CHECK_AND_UPDATE, req_ns, ent_off, ent_key, ent_val, req_delta, n_req, now_us, load_counters,
                            verdict, limited_limit, req_off_out, hits_out, hits_cap, n_hits_out, remaining, expires_in_us);
} RL_ABI_CATCH

int32_t rl_match_batch_op(rl_engine* e, int32_t op, const uint32_t* req_ns, const uint32_t* ent_off, const uint32_t* ent_key,
                          const uint32_t* ent_val, const uint32_t* req_delta, uint32_t n_req, uint64_t now_us, uint8_t* verdict,
                          int32_t* limited_limit) try {
    return match_batch_host(e, op, req_ns, ent_off, ent_key, ent_val, req_delta, n_req, now_us, 0, verdict, limited_limit, nullptr,
                            nullptr, 0u, nullptr, nullptr, nullptr);
} RL_ABI_CATCH

int32_t rl_wire_table_set(rl_engine* e, const uint8_t* blob, uint32_t blob_len, const rl_wire_str* ns, uint32_t n_ns,
                          const rl_wire_str* keys, uint32_t n_keys, const rl_wire_str* vals, uint32_t n_vals,
                          const uint64_t* limit_prefix, uint32_t n_limits, const uint64_t* hash_key) try {
    if (!e || (blob_len && !blob) || !ns || (n_keys && !keys) || (n_vals && !vals) || (n_limits && !limit_prefix) || !hash_key) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    if (!e->d_match_limits || !e->match_fast || !e->match_one)
        return fail(e, RL_ERR_INVALID, "rl_wire_table_set needs the slot form of the match table (rl_match_table_set first: at most %u descriptor keys, 64 limits per namespace)", MATCH_SLOTS);
    if (blob_len > WIRE_BLOB_MAX) return fail(e, RL_ERR_INVALID, "the table's strings take %u bytes > %u", blob_len, WIRE_BLOB_MAX);
    if (n_ns != e->n_match_ns) return fail(e, RL_ERR_INVALID, "%u namespace strings for a match table of %u namespaces", n_ns, e->n_match_ns);
    if (n_vals > WIRE_LIT_TAB / 2) return fail(e, RL_ERR_INVALID, "%u condition literals > %u", n_vals, WIRE_LIT_TAB / 2);
    if (n_limits <= e->match_max_limit_id)
        return fail(e, RL_ERR_INVALID, "limit_prefix has %u rows, the match table names limit id %u", n_limits, e->match_max_limit_id);
    // The hash key NAMES the cells: a table that holds counters keyed under one key is not served under another (a restart
    // that reloaded a snapshot without bringing the ingest's key back, a second front-end with a key of its own: every limit
    // would silently start from zero and the old cells would sit there until they expire — ADVICE r05).  What is kept is a
    // fingerprint — SipHash of a constant under the key — in the engine and in the snapshot's header; never the key itself.
    static const uint8_t fp_text[] = "limitador_amd: hash key fingerprint";
    const u64 key_fp = rl_kh_bytes(fp_text, sizeof(fp_text) - 1, rl_hkey{hash_key[0], hash_key[1]}).h1 | 1ull;
    if (e->wire_key_fp && e->wire_key_fp != key_fp && e->live)
        return fail(e, RL_ERR_INVALID,
                    "the table holds %llu cells named under ANOTHER hash key: give this ingest the key the table was filled with "
                    "(rli_set_hash_key with the first ingest's rli_hash_key — before a snapshot is reloaded too), or clear the table",
                    (unsigned long long)e->live);
    auto inside = [&](const rl_wire_str& s) { return (u64)s.off + s.len <= blob_len && s.len <= 0xFFFFu; };
    for (u32 i = 0; i < n_ns; ++i)
        if (!inside(ns[i])) return fail(e, RL_ERR_INVALID, "namespace string %u lies outside the blob", i);
    for (u32 i = 0; i < n_keys; ++i)  // (a key's `len`: bits 0..23 the length, bits 24..31 the descriptor it is read from)
        if (!inside(rl_wire_str{keys[i].off, keys[i].len & 0xFFFFFFu})) return fail(e, RL_ERR_INVALID, "key string %u lies outside the blob", i);
    for (u32 i = 0; i < n_vals; ++i)
        if (!inside(vals[i])) return fail(e, RL_ERR_INVALID, "value string %u lies outside the blob", i);
    HIP_TRY(e, hipSetDevice(e->device));
    WireTables W{};
    W.desc_mask = 1ull;
    for (u32 sl = 0; sl < e->match_slots.n; ++sl) {
        const u32 kid = e->match_slots.key[sl];
        if (kid >= n_keys) return fail(e, RL_ERR_INVALID, "the match table reads key id %u, %u key strings were given", kid, n_keys);
        W.slot_key[sl] = WireStr{keys[kid].off, keys[kid].len & 0xFFFFFFu};
        W.slot_desc[sl] = keys[kid].len >> 24;
        if (W.slot_desc[sl] > 63u) return fail(e, RL_ERR_INVALID, "key %u reads descriptors[%u]: at most descriptors[63]", kid, W.slot_desc[sl]);
        W.desc_mask |= 1ull << W.slot_desc[sl];
    }
    std::vector<WireLit> lit(WIRE_LIT_TAB, WireLit{0ull, 0u, 0, 0xFFFFu});
    for (u32 id = 0; id < n_vals; ++id) {
        const rl_h128 h = rl_kh_bytes(blob + vals[id].off, vals[id].len, rl_hkey{hash_key[0], hash_key[1]});
        u32 q = (u32)h.h1 & (WIRE_LIT_TAB - 1u);
        while (lit[q].id != 0xFFFFu) q = (q + 1u) & (WIRE_LIT_TAB - 1u);
        lit[q] = WireLit{h.h1, vals[id].off, (unsigned short)vals[id].len, (unsigned short)id};
    }
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    const size_t mb = e->max_batch;
    auto need = [&](void** p, size_t bytes) { return *p || hipMalloc(p, bytes ? bytes : 16) == hipSuccess; };
    if (!need((void**)&e->d_w_blob, WIRE_BLOB_MAX) || !need((void**)&e->d_w_ns, MATCH_LDS_NS * sizeof(WireStr)) ||
        !need((void**)&e->d_w_lit, WIRE_LIT_TAB * sizeof(WireLit)) || !need((void**)&e->d_w_prefix, 4096 * 2 * sizeof(u64)) ||
        !need((void**)&e->d_w_off, (mb + 1) * sizeof(u32)) || !need((void**)&e->d_w_status, mb * sizeof(int32_t)) ||
        !need((void**)&e->d_w_slot_h, mb * MATCH_SLOTS * sizeof(uint4)) || !need((void**)&e->d_hit_check, mb * sizeof(u32)))
        return fail(e, RL_ERR_NOMEM, "hipMalloc of the wire path's tables / staging failed");
    if (blob_len) HIP_TRY(e, hipMemcpy(e->d_w_blob, blob, blob_len, hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->d_w_ns, ns, n_ns * sizeof(WireStr), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->d_w_lit, lit.data(), lit.size() * sizeof(WireLit), hipMemcpyHostToDevice));
    if (n_limits) HIP_TRY(e, hipMemcpy(e->d_w_prefix, limit_prefix, (size_t)std::min(n_limits, 4096u) * 2 * sizeof(u64), hipMemcpyHostToDevice));
    W.blob = e->d_w_blob;
    W.blob_len = blob_len;
    W.ns = e->d_w_ns;
    W.n_ns = n_ns;
    W.lit = e->d_w_lit;
    W.prefix = e->d_w_prefix;
    W.var_slot_mask = e->match_var_slots;
    W.hkey = rl_hkey{hash_key[0], hash_key[1]};
    e->wire_t = W;
    e->wire_ready = true;
    e->wire_key_fp = key_fp;
    return RL_OK;
} RL_ABI_CATCH

static int32_t host_staging_locked(rl_engine* e, uint32_t slot, uint64_t bytes, void** out);
int32_t rl_host_staging(rl_engine* e, uint32_t slot, uint64_t bytes, void** out) try {
    if (!e || !out || slot >= 4u * RL_SERVE_SETS) return RL_ERR_INVALID;
    // A buffer that is large enough already is handed out WITHOUT the engine's mutex: a slot belongs to one serving set, a set
    // to one caller at a time (rl_wire_serve_batch_set), so nobody else resizes it — and the caller of the other set, which
    // holds the mutex for the whole of its copy-in + decide, must not keep this one from packing its messages meanwhile.
    if (bytes <= e->h_stage_cap[slot]) {
        *out = e->h_stage[slot];
        return RL_OK;
    }
    EngineLock g(e);
    return host_staging_locked(e, slot, bytes, out);
} RL_ABI_CATCH

static int32_t host_staging_locked(rl_engine* e, uint32_t slot, uint64_t bytes, void** out) {
    if (bytes > e->h_stage_cap[slot]) {
        HIP_TRY(e, hipSetDevice(e->device));
        HIP_TRY(e, hipStreamSynchronize(e->stream));
        if (e->h_stage[slot]) (void)hipHostFree(e->h_stage[slot]);
        e->h_stage[slot] = nullptr;
        e->h_stage_cap[slot] = 0;
        u64 cap = 1u << 20;
        while (cap < bytes) cap <<= 1;
        if (hipHostMalloc(&e->h_stage[slot], cap, hipHostMallocDefault) != hipSuccess)
            return fail(e, RL_ERR_NOMEM, "hipHostMalloc of %llu bytes of host staging failed", (unsigned long long)cap);
        e->h_stage_cap[slot] = cap;
    }
    *out = e->h_stage[slot];
    return RL_OK;
}

static int32_t wire_match_host(rl_engine* e, int op, const uint8_t* wire, const uint32_t* msg_off, uint32_t n, uint64_t now_us,
                               int32_t load_counters, uint8_t* verdict, int32_t* limited_limit, int32_t* status,
                               uint32_t* req_off_out, rl_hit* hits_out, uint32_t hits_cap, uint32_t* n_hits_out,
                               uint64_t* remaining, uint64_t* expires_in_us, int64_t* collided_message,
                               const ServeOut* so = nullptr) {
    if (!e || !n || !msg_off || !verdict || !status) return RL_ERR_INVALID;
    if (so && (!so->resp_off || !so->resp)) return RL_ERR_INVALID;
    if (op != RL_OP_CHECK_AND_UPDATE && op != RL_OP_CHECK && op != RL_OP_UPDATE) return RL_ERR_INVALID;
    if (collided_message) *collided_message = -1;
    if (n > e->max_batch) {
        EngineLock gl(e);
        return fail(e, RL_ERR_BATCH_TOO_LARGE, "n %u > max_batch_hits %u", n, e->max_batch);
    }
    const u64 bytes = msg_off[n];
    bool offsets_ok = msg_off[0] == 0 && (!bytes || wire) && bytes <= 0xFFFFFFFFull - 64;
    for (u32 i = 0; i < n && offsets_ok; ++i)  // (the device walks [msg_off[i], msg_off[i + 1]) of the staging: never outside it)
        offsets_ok = msg_off[i] <= msg_off[i + 1];
    // ---- a serving call's messages travel BEFORE the engine's mutex is taken (round 6): on the copy stream, into the set's
    //      own device buffers — 16 MB / 0.3 ms for 262 144 messages that used to sit inside the serialised part of the call,
    //      in front of its kernels, now run beside the other set's decide phase.  Only when the set's buffers are large
    //      enough already (they are only ever resized by the set's own calls, under the mutex, below). ---------------------
    const u32 set = so && so->set < rl_engine::SERVE_SETS ? so->set : 0u;
    uint8_t*& d_bytes_set = set ? e->d_w_bytes_more[set - 1] : e->d_w_bytes;
    u64& cap_set = set ? e->w_bytes_cap_more[set - 1] : e->w_bytes_cap;
    u32*& d_off_set = set ? e->d_w_off_more[set - 1] : e->d_w_off;
    bool copied_early = false;
    if (so && offsets_ok && e->wire_ready && e->in_stream && e->in_ev[set] && d_off_set && bytes <= cap_set) {
        if (hipSetDevice(e->device) == hipSuccess &&
            (!bytes || hipMemcpyAsync(d_bytes_set, wire, bytes, hipMemcpyHostToDevice, e->in_stream) == hipSuccess) &&
            hipMemcpyAsync(d_off_set, msg_off, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, e->in_stream) == hipSuccess &&
            hipEventRecord(e->in_ev[set], e->in_stream) == hipSuccess)
            copied_early = true;
        else
            (void)hipGetLastError();
    }
    const bool wm_trace = RL_EXP_ENV("RL_WIRE_TRACE") != nullptr;
    const auto wm_t0 = std::chrono::steady_clock::now();
    auto wm_lap = [&](const char* what) {
        if (wm_trace) {
            const auto t = std::chrono::steady_clock::now();
            const double abs_us = std::chrono::duration<double, std::micro>(t.time_since_epoch()).count();
            std::fprintf(stderr, "[wm] set %u %-12s +%8.1f us  (clock %12.1f us)\n", so ? so->set : 0u, what,
                         std::chrono::duration<double, std::micro>(t - wm_t0).count(), abs_us - 1e8 * (double)(long long)(abs_us / 1e8));
        }
    };
    EngineLock g(e);
    wm_lap("locked");
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    if (!e->wire_ready) return fail(e, RL_ERR_INVALID, "rl_wire_table_set was not called for the installed match table");
    if (msg_off[0] != 0 || (bytes && !wire)) return fail(e, RL_ERR_INVALID, "msg_off[0] must be 0 and wire non-null");
    if (bytes > 0xFFFFFFFFull - 64) return fail(e, RL_ERR_BATCH_TOO_LARGE, "the messages take %llu bytes", (unsigned long long)bytes);
    if (!offsets_ok) return fail(e, RL_ERR_INVALID, "msg_off is not non-decreasing");
    HIP_TRY(e, hipSetDevice(e->device));
    if (!d_off_set && hipMalloc((void**)&d_off_set, ((size_t)e->max_batch + 1) * sizeof(u32)) != hipSuccess)
        return fail(e, RL_ERR_NOMEM, "hipMalloc of the message offsets failed");
    if (so && !e->in_stream) HIP_TRY(e, hipStreamCreateWithFlags(&e->in_stream, hipStreamNonBlocking));
    if (so && !e->in_ev[set]) HIP_TRY(e, hipEventCreateWithFlags(&e->in_ev[set], hipEventDisableTiming));
    if (bytes > cap_set) {
        HIP_TRY(e, hipStreamSynchronize(e->stream));
        if (d_bytes_set) (void)hipFree(d_bytes_set);
        d_bytes_set = nullptr;
        cap_set = 0;
        u64 cap = 1u << 16;
        while (cap < bytes) cap <<= 1;
        if (hipMalloc((void**)&d_bytes_set, cap) != hipSuccess) return fail(e, RL_ERR_NOMEM, "hipMalloc of %llu bytes of message staging failed", (unsigned long long)cap);
        cap_set = cap;
    }
    if (copied_early) {
        // (a wait command costs its stream ~9 us even when its event completed long ago: where the host can see the copy
        // complete — the usual case when the other set's call held the mutex meanwhile — the stream is not made to wait)
        if (hipEventQuery(e->in_ev[set]) != hipSuccess) {
            (void)hipGetLastError();
            HIP_TRY(e, hipStreamWaitEvent(e->stream, e->in_ev[set], 0));
        }
    } else {
        ResultsOut In(e);
        In.add(d_bytes_set, wire, bytes, wire);
        In.add(d_off_set, msg_off, ((u64)n + 1) * 4, msg_off);
        const int32_t irc = In.go();
        if (irc) return irc;
    }
    const u32 gq = cdiv(n, 256);
    const u32 call = ++e->m_call ? e->m_call : ++e->m_call;
    const MatchTables T{e->d_match_flimits, e->n_match_limits, e->d_match_ns_off, e->n_match_ns, e->d_match_fconds,
                        e->n_match_conds, e->match_slots};
    k_wire_count<<<gq, 256, 0, e->stream>>>(d_bytes_set, d_off_set, n, e->wire_t, T, e->d_m_ns, e->d_m_delta, e->d_w_status,
                                            e->d_m_mask, e->d_w_slot_h, e->d_m_scan1);
    k_match_scan2<<<1, 1024, 0, e->stream>>>(e->d_m_scan1, gq, e->d_req_off + n, e->h_m_word, call);
    k_wire_fill<<<gq, 256, 0, e->stream>>>(e->d_m_ns, e->d_m_delta, n, T, e->d_w_prefix, e->wire_t.hkey, e->d_m_mask, e->d_w_slot_h, e->d_m_scan1,
                                           e->d_req_off, e->d_hits, e->d_hit_check, e->d_hit_req, e->max_batch);
    HIP_TRY(e, hipGetLastError());
    wm_lap("enqueued");
    int rc = wait_word(e, e->h_m_word + 3, call, "the wire path's count pass");
    wm_lap("counted");
    if (rc) return rc;
    const u32 n_hits = e->h_m_word[0], m_err = e->h_m_word[1];
    if (n_hits_out) *n_hits_out = n_hits;
    if (m_err || n_hits > e->max_batch) HIP_TRY(e, hipStreamSynchronize(e->stream));  // refused
    if (m_err) return status_to_error(e, m_err);
    if (n_hits > e->max_batch) return fail(e, RL_ERR_BATCH_TOO_LARGE, "the messages expand to %u counters > max_batch_hits %u", n_hits, e->max_batch);
    // (CheckRateLimit over the wire checks with delta 1 whatever hits_addend says: kuadrant_service.rs:62-64)
    rc = matched_op_locked(e, op, n_hits, n, now_us, load_counters != 0, e->d_verdict, limited_limit ? e->d_m_limited : nullptr,
                           true, e->d_hit_check, e->d_w_status, op == RL_OP_CHECK ? 1u : 0u);
    if (rc == RL_ERR_KEY_COLLISION) {
        // status[] names EVERY message that carries a colliding counter (-103); *collided_message is one of them
        if (collided_message && e->collide_hit < n_hits) {
            u32 req = 0;
            if (hipMemcpy(&req, e->d_hit_req + e->collide_hit, sizeof(u32), hipMemcpyDeviceToHost) == hipSuccess) *collided_message = req;
        }
        (void)hipMemcpy(status, e->d_w_status, (size_t)n * 4, hipMemcpyDeviceToHost);
    }
    if (rc) return rc;
    wm_lap("decided");
    ResultsOut R(e);
    R.add(verdict, e->d_verdict, n);
    R.add(status, e->d_w_status, (u64)n * 4);
    if (so) {  // the answer as RateLimitResponse bytes instead of the counters' arrays
        rc = R.go();
        if (rc) return rc;
        rc = responses_locked(e, n, n_hits, e->d_w_status, e->d_verdict, *so);  // (synchronises behind the verdicts' / statuses' copies)
        wm_lap("responses");
        if (rc) return rc;
        if (!e->resp_n_chunks[so->set < rl_engine::SERVE_SETS ? so->set : 0u]) HIP_TRY(e, hipStreamSynchronize(e->stream));
        return RL_OK;
    }
    if (limited_limit) R.add(limited_limit, e->d_m_limited, (u64)n * 4);
    if (req_off_out) R.add(req_off_out, e->d_req_off, ((u64)n + 1) * 4);
    const u32 n_copy = n_hits < hits_cap ? n_hits : hits_cap;
    if (hits_out && n_copy) R.add(hits_out, e->d_hits, (u64)n_copy * sizeof(Hit));
    if (load_counters && n_copy && remaining && expires_in_us) {
        R.add(remaining, e->d_remaining, (u64)n_copy * 8);
        R.add(expires_in_us, e->d_expires, (u64)n_copy * 8);
    }
    rc = R.go();
    if (rc) return rc;
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return RL_OK;
}

int32_t rl_wire_match_and_check_batch(rl_engine* e, const uint8_t* wire, const uint32_t* msg_off, uint32_t n, uint64_t now_us,
                                      int32_t load_counters, uint8_t* verdict, int32_t* limited_limit, int32_t* status,
                                      uint32_t* req_off_out, rl_hit* hits_out, uint32_t hits_cap, uint32_t* n_hits_out,
                                      uint64_t* remaining, uint64_t* expires_in_us, int64_t* collided_message) try {
    return wire_match_host(e, RL_OP_CHECK_AND_UPDATE, wire, msg_off, n, now_us, load_counters, verdict, limited_limit, status,
                           req_off_out, hits_out, hits_cap, n_hits_out, remaining, expires_in_us, collided_message);
} RL_ABI_CATCH

int32_t rl_resp_table_set(rl_engine* e, const uint8_t* blob, uint32_t blob_len, const rl_wire_str* frag, uint32_t n_limits) try {
    if (!e || (blob_len && !blob) || (n_limits && !frag)) return RL_ERR_INVALID;
    EngineLock g(e);
    if (engine_busy(e)) return fail(e, RL_ERR_BUSY, "batches are in flight (rl_check_and_update_collect) or a phased pass is open (rl_gen_commit_device / rl_gen_abort)");
    for (u32 i = 0; i < n_limits; ++i)
        if ((u64)frag[i].off + frag[i].len > blob_len) return fail(e, RL_ERR_INVALID, "fragment %u lies outside the blob", i);
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (e->resp_stream) HIP_TRY(e, hipStreamSynchronize(e->resp_stream));  // (response kernels of a call still in flight read the fragments)
    if (e->d_resp_blob) (void)hipFree(e->d_resp_blob);
    if (e->d_resp_frag) (void)hipFree(e->d_resp_frag);
    e->d_resp_blob = nullptr;
    e->d_resp_frag = nullptr;
    e->resp_ready = false;
    if (hipMalloc((void**)&e->d_resp_blob, ((size_t)blob_len + 31) & ~(size_t)15) != hipSuccess ||
        hipMalloc((void**)&e->d_resp_frag, (n_limits ? n_limits : 1) * sizeof(WireStr)) != hipSuccess)
        return fail(e, RL_ERR_NOMEM, "hipMalloc of the response fragments failed");
    if (blob_len) HIP_TRY(e, hipMemcpy(e->d_resp_blob, blob, blob_len, hipMemcpyHostToDevice));
    if (n_limits) HIP_TRY(e, hipMemcpy(e->d_resp_frag, frag, (size_t)n_limits * sizeof(WireStr), hipMemcpyHostToDevice));
    e->n_resp_frag = n_limits;
    e->resp_max_frag = 0;
    for (u32 i = 0; i < n_limits; ++i) e->resp_max_frag = std::max(e->resp_max_frag, frag[i].len);
    e->resp_blob_len = blob_len;
    e->resp_ready = true;
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_resp_table_ready(rl_engine* e) { return e && e->resp_ready ? 1 : 0; }

int32_t rl_match_serve_batch(rl_engine* e, const uint32_t* req_ns, const uint32_t* ent_off, const uint32_t* ent_key,
                             const uint32_t* ent_val, const uint32_t* req_delta, uint32_t n_req, uint64_t now_us,
                             uint32_t flags, uint8_t* verdict, const uint32_t** resp_off, const uint8_t** resp) try {
    const int32_t with_headers = (flags & RL_SERVE_HEADERS) ? 1 : 0;
    const ServeOut so{with_headers, (flags & RL_SERVE_ASYNC) != 0, resp_off, resp};
    return match_batch_host(e, RL_OP_CHECK_AND_UPDATE, req_ns, ent_off, ent_key, ent_val, req_delta, n_req, now_us, with_headers,
                            verdict, nullptr, nullptr, nullptr, 0u, nullptr, nullptr, nullptr, &so);
} RL_ABI_CATCH

int32_t rl_wire_serve_batch(rl_engine* e, const uint8_t* wire, const uint32_t* msg_off, uint32_t n, uint64_t now_us,
                            uint32_t flags, uint8_t* verdict, int32_t* status, const uint32_t** resp_off,
                            const uint8_t** resp, int64_t* collided_message) try {
    const int32_t with_headers = (flags & RL_SERVE_HEADERS) ? 1 : 0;
    const ServeOut so{with_headers, (flags & RL_SERVE_ASYNC) != 0, resp_off, resp};
    return wire_match_host(e, RL_OP_CHECK_AND_UPDATE, wire, msg_off, n, now_us, with_headers, verdict, nullptr, status, nullptr,
                           nullptr, 0u, nullptr, nullptr, nullptr, collided_message, &so);
} RL_ABI_CATCH

// (no engine lock: called by the host layer's scatter threads side by side, between a serving call on the set and the next
// call on the SAME set; the events and the chunk ends were written by that serving call)
int32_t rl_serve_wait_set(rl_engine* e, uint32_t set, uint64_t upto) try {
    if (!e || set >= rl_engine::SERVE_SETS) return RL_ERR_INVALID;
    if (!upto || !e->resp_n_chunks[set]) return RL_OK;
    // (a piece's event says the pieces before it are complete too: one stream, in order)
    u32 c = 0;
    while (c + 1 < e->resp_n_chunks[set] && e->resp_chunk_end[set][c] < upto) ++c;
    rl_engine::LazyCopy& L = e->lazy[set];
    if (L.n && L.issued.load(std::memory_order_acquire) <= c) {
        // (lazily issued pieces: whoever needs piece c first issues the pieces up to it, each behind the one before; a piece that
        // has been issued is waited for without the lock)
        std::lock_guard<std::mutex> lg(L.mu);
        while (L.issued.load(std::memory_order_relaxed) <= c && L.issued.load(std::memory_order_relaxed) < L.n) {
            const u32 k = L.issued.load(std::memory_order_relaxed);
            if (k >= e->resp_lazy_depth && hipEventSynchronize(e->resp_ev[set][k - e->resp_lazy_depth]) != hipSuccess) return RL_ERR_DEVICE;  // resp_lazy_depth pieces in flight
            if (L.hi[k] > L.lo[k] &&
                hipMemcpyAsync(L.dst + L.lo[k], e->d_resp_set[set] + L.lo[k], L.hi[k] - L.lo[k], hipMemcpyDeviceToHost, L.stream) != hipSuccess)
                return RL_ERR_DEVICE;
            if (hipEventRecord(e->resp_ev[set][k], L.stream) != hipSuccess) return RL_ERR_DEVICE;
            L.issued.store(k + 1, std::memory_order_release);
        }
    }
    return hipEventSynchronize(e->resp_ev[set][c]) == hipSuccess ? (int32_t)RL_OK : (int32_t)RL_ERR_DEVICE;
} RL_ABI_CATCH

int32_t rl_serve_wait(rl_engine* e, uint64_t upto) try { return rl_serve_wait_set(e, 0, upto); } RL_ABI_CATCH

int32_t rl_wire_serve_batch_set(rl_engine* e, uint32_t set, const uint8_t* wire, const uint32_t* msg_off, uint32_t n, uint64_t now_us,
                                uint32_t flags, uint8_t* verdict, int32_t* status, const uint32_t** resp_off,
                                const uint8_t** resp, int64_t* collided_message) try {
    if (set >= rl_engine::SERVE_SETS) return RL_ERR_INVALID;
    const int32_t with_headers = (flags & RL_SERVE_HEADERS) ? 1 : 0;
    const ServeOut so{with_headers, (flags & RL_SERVE_ASYNC) != 0, resp_off, resp, set};
    return wire_match_host(e, RL_OP_CHECK_AND_UPDATE, wire, msg_off, n, now_us, with_headers, verdict, nullptr, status, nullptr,
                           nullptr, 0u, nullptr, nullptr, nullptr, collided_message, &so);
} RL_ABI_CATCH

int32_t rl_wire_match_batch_op(rl_engine* e, int32_t op, const uint8_t* wire, const uint32_t* msg_off, uint32_t n, uint64_t now_us,
                               uint8_t* verdict, int32_t* limited_limit, int32_t* status, int64_t* collided_message) try {
    return wire_match_host(e, op, wire, msg_off, n, now_us, 0, verdict, limited_limit, status, nullptr, nullptr, 0u, nullptr,
                           nullptr, nullptr, collided_message);
} RL_ABI_CATCH

uint32_t rl_owner_of(uint64_t key, uint64_t hash_seed, uint32_t world) { return owner_of(key, hash_seed, world); }

static int32_t route_partition_on(rl_engine* e, hipStream_t st, bool block, const rl_hit* d_hits, uint32_t n_hits,
                                  uint32_t world, rl_hit* d_out, uint32_t* d_perm, uint32_t* d_counts) {
    if (!e || !d_counts || (n_hits && (!d_hits || !d_out || !d_perm))) return RL_ERR_INVALID;
    // The caller's-stream form takes no engine lock and does not switch devices: it reads two immutable fields and a scratch
    // that stream order protects (every routed step of a communicator goes through ONE stream), and a router calls it while
    // its helper thread hands a batch to the same engine — behind the engine's mutex each launch here waited for that.
    std::unique_lock<std::mutex> g(e->mu, std::defer_lock);
    if (block) {
        g.lock();
        serve_stop(e);
    }
    if (world == 0 || world > ROUTE_MAX_WORLD) return fail(e, RL_ERR_INVALID, "world %u not in [1,%d]", world, ROUTE_MAX_WORLD);
    if (block) HIP_TRY(e, hipSetDevice(e->device));
    const u32 nblk = n_hits ? cdiv(n_hits, ROUTE_TILE) : 0;
    if (nblk > ROUTE_MAX_BLOCKS) return fail(e, RL_ERR_BATCH_TOO_LARGE, "n_hits %u too large for the router", n_hits);
    // count / scan / scatter
    if (nblk)
        k_route_count<<<nblk, ROUTE_BLOCK, 0, st>>>(reinterpret_cast<const Hit*>(d_hits), n_hits, e->seed, world, e->d_route_cnt);
    k_route_scan<<<1, 256, 0, st>>>(e->d_route_cnt, nblk, world, d_counts);
    if (nblk)
        k_route_scatter<<<nblk, ROUTE_BLOCK, 0, st>>>(reinterpret_cast<const Hit*>(d_hits), n_hits, e->seed, world, e->d_route_cnt,
                                                      reinterpret_cast<Hit*>(d_out), d_perm);
    HIP_TRY(e, hipGetLastError());
    if (block) HIP_TRY(e, hipStreamSynchronize(st));
    return RL_OK;
}

int32_t rl_route_partition_device(rl_engine* e, const rl_hit* d_hits, uint32_t n_hits, uint32_t world,
                                  rl_hit* d_out, uint32_t* d_perm, uint32_t* d_counts) try {
    if (!e) return RL_ERR_INVALID;
    return route_partition_on(e, e->stream, !e->external_stream, d_hits, n_hits, world, d_out, d_perm, d_counts);
} RL_ABI_CATCH

int32_t rl_route_partition_stream(rl_engine* e, void* stream, const rl_hit* d_hits, uint32_t n_hits, uint32_t world,
                                  rl_hit* d_out, uint32_t* d_perm, uint32_t* d_counts) try {
    if (!e) return RL_ERR_INVALID;
    return route_partition_on(e, reinterpret_cast<hipStream_t>(stream), false, d_hits, n_hits, world, d_out, d_perm, d_counts);
} RL_ABI_CATCH

static int32_t unpermute_on(rl_engine* e, hipStream_t st, bool block, const uint8_t* d_src, const uint32_t* d_perm,
                            uint32_t n, uint8_t* d_dst) {
    if (!e || (n && (!d_src || !d_perm || !d_dst))) return RL_ERR_INVALID;
    std::unique_lock<std::mutex> g(e->mu, std::defer_lock);  // (the caller's-stream form: no lock, see route_partition_on)
    if (block) {
        g.lock();
        serve_stop(e);
        HIP_TRY(e, hipSetDevice(e->device));
    }
    if (n) k_unpermute_u8<<<cdiv(n, 256), 256, 0, st>>>(d_src, d_perm, n, d_dst);
    HIP_TRY(e, hipGetLastError());
    if (block) HIP_TRY(e, hipStreamSynchronize(st));
    return RL_OK;
}

int32_t rl_unpermute_u8_device(rl_engine* e, const uint8_t* d_src, const uint32_t* d_perm, uint32_t n,
                               uint8_t* d_dst) try {
    if (!e) return RL_ERR_INVALID;
    return unpermute_on(e, e->stream, !e->external_stream, d_src, d_perm, n, d_dst);
} RL_ABI_CATCH

int32_t rl_unpermute_u8_stream(rl_engine* e, void* stream, const uint8_t* d_src, const uint32_t* d_perm, uint32_t n,
                               uint8_t* d_dst) try {
    if (!e) return RL_ERR_INVALID;
    return unpermute_on(e, reinterpret_cast<hipStream_t>(stream), false, d_src, d_perm, n, d_dst);
} RL_ABI_CATCH

int32_t rl_copy_segments_stream(rl_engine* e, void* stream, const rl_copy_seg* segs, uint32_t n) try {
    if (!e || (n && !segs) || n > COPY_SEGS_MAX) return RL_ERR_INVALID;
    if (!n) return RL_OK;
    CopySegs S{};
    u64 most = 0;
    for (u32 k = 0; k < n; ++k) {
        S.dst[k] = segs[k].dst;
        S.src[k] = segs[k].src;
        S.bytes[k] = segs[k].bytes;
        most = std::max<u64>(most, segs[k].bytes);
    }
    S.n = n;
    if (!most) return RL_OK;
    const u32 grid = (u32)std::min<u64>(std::max<u64>(most >> 14, 1), 2048);
    // (no engine lock: the launch touches nothing of the engine's but its device)
    k_copy_segs<<<grid, 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(S);
    return hipGetLastError() == hipSuccess ? RL_OK : RL_ERR_DEVICE;
} RL_ABI_CATCH

int32_t rl_engine_info(rl_engine* e, int32_t* device, uint32_t* max_batch_hits) try {
    if (!e) return RL_ERR_INVALID;
    if (device) *device = e->device;
    if (max_batch_hits) *max_batch_hits = e->max_batch;
    return RL_OK;
} RL_ABI_CATCH

// ---- ingress side of the key-sharded multi-counter step (rl_route.hpp): enqueue on the caller's stream and return -----
int32_t rl_req_ids_stream(rl_engine* e, void* stream, const uint32_t* d_req_off, uint32_t n_req, uint32_t n_hits,
                          uint32_t base, const uint32_t* d_perm, uint32_t* d_req_of_hit, uint32_t* d_req_id_sorted) try {
    if (!e || !d_req_off || (n_hits && (!d_perm || !d_req_of_hit || !d_req_id_sorted))) return RL_ERR_INVALID;
    EngineLock g(e);
    HIP_TRY(e, hipSetDevice(e->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (n_req && n_hits) k_req_of_hit<<<cdiv(n_req, 256), 256, 0, st>>>(d_req_off, n_req, d_req_of_hit);
    if (n_hits) k_req_id_sorted<<<cdiv(n_hits, 256), 256, 0, st>>>(d_req_of_hit, d_perm, n_hits, base, d_req_id_sorted);
    HIP_TRY(e, hipGetLastError());
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_req_round_stream(rl_engine* e, void* stream, const uint8_t* d_pass_sorted, const uint32_t* d_perm,
                            const uint32_t* d_req_off, const uint32_t* d_req_of_hit, uint32_t n_req, uint32_t n_hits,
                            int32_t first_round, uint8_t* d_pass_home, uint8_t* d_adm, int32_t* d_first, uint8_t* d_verdict,
                            uint32_t* d_changed, uint8_t* d_adm_sorted) try {
    if (!e || !d_changed || (n_req && (!d_req_off || !d_adm || !d_first || !d_verdict)) ||
        (n_hits && (!d_pass_sorted || !d_perm || !d_req_of_hit || !d_pass_home || !d_adm_sorted)))
        return RL_ERR_INVALID;
    EngineLock g(e);
    HIP_TRY(e, hipSetDevice(e->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (n_hits) k_unpermute_u8<<<cdiv(n_hits, 256), 256, 0, st>>>(d_pass_sorted, d_perm, n_hits, d_pass_home);
    if (n_req)
        k_req_and<<<cdiv(n_req, 256), 256, 0, st>>>(d_pass_home, d_req_off, n_req, first_round ? 1u : 0u, d_adm, d_first, d_verdict,
                                                    d_changed);
    if (n_hits) k_req_spread<<<cdiv(n_hits, 256), 256, 0, st>>>(d_adm, d_req_of_hit, d_perm, n_hits, d_adm_sorted);
    HIP_TRY(e, hipGetLastError());
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_req_reached_stream(rl_engine* e, void* stream, const int32_t* d_first, const uint32_t* d_req_of_hit,
                              const uint32_t* d_perm, uint32_t n_hits, uint8_t* d_reached_sorted) try {
    if (!e || (n_hits && (!d_first || !d_req_of_hit || !d_perm || !d_reached_sorted))) return RL_ERR_INVALID;
    EngineLock g(e);
    HIP_TRY(e, hipSetDevice(e->device));
    if (n_hits)
        k_req_reached<<<cdiv(n_hits, 256), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(d_first, d_req_of_hit, d_perm, n_hits,
                                                                                             d_reached_sorted);
    HIP_TRY(e, hipGetLastError());
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_unpermute_u64_stream(rl_engine* e, void* stream, const uint64_t* d_src, const uint32_t* d_perm, uint32_t n,
                                uint64_t* d_dst) try {
    if (!e || (n && (!d_src || !d_perm || !d_dst))) return RL_ERR_INVALID;
    EngineLock g(e);
    HIP_TRY(e, hipSetDevice(e->device));
    if (n)
        k_unpermute_u64<<<cdiv(n, 256), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(reinterpret_cast<const u64*>(d_src), d_perm, n,
                                                                                          reinterpret_cast<u64*>(d_dst));
    HIP_TRY(e, hipGetLastError());
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_host_register(rl_engine* e, void* ptr, uint64_t bytes) try {
    if (!e || !ptr || !bytes) return RL_ERR_INVALID;
    EngineLock g(e);
    HIP_TRY(e, hipSetDevice(e->device));
    if (hipHostRegister(ptr, bytes, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();
        return fail(e, RL_ERR_INVALID, "hipHostRegister refused %llu bytes at %p", (unsigned long long)bytes, ptr);
    }
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_host_unregister(rl_engine* e, void* ptr) try {
    if (!e || !ptr) return RL_ERR_INVALID;
    EngineLock g(e);
    HIP_TRY(e, hipSetDevice(e->device));
    // (copies of host-buffer calls are complete when the call returns: nothing of ours still reads the range)
    if (hipHostUnregister(ptr) != hipSuccess) {
        (void)hipGetLastError();
        return fail(e, RL_ERR_INVALID, "hipHostUnregister: %p is not a registered range", ptr);
    }
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_kernel_timing(rl_engine* e, int32_t enable) try {
    if (!e) return RL_ERR_INVALID;
    EngineLock g(e);
    if (enable < 0 || enable > 3) return fail(e, RL_ERR_INVALID, "timing mode %d", enable);
    e->timing = enable;
    return RL_OK;
} RL_ABI_CATCH

int32_t rl_kernel_timing_read(rl_engine* e, double* ms, uint64_t* launches, int32_t reset) try {
    if (!e) return RL_ERR_INVALID;
    EngineLock g(e);
    if (ms)
        for (int q = 0; q < RL_TIMING_SLOTS; ++q) ms[q] = e->ms_slot[q];
    if (launches) *launches = e->timed_launches;
    if (reset) {
        for (auto& v : e->ms_slot) v = 0;
        e->timed_launches = 0;
    }
    return RL_OK;
} RL_ABI_CATCH

}  // extern "C"
