// rl_cell.hpp — HBM data layout of the counter table and the device-side helpers every
// kernel shares.  gfx950 only.
//
// One counter cell is ONE 64-byte line.  Every access on the hot path is a random access to
// a single cell, so the cell's persistent state (what the reference keeps in the 16-byte
// AtomicExpiringValue, limitador/src/storage/atomic_expiring_value.rs:5-9, plus its key) and
// the per-batch scratch the kernels need (pending sum, hit count, flags) share the line: a hit
// costs one line from HBM, and the later phases of the same batch find it in L2 / MALL.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rl {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr u64 TAG_EMPTY = 0xFFFFFFFFFFFFFFFFull;
constexpr u64 TAG_TOMB = 0xFFFFFFFFFFFFFFFEull;
constexpr u32 SIMPLE_FLAG = 0x80000000u;

// amb states (per-batch scratch)
constexpr u32 AMB_NONE = 0;     // cell decided without looking at trace order
constexpr u32 AMB_PENDING = 1;  // some hit of this batch needs trace-order resolution
constexpr u32 AMB_ADMIT = 2;    // resolved: at least one hit admitted, aux = final value
constexpr u32 AMB_DENY = 3;     // resolved: nothing admitted, cell unchanged

struct alignas(64) Cell {
    u64 tag;     //  0  key, TAG_EMPTY or TAG_TOMB                       } one dwordx4
    u64 value;   //  8  AtomicExpiringValue.value                        }
    u64 expiry;  // 16  AtomicExpiringValue.expiry (us since epoch)      } one dwordx4: what the hot path
    u32 limit;   // 24  limit id | SIMPLE_FLAG (attribute of the cell)   } reads besides tag + value
    u32 cnt;     // 28  reserved                                         }
    // ---- per-batch scratch of the general resolver / first-generation pipeline: 32 contiguous bytes,
    //      zero between batches, cleared with two 16-byte stores ------------------------------------
    u64 pend;    // 32  (hit count << 40) | sum of deltas of this batch
    u64 aux;     // 40  resolver's final value, or (idx+1)<<32|delta for 0-second windows
    u32 amb;     // 48  AMB_*
    u32 nonuni;  // 52  ordered segment has non-uniform deltas / needs sequential walk
    u32 seg;     // 56  start of this cell's segment in the sorted ordered list
    u32 pad;     // 60  general resolver: 1 = created by this batch, not yet reached; 2 = reached
};
static_assert(sizeof(Cell) == 64, "one cell = one 64-byte line");

// Device copy of one rl_limit_row, window pre-multiplied to microseconds
// (counter.rs:76-78 Duration::from_secs; atomic_expiring_value.rs:88 as_micros).
struct LimitDev {
    u64 max_value;
    u64 window_us;
};

// Batch status word block, zeroed before each batch.
struct Status {
    u32 err;        // bitmask of ERRBIT_*
    u32 n_ord;      // hits appended to the ordered list
    u32 n_inserted; // cells created by this batch
    u32 n_removed;  // cells tombstoned by a scan
    u32 n_out;      // rows appended by a scan
    u32 n_rounds;   // reserved
    u32 pad[10];
};
constexpr u32 ERRBIT_BAD_LIMIT = 1u;
constexpr u32 ERRBIT_MISSING_SIMPLE = 2u;
constexpr u32 ERRBIT_TABLE_FULL = 4u;
constexpr u32 ERRBIT_KEY_LIMIT = 8u;
constexpr u32 ERRBIT_RESERVED_KEY = 16u;
constexpr u32 ERRBIT_BIG_DELTA = 32u;  // not an error: the batch must take the exact general path

// One 64-bit atomic per (tile, cell) carries both the hit count (leader election) and the delta
// sum: count in the top 24 bits, sum in the low 40.  Exact as long as the deltas of a whole batch
// cannot carry out of 40 bits; k_probe flags any delta >= 2^40 / n_hits and the host then reruns
// the batch through the general path, which does not use the sum.
constexpr u32 PEND_SHIFT = 40;
constexpr u64 PEND_SUM_MASK = (1ull << PEND_SHIFT) - 1ull;
constexpr u32 MAX_BATCH_HITS = (1u << 24) - 1u;

constexpr u32 SLOT_INVALID = 0x7FFFFFFFu;
constexpr u32 SLOT_MASK = 0x7FFFFFFFu;
constexpr u32 LEADER_BIT = 0x80000000u;

// Zero the scratch half of a cell (bytes 32..63).
__device__ __forceinline__ void cell_clear_scratch(Cell* c) {
    uint4* p = reinterpret_cast<uint4*>(&c->pend);
    p[0] = make_uint4(0, 0, 0, 0);
    p[1] = make_uint4(0, 0, 0, 0);
}

__host__ __device__ inline u64 fmix64(u64 x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}
// Slot from the TOP bits of the hash, owner shard from the LOW 32 bits: independent.
__host__ __device__ inline u32 slot_of(u64 key, u64 seed, u32 log2cap) {
    return (u32)(fmix64(key ^ seed) >> (64 - log2cap));
}
__host__ __device__ inline u32 owner_of(u64 key, u64 seed, u32 world) {
    return (u32)(((fmix64(key ^ seed) & 0xFFFFFFFFull) * (u64)world) >> 32);
}

}  // namespace rl
