// rl_cell.hpp — HBM data layout of the counter table and the device-side helpers every
// kernel shares.  gfx950 only.
//
// One counter cell is 32 bytes, two per 64-byte line: the key, the 16-byte AtomicExpiringValue of the
// reference (limitador/src/storage/atomic_expiring_value.rs:5-9) and the limit id.  Every access on the hot
// path is a random access to a single cell, read as two 16-byte loads of one 32-byte sector; the streaming
// operations (sweep, get_counters, delete, dump, rehash) move 32 B per slot.  No per-batch scratch lives in
// the table: the batch pipelines keep theirs in LDS (rl_apply.hpp) or in per-pass arrays (rl_general.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rl {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr u64 TAG_EMPTY = 0xFFFFFFFFFFFFFFFFull;
constexpr u64 TAG_TOMB = 0xFFFFFFFFFFFFFFFEull;
constexpr u32 SIMPLE_FLAG = 0x80000000u;

struct alignas(32) Cell {
    u64 tag;     //  0  key, TAG_EMPTY or TAG_TOMB                       } one dwordx4
    u64 value;   //  8  AtomicExpiringValue.value                        }
    u64 expiry;  // 16  AtomicExpiringValue.expiry (us since epoch)      } one dwordx4
    u32 limit;   // 24  limit id | SIMPLE_FLAG (attribute of the cell)   }
    u32 pad;     // 28  check word of a hashed key (rl_keyhash.h), 0 = none      }
};
static_assert(sizeof(Cell) == 32, "one cell = one 32-byte sector");

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
constexpr u32 ERRBIT_KEY_COLLISION = 32u;  // two counters share a 64-bit key (hashed keys, include/rl_keyhash.h): their check words differ

constexpr u32 MAX_BATCH_HITS = (1u << 24) - 1u;  // hit indices travel in 24 bits (BHit.idx_tag)

constexpr u32 SLOT_INVALID = 0x7FFFFFFFu;

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
