// Microbenchmark: the chip's random-transaction RATE as a SLOPE, not a one-launch intercept.
// One launch of N random accesses to a table of 32-byte cells (the engine's cell, rl_cell.hpp) for N = 1 M, 4 M, 16 M,
// 64 M; timed with the launch's own start / stop events (hipExtLaunchKernelGGL — two hipEventRecord markers around a
// launch add 4-6 us).  Reported: us per launch for every N and the slope between consecutive N (ns per access, and
// G accesses/s) — the intercept (launch, ramp, one dependent trip) drops out of the slope.
// Modes (what the replay of check_and_update does to a cell, reference in_memory.rs:72-156 / atomic_expiring_value.rs:36-42):
//   read32            the cell as two dwordx4
//   read32+store8     + the 8-byte value write-back
//   read32_sc1        the same reads as agent-scope relaxed atomic loads (what a hand-over between two kernels that
//                     overlap on different XCDs would need: the L2s only meet in memory)
//   read32+store8_sc1 reads and the write-back both agent-scope
//   atomic_add_noret  one agent-scope 64-bit atomicAdd without return per access
//   atomic_add_ret    ... with return
//   read16            only the first dwordx4 of the cell (tag + value)
//   read32_pair       the cell by a PAIR of lanes, 16 bytes each, in ONE instruction (lanes 2k, 2k+1 = cell k of the instruction's 32
//                     cells): the two halves of a cell are never two requests to one line in flight (TCP_PENDING_STALL_CYCLES is
//                     half of k_bkt_step's TCP time, profiles/r04h_tcp_counters.md) — same number of load instructions per cell
//   read32_pair+exchange  ... and every lane gets both halves of ITS cell back through ds_bpermute (what a replay round would need)
//   read32 (2nd half late) the second dwordx4 only after the first has returned
// Tables: 2^25 cells (1.07 GB: the bench's table, far beyond the 256 MiB Infinity Cache) and 2^21 cells (67 MB: inside it).
// build: hipcc -O3 --offload-arch=gfx950 random_slope.hip -o bin/random_slope
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;

#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(r)); exit(1);} } while (0)

__host__ __device__ inline u64 fmix64(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x;
}

enum { M_READ32 = 0, M_READ32_ST8, M_READ32_SC1, M_READ32_ST8_SC1, M_ATOMIC_NORET, M_ATOMIC_RET, M_READ16, M_READ32_PAIR, M_READ32_PAIR_X, M_READ32_LATE, M_COUNT };
static const char* kModeName[M_COUNT] = {"read32", "read32+store8", "read32_sc1", "read32+store8_sc1", "atomic_add_noret",
                                         "atomic_add_ret", "read16", "read32_pair", "read32_pair+exchange", "read32 (2nd half late)"};

template <int MODE, int HPT>
__global__ __launch_bounds__(256) void k_access(char* __restrict__ table, u32 log2cells, u64 salt, u64 n, u64* sink) {
    const u64 base = (u64)blockIdx.x * 256 * HPT;
    u64 acc = 0;
    char* p[HPT];
    bool ok[HPT];
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
        const u64 i = base + (u64)u * 256 + threadIdx.x;
        ok[u] = i < n;
        p[u] = table + (fmix64(i + salt) >> (64 - log2cells)) * 32;
    }
    if (MODE == M_READ32 || MODE == M_READ32_ST8) {
        uint4 a[HPT], b[HPT];
#pragma unroll
        for (int u = 0; u < HPT; ++u) {  // (unconditional: every lane has a valid address)
            a[u] = reinterpret_cast<const uint4*>(p[u])[0];
            b[u] = reinterpret_cast<const uint4*>(p[u])[1];
        }
#pragma unroll
        for (int u = 0; u < HPT; ++u) acc += a[u].x + a[u].z + b[u].y;
        if (MODE == M_READ32_ST8) {
#pragma unroll
            for (int u = 0; u < HPT; ++u)
                if (ok[u]) *reinterpret_cast<u64*>(p[u] + 8) = acc + u;
        }
    } else if (MODE == M_READ16) {
        uint4 a[HPT];
#pragma unroll
        for (int u = 0; u < HPT; ++u) a[u] = reinterpret_cast<const uint4*>(p[u])[0];
#pragma unroll
        for (int u = 0; u < HPT; ++u) acc += a[u].x + a[u].z;
    } else if (MODE == M_READ32_LATE) {
        uint4 a[HPT], b[HPT];
#pragma unroll
        for (int u = 0; u < HPT; ++u) a[u] = reinterpret_cast<const uint4*>(p[u])[0];
#pragma unroll
        for (int u = 0; u < HPT; ++u) acc += a[u].x + a[u].z;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < HPT; ++u) b[u] = reinterpret_cast<const uint4*>(p[u] + (acc == 77 ? 32 : 0))[1];
#pragma unroll
        for (int u = 0; u < HPT; ++u) acc += b[u].y;
    } else if (MODE == M_READ32_PAIR || MODE == M_READ32_PAIR_X) {
        const u32 lane = threadIdx.x & 63u;
        uint4 q[HPT][2];
#pragma unroll
        for (int u = 0; u < HPT; ++u)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u32 src = (lane >> 1) + 32u * j;  // the lane whose cell this pair of lanes reads
                const u64 pv = (u64)p[u];
                const u64 ps = ((u64)(u32)__shfl((int)(pv >> 32), (int)src) << 32) | (u32)__shfl((int)(u32)pv, (int)src);
                q[u][j] = *reinterpret_cast<const uint4*>((const char*)ps + (lane & 1u) * 16);
            }
        if (MODE == M_READ32_PAIR) {
#pragma unroll
            for (int u = 0; u < HPT; ++u) acc += q[u][0].x + q[u][0].z + q[u][1].y + q[u][1].w;
        } else {
#pragma unroll
            for (int u = 0; u < HPT; ++u) {
                // lane l's cell sits in instruction j = l >> 5, lanes 2 (l & 31) (first half) and 2 (l & 31) + 1 (second half)
                const int s0 = (int)(2u * (lane & 31u)), s1 = s0 + 1;
                const bool hi = lane >= 32u;
                u32 a[4], b[4];
                const u32* q0 = reinterpret_cast<const u32*>(&q[u][0]);
                const u32* q1 = reinterpret_cast<const u32*>(&q[u][1]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u32 a0 = (u32)__shfl((int)q0[k], s0), a1 = (u32)__shfl((int)q1[k], s0);
                    const u32 b0 = (u32)__shfl((int)q0[k], s1), b1 = (u32)__shfl((int)q1[k], s1);
                    a[k] = hi ? a1 : a0;
                    b[k] = hi ? b1 : b0;
                }
                acc += a[0] + a[2] + b[1];
            }
        }
    } else if (MODE == M_READ32_SC1 || MODE == M_READ32_ST8_SC1) {
        u64 q[HPT][4];
#pragma unroll
        for (int u = 0; u < HPT; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                q[u][k] = __hip_atomic_load(reinterpret_cast<u64*>(p[u]) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < HPT; ++u) acc += q[u][0] + q[u][1] + q[u][2] + q[u][3];
        if (MODE == M_READ32_ST8_SC1) {
#pragma unroll
            for (int u = 0; u < HPT; ++u)
                if (ok[u]) __hip_atomic_store(reinterpret_cast<u64*>(p[u]) + 1, acc + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (MODE == M_ATOMIC_NORET) {
#pragma unroll
        for (int u = 0; u < HPT; ++u)
            if (ok[u]) __hip_atomic_fetch_add(reinterpret_cast<u64*>(p[u]) + 3, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
#pragma unroll
        for (int u = 0; u < HPT; ++u)
            if (ok[u]) acc += __hip_atomic_fetch_add(reinterpret_cast<u64*>(p[u]) + 3, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (acc == 0x1234567ull) *sink = acc;
}

template <int MODE, int HPT>
static float run(char* table, u32 log2cells, u64 n, u64* sink, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const u32 grid = (u32)((n + 256ull * HPT - 1) / (256ull * HPT));
    hipLaunchKernelGGL((k_access<MODE, HPT>), dim3(grid), dim3(256), 0, 0, table, log2cells, 999ull, n, sink);
    CK(hipDeviceSynchronize());
    float tot = 0;
    for (int r = 0; r < reps; ++r) {
        hipExtLaunchKernelGGL((k_access<MODE, HPT>), dim3(grid), dim3(256), 0, 0, a, b, 0, table, log2cells,
                              1000003ull * (r + 1), n, sink);
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        tot += ms;
    }
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return tot / reps * 1e3f;  // us
}

template <int MODE>
static void sweep(char* table, u32 log2cells, u64* sink) {
    const u64 ns[4] = {1ull << 20, 4ull << 20, 16ull << 20, 64ull << 20};
    float t[4];
    for (int k = 0; k < 4; ++k) t[k] = run<MODE, 4>(table, log2cells, ns[k], sink, k < 2 ? 10 : 4);
    printf("  %-20s", kModeName[MODE]);
    for (int k = 0; k < 4; ++k) printf("  %3lluM %8.1f us", (unsigned long long)(ns[k] >> 20), t[k]);
    printf("   slope ns/access:");
    for (int k = 1; k < 4; ++k) printf(" %.4f", (t[k] - t[k - 1]) * 1e3 / (double)(ns[k] - ns[k - 1]));
    const double s = (t[3] - t[1]) * 1e3 / (double)(ns[3] - ns[1]);
    printf("   => %.1f G accesses/s (4M..64M), 1 M accesses = %.1f us at that rate\n", 1.0 / s, s * (1 << 20) / 1e3);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const bool only_reads = argc > 1;
    u64* sink;
    CK(hipMalloc(&sink, 4096));
    const u32 cfgs[2] = {25, 21};
    for (u32 log2cells : cfgs) {
        const size_t bytes = ((size_t)1 << log2cells) * 32;
        char* table;
        CK(hipMalloc(&table, bytes));
        CK(hipMemset(table, 0, bytes));
        printf("== table 2^%u cells x 32 B = %.0f MB; 4 independent accesses per thread, 256-thread workgroups\n", log2cells,
               bytes / 1e6);
        sweep<M_READ32>(table, log2cells, sink);
        if (!only_reads) {
        sweep<M_READ32_ST8>(table, log2cells, sink);
        sweep<M_READ32_SC1>(table, log2cells, sink);
        sweep<M_READ32_ST8_SC1>(table, log2cells, sink);
        sweep<M_ATOMIC_NORET>(table, log2cells, sink);
        sweep<M_ATOMIC_RET>(table, log2cells, sink);
        }
        sweep<M_READ16>(table, log2cells, sink);
        sweep<M_READ32_PAIR>(table, log2cells, sink);
        sweep<M_READ32_PAIR_X>(table, log2cells, sink);
        sweep<M_READ32_LATE>(table, log2cells, sink);
        CK(hipFree(table));
    }
    return 0;
}
