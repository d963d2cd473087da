// Microbenchmark: the LATENCY of a dependent random cell read at the replay's occupancy — 1280 workgroups of 256 threads (five per
// CU, 21.5 KB of LDS each, like k_bkt_step), every thread a chain of `iters` reads whose address depends on the data of the one
// before (what a replay round is: record -> cell -> decide -> next round).  Reported: us per link of the chain (slope between two
// chain lengths), for the forms a 32-byte cell can be read in:
//   read32       two dwordx4 per lane, back to back (what apply2_round does: ca, cb)
//   read16       the first dwordx4 only
//   pair         lanes 2k, 2k+1 read the two halves of cell k: one instruction per 32 cells, never two requests of one wave to a
//                line in flight
//   pair+xchg    ... and every lane gets both halves of its own cell back (8 ds_bpermute per dword pair)
//   read32+touch the two dwordx4 plus a 4-byte read of ANOTHER random cell (the read-ahead touch of profiles/r04h)
// Tables: 2^25 cells (1.07 GB, the bench's) and 2^17 cells (4 MB: L2-resident) — HBM latency against L2 latency.
// build: hipcc -O3 --offload-arch=gfx950 random_chain.hip -o bin/random_chain
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(r)); exit(1);} } while (0)
__device__ inline u64 fmix64(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
enum { C_READ32 = 0, C_READ16, C_PAIR, C_PAIR_X, C_TOUCH, C_COUNT };
static const char* kName[C_COUNT] = {"read32", "read16", "pair", "pair+xchg", "read32+touch"};

template <int MODE>
__global__ __launch_bounds__(256) void k_chain(const char* __restrict__ table, u32 log2cells, u64 salt, u32 iters, u64* sink) {
    extern __shared__ unsigned char s_dyn[];
    const u32 lane = threadIdx.x & 63u;
    u64 x = ((u64)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + salt;
    u64 acc = 0;
    for (u32 it = 0; it < iters; ++it) {
        const char* p = table + (fmix64(x) >> (64 - log2cells)) * 32;
        u32 v;
        if (MODE == C_READ32 || MODE == C_TOUCH) {
            const uint4 a = reinterpret_cast<const uint4*>(p)[0];
            const uint4 b = reinterpret_cast<const uint4*>(p)[1];
            v = a.x + a.z + b.y;
            if (MODE == C_TOUCH) {
                const char* p2 = table + (fmix64(x + 12345) >> (64 - log2cells)) * 32;
                acc += *reinterpret_cast<const u32*>(p2);
            }
        } else if (MODE == C_READ16) {
            const uint4 a = reinterpret_cast<const uint4*>(p)[0];
            v = a.x + a.z;
        } else {
            uint4 q[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int src = (int)((lane >> 1) + 32u * j);
                const u64 pv = (u64)p;
                const u64 ps = ((u64)(u32)__shfl((int)(pv >> 32), src) << 32) | (u32)__shfl((int)(u32)pv, src);
                q[j] = *reinterpret_cast<const uint4*>((const char*)ps + (lane & 1u) * 16);
            }
            if (MODE == C_PAIR) {
                // (not a replay: the lane keeps what it loaded; the chain still depends on it)
                v = q[0].x + q[0].z + q[1].y + q[1].w;
            } else {
                const int s0 = (int)(2u * (lane & 31u)), s1 = s0 + 1;
                const bool hi = lane >= 32u;
                const u32* q0 = reinterpret_cast<const u32*>(&q[0]);
                const u32* q1 = reinterpret_cast<const u32*>(&q[1]);
                u32 a[4], b[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u32 a0 = (u32)__shfl((int)q0[k], s0), a1 = (u32)__shfl((int)q1[k], s0);
                    const u32 b0 = (u32)__shfl((int)q0[k], s1), b1 = (u32)__shfl((int)q1[k], s1);
                    a[k] = hi ? a1 : a0;
                    b[k] = hi ? b1 : b0;
                }
                v = a[0] + a[1] + a[2] + a[3] + b[0] + b[1] + b[2];
            }
        }
        x = x * 6364136223846793005ull + v + it + 1;
        acc += v;
    }
    if (acc == 0x1234567ull) { *sink = acc; s_dyn[threadIdx.x] = 1; }
}

template <int MODE>
static float run(const char* table, u32 log2cells, u32 iters, u64* sink, u32 grid) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_chain<MODE>), dim3(grid), dim3(256), 21504, 0, table, log2cells, 7ull, iters, sink);
    CK(hipDeviceSynchronize());
    float tot = 0;
    const int reps = 10;
    for (int r = 0; r < reps; ++r) {
        hipExtLaunchKernelGGL((k_chain<MODE>), dim3(grid), dim3(256), 21504, 0, a, b, 0, table, log2cells, 1000003ull * (r + 1), iters, sink);
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        tot += ms;
    }
    return tot / reps * 1e3f;
}

template <int MODE>
static void line(const char* table, u32 log2cells, u64* sink, u32 grid) {
    const float t8 = run<MODE>(table, log2cells, 8, sink, grid), t24 = run<MODE>(table, log2cells, 24, sink, grid);
    printf("  %-14s 8 links %7.1f us   24 links %7.1f us   => %.2f us per link\n", kName[MODE], t8, t24, (t24 - t8) / 16.0);
    fflush(stdout);
}

int main() {
    u64* sink;
    CK(hipMalloc(&sink, 4096));
    const u32 cfgs[2] = {25, 17};
    for (u32 log2cells : cfgs) {
        const size_t bytes = ((size_t)1 << log2cells) * 32;
        char* table;
        CK(hipMalloc(&table, bytes));
        CK(hipMemset(table, 0, bytes));
        for (u32 grid : {1280u, 640u, 256u}) {
            printf("== table 2^%u cells (%.0f MB), %u workgroups x 256 threads, 21.5 KB LDS each\n", log2cells, bytes / 1e6, grid);
            line<C_READ32>(table, log2cells, sink, grid);
            line<C_READ16>(table, log2cells, sink, grid);
            line<C_PAIR>(table, log2cells, sink, grid);
            line<C_PAIR_X>(table, log2cells, sink, grid);
            line<C_TOUCH>(table, log2cells, sink, grid);
        }
        CK(hipFree(table));
    }
    return 0;
}
