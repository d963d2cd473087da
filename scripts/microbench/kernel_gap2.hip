// Microbenchmark (prepared in round 4, not yet run): what the 9.4-9.9 us between two replay launches of the pipeline consist
// of.  kernel_gap.hip showed 3.4-4.7 us between two dependent kernels of one stream when the first leaves >= 16 MB of plain
// stores behind; the pipeline's replay stream carries more than two kernels back to back: a cross-stream wait on the partition's
// event in front of every replay, a stop event on every replay, start + stop events on every fourth, and another kernel (the
// partition) running on a second stream across the boundary.  Each row adds ONE of those to the plain pair:
//   pair              A ; B
//   stop_event        A launched with a stop event (hipExtLaunchKernelGGL) ; B
//   start_stop        A with start and stop events ; B
//   wait_done_event   A ; hipStreamWaitEvent(event of a kernel that ended long ago) ; B
//   wait_live_event   A ; hipStreamWaitEvent(event of a kernel on another stream that ends while A runs) ; B
//   beside_other      the plain pair while a kernel on another stream keeps one 512-thread workgroup per CU busy across the boundary
//   all               stop event on A + wait on a live event + the other stream's kernel (what a replay launch sees)
// A writes 64 MB (coalesced 16-byte stores, ~40 us) so that B is queued long before A ends; the gap is measured on the device:
// wall clock (100 MHz) at A's last workgroup end -> at B's first workgroup start.
// build: hipcc -O3 --offload-arch=gfx950 kernel_gap2.hip -o bin/kernel_gap2
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(r)); exit(1);} } while (0)
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_a(u32x4* __restrict__ buf, u64 n16, u32 passes, u64* stamps) {
    const u32x4 v = {blockIdx.x, threadIdx.x, 3u, 4u};
    for (u32 p = 0; p < passes; ++p)
        for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n16; i += (u64)gridDim.x * 256) buf[i] = v;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&stamps[1], wall_clock64());
}
__global__ __launch_bounds__(256) void k_b(u64* stamps, const u32* probe) {
    if (threadIdx.x == 0) atomicMin(&stamps[2], wall_clock64() + (probe[0] == 0xFFFFFFFFu ? 1 : 0));
}
// the "partition": one 512-thread workgroup per CU spinning on the wall clock for `us` microseconds
__global__ __launch_bounds__(512) void k_other(u64* sink, u32 us) {
    const u64 t0 = wall_clock64();
    u64 t = t0;
    while (t - t0 < (u64)us * 100ull) t = wall_clock64();
    if (t == 1) *sink = t;
}
__global__ void k_short(u64* sink) {
    if (wall_clock64() == 1) *sink = 1;
}

enum { PAIR = 0, STOP_EVENT, START_STOP, WAIT_DONE, WAIT_LIVE, BESIDE, ALL, N_MODES };
static const char* kName[N_MODES] = {"pair", "stop_event", "start_stop", "wait_done_event", "wait_live_event", "beside_other", "all"};

int main() {
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    u32x4* buf;
    const u64 bytes = 64ull << 20;
    CK(hipMalloc(&buf, bytes));
    CK(hipMemset(buf, 0, bytes));
    u64 *d_st, *d_sink;
    CK(hipMalloc(&d_st, 64));
    CK(hipMalloc(&d_sink, 64));
    hipEvent_t e0, e1, e_old, e_live;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&e_old, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e_live, hipEventDisableTiming));
    hipLaunchKernelGGL(k_short, dim3(1), dim3(64), 0, s2, d_sink);
    CK(hipEventRecord(e_old, s2));
    CK(hipDeviceSynchronize());
    for (int mode = 0; mode < N_MODES; ++mode) {
        double gap = 0;
        const int reps = 12;
        for (int r = 0; r < reps + 3; ++r) {
            const u64 init[4] = {0, 0, ~0ull, 0};
            CK(hipMemcpy(d_st, init, sizeof(init), hipMemcpyHostToDevice));
            CK(hipDeviceSynchronize());
            // keep the clocks up: a throw-away A in front
            hipLaunchKernelGGL(k_a, dim3(1280), dim3(256), 0, s, buf, bytes / 16, 1u, d_st + 4);
            const bool other = mode == BESIDE || mode == ALL, live = mode == WAIT_LIVE || mode == ALL;
            if (live) {  // a kernel on the other stream that ends while A runs
                hipLaunchKernelGGL(k_other, dim3(245), dim3(512), 0, s2, d_sink, 30u);
                CK(hipEventRecord(e_live, s2));
            }
            if (mode == STOP_EVENT || mode == ALL)
                hipExtLaunchKernelGGL(k_a, dim3(1280), dim3(256), 0, s, nullptr, e1, 0, buf, bytes / 16, 1u, d_st);
            else if (mode == START_STOP)
                hipExtLaunchKernelGGL(k_a, dim3(1280), dim3(256), 0, s, e0, e1, 0, buf, bytes / 16, 1u, d_st);
            else
                hipLaunchKernelGGL(k_a, dim3(1280), dim3(256), 0, s, buf, bytes / 16, 1u, d_st);
            if (other) hipLaunchKernelGGL(k_other, dim3(245), dim3(512), 0, s2, d_sink, 90u);  // (spans A's end)
            if (mode == WAIT_DONE) CK(hipStreamWaitEvent(s, e_old, 0));
            if (live) CK(hipStreamWaitEvent(s, e_live, 0));
            hipLaunchKernelGGL(k_b, dim3(1280), dim3(256), 0, s, d_st, reinterpret_cast<const u32*>(buf));
            CK(hipDeviceSynchronize());
            u64 st[4];
            CK(hipMemcpy(st, d_st, sizeof(st), hipMemcpyDeviceToHost));
            if (r >= 3) gap += (double)(st[2] - st[1]) / 100.0;
        }
        printf("  %-16s gap between A's last workgroup and B's first: %6.2f us\n", kName[mode], gap / reps);
        fflush(stdout);
    }
    return 0;
}
