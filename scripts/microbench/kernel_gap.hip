// Microbenchmark: what lies between two dependent kernels of one stream — does the gap grow with the bytes the first one
// leaves dirty in the XCDs' L2s (the write-back at the end of a kernel), and do store flavours that write through shrink it?
// k_write stores N bytes (coalesced dwordx4, one flavour) and records the wall clock (100 MHz) at which its last workgroup
// ends; k_next records the wall clock at which it starts.  Reported: the gap (us) and k_write's own duration.
// build: hipcc -O3 --offload-arch=gfx950 kernel_gap.hip -o bin/kernel_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(r)); exit(1);} } while (0)
enum { F_PLAIN = 0, F_NT, F_SC1, F_SC0SC1, F_COUNT };
static const char* kName[F_COUNT] = {"plain", "nt", "sc1", "sc0 sc1"};
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

template <int F>
__global__ __launch_bounds__(256) void k_write(u32x4* __restrict__ buf, u64 n16, u64* stamps) {
    const u32x4 v = {blockIdx.x, threadIdx.x, 3u, 4u};
    const u64 t0 = wall_clock64();
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n16; i += (u64)gridDim.x * 256) {
        if (F == F_PLAIN) buf[i] = v;
        else if (F == F_NT) __builtin_nontemporal_store(v, &buf[i]);
        else if (F == F_SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&buf[i]), "v"(v) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(&buf[i]), "v"(v) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&stamps[0], t0);
        atomicMax(&stamps[1], wall_clock64());
    }
}
__global__ void k_next(u64* stamps, const u32* probe) {
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2] = wall_clock64() + (probe[0] == 0xFFFFFFFFu ? 1 : 0);
}

template <int F>
static void run(u32x4* buf, u64 bytes, u64* d_st, hipStream_t s) {
    double gap = 0, dur = 0;
    const int reps = 12;
    for (int r = 0; r < reps + 2; ++r) {
        const u64 init[4] = {~0ull, 0, 0, 0};
        CK(hipMemcpy(d_st, init, sizeof(init), hipMemcpyHostToDevice));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL((k_write<F>), dim3(1280), dim3(256), 0, s, buf, bytes / 16, d_st);
        hipLaunchKernelGGL(k_next, dim3(1280), dim3(256), 0, s, d_st, reinterpret_cast<const u32*>(buf));
        CK(hipStreamSynchronize(s));
        u64 st[4];
        CK(hipMemcpy(st, d_st, sizeof(st), hipMemcpyDeviceToHost));
        if (r >= 2) { gap += (double)(st[2] - st[1]) / 100.0; dur += (double)(st[1] - st[0]) / 100.0; }
    }
    printf("  %-8s %6.1f MB written: kernel %7.1f us, gap to the next kernel's first wave %6.2f us\n", kName[F], bytes / 1e6, dur / reps, gap / reps);
    fflush(stdout);
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    u32x4* buf;
    CK(hipMalloc(&buf, 256ull << 20));
    CK(hipMemset(buf, 0, 256ull << 20));
    u64* d_st;
    CK(hipMalloc(&d_st, 64));
    for (u64 mb : {0ull, 1ull, 4ull, 16ull, 35ull, 64ull, 128ull}) {
        run<F_PLAIN>(buf, mb << 20, d_st, s);
        if (mb == 0) continue;
        run<F_NT>(buf, mb << 20, d_st, s);
        run<F_SC1>(buf, mb << 20, d_st, s);
        run<F_SC0SC1>(buf, mb << 20, d_st, s);
    }
    return 0;
}
