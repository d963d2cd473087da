// Does a device -> host copy slow the kernels of another stream — as the runtime's hipMemcpyAsync (a blit kernel), and as an SDMA copy
// issued through HSA directly (hsa_amd_memory_async_copy)?  A chain of 40 small HBM-bound kernels (each ~15 us) on one stream, timed
// alone, beside a loop of 40 MB hipMemcpyAsync D2H on another stream, and beside the same bytes through HSA.
// usage: sdma_beside [counters (default 8 M: ~240 us per kernel; 262144: ~6 us, the decide phase's kind)] [kernels per chain] [sync every]
// build: hipcc -O2 --offload-arch=gfx950 sdma_beside.hip -o bin/sdma_beside -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#define CK(x)                                                             \
    do {                                                                  \
        hipError_t r_ = (x);                                              \
        if (r_ != hipSuccess) {                                           \
            printf("%s: %s\n", #x, hipGetErrorString(r_));                \
            return 1;                                                     \
        }                                                                 \
    } while (0)

__global__ void k_copy(const uint4* s, uint4* d, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) d[i] = s[i];
}

__global__ void k_touch(unsigned* p, size_t n, unsigned mul) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[(i * 2654435761ull) % n] += mul;  // scattered read-modify-write
}

static hsa_agent_t g_gpu{}, g_cpu{};
static hsa_status_t pick(hsa_agent_t a, void*) {
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_gpu.handle) g_gpu = a;
    if (t == HSA_DEVICE_TYPE_CPU && !g_cpu.handle) g_cpu = a;
    return HSA_STATUS_SUCCESS;
}

int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : (8u << 20);  // 32 MB of counters
    const int n_k = argc > 2 ? atoi(argv[2]) : 40, sync_every = argc > 3 ? atoi(argv[3]) : 8;
    const unsigned grid = (unsigned)std::min<size_t>(1024, (n + 255) / 256);
    unsigned* d = nullptr;
    CK(hipMalloc(&d, n * 4));
    CK(hipMemset(d, 0, n * 4));
    const size_t bytes = 40u << 20;
    void *d_src = nullptr, *h_dst = nullptr;
    CK(hipMalloc(&d_src, bytes));
    CK(hipHostMalloc(&h_dst, bytes, hipHostMallocDefault));
    hipStream_t sk, sc;
    CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    if (hsa_init() != HSA_STATUS_SUCCESS) return printf("hsa_init failed\n"), 1;
    hsa_iterate_agents(pick, nullptr);
    hsa_signal_t sig;
    hsa_signal_create(1, 0, nullptr, &sig);
    auto chain = [&]() -> double {  // ms for 40 kernels, each waited for by the host like a decide phase's round trips (every 8th)
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < n_k; ++k) {
            k_touch<<<grid, 256, 0, sk>>>(d, n, (unsigned)k);
            if (k % sync_every == sync_every - 1) (void)hipStreamSynchronize(sk);
        }
        (void)hipStreamSynchronize(sk);
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    for (int w = 0; w < 3; ++w) chain();
    double alone = 1e9;
    for (int r = 0; r < 10; ++r) alone = std::min(alone, chain());
    printf("%d kernels over %zu counters, a synchronise every %d, alone: %.3f ms\n", n_k, n, sync_every, alone);
    for (int mode = 0; mode < 5; ++mode) {
        std::atomic<bool> stop{false};
        std::atomic<int> copies{0};
        std::thread pump([&] {
            (void)hipSetDevice(0);
            while (!stop.load()) {
                if (mode == 0) {
                    (void)hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, sc);
                    (void)hipStreamSynchronize(sc);
                } else if (mode == 2) {
                    k_copy<<<32, 256, 0, sc>>>(static_cast<const uint4*>(d_src), static_cast<uint4*>(h_dst), bytes / 16);
                    (void)hipStreamSynchronize(sc);
                } else if (mode == 3) {  // the other direction as a kernel: loads from host memory, stores to the device
                    k_copy<<<32, 256, 0, sc>>>(static_cast<const uint4*>(h_dst), static_cast<uint4*>(d_src), bytes / 16);
                    (void)hipStreamSynchronize(sc);
                } else if (mode == 4) {  // ... and as a copy command
                    (void)hipMemcpyAsync(d_src, h_dst, bytes, hipMemcpyHostToDevice, sc);
                    (void)hipStreamSynchronize(sc);
                } else {
                    hsa_signal_store_relaxed(sig, 1);
                    if (hsa_amd_memory_async_copy(h_dst, g_cpu, d_src, g_gpu, bytes, 0, nullptr, sig) != HSA_STATUS_SUCCESS) {
                        printf("hsa_amd_memory_async_copy failed\n");
                        return;
                    }
                    while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, 1000000, HSA_WAIT_STATE_BLOCKED) != 0) {
                    }
                }
                copies.fetch_add(1);
            }
        });
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        double best = 1e9, sum = 0;
        const int c0 = copies.load();
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 20; ++r) {
            const double t = chain();
            best = std::min(best, t);
            sum += t;
        }
        const double el = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const int nc = copies.load() - c0;
        stop.store(true);
        pump.join();
        printf("beside %s: best %.3f ms, mean %.3f ms; %d copies of 40 MB in %.1f ms = %.1f GB/s\n",
               mode == 0 ? "hipMemcpyAsync device -> host        " : mode == 1 ? "hsa_amd_memory_async_copy (SDMA)" : mode == 2 ? "a 32-workgroup kernel storing to host" : mode == 3 ? "a 32-workgroup kernel loading from host" : "hipMemcpyAsync host -> device", best, sum / 20, nc, el, nc * 41.943 / el);
    }
    return 0;
}
