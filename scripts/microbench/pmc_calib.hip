// PMC calibration: kernels with KNOWN HBM byte counts, in the access patterns the engine uses, so
// that rocprofv3's FETCH_SIZE / WRITE_SIZE (uncalibrated on gfx950 for anything but wide streaming
// reads, MI355X_MICROARCH.md §HBM) can be turned into bytes for k_bkt_apply and friends.
//   k_calib_stream_read    1 GiB, 16 B per lane, coalesced          -> 1 GiB fetched
//   k_calib_stream_write   1 GiB, 16 B per lane, coalesced          -> 1 GiB written
//   k_calib_random_read32  2^20 random 64-byte cells of a 4 GiB table, 32 B read of each
//                          (tag+value, expiry+limit: what k_bkt_apply reads)  -> 64 MiB fetched
//   k_calib_random_store8  2^20 random 8-byte stores into the same table    -> 8 MiB useful,
//                          64 MiB if the memory side writes whole 64-byte lines
// build: hipcc -O3 --offload-arch=gfx950 pmc_calib.hip -o bin/pmc_calib ; run under rocprofv3 --pmc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(r)); exit(1);} } while (0)
__device__ inline u64 fmix64(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x;
}
__global__ __launch_bounds__(256) void k_calib_stream_read(const uint4* p, size_t n16, u64* sink) {
    u64 acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc += v.x ^ v.w;
    }
    if (acc == 0x1234567) *sink = acc;
}
__global__ __launch_bounds__(256) void k_calib_stream_write(uint4* p, size_t n16, u32 salt) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        p[i] = make_uint4(salt, (u32)i, salt, (u32)i);
}
__global__ __launch_bounds__(256) void k_calib_random_read32(const char* table, u32 log2cells, u64 salt, u32 n, u64* sink) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const char* c = table + (fmix64(i + salt) >> (64 - log2cells)) * 64;
    const uint4 a = *(const uint4*)c;
    const uint4 b = *(const uint4*)(c + 16);
    if ((a.x ^ b.x ^ b.z) == 0x1234567) *sink = b.x;
}
__global__ __launch_bounds__(256) void k_calib_random_store8(char* table, u32 log2cells, u64 salt, u32 n) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    *(u64*)(table + (fmix64(i + salt) >> (64 - log2cells)) * 64 + 8) = salt + i;
}
int main() {
    const u32 log2cells = 26;  // 4 GiB of 64-byte cells: far beyond the 256 MiB Infinity Cache
    const size_t bytes = (size_t)64 << log2cells;
    char* table; CK(hipMalloc(&table, bytes)); CK(hipMemset(table, 0, bytes));
    u64* sink; CK(hipMalloc(&sink, 8));
    const size_t n16 = ((size_t)1 << 30) / 16;
    const u32 n = 1u << 20;
    for (int r = 0; r < 5; ++r) {
        k_calib_stream_read<<<4096, 256>>>((const uint4*)table + (size_t)r * n16 % (bytes / 16 - n16), n16, sink);
        k_calib_stream_write<<<4096, 256>>>((uint4*)table + (size_t)(r + 1) * n16 % (bytes / 16 - n16), n16, r);
        k_calib_random_read32<<<n / 256, 256>>>(table, log2cells, 7777ull * (r + 1), n, sink);
        k_calib_random_store8<<<n / 256, 256>>>(table, log2cells, 9999ull * (r + 1), n);
        CK(hipDeviceSynchronize());
    }
    printf("pmc_calib done: stream 1 GiB r/w, 2^20 random 64-B-cell reads (32 B each) and 8-B stores, 5 rounds\n");
    return 0;
}
