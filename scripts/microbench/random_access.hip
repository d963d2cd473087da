// Microbenchmark: what does one random 64-byte-cell access cost on MI355X, and what do the
// global atomics of k_probe add?  Informs the table layout / kernel structure.
// build: hipcc -O3 --offload-arch=gfx950 random_access.hip -o random_access
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;

#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(r)); exit(1);} } while (0)

__host__ __device__ inline u64 fmix64(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x;
}

// mode bits: 1 = read tag (8 B), 2 = returning u32 atomic, 4 = non-returning u64 atomic, 8 = plain 8-B store,
//            16 = read 48 B (3 x 16 B), 32 = zipf-ish skew (hot lines)
template <int HPT>
__global__ __launch_bounds__(256) void k_access(char* table, u32 log2cells, u32 cell_bytes, u64 salt, u32 n,
                                                int mode, u64* sink) {
    const u32 base = blockIdx.x * 256 * HPT;
    u64 acc = 0;
    u64 addr[HPT];
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
        const u32 i = base + u * 256 + threadIdx.x;
        u64 h = fmix64(i + salt);
        if (mode & 32) {  // skew: 30% of accesses go to 64 hot cells
            if ((h & 0xFF) < 77) h = fmix64((h >> 8) & 63);
        }
        addr[u] = (u64)(h >> (64 - log2cells)) * cell_bytes;
    }
    if (mode & 1) {
#pragma unroll
        for (int u = 0; u < HPT; ++u) if (base + u * 256 + threadIdx.x < n) acc += *(const u64*)(table + addr[u]);
    }
    if (mode & 16) {
#pragma unroll
        for (int u = 0; u < HPT; ++u) if (base + u * 256 + threadIdx.x < n) {
            const uint4* p = (const uint4*)(table + addr[u]);
            uint4 a = p[0], b = p[1];
            acc += a.x + b.y;
            if (cell_bytes >= 64) { uint4 c = p[2]; acc += c.z; }
        }
    }
    if (mode & 2) {
#pragma unroll
        for (int u = 0; u < HPT; ++u) if (base + u * 256 + threadIdx.x < n) acc += atomicAdd((u32*)(table + addr[u] + (cell_bytes >= 64 ? 32 : 16)), 1u);
    }
    if (mode & 4) {
#pragma unroll
        for (int u = 0; u < HPT; ++u) if (base + u * 256 + threadIdx.x < n) atomicAdd((u64*)(table + addr[u] + 24), 1ull);
    }
    if (mode & 8) {
#pragma unroll
        for (int u = 0; u < HPT; ++u) if (base + u * 256 + threadIdx.x < n) *(u64*)(table + addr[u] + 8) = acc + u;
    }
    if (mode & 64) {  // 16-byte store
#pragma unroll
        for (int u = 0; u < HPT; ++u) if (base + u * 256 + threadIdx.x < n) *(uint4*)(table + addr[u]) = make_uint4((u32)acc, u, 3, 4);
    }
    if (mode & 128) {  // the whole 32-byte sector (two 16-byte stores)
#pragma unroll
        for (int u = 0; u < HPT; ++u) if (base + u * 256 + threadIdx.x < n) {
            uint4* p = (uint4*)(table + addr[u]);
            p[0] = make_uint4((u32)acc, u, 3, 4);
            p[1] = make_uint4((u32)acc, u, 5, 6);
        }
    }
    if (mode & 256) {  // read 32 B (2 x 16 B)
#pragma unroll
        for (int u = 0; u < HPT; ++u) if (base + u * 256 + threadIdx.x < n) {
            const uint4* p = (const uint4*)(table + addr[u]);
            uint4 a = p[0], b = p[1];
            acc += a.x + b.y;
        }
    }
    if (mode & 512) {  // 1-byte result scatter into a 1 MB array (the verdict pattern), index = hash
#pragma unroll
        for (int u = 0; u < HPT; ++u) if (base + u * 256 + threadIdx.x < n) ((unsigned char*)sink)[64 + (fmix64(base + u * 256 + threadIdx.x + salt + 7) & 0xFFFFF)] = (unsigned char)acc;
    }
    if (acc == 0x1234567) *sink = acc;
}

template <int HPT>
float run(char* table, u32 log2cells, u32 cell_bytes, u32 n, int mode, u64* sink, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const u32 grid = (n + 256 * HPT - 1) / (256 * HPT);
    k_access<HPT><<<grid, 256>>>(table, log2cells, cell_bytes, 999, n, mode, sink);
    CK(hipDeviceSynchronize());
    float best = 1e9, tot = 0;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        k_access<HPT><<<grid, 256>>>(table, log2cells, cell_bytes, 1000003ull * (r + 1), n, mode, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        tot += ms; if (ms < best) best = ms;
    }
    return tot / reps;
}

int main(int argc, char** argv) {
    const u32 n = 1u << 20;
    u64* sink; CK(hipMalloc(&sink, (1 << 20) + 128));
    struct Cfg { u32 log2cells; u32 cell_bytes; };
    Cfg cfgs[] = {{25, 64}, {26, 32}, {25, 32}};
    struct Mode { int m; const char* name; };
    Mode modes[] = {{1, "read8"}, {16, "read48"}, {1 | 8, "read8+store8"},
                    {256, "read32"}, {256 | 8, "read32+store8"}, {256 | 64, "read32+store16"}, {256 | 128, "read32+store32"},
                    {512, "byte scatter only (1 MB array)"}, {256 | 8 | 512, "read32+store8+byte scatter"}};
    for (auto c : cfgs) {
        const size_t bytes = ((size_t)1 << c.log2cells) * c.cell_bytes;
        char* table; CK(hipMalloc(&table, bytes)); CK(hipMemset(table, 0, bytes));
        printf("== table 2^%u cells x %u B = %.0f MB, %u random accesses per launch\n", c.log2cells, c.cell_bytes, bytes / 1e6, n);
        for (auto m : modes) {
            float t1 = run<1>(table, c.log2cells, c.cell_bytes, n, m.m, sink, 10);
            float t4 = run<4>(table, c.log2cells, c.cell_bytes, n, m.m, sink, 10);
            float t8 = run<8>(table, c.log2cells, c.cell_bytes, n, m.m, sink, 10);
            printf("  %-44s  HPT1 %7.1f us   HPT4 %7.1f us   HPT8 %7.1f us   (%.1f G acc/s best)\n", m.name, t1 * 1e3, t4 * 1e3, t8 * 1e3,
                   n / (fminf(t1, fminf(t4, t8)) * 1e-3) / 1e9);
            fflush(stdout);
        }
        CK(hipFree(table));
    }
    return 0;
}
