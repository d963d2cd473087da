// host vs device rl_kh_bytes on a few strings (debugging aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../../include/rl_keyhash.h"
__global__ void k(const uint8_t* s, const uint32_t* off, int n, rl_h128* out) {
    int i = threadIdx.x;
    if (i < n) out[i] = rl_kh_bytes(s + off[i], off[i + 1] - off[i], 0ull);
}
int main() {
    const char* strs[] = {"PUT", "/admin", "", "a-user-name-of-more-than-sixteen-bytes", "GET", "0123456789abcdef", "0123456789abcdefg"};
    const int n = 7;
    uint8_t buf[256]; uint32_t off[8]; uint32_t p = 0;
    for (int i = 0; i < n; ++i) { off[i] = p; memcpy(buf + p, strs[i], strlen(strs[i])); p += strlen(strs[i]); }
    off[n] = p;
    uint8_t* d; uint32_t* doff; rl_h128* dout; rl_h128 out[8];
    hipMalloc(&d, 256); hipMalloc(&doff, 64); hipMalloc(&dout, sizeof(out));
    hipMemcpy(d, buf, 256, hipMemcpyHostToDevice); hipMemcpy(doff, off, sizeof(off), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, doff, n, dout);
    hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        rl_h128 h = rl_kh_bytes((const uint8_t*)strs[i], strlen(strs[i]), 0ull);
        printf("%-40s host %016llx %016llx dev %016llx %016llx %s\n", strs[i], (unsigned long long)h.h1, (unsigned long long)h.h2,
               (unsigned long long)out[i].h1, (unsigned long long)out[i].h2, (h.h1 == out[i].h1 && h.h2 == out[i].h2) ? "ok" : "DIFF");
        bad += !(h.h1 == out[i].h1 && h.h2 == out[i].h2);
    }
    return bad;
}
