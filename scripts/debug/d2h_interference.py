"""Does a device -> pinned-host copy command slow a memory-bound kernel on another stream?  (r15b / r15e: while a kernel writes host
memory — k_resp<true> or the runtime's blit copy — every kernel of the other queue runs 3-8 x longer.)
Times x.add_(1) on 256 MB alone, beside a 400 MB hipMemcpyAsync D2H, and beside a H2D of the same size."""
import time

import torch

dev = torch.device("cuda", 0)
x = torch.zeros(64 << 20, dtype=torch.float32, device=dev)  # 256 MB: read + write = 512 MB per add_
big = torch.zeros(100 << 20, dtype=torch.float32, device=dev)  # 400 MB
host = torch.empty(100 << 20, dtype=torch.float32).pin_memory()
s_k, s_c = torch.cuda.Stream(), torch.cuda.Stream()


def kernel_times(n=12):
    ts = []
    with torch.cuda.stream(s_k):
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            x.add_(1.0)
            b.record()
            ts.append((a, b))
    s_k.synchronize()
    return [round(a.elapsed_time(b), 3) for a, b in ts]


for _ in range(2):
    kernel_times(3)
print("alone            ", kernel_times())
for name, fn in (("beside D2H copy ", lambda: host.copy_(big, non_blocking=True)), ("beside H2D copy ", lambda: big.copy_(host, non_blocking=True))):
    with torch.cuda.stream(s_c):
        t0 = time.perf_counter()
        fn()
    ts = kernel_times()
    s_c.synchronize()
    print(name, ts, "copy+kernels wall %.2f ms" % ((time.perf_counter() - t0) * 1e3))
with torch.cuda.stream(s_c):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    host.copy_(big, non_blocking=True)
    b.record()
s_c.synchronize()
print("D2H 400 MB alone: %.2f ms = %.1f GB/s" % (a.elapsed_time(b), 0.4194 / a.elapsed_time(b) * 1e3))
# the same 400 MB as 80 pieces of 5 MB (what a serving call's responses look like), and as 8 of 50 MB
for pieces in (80, 8):
    per = (100 << 20) // pieces
    with torch.cuda.stream(s_c):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for p in range(pieces):
            host[p * per:(p + 1) * per].copy_(big[p * per:(p + 1) * per], non_blocking=True)
        b.record()
    ts = kernel_times()
    s_c.synchronize()
    print("beside %d D2H pieces of %.1f MB: kernels %s; the copies %.2f ms = %.1f GB/s" % (pieces, per * 4 / 1e6, ts, a.elapsed_time(b), 0.4194 / a.elapsed_time(b) * 1e3))
