"""An HBM-random-bound kernel (a gather of 64 M random rows of 16 bytes out of a 4 GB table) alone, beside a device -> pinned-host
copy command, and beside a host -> device one: does the copy slow it?"""
import torch

dev = torch.device("cuda", 0)
table = torch.zeros((256 << 20, 4), dtype=torch.float32, device=dev)  # 4 GB
idx = torch.randint(0, 256 << 20, (32 << 20,), device=dev)
big = torch.zeros(100 << 20, dtype=torch.float32, device=dev)  # 400 MB
host = torch.empty(100 << 20, dtype=torch.float32).pin_memory()
s_k, s_c = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()


def kernel_times(n=6):
    ts = []
    with torch.cuda.stream(s_k):
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = table[idx]
            b.record()
            ts.append((a, b))
    s_k.synchronize()
    del out
    return [round(a.elapsed_time(b), 3) for a, b in ts]


kernel_times(2)
print("alone          ", kernel_times())
for name, fn in (("beside D2H copy", lambda: host.copy_(big, non_blocking=True)), ("beside H2D copy", lambda: big.copy_(host, non_blocking=True))):
    with torch.cuda.stream(s_c):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
    ts = kernel_times()
    s_c.synchronize()
    print(name, ts, "the copy: %.2f ms" % a.elapsed_time(b))
