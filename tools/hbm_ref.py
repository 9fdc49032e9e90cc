"""Reference bandwidths of simple streaming kernels on this GPU (torch elementwise kernels): copy, add, strided-slice copy."""
import torch
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
N = 32 * 256 * 256
a = torch.randn(N, 64, device=dev); b = torch.randn(N, 64, device=dev); c = torch.empty(N, 64, device=dev)
gb = a.numel() * 4 / 1e9
print(f"copy  (1 read + 1 write of {gb:.2f} GB): {2 * gb / t(lambda: c.copy_(a)) / 1e3:.2f} TB/s")
print(f"add   (2 reads + 1 write):              {3 * gb / t(lambda: torch.add(a, b, out=c)) / 1e3:.2f} TB/s")
print(f"fill  (1 write):                         {gb / t(lambda: c.fill_(1.0)) / 1e3:.2f} TB/s")
d = torch.empty(N, 128, device=dev)
print(f"copy into a 64-of-128 channel slice:     {2 * gb / t(lambda: d[:, 32:96].copy_(a)) / 1e3:.2f} TB/s")
e2 = torch.empty(N, 112, device=dev); f = torch.randn(N, 28, device=dev)
print(f"copy into a 28-of-112 channel slice:     {2 * f.numel() * 4 / 1e9 / t(lambda: e2[:, 28:56].copy_(f)) / 1e3:.2f} TB/s")
