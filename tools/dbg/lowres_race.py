"""does esr_esa_lowres_f32 alone (bf16 storage) give different results when launches overlap on several streams?"""
import ctypes, os, sys, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
from ntire2022_esr_amd import _lib as L
from ntire2022_esr_amd.engine import pack_dense
lib = L.lib(); DEV = "cuda:0"
g = torch.Generator().manual_seed(5)
f = 16
w2, b2 = torch.randn(f, f, 3, 3, generator=g) * 0.2, torch.randn(f, generator=g) * 0.1
wl, bl = torch.randn(f, f, 3, 3, generator=g) * 0.2, torch.randn(f, generator=g) * 0.1
blobs = [pack_dense(w2, b2, 16, 16).to(DEV), pack_dense(wl, bl, 16, 16).to(DEV)]
shapes = [(85, 128), (96, 128), (128, 85), (74, 128), (339, 510), (87, 128), (128, 96), (64, 64)]
jobs = []
for (h, w) in shapes:
    x = torch.randn(1, h, w, 16, generator=g).to(DEV).to(torch.bfloat16).contiguous()
    h2, w2_ = (h - 3) // 2 + 1, (w - 3) // 2 + 1; h3, w3 = (h2 - 7) // 3 + 1, (w2_ - 7) // 3 + 1
    pooled = torch.zeros(1, h3, w3, 16, device=DEV); y = torch.zeros(1, h3, w3, 16, device=DEV)
    d = L.EsaLowresDesc(); d.n, d.h, d.w, d.f, d.storage, d.n_layers = 1, h, w, f, 1, 1
    d.x = L.View(ctypes.c_void_p(x.data_ptr()), 16, 0); d.w_s2, d.pooled, d.y = blobs[0].data_ptr(), pooled.data_ptr(), y.data_ptr()
    d.layer[0].kind, d.layer[0].act, d.layer[0].w = 0, 0, blobs[1].data_ptr()
    jobs.append((d, x, pooled, y))
def run(d, stream): L.check(lib.esr_esa_lowres_f32(ctypes.byref(d), ctypes.c_void_p(stream.cuda_stream)), "lowres")
s0 = torch.cuda.current_stream()
want = []
for d, x, pooled, y in jobs:
    run(d, s0); torch.cuda.synchronize(); want.append((pooled.clone(), y.clone()))
streams = [torch.cuda.Stream(DEV) for _ in range(4)]
bad = 0
noise = torch.randn(4096, 4096, device=DEV)
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    for d, x, pooled, y in jobs: pooled.zero_(); y.zero_()
    torch.cuda.synchronize()
    for i, (d, x, pooled, y) in enumerate(jobs):
        st = streams[(i + rnd) % 4]
        with torch.cuda.stream(st):
            if "noise" in sys.argv and i % 2 == 0: noise.mul_(1.0001)
            run(d, st)
    torch.cuda.synchronize()
    for i, (d, x, pooled, y) in enumerate(jobs):
        if not torch.equal(pooled, want[i][0]) or not torch.equal(y, want[i][1]):
            bad += 1
            dp = (pooled - want[i][0]).abs(); nz = dp.nonzero()
            if bad < 6: print(f"round {rnd} job {i} shape {shapes[i]}: pooled differs at {len(nz)} values, rows {nz[:,1].min().item()}..{nz[:,1].max().item()} cols {nz[:,2].min().item()}..{nz[:,2].max().item()}" if len(nz) else f"round {rnd} job {i}: y differs only")
print("mismatches:", bad)
