#!/usr/bin/env python3
"""where does wino8_f32_kernel differ from wino_f32_kernel?  (research)"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
from ntire2022_esr_amd import ops, _lib as L
dev = "cuda:0"
n, cin, cout, h, w = 4, 48, 64, 256, 256
g = torch.Generator().manual_seed(1)
x = torch.randn(n, h, w, cin, generator=g).to(dev)
wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
b = torch.randn(cout, generator=g)
lib = L.lib()
lib.esr_dbg_wino8(ctypes.c_int(0)); y4 = ops.conv2d(x, wt, b, wino=True)
lib.esr_dbg_wino8(ctypes.c_int(1)); y8 = ops.conv2d(x, wt, b, wino=True)
torch.cuda.synchronize()
d = (y8 - y4).abs()
print("max", float(d.max()), "frac wrong", float((d > 1e-4).float().mean()))
bad = d > 1e-4
print("per image", bad.flatten(1).float().mean(1).tolist())
print("per channel (first 64)", [round(v, 2) for v in bad.permute(3, 0, 1, 2).flatten(1).float().mean(1).tolist()])
rows = bad[0].any(2).float().mean(1)     # per row fraction of bad pixels, image 0
print("rows of image 0 with bad pixels:", [(i, round(float(v), 2)) for i, v in enumerate(rows.tolist()) if v > 0][:40])
cols = bad[0].any(2).float().mean(0)
print("cols of image 0 with bad pixels:", [(i, round(float(v), 2)) for i, v in enumerate(cols.tolist()) if v > 0][:40])
print("y8 nan?", bool(torch.isnan(y8).any()), "sample", y8[0, 5, 5, :4].tolist(), y4[0, 5, 5, :4].tolist())
idx = bad.nonzero()
import collections
print("wrong values:", len(idx))
print("y%4:", collections.Counter((idx[:, 1] % 4).tolist()))
print("x%16:", sorted(collections.Counter((idx[:, 2] % 16).tolist()).items()))
print("c:", sorted(collections.Counter((idx[:, 3]).tolist()).items())[:70])
# strips: (n, y//4, x//16)
st = collections.Counter(zip(idx[:, 0].tolist(), (idx[:, 1] // 4).tolist(), (idx[:, 2] // 16).tolist()))
print("strips hit:", len(st), "of", n * (h // 4) * (w // 16), "; values per strip:", collections.Counter(st.values()))
# which step k of the wave? strip index s = (n*64 + y//4)*16 + x//16 ; per_block = ceil(S/128); k = ((s % per_block) // 8)
S = n * (h // 4) * (w // 16); pb = (S + 127) // 128
ks = collections.Counter((((nn * (h // 4) + yy) * (w // 16) + xx) % pb) // 8 for (nn, yy, xx) in st)
print("per_block", pb, "step k of the wave:", sorted(ks.items()))
