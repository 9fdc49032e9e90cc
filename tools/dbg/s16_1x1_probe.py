#!/usr/bin/env python3
"""why are BSRN's 1x1 convolutions slow? act / slice-store / residual variants of a 48 -> 24|48 1x1 at 32x270x480 (research tooling)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ntire2022_esr_amd import ops, _lib as L
DEV = "cuda:0"
n, h, w = 32, 270, 480
dt = torch.float16
x = torch.randn(n, h, w, 48, device=DEV).to(dt)
r = torch.randn(n, h, w, 48, device=DEV).to(dt)
cat = torch.zeros(n, h, w, 96, device=DEV, dtype=dt)
def bench(label, fn, nbytes):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{label:60s} {ms:.4f} ms  {nbytes / ms / 1e9:.2f} TB/s", flush=True)
npx = n * h * w
for cout in (24, 48):
    wt = torch.randn(cout, 48) * 0.2; b = torch.randn(cout)
    y = torch.zeros(n, h, w, (cout + 7) // 8 * 8, device=DEV, dtype=dt)
    for act, an in ((L.ACT_LRELU, "lrelu"), (L.ACT_GELU, "gelu"), (L.ACT_NONE, "none")):
        bench(f"48->{cout} 1x1 {an}, dense out", lambda: ops.conv2d(x, wt, b, act=act, out=y), npx * (96 + 2 * cout))
    bench(f"48->{cout} 1x1 gelu, out = slice of a 96-ch buffer", lambda: ops.conv2d(x, wt, b, act=L.ACT_GELU, out=cat, out_coff=24), npx * (96 + 2 * cout))
    bench(f"48->{cout} 1x1 lrelu, out = slice of a 96-ch buffer", lambda: ops.conv2d(x, wt, b, act=L.ACT_LRELU, out=cat, out_coff=24), npx * (96 + 2 * cout))
wt = torch.randn(48, 48) * 0.2; b = torch.randn(48)
y = torch.zeros(n, h, w, 48, device=DEV, dtype=dt)
bench("48->48 1x1 none + residual (HBM, staged)", lambda: ops.conv2d(x, wt, b, act=L.ACT_NONE, res=r, res_mode=1, out=y), npx * 96 * 3)
wt3 = torch.randn(48, 48, 3, 3) * 0.05
bench("48->48 3x3 lrelu", lambda: ops.conv2d(x, wt3, b, act=L.ACT_LRELU, out=y), npx * 96 * 2)
bench("48->48 3x3 gelu", lambda: ops.conv2d(x, wt3, b, act=L.ACT_GELU, out=y), npx * 96 * 2)
bench("48->48 3x3 gelu + res == input", lambda: ops.conv2d(x, wt3, b, act=L.ACT_GELU, res=x, res_mode=1, out=y), npx * 96 * 2)
