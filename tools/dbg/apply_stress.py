#!/usr/bin/env python3
"""race hunt: esa_apply (RLFN shape, bf16) looping on stream A while stream B loops ONE partner kernel; every apply result is compared
with the serial one.  usage: apply_stress.py [partner ...]   partners: none apply conv3 conv3post conv1 pack"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
from ntire2022_esr_amd import _lib as L, ops
DEV = "cuda:0"
dt = torch.bfloat16
g = torch.Generator().manual_seed(1)
H, W, C, F_ = 85, 128, 46, 16
x = (torch.randn(1, H, W, 48, generator=g) * 30).to(dt).to(DEV)
c1 = torch.randn(1, H, W, 16, generator=g).to(dt).to(DEV)
c3 = torch.zeros(1, 12, 19, 16); c3[..., :F_] = torch.randn(1, 12, 19, F_, generator=g); c3 = c3.to(DEV)
wf, bf = torch.randn(F_, F_, generator=g) * 0.3, torch.randn(F_, generator=g)
w4, b4 = torch.randn(C, F_, generator=g) * 0.3, torch.randn(C, generator=g)
from ntire2022_esr_amd.engine import pack_dense, pack_conv_s16
import ctypes
pf, p4 = pack_dense(wf.reshape(F_, F_, 1, 1), bf, 16, 16).to(DEV), pack_dense(w4.reshape(C, F_, 1, 1), b4, 16, 48).to(DEV)

def apply_into(y):
    d = L.EsaDesc()
    d.n, d.h, d.w, d.c, d.f, d.h_lo, d.w_lo = 1, H, W, C, F_, 12, 19
    d.storage = L.STORE["bf16"]
    d.x = L.View(ctypes.c_void_p(x.data_ptr()), 48, 0); d.y = L.View(ctypes.c_void_p(y.data_ptr()), 48, 0)
    d.c1, d.c3, d.w0, d.w1 = c1.data_ptr(), c3.data_ptr(), pf.data_ptr(), p4.data_ptr()
    L.check(L.lib().esr_esa_apply_f32(ctypes.byref(d), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "apply")

ref = torch.zeros(1, H, W, 48, dtype=dt, device=DEV); apply_into(ref); torch.cuda.synchronize()
# partner inputs
xb = torch.randn(1, 96, 128, 48, generator=g).to(dt).to(DEV)
w3 = torch.randn(48, 48, 3, 3, generator=g) * 0.05; b3 = torch.randn(48, generator=g)
pk3 = pack_conv_s16(w3, b3, "bf16").to(DEV)
w1 = torch.randn(48, 48, 1, 1, generator=g) * 0.05
pk1 = pack_conv_s16(w1, b3, "bf16").to(DEV)
yb = torch.zeros(1, 96, 128, 48, dtype=dt, device=DEV)
xin = torch.rand(1, 3, 96, 128, device=DEV)
x2 = (torch.randn(1, 96, 128, 48, generator=g) * 30).to(dt).to(DEV); c12 = torch.randn(1, 96, 128, 16, generator=g).to(dt).to(DEV)
y2 = torch.zeros_like(x2)

def partner(name):
    if name == "conv3":
        ops.conv2d(xb, w3, b3, act=L.ACT_LRELU, packed=pk3, out=yb)
    elif name == "conv3res":
        ops.conv2d(xb, w3, b3, act=L.ACT_LRELU, packed=pk3, out=yb, res=x2, res_mode=L.RES_POST_ACT)
    elif name == "conv1":
        ops.conv2d(xb, w1, b3, packed=pk1, out=yb)
    elif name == "apply":
        d = L.EsaDesc()
        d.n, d.h, d.w, d.c, d.f, d.h_lo, d.w_lo = 1, 96, 128, C, F_, 12, 19
        d.storage = L.STORE["bf16"]
        d.x = L.View(ctypes.c_void_p(x2.data_ptr()), 48, 0); d.y = L.View(ctypes.c_void_p(y2.data_ptr()), 48, 0)
        d.c1, d.c3, d.w0, d.w1 = c12.data_ptr(), c3.data_ptr(), pf.data_ptr(), p4.data_ptr()
        L.check(L.lib().esr_esa_apply_f32(ctypes.byref(d), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "apply")

sa, sb = torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)
for name in (sys.argv[1:] or ["none", "apply", "conv3", "conv3res", "conv1"]):
    ys = [torch.zeros(1, H, W, 48, dtype=dt, device=DEV) for _ in range(200)]
    torch.cuda.synchronize()
    for i in range(200):
        with torch.cuda.stream(sa):
            apply_into(ys[i])
        if name != "none":
            with torch.cuda.stream(sb):
                partner(name); partner(name)
    torch.cuda.synchronize()
    bad = [i for i in range(200) if not torch.equal(ys[i], ref)]
    detail = ""
    if bad:
        d = (ys[bad[0]].float() - ref.float()).abs()
        nz = d.nonzero()
        detail = f" first: iter {bad[0]}, {int((d > 0).sum())} values, rows {int(nz[:,1].min())}..{int(nz[:,1].max())} cols {int(nz[:,2].min())}..{int(nz[:,2].max())} ch {int(nz[:,3].min())}..{int(nz[:,3].max())}"
    print(f"partner {name:9s}: {len(bad)} of 200 apply results differ from the serial one{detail}", flush=True)
