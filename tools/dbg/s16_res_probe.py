#!/usr/bin/env python3
"""what do the wrong outputs of a residual conv contain? (research tooling)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from ntire2022_esr_amd import ops
from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16
DEV = "cuda:0"
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
compute, dt = "f16", torch.float16
n, cin, cout, k, hw = 1, 64, 64, 1, (64, 64)
g = torch.Generator().manual_seed(1)
x = torch.randn(n, cin, *hw, generator=g).to(dt)
r = (torch.arange(hw[0]).reshape(1, 1, -1, 1) * 1.0 + torch.arange(hw[1]).reshape(1, 1, 1, -1) / 64.0 + torch.zeros(n, cout, 1, 1)).to(dt)   # r[y, x] = y + x/64
w = torch.zeros(cout, cin, k, k); b = torch.zeros(cout)
blob = pack_conv_s16(w, b, compute, cin_phys=cin)
y = ops.conv2d(nhwc(x).to(DEV), w, b, act=1, res=nhwc(r).to(DEV), res_mode=2, cin=cin, packed=blob.to(DEV))
got = y.float().permute(0, 3, 1, 2)[0].cpu()
ref = r[0].float()
bad = (got - ref).abs() > 1e-3
print("bad", int(bad.sum()), "of", bad.numel())
for ch in (0, 5, 17, 40):
    print("channel", ch)
    for yy in (0, 3, 15, 16, 17, 20, 31, 32, 40):
        print("  row", yy, "got", [round(float(v), 3) for v in got[ch, yy, :4]], "ref", [round(float(v), 3) for v in ref[ch, yy, :4]])
