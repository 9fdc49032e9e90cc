#!/usr/bin/env python3
"""conv_s16 against an fp64 reference at shapes with MORE tiles than blocks and partial tiles (research tooling)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from ntire2022_esr_amd import ops
from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16
DEV = "cuda:0"
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous()
for compute, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
    for (n, cin, cout, k, hw, act, res_mode) in ((1, 16, 16, 3, (270, 480), 1, 1), (1, 16, 16, 3, (270, 480), 1, 2), (1, 16, 16, 1, (270, 480), 1, 2),
                                                 (1, 16, 32, 3, (270, 480), 1, 2), (1, 64, 64, 3, (270, 480), 1, 2), (1, 64, 64, 1, (270, 480), 1, 2),
                                                 (1, 48, 48, 3, (339, 510), 1, 2), (1, 32, 32, 3, (339, 510), 1, 1), (1, 64, 64, 3, (339, 510), 1, 0),
                                                 (2, 48, 48, 1, (339, 510), 1, 2), (1, 128, 64, 1, (270, 480), 1, 2), (4, 64, 64, 3, (256, 256), 1, 2)):
        g = torch.Generator().manual_seed(cin + cout + hw[0] + k)
        x = torch.randn(n, cin, *hw, generator=g).to(dt)
        r = torch.randn(n, cout, *hw, generator=g).to(dt)  # residual != input: loaded from HBM
        w = torch.randn(cout, cin, k, k, generator=g) * (0.1 if k == 3 else 0.2)
        b = torch.randn(cout, generator=g)
        blob = pack_conv_s16(w, b, compute, cin_phys=cin)
        weff, _ = unpack_conv_s16(blob, cin, cout, k, compute, cin_phys=cin)
        conv = F.conv2d(x.double().to(DEV), weff.double().to(DEV), b.double().to(DEV), padding=k // 2)
        lr = lambda t: F.leaky_relu(t, 0.05)
        rd = r.double().to(DEV)
        ref = lr(conv + rd) if res_mode == 1 else (lr(conv) + rd if res_mode == 2 else lr(conv))
        worst = 0.0; bad = 0
        for it in range(5):
            y = ops.conv2d(nhwc(x).to(DEV), w, b, act=act, res=nhwc(r).to(DEV) if res_mode else None, res_mode=res_mode, cin=cin, packed=blob.to(DEV))
            got = y.double().permute(0, 3, 1, 2)[:, :cout]
            eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
            err = (got - ref).abs() - (ref.abs() * eps * 1.01 + 3e-5 * max(1.0, float(ref.abs().max())))
            worst = max(worst, float(err.max())); bad = max(bad, int((err > 0).sum()))
            if it == 0 and bad:
                idx = (err > 0).nonzero()
                print("   first bad (n,c,y,x):", idx[:3].tolist(), " last:", idx[-3:].tolist(), " rows:", sorted(set(idx[:, 2].tolist()))[:12])
        print(f"{compute} n{n} {cin}->{cout} k{k} {hw} res{res_mode}: worst excess {worst:.3e}, bad elements {bad}", flush=True)
