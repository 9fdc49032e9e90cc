#!/usr/bin/env python3
"""conv48r_kernel at batch 32: 16 x 32 tiles / two stages (product) against 16 x 16 tiles with two and three stages (esr_dbg_c48) -- same
arithmetic, the outputs must be bit-identical; times round-robin"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ntire2022_esr_amd import _lib as L, ops
from ntire2022_esr_amd.engine import pack_conv_s16
lib = L.lib()
lib.esr_dbg_c48.argtypes = [ctypes.c_int]
dev = "cuda:0"
for (n, h, w) in ((32, 256, 256), (32, 270, 480), (4, 339, 510)):
    x = torch.randn(n, h, w, 48, device=dev).to(torch.bfloat16)
    wt, b = torch.randn(48, 48, 3, 3) * 0.1, torch.randn(48)
    blob = pack_conv_s16(wt, b, "bf16").to(dev)
    ref = None
    for rnd in range(2):
        for mode in (0, 2, 3):
            lib.esr_dbg_c48(mode)
            for _ in range(3): y = ops.conv2d(x, wt, b, act=1, packed=blob)
            torch.cuda.synchronize()
            if ref is None: ref = y.clone()
            assert torch.equal(y, ref), mode
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            out = torch.empty_like(y)
            e0.record()
            for _ in range(20): ops.conv2d(x, wt, b, act=1, packed=blob, out=out)
            e1.record(); torch.cuda.synchronize()
            print(f"{n}x{h}x{w} mode {mode}: {e0.elapsed_time(e1) / 20 * 1000:7.1f} us", flush=True)
lib.esr_dbg_c48(0)
