#!/usr/bin/env python3
"""one conv_s16 launch config in a loop (for rocprofv3 --pmc): python tools/abl/probe_one.py cin cout k res [lib]"""
import ctypes, os, sys
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(os.path.dirname(HERE)); sys.path.insert(0, REPO)
import torch
from ntire2022_esr_amd import _lib as L
from ntire2022_esr_amd.engine import pack_conv_s16
cin, cout, k, res = (int(v) for v in sys.argv[1:5])
libp = sys.argv[5] if len(sys.argv) > 5 else os.path.join(REPO, "ntire2022_esr_amd", "libesr_hip.so")
dev = "cuda:0"
x = torch.randn(32, 256, 256, cin, device=dev).to(torch.bfloat16)
y = torch.zeros(32, 256, 256, cout, device=dev, dtype=torch.bfloat16)
r = torch.randn(32, 256, 256, cout, device=dev).to(torch.bfloat16)
blob = pack_conv_s16(torch.randn(cout, cin, k, k) * 0.1, torch.randn(cout), "bf16").to(dev)
lib = ctypes.CDLL(libp)
lib.esr_conv2d_f32.argtypes = [ctypes.POINTER(L.ConvDesc), ctypes.c_void_p]
d = L.ConvDesc()
d.n, d.h, d.w, d.cin, d.cout, d.ksize = 32, 256, 256, cin, cout, k
d.act, d.slope, d.storage, d.compute = 1, 0.05, 1, 1
d.inp = L.View(x.data_ptr(), cin, 0); d.out0 = L.View(y.data_ptr(), cout, 0)
if res:
    d.res_mode, d.res = res, L.View(r.data_ptr(), cout, 0)
d.wpacked = blob.data_ptr()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(10):
    assert lib.esr_conv2d_f32(ctypes.byref(d), st) == 0
torch.cuda.synchronize()
