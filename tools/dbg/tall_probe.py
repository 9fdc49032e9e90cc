"""Run the 'tall' instrumented 8-wave conv (64->64, B=32, warm) and print per-chunk cycle sums per wave group."""
import ctypes, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from ntire2022_esr_amd import _lib as L
L.SO_PATH = os.path.join(here, "libesr_dbg_tall.so")
from ntire2022_esr_amd import ops
from ntire2022_esr_amd.engine import pack_conv
lib = L.lib(); lib.esr_set_dbg.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
x = torch.randn(32, 256, 256, 64, device=dev); w = torch.randn(64, 64, 3, 3) * 0.05; b = torch.randn(64)
pk = pack_conv(w, b).to(dev); out = torch.empty(32, 256, 256, 64, device=dev)
dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
for _ in range(20): ops.conv2d(x, w, b, act=1, packed=pk, out=out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): ops.conv2d(x, w, b, act=1, packed=pk, out=out)
e.record(); torch.cuda.synchronize()
print(f"instrumented kernel: {s.elapsed_time(e)/20:.4f} ms per launch ({154.6/(s.elapsed_time(e)/20)/155:.3f} of 155)")
lib.esr_set_dbg(ctypes.c_void_p(dbg.data_ptr()))
ops.conv2d(x, w, b, act=1, packed=pk, out=out)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(256, 8, 8).astype(np.float64)
ntile = 16; nchunk = ntile * 8
names = ["staging block", "stage-end vmcnt/lgkm wait", "s_barrier", "whole chunk (2 waves/SIMD: ideal 18432)", "epilogue / tile", "first-frag wait"]
for grp, sl in (("early waves 0-3", slice(0, 4)), ("late  waves 4-7", slice(4, 8))):
    print(grp)
    for i, n in enumerate(names):
        v = d[:, sl, i] / (ntile if i == 4 else nchunk)
        print(f"   {n:44s} mean {v.mean():8.0f}  p10 {np.percentile(v,10):8.0f}  p90 {np.percentile(v,90):8.0f}")
