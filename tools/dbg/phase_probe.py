"""Run the 'waits' instrumented conv (64->64, B=32, warm) and print per-chunk cycle sums per wave."""
import ctypes, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from ntire2022_esr_amd import _lib as L
L.SO_PATH = os.path.join(here, "libesr_dbg_waits.so")
from ntire2022_esr_amd import ops
from ntire2022_esr_amd.engine import pack_conv
lib = L.lib(); lib.esr_set_dbg.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
x = torch.randn(32, 256, 256, 64, device=dev); w = torch.randn(64, 64, 3, 3) * 0.05; b = torch.randn(64)
pk = pack_conv(w, b).to(dev); out = torch.empty(32, 256, 256, 64, device=dev)
dbg = torch.zeros(512 * 4 * 8, dtype=torch.int64, device=dev)
for _ in range(20): ops.conv2d(x, w, b, act=1, packed=pk, out=out)
torch.cuda.synchronize()
lib.esr_set_dbg(ctypes.c_void_p(dbg.data_ptr()))
for _ in range(5): ops.conv2d(x, w, b, act=1, packed=pk, out=out)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(512, 4, 8).astype(np.float64)
nchunk = 16 * 8
names = ["wait in_reg (vmcnt)", "ds_write issue", "request input g+2", "wait weight DMA + lgkm", "s_barrier", "chunk top: DMA issue + first frags", "MFMA phase", "total loop+epilogues"]
raw7 = dbg.cpu().numpy().reshape(512, 4, 8)[:, :, 7]
print(f"   chunk top split: weight DMA issue (+fallback loads) {np.mean(raw7 & 0xffffffff) / nchunk:8.0f}   first-frag ds_read issue {np.mean(raw7 >> 32) / nchunk:8.0f}   (rest = lgkmcnt wait)")
for i, n in enumerate(names[:7]):
    v = d[:, :, i] / (16 if i == 7 else nchunk)
    print(f"{n:36s} per {'tile ' if i == 7 else 'chunk'}: mean {v.mean():8.0f}  p10 {np.percentile(v,10):8.0f}  p90 {np.percentile(v,90):8.0f}")
