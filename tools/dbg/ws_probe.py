"""Run the instrumented wave-specialised conv (64->64, B=32, warm) and print per-chunk cycle sums."""
import ctypes, os, sys
os.environ["ESR_WS_MIN_NT"] = "1"
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from ntire2022_esr_amd import _lib as L
L.SO_PATH = os.path.join(here, "libesr_dbg_ws.so")
from ntire2022_esr_amd import ops
from ntire2022_esr_amd.engine import pack_conv
lib = L.lib(); lib.esr_set_dbg.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
x = torch.randn(32, 256, 256, 64, device=dev); w = torch.randn(64, 64, 3, 3) * 0.05; b = torch.randn(64)
pk = pack_conv(w, b).to(dev); out = torch.empty(32, 256, 256, 64, device=dev)
MODE = 1; NB = 256
dbg = torch.zeros(NB * 8 * 4, dtype=torch.int64, device=dev)
for _ in range(20): ops.conv2d(x, w, b, act=1, packed=pk, out=out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): ops.conv2d(x, w, b, act=1, packed=pk, out=out)
e.record(); torch.cuda.synchronize()
print(f"instrumented kernel: {s.elapsed_time(e)/20:.4f} ms per launch ({154.6/(s.elapsed_time(e)/20):.1f} TFLOP/s)")
lib.esr_set_dbg(ctypes.c_void_p(dbg.data_ptr()))
ops.conv2d(x, w, b, act=1, packed=pk, out=out)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(NB, 8, 4).astype(np.float64)
ntile = 8192 // NB; nchunk = ntile * 8
c, l = d[:, :4, :], d[:, 4:(8 if MODE == 1 else 6), :]
def line(n, v): print(f"{n:52s} mean {v.mean():9.0f}  p10 {np.percentile(v,10):9.0f}  p90 {np.percentile(v,90):9.0f}")
line("consumer: wait for stage counter / chunk", c[:, :, 0] / nchunk)
line("consumer: chunk loop / chunk (ideal 9216)", c[:, :, 1] / nchunk)
line("consumer: hand-over or own epilogue / tile", c[:, :, 2] / ntile)
line("consumer: wait for loader to free scratch / tile", c[:, :, 3] / ntile)
line("loader: wait for freed buffer / chunk", l[:, :, 0] / nchunk)
line("loader: vmcnt wait + ds_write / chunk", l[:, :, 1] / nchunk)
line("loader: request issue / chunk", l[:, :, 2] / nchunk)
line("loader: epilogue duty / chunk", l[:, :, 3] / nchunk)
