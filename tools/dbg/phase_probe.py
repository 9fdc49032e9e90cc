"""Debug helper: run the instrumented conv kernel (tools/dbg/libesr_dbg.so) and dump per-wave phase stamps."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ntire2022_esr_amd import _lib as L
L.SO_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libesr_dbg.so")
from ntire2022_esr_amd import ops
lib = L.lib()
lib.esr_set_dbg.argtypes = [ctypes.c_void_p]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
x = torch.randn(B, 256, 256, 64, device=dev)
w = torch.randn(64, 64, 3, 3) * 0.05
b = torch.randn(64)
from ntire2022_esr_amd.engine import pack_conv
pk = pack_conv(w, b).to(dev)
out = torch.empty(B, 256, 256, 64, device=dev)
nblk = min(B * 256, 512)
dbg = torch.zeros(nblk * 4 * 8 * 2 + 512 * 4 * 8 * 2, dtype=torch.int64, device=dev)
for it in range(3):
    ops.conv2d(x, w, b, act=1, packed=pk, out=out)
torch.cuda.synchronize()
lib.esr_set_dbg(ctypes.c_void_p(dbg.data_ptr()))
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); ops.conv2d(x, w, b, act=1, packed=pk, out=out); e.record(); torch.cuda.synchronize()
print("instrumented launch ms", s.elapsed_time(e))
lib.esr_set_dbg(None)
os.makedirs("gpurun_out/dbg", exist_ok=True)
a = dbg.cpu().numpy()
np.save("gpurun_out/dbg/phase.npy", a[:nblk * 32].reshape(nblk, 4, 8))
np.save("gpurun_out/dbg/epi.npy", a[512 * 32:512 * 32 + nblk * 32].reshape(nblk, 4, 8))
