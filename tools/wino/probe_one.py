"""a few launches of one wino_f32_kernel shape (for rocprofv3 --pmc): probe_one.py [cin] [dyn]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ntire2022_esr_amd import _lib as L
from ntire2022_esr_amd.engine import pack_conv, pack_wino
lib = L.lib(); dev = "cuda:0"
cin = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dyn = len(sys.argv) > 2 and sys.argv[2] == "dyn"
B, H, W, cout = 32, 256, 256, 64
x = torch.randn(B, H, W, cin, device=dev); y = torch.zeros(B, H, W, cout, device=dev)
w = torch.randn(cout, cin, 3, 3) * 0.05; bias = torch.randn(cout)
blob = pack_conv(w, bias).to(dev); wb = pack_wino(w, bias).to(dev)
sched = torch.zeros(8, dtype=torch.int64, device=dev)
d = L.ConvDesc(); d.n, d.h, d.w, d.cin, d.cout, d.ksize = B, H, W, cin, cout, 3
d.act, d.slope = 1, 0.05
d.inp = L.View(x.data_ptr(), cin, 0); d.out0 = L.View(y.data_ptr(), cout, 0)
d.wpacked = blob.data_ptr(); d.wino_wpacked = wb.data_ptr()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(12): assert lib.esr_conv2d_f32(ctypes.byref(d), st) == 0
torch.cuda.synchronize()
