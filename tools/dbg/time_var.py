"""Warm timing of the IMDN conv shapes (B=32) for library variants: time_var.py name1 name2 ...  ('prod' = the product .so)"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
code = r'''
import os, sys, torch
sys.path.insert(0, "{root}")
from ntire2022_esr_amd import _lib as L
L.SO_PATH = "{so}"
from ntire2022_esr_amd import ops
from ntire2022_esr_amd.engine import pack_conv
dev = torch.device("cuda:0")
B = int(os.environ.get("BATCH", "32"))
res = []
for cin, cout in ((64, 64), (48, 64)):
    x = torch.randn(B, 256, 256, cin, device=dev); w = torch.randn(cout, cin, 3, 3) * 0.05; b = torch.randn(cout)
    pk = pack_conv(w, b).to(dev); out = torch.empty(B, 256, 256, cout, device=dev)
    for _ in range(10): ops.conv2d(x, w, b, packed=pk, out=out, act=1)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30): ops.conv2d(x, w, b, packed=pk, out=out, act=1)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 30
    gf = 2 * B * 65536 * 9 * cin * cout / 1e9
    res.append("%d->%d %.4f ms %.3f" % (cin, cout, ms, gf / ms / 155))
print("{var:14s} " + "   ".join(res))
'''
for var in sys.argv[1:]:
    so = os.path.join(here, f"libesr_var_{var}.so") if var != "prod" else os.path.join(root, "ntire2022_esr_amd/libesr_hip.so")
    subprocess.run([sys.executable, "-c", code.format(root=root, so=so, var=var)])
