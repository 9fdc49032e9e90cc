"""Warm timing of one 64->64 3x3 conv (B=32) through the wave-specialised kernel of each ablated library."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
code = r'''
import os, sys, torch
os.environ["ESR_WS_MIN_NT"] = "1"
sys.path.insert(0, "{root}")
from ntire2022_esr_amd import _lib as L
L.SO_PATH = "{so}"
from ntire2022_esr_amd import ops
from ntire2022_esr_amd.engine import pack_conv
dev = torch.device("cuda:0")
x = torch.randn(32, 256, 256, 64, device=dev); w = torch.randn(64, 64, 3, 3) * 0.05; b = torch.randn(64)
pk = pack_conv(w, b).to(dev); out = torch.empty(32, 256, 256, 64, device=dev)
for _ in range(20): ops.conv2d(x, w, b, packed=pk, out=out, act=1)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(30): ops.conv2d(x, w, b, packed=pk, out=out, act=1)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 30
print("{var:12s} {{:7.4f}} ms  {{:6.1f}} TFLOP/s  {{:.3f}} of 155".format(ms, 154.6 / ms, 154.6 / ms / 155))
'''
for var in sys.argv[1:]:
    so = os.path.join(here, f"libesr_dbg_ws_{var}.so") if var != "prod" else os.path.join(os.path.dirname(os.path.dirname(here)), "ntire2022_esr_amd/libesr_hip.so")
    subprocess.run([sys.executable, "-c", code.format(root=os.path.dirname(os.path.dirname(here)), so=so, var=var)])
