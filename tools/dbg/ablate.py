"""Warm back-to-back timing of one 64->64 3x3 conv (B=32) for each ablated library variant."""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
code = r'''
import ctypes, os, sys, torch
sys.path.insert(0, "{root}")
from ntire2022_esr_amd import _lib as L
L.SO_PATH = "{so}"
from ntire2022_esr_amd import ops
from ntire2022_esr_amd.engine import pack_conv
dev = torch.device("cuda:0")
x = torch.randn(32, 256, 256, 64, device=dev); w = torch.randn(64, 64, 3, 3) * 0.05; b = torch.randn(64)
pk = pack_conv(w, b).to(dev); out = torch.empty(32, 256, 256, 64, device=dev); r = torch.randn(32, 256, 256, 64, device=dev)
for mode, kw in (("lrelu", dict(act=1)), ("res_pre", dict(act=0, res=r, res_mode=1))):
    for _ in range(20): ops.conv2d(x, w, b, packed=pk, out=out, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30): ops.conv2d(x, w, b, packed=pk, out=out, **kw)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 30
    print("{var:8s} {{:8s}} {{:7.4f}} ms  {{:6.1f}} TFLOP/s".format(mode, ms, 154.6 / ms))
'''
for var in sys.argv[1:]:
    so = os.path.join(here, f"libesr_dbg_{var}.so")
    subprocess.run([sys.executable, "-c", code.format(root=os.path.dirname(os.path.dirname(here)), so=so, var=var)])
