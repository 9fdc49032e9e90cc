"""Bit-compare a library variant against the product .so on a few conv shapes: cmp_var.py <variant>"""
import os, subprocess, sys, tempfile
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
torch.manual_seed(1)
outs = []
for (B, H, W, cin, cout, act, res) in ((8, 256, 256, 64, 64, 1, 0), (3, 200, 136, 48, 64, 1, 0), (8, 256, 256, 64, 64, 0, 2), (9, 96, 80, 48, 48, 1, 1)):
    x = torch.randn(B, H, W, cin, device=dev); w = torch.randn(cout, cin, 3, 3) * 0.05; b = torch.randn(cout)
    r = torch.randn(B, H, W, cout, device=dev)
    out = torch.full((B, H, W, cout), float("nan"), device=dev)
    for _ in range(3):
        ops.conv2d(x, w, b, act=act, out=out, **(dict(res=r, res_mode=res) if res else dict()))
    torch.cuda.synchronize()
    outs.append(out.cpu())
torch.save(outs, "{dump}")
'''
var = sys.argv[1]
dumps = []
for name, so in (("prod", os.path.join(root, "ntire2022_esr_amd/libesr_hip.so")), (var, os.path.join(here, f"libesr_var_{var}.so"))):
    d = os.path.join(tempfile.gettempdir(), f"cmp_{name}.pt"); dumps.append(d)
    env = dict(os.environ)
    if name == "prod": env["ESR_TALL_MIN"] = "1000000000"      # reference = the 4-wave kernel
    else: env["ESR_TALL_MIN"] = "1"
    subprocess.run([sys.executable, "-c", code.format(root=root, so=so, dump=d)], check=True, env=env)
import torch
a, b = torch.load(dumps[0]), torch.load(dumps[1])
for i, (x, y) in enumerate(zip(a, b)):
    print(f"case {i}: identical={torch.equal(x, y)} nan={int(torch.isnan(y).sum())} maxdiff={float((x - y).abs().max()):.3g}")
