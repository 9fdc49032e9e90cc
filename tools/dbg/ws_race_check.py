import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ntire2022_esr_amd import _lib as L
L.SO_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libesr_dbg_ws_plain.so")
from ntire2022_esr_amd import ops
from ntire2022_esr_amd.engine import pack_conv
dev = torch.device("cuda:0")
torch.manual_seed(0)
for B in (1, 2, 4, 8):
    x = torch.randn(B, 256, 256, 64, device=dev); w = torch.randn(64, 64, 3, 3) * 0.05; b = torch.randn(64)
    pk = pack_conv(w, b).to(dev)
    outs = []
    for rep in range(4):
        out = torch.full((B, 256, 256, 64), float("nan"), device=dev)
        ops.conv2d(x, w, b, act=1, packed=pk, out=out)
        torch.cuda.synchronize()
        outs.append(out)
    for i in range(1, 4):
        d = (outs[i] - outs[0])
        bad = ~(d == 0)
        if bad.any():
            idx = bad.nonzero()
            print("B", B, "rep", i, "mismatch count", int(bad.sum()), "first", idx[0].tolist(), "last", idx[-1].tolist(),
                  "rows", sorted(set((idx[:, 1] % 16).tolist()))[:16], "nan", int(torch.isnan(outs[i]).sum()))
        else:
            print("B", B, "rep", i, "identical")

    if B > 1:
        for i in range(B):
            o1 = torch.full((1, 256, 256, 64), float("nan"), device=dev)
            ops.conv2d(x[i:i+1].contiguous(), w, b, act=1, packed=pk, out=o1)
            bad = ~(o1[0] == outs[0][i])
            if bad.any():
                idx = bad.nonzero()
                print("  B", B, "img", i, "vs single: mismatches", int(bad.sum()), "first", idx[0].tolist(), "last", idx[-1].tolist(), "maxdiff", float((o1[0]-outs[0][i]).abs().max()))
