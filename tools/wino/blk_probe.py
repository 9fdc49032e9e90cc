#!/usr/bin/env python3
"""NHWC vs channel-blocked INPUT for one Winograd layer (research): event-timed, alternating.  blk_probe.py [cin cout]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
from ntire2022_esr_amd import ops
from ntire2022_esr_amd.engine import pack_conv, pack_wino
dev = "cuda:0"
cin, cout = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 64)
n, h, w = 32, 256, 256
g = torch.Generator().manual_seed(0)
x = torch.randn(n, h, w, cin, generator=g).to(dev)
xb = x.view(n, h, w, cin // 8, 8).permute(0, 3, 1, 2, 4).contiguous()
wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
b = torch.randn(cout, generator=g)
y = torch.empty(n, h, w, cout, device=dev)
other = [torch.randn(n, h, w, 64, device=dev) for _ in range(4)]       # cold caches between launches
def run(blocked):
    ts = []
    for i in range(12):
        other[i % 4].add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv2d(xb if blocked else x, wt, b, act=1, blocked_in=blocked, out=y, wino=True)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
ya = ops.conv2d(x, wt, b, act=1, wino=True).clone(); yb_ = ops.conv2d(xb, wt, b, act=1, blocked_in=True, wino=True)
print("equal:", bool(torch.equal(ya, yb_)))
for rep in range(3):
    print(f"{cin}->{cout}: NHWC {run(False):.4f} ms   blocked {run(True):.4f} ms", flush=True)
