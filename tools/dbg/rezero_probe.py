#!/usr/bin/env python3
"""does a forward depend on what another shape left in the workspace's pad channels?  (research tooling)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from ntire2022_esr_amd.registry import select_model
dev = torch.device("cuda:0")
for name, compute in (("team04_rlfn", "bf16"), ("rfdn_baseline", "bf16"), ("team18_bsrn", "f16"), ("imdn_baseline", "bf16"), ("imdn_baseline", "f32"), ("rfdn_baseline", "f32"), ("team18_bsrn", "f32"), ("team04_rlfn", "f32")):
    m, _, dr, _ = select_model(bench.MODELS[name][0], dev)
    m.set_compute(compute)
    g = torch.Generator().manual_seed(1)
    shapes = [(1, 3, 64, 80), (2, 3, 33, 47), (1, 3, 120, 40), (1, 3, 17, 15), (3, 3, 40, 40)]
    xs = [(torch.rand(*s, generator=g) * dr).to(dev) for s in shapes]
    m.rezero_on_switch = True
    ref = [m(x).clone() for x in xs]
    m.rezero_on_switch = False
    ok = True
    for rep in range(3):
        for i in (3, 0, 4, 1, 2, 0, 3):
            y = m(xs[i])
            same = torch.equal(y, ref[i])
            ok &= same
            if not same: print("   differs:", name, compute, shapes[i], float((y - ref[i]).abs().max()))
    print(name, compute, "independent of the workspace's previous content:", ok, flush=True)
