#!/usr/bin/env python3
"""wino8_f32_kernel on / off inside one process (esr_dbg_wino8): per-kernel event times of IMDN fp32 at batch 32, alternating"""
import ctypes, os, sys, collections
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
from ntire2022_esr_amd import _lib as L
if len(sys.argv) > 2:
    L.SO_PATH = sys.argv[2]
from ntire2022_esr_amd.registry import select_model
dev = torch.device("cuda:0")
m, _, dr, _ = select_model(-1, dev)
x = torch.rand(32, 3, 256, 256, device=dev) * dr
lib = L.lib()
for _ in range(3): m(x)
torch.cuda.synchronize()
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for on in (0, 1):
        lib.esr_dbg_wino8(ctypes.c_int(on))
        m.enable_profiling(10)
        m(x); torch.cuda.synchronize(); m.collect_profile()
        for _ in range(10): m(x)
        torch.cuda.synchronize()
        prof = m.collect_profile(); m.disable_profiling()
        by = collections.defaultdict(lambda: [0.0, 0])
        for o in prof:
            by[o["kernel"]][0] += o["ms_sum"]; by[o["kernel"]][1] += o["passes"]
        tot = sum(v[0] for v in by.values()) / 10
        print(f"wino8={on}: {tot:.3f} ms/step  " + "  ".join(f"{k.replace('_f32_kernel','')}: {v[0]/v[1]:.4f}x{v[1]//10}" for k, v in sorted(by.items(), key=lambda kv: -kv[1][0])[:6]), flush=True)
