#!/usr/bin/env python3
"""per-op kernel times of one forward (HIP events around every op): per_op.py <registry id> <compute> [batch] [HxW] [--no-hilo-skip] [--no-tight-pitch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntire2022_esr_amd.registry import select_model
mid, comp = int(sys.argv[1]), sys.argv[2]
b = int(sys.argv[3]) if len(sys.argv) > 3 else 32
h, w = (int(v) for v in sys.argv[4].split("x")) if len(sys.argv) > 4 else (256, 256)
dev = torch.device("cuda:0")
m, name, dr, _ = select_model(mid, dev)
m.set_compute(comp)
if "--no-hilo-skip" in sys.argv:
    m.hilo_skip = False
if "--no-tight-pitch" in sys.argv:
    m.tight_pitch = False
x = torch.rand(b, 3, h, w, device=dev) * dr
for _ in range(3): m(x)
torch.cuda.synchronize()
m.enable_profiling(10)
for _ in range(11): m(x)
torch.cuda.synchronize()
prof = m.collect_profile(); m.disable_profiling()
tot = sum(o["ms_sum"] / o["passes"] for o in prof)
print(f"{name} {comp} {b}x{h}x{w}: {tot:.3f} ms of kernels per forward")
for o in prof:
    ms = o["ms_sum"] / o["passes"]
    gb = (o["read_bytes"] + o["write_bytes"]) / 1e9
    print(f"  {o['name'][:34]:34s} {o['kernel'][:52]:52s} {ms * 1000:8.1f} us  {gb / ms:6.0f} GB/s  {o['flops_exec'] / ms / 1e9:6.0f} TF/s")
