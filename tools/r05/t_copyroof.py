#!/usr/bin/env python3
"""the read + write rate a plain copy kernel reaches on this device by working set (esr_bw_probe): the practical memory roof of a streaming launch"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ntire2022_esr_amd import _lib as L
lib = L.lib()
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for mb in (16, 64, 128, 256, 512, 1024, 2048):
    n = mb << 20
    buf = torch.zeros(2 * n, dtype=torch.uint8, device=dev)
    gbs = ctypes.c_double(0)
    reps = max(1, (1 << 30) // n)
    for _ in range(2):
        L.check(lib.esr_bw_probe(ctypes.c_void_p(buf.data_ptr()), n, reps, st, ctypes.byref(gbs)), "esr_bw_probe")
    print(f"working set 2 x {mb:5d} MiB, {reps:3d} passes per launch: {gbs.value:8.1f} GB/s read + write")
    del buf
