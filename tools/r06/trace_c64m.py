"""segment times of conv64m_kernel (C64M_TRACE builds): python tools/r06/trace_c64m.py  (ESR_HIP_LIB = a trace build)"""
import ctypes, sys, torch
sys.path.insert(0, ".")
from ntire2022_esr_amd.registry import select_model
from ntire2022_esr_amd import _lib as L
m = select_model(0, torch.device("cuda:0"))[0]
m.set_compute("bf16")
x = (torch.rand(32, 3, 256, 256) * 255.0).cuda()
for _ in range(3):
    m(x)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
lib = L.lib()
assert lib.esr_c64m_trace_read(buf) == 0
names = ["bookkeeping", "pair 0, k steps 0-12", "pair 0, rest", "pair 1", "wait + barrier", "DMA burst"]
for base, kn in ((0, "plain"), (16, "post")):
    tiles = buf[base + 6]
    if not tiles:
        continue
    tot = sum(buf[base + i] for i in range(6))
    print(f"{kn}: {tiles} tiles of block 0 wave 0 (last launch), {tot / tiles:.0f} ticks per tile")
    for i in range(6):
        print(f"   {names[i]:32s} {buf[base + i] / tiles:8.0f} ticks")
