import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntire2022_esr_amd import _lib as L
from ntire2022_esr_amd.registry import select_model
m, name, dr, _ = select_model(-1, torch.device("cuda:0"))
x = torch.rand(32, 3, 256, 256, device="cuda:0")
lib = L.lib()
for slot, spread in [(0, 0), (5, 1), (3, 1), (8, 1), (5, 0), (10, 2), (5, 2), (2, 0), (16, 1)]:
    lib.esr_set_tuning(slot, spread)
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): m(x)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"stagger slot={slot:2d} spread={spread}: {ms:7.3f} ms/fwd  {32/ms*1e3:7.1f} img/s")
