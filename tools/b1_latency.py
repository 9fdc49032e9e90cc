import sys, os, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntire2022_esr_amd.registry import select_model
m, name, dr, _ = select_model(-1, torch.device("cuda:0"))
x = torch.rand(1, 3, 256, 256, device="cuda:0")
for _ in range(5): m(x)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(200): y = m(x)
e.record(); torch.cuda.synchronize()
print("B=1 wall per forward: %.3f ms" % (s.elapsed_time(e) / 200))
import time
t0 = time.perf_counter()
for _ in range(200): y = m(x)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("host enqueue time per forward: %.3f ms" % ((t1 - t0) / 200 * 1e3))
m.enable_profiling(5)
for _ in range(5): m(x)
torch.cuda.synchronize()
tot = sum(o["ms_sum"] for o in m.collect_profile()) / 5
print("sum of per-kernel event times per forward: %.3f ms" % tot)
