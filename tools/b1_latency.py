#!/usr/bin/env python3
"""one image per forward: wall time, host enqueue time, sum of kernel times, and the GPU time of the whole op list as ONE interval
(events around esr_run_ops): b1_latency.py [registry id] [compute] [HxW]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntire2022_esr_amd.registry import select_model
mid = int(sys.argv[1]) if len(sys.argv) > 1 else -1
comp = sys.argv[2] if len(sys.argv) > 2 else "f32"
h, w = (int(v) for v in sys.argv[3].split("x")) if len(sys.argv) > 3 else (256, 256)
m, name, dr, _ = select_model(mid, torch.device("cuda:0"))
m.set_compute(comp)
x = torch.rand(1, 3, h, w, device="cuda:0") * dr
for _ in range(5): m(x)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(200): y = m(x)
e.record(); torch.cuda.synchronize()
print(f"{name} {comp} 1x{h}x{w}: wall per forward (back to back): {s.elapsed_time(e) / 200:.3f} ms")
t0 = time.perf_counter()
for _ in range(200): y = m(x)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("host enqueue time per forward (back-pressured queue): %.3f ms" % ((t1 - t0) / 200 * 1e3))
# against an IDLE queue: one forward at a time, the device drained in between -- what the host alone costs
for graphs in (True, False):
    m.use_graphs = graphs
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(100):
        t0 = time.perf_counter(); y = m(x); tot += time.perf_counter() - t0
        torch.cuda.synchronize()
    s.record()
    for _ in range(200): y = m(x)
    e.record(); torch.cuda.synchronize()
    print(f"use_graphs={graphs}: host time per forward, idle queue: {tot / 100 * 1e6:.1f} us; wall per forward back to back: {s.elapsed_time(e) / 200:.3f} ms")
m.use_graphs = True
m.enable_profiling(5)
for _ in range(5): m(x)
torch.cuda.synchronize()
prof = m.collect_profile(); m.disable_profiling()
print("sum of per-kernel event times per forward: %.3f ms over %d ops" % (sum(o["ms_sum"] / o["passes"] for o in prof), len(prof)))
