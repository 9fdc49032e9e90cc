"""per-op times of one-image forwards (HIP events around every op, esr_run_ops_profiled): b1_profile.py <registry id> <compute> [H W]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntire2022_esr_amd.registry import select_model
mid, comp = int(sys.argv[1]), sys.argv[2]
H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (339, 510)
m, name, dr, _ = select_model(mid, torch.device("cuda:0"))
m.set_compute(comp)
if os.environ.get("NOFUSE"): m.fuse_esa_lowres = False
x = torch.rand(1, 3, H, W, device="cuda:0") * dr
for _ in range(5): m(x)
torch.cuda.synchronize()
m.enable_profiling(20)
for _ in range(20): m(x)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for o in m.collect_profile():
    a = agg[o["kernel"]]; a[0] += o["ms_sum"] / o["passes"]; a[1] += 1
tot = sum(v[0] for v in agg.values())
print(f"{name} {comp} 1x3x{H}x{W}: sum of kernel times {tot * 1e3:.1f} us in {sum(v[1] for v in agg.values())} launches")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"   {k:58s} {v[1]:3d} x {v[0] / v[1] * 1e3:7.1f} us = {v[0] * 1e3:7.1f} us  {v[0] / tot * 100:5.1f}%")
