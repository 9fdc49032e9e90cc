"""Rough timing of the forward at a few batch sizes (development helper, not the bench)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntire2022_esr_amd.registry import select_model
GF = {-1: 116.86, 0: 54.07, 4: 39.32, 18: 18.86}
ids = [int(a) for a in sys.argv[1:]] or [-1]
for mid in ids:
    m, name, dr, _ = select_model(mid, torch.device("cuda:0"))
    for B in [1, 8, 32]:
        x = torch.rand(B, 3, 256, 256, device="cuda:0") * dr
        for _ in range(3): y = m(x)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 10
        s.record()
        for _ in range(it): y = m(x)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / it
        print(f"{name:18s} B={B:3d} {ms:8.3f} ms/fwd {B/ms*1e3:8.1f} img/s {GF[mid]*B/ms:7.1f} TFLOP/s ({GF[mid]*B/ms/157.3*100:.1f}% fp32 MFMA peak)")
    if len(ids) > 1 or os.environ.get("PROF"):
        m.enable_profiling(3)
        for _ in range(3): m(x)
        torch.cuda.synchronize()
        import collections
        agg = collections.defaultdict(lambda: [0.0, 0])
        for o in m.collect_profile():
            agg[o["kernel"]][0] += o["ms_sum"]; agg[o["kernel"]][1] += o["passes"]
        tot = sum(v[0] for v in agg.values())
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            print(f"      {k:44s} {v[1]//3:3d} launches/fwd {v[0]/3:8.3f} ms/fwd {v[0]/tot*100:5.1f}%")
        m.disable_profiling()
