"""Per-op average launch time of one model/compute at batch 32 (HIP events around every launch): python tools/dbg/per_op.py [model compute]"""
import collections, os, sys
import torch
sys.path.insert(0, os.getcwd())
import bench
name, compute = (sys.argv[1:3] + ["imdn_baseline", "f32"])[:2] if len(sys.argv) >= 3 else ("imdn_baseline", "f32")
dev = torch.device("cuda", 0)
m, _ = bench.build_model(name, dev, compute)
x = (torch.rand(32, 3, 256, 256) * bench.MODELS[name][1]).to(dev)
with torch.no_grad():
    for _ in range(3):
        m(x)
    m.enable_profiling(10)
    m(x); torch.cuda.synchronize(); m.collect_profile()
    for _ in range(10):
        m(x)
    torch.cuda.synchronize()
acc = collections.OrderedDict()
for o in m.collect_profile():
    key = o["name"].split(".")[-2] + "." + o["name"].split(".")[-1] if "sub" in o["name"] else o["name"]
    a = acc.setdefault((key, o["kernel"]), [0.0, 0])
    a[0] += o["ms_sum"]; a[1] += o["passes"]
for (k, kern), (ms, n) in acc.items():
    print(f"{k:24s} {kern:48s} {ms / n:.4f} ms x {n}")
