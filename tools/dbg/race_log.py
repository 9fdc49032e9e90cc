#!/usr/bin/env python3
"""multi-stream forwards against the checksum-log library (tools/dbg/race_variants.py log): which op's output differs FIRST from the
serial forward of the same image?   race_log.py <model> <compute> [rounds]"""
import ctypes, os, sys, collections
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import ntire2022_esr_amd._lib as L
L.SO_PATH = os.path.join(REPO, "tools", "abl", "libesr_r_log.so")
import numpy as np, torch
from test_gpu_big import _model
name, compute = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 100
m, dr = _model(name, compute)
DEV = "cuda:0"
g = torch.Generator().manual_seed(3)
shapes = [(85, 128), (96, 128), (128, 85), (74, 128), (85, 128), (87, 128), (128, 96), (85, 128), (85, 128), (64, 64)]
xs = [(torch.rand(1, 3, h, w, generator=g) * dr).to(DEV) for h, w in shapes]
want = [m(x).clone() for x in xs]
torch.cuda.synchronize()
streams = [torch.cuda.Stream(DEV) for _ in range(4)]
badf = []
for rnd in range(rounds):
    got = []
    for i, x in enumerate(xs):
        with torch.cuda.stream(streams[(i + rnd) % 4]):
            got.append(m(x))
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(got, want)):
        if not torch.equal(a, b): badf.append((rnd, i))
print(len(badf), "mismatching forwards of", rounds * 10)
lib = L.lib()
ncall = 10 + rounds * 10
buf = np.zeros((ncall, 64, 4), dtype=np.uint64)
lib.esr_dbg_log_read.restype = ctypes.c_uint
n = lib.esr_dbg_log_read(ctypes.c_void_p(buf.ctypes.data), ctypes.c_uint(ncall))
print("calls logged:", n)
plan = next(iter(m._plans.values())).plan
ops = plan.ops
def opname(k):
    o = ops[k]; return f"{k}:{o['kind']}:{o.get('w', '')}"
first = collections.Counter(); anydiff = 0
for rnd in range(rounds):
    for i in range(10):
        c = 10 + rnd * 10 + i
        if c >= n: continue
        d = np.argwhere(buf[c] != buf[i])
        if len(d):
            anydiff += 1
            k, slot = d[0]
            first[(int(k), int(slot))] += 1
            if anydiff <= 8: print(f"round {rnd} image {i} {shapes[i]}: first differing op {opname(int(k))} slot {slot}; all differing ops {sorted(set(int(a) for a, b in d))[:12]}  final mismatch: {(rnd, i) in badf}")
print("forwards with any differing checksum:", anydiff)
for (k, slot), cnt in sorted(first.items()): print(f"  first diff at op {opname(k)} slot {slot}: {cnt}x")
print("ops:", [opname(k) for k in range(len(ops))])
