#!/usr/bin/env python3
"""repeat tests/test_gpu_big.py::test_forwards_on_several_streams_equal_serial and report WHERE serial and overlapped forwards differ
usage: streams_race.py <model> <compute> [rounds] [so path | -] [nolowres] [big]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import ntire2022_esr_amd._lib as L
if len(sys.argv) > 4 and sys.argv[4] != "-":
    L.SO_PATH = sys.argv[4]
import torch
from test_gpu_big import _model
name, compute = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 30
m, dr = _model(name, compute)
if "nolowres" in sys.argv:
    m.fuse_esa_lowres = False
DEV = "cuda:0"
g = torch.Generator().manual_seed(3)
shapes = [(85, 128), (96, 128), (128, 85), (74, 128), (85, 128), (87, 128), (128, 96), (85, 128), (85, 128), (64, 64)]
if "big" in sys.argv:          # DIV2K-val-shaped images: >= 256 tiles of 16 x 16, the launches take conv48r / conv48rp / conv64r_kernel
    shapes = [(339, 510), (339, 510), (384, 510), (339, 510), (510, 339), (294, 510), (339, 510), (345, 510), (510, 384), (339, 510)]
xs = [(torch.rand(1, 3, h, w, generator=g) * dr).to(DEV) for h, w in shapes]
want = [m(x).clone() for x in xs]
torch.cuda.synchronize()
again = [m(x).clone() for x in xs]
torch.cuda.synchronize()
print("serial repeat equal:", all(torch.equal(a, b) for a, b in zip(again, want)))
streams = [torch.cuda.Stream(DEV) for _ in range(4)]
bad = 0
for rnd in range(rounds):
    got = []
    for i, x in enumerate(xs):
        with torch.cuda.stream(streams[(i + rnd) % 4]):
            got.append(m(x))
        if "sync" in sys.argv:
            torch.cuda.synchronize()        # same stream assignment (same per-stream workspace history), no overlap
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(got, want)):
        if not torch.equal(a, b):
            d = (a - b).abs()
            nz = d.nonzero()
            bad += 1
            print(f"round {rnd} image {i} shape {tuple(a.shape)}: {int((d > 0).sum())} differing values, max {float(d.max()):.3e}, "
                  f"rows {int(nz[:, 2].min())}..{int(nz[:, 2].max())} cols {int(nz[:, 3].min())}..{int(nz[:, 3].max())} chans {sorted(set(nz[:, 1].tolist()))}")
print(f"{bad} mismatching forwards in {rounds} rounds x {len(xs)} images")
try:
    f = L.lib().esr_dbg_pool_bad
    f.restype = __import__("ctypes").c_uint
    print("s2pool16 vs fp32-MFMA s2pool on the same input: differing elements", f(0), "max index", f(1), "launches compared", f(2))
except AttributeError:
    pass
try:
    f = L.lib().esr_dbg_lds_bad
    f.restype = __import__("ctypes").c_uint
    print("corrupted LDS image elements seen by esa_apply blocks:", f())
except AttributeError:
    pass
