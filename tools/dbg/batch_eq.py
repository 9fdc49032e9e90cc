"""model(x[B = 4]) == cat(model(x[i])) for a library given on the command line:  batch_eq.py <so | -> [model compute ...]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import ntire2022_esr_amd._lib as L
if sys.argv[1] != "-": L.SO_PATH = sys.argv[1]
import torch
from test_gpu_big import _model
for name, compute in (("rfdn_baseline", "bf16"), ("team18_bsrn", "f16"), ("team04_rlfn", "bf16"), ("rfdn_baseline", "f16")):
    m, dr = _model(name, compute)
    x = torch.rand(4, 3, 256, 256, generator=torch.Generator().manual_seed(3)).to("cuda:0") * dr
    y = m(x); ys = torch.cat([m(x[i:i + 1]) for i in range(4)])
    d = (y - ys).abs()
    print(name, compute, "equal" if torch.equal(y, ys) else f"DIFFERENT: {int((d > 0).sum())} values, max {float(d.max()):.3e}, images {sorted(set(d.nonzero()[:, 0].tolist()))}")
