"""hash of a model's outputs on fixed inputs (compare runs of different libraries / environment switches bit for bit):  out_hash.py <model> <compute>"""
import hashlib, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from test_gpu_big import _model
m, dr = _model(sys.argv[1], sys.argv[2])
g = torch.Generator().manual_seed(11)
h = hashlib.sha256()
for shape in ((4, 3, 256, 256), (1, 3, 339, 510), (2, 3, 270, 480), (1, 3, 85, 128)):
    y = m((torch.rand(*shape, generator=g) * dr).to("cuda:0"))
    h.update(y.cpu().numpy().tobytes())
print(sys.argv[1], sys.argv[2], h.hexdigest()[:16])
