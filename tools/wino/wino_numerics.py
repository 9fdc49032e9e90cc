"""CPU emulation of a Winograd F(2x2,3x3) fp32 convolution inside the IMDN graph: is the rounding inside the 2e-5 budget?

U = G g G^T is computed in fp64 and rounded once (as the host packer will); V = B^T d B, the 16 position GEMMs and
Y = A^T M A run in fp32 (torch CPU).  Compares the network output with the committed reference fixture and with the
fp64 evaluation of the direct graph."""
import sys, os
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from safetensors.torch import load_file
from oracle import torch_port as TP

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def wino_conv(x, w, b):
    n, c, h, wd = x.shape
    U = torch.einsum('ij,ocjk,lk->ocil', G, w.double(), G).float()          # [o][c][4][4], rounded once
    hp, wp = (h + 1) // 2 * 2, (wd + 1) // 2 * 2
    xp = F.pad(x, (1, 1 + wp - wd, 1, 1 + hp - h))
    # patches: [n, c, th, tw, 4, 4]
    p = xp.unfold(2, 4, 2).unfold(3, 4, 2)
    t = torch.einsum('ij,nchwjk->nchwik', Bt, p)                             # exact adds in fp32
    V = torch.einsum('nchwik,lk->nchwil', t, Bt)
    M = torch.einsum('ocil,nchwil->nohwil', U, V)                           # fp32 GEMMs (order differs from MFMA's, same magnitude)
    t2 = torch.einsum('ij,nohwjk->nohwik', At, M)
    Y = torch.einsum('nohwik,lk->nohwil', t2, At)                           # [n,o,th,tw,2,2]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(n, w.shape[0], hp, wp)[:, :, :h, :wd]
    return y + b.view(1, -1, 1, 1)


def imdn_wino(sd, x, nb=8):
    def c(name, t, wino):
        w, b = sd[name + ".weight"], sd[name + ".bias"]
        return wino_conv(t, w, b) if wino else F.conv2d(t, w, b, padding=w.shape[2] // 2)
    act = lambda t: F.leaky_relu(t, 0.05)
    head = c("model.0", x, False)
    t = head
    for i in range(nb):
        p = f"model.1.sub.{i}."
        d1, r1 = torch.split(act(c(p + "conv1.0", t, True)), (16, 48), dim=1)
        d2, r2 = torch.split(act(c(p + "conv2.0", r1, True)), (16, 48), dim=1)
        d3, r3 = torch.split(act(c(p + "conv3.0", r2, True)), (16, 48), dim=1)
        d4 = c(p + "conv4", r3, False)
        t = t + c(p + "conv1x1", torch.cat((d1, d2, d3, d4), 1), False)
    t = head + c(f"model.1.sub.{nb}", t, True)
    return F.pixel_shuffle(c("model.2", t, WINO_LAST), 4)


WINO_LAST = True
root = os.path.join(os.path.dirname(__file__), "..", "..")
sd = load_file(os.path.join(root, "weights", "imdn_baseline.safetensors"))
sd64 = {k: v.double() for k, v in sd.items()}
torch.manual_seed(0)
for name, x in (("rand 2x3x64x64", torch.rand(2, 3, 64, 64)), ("rand 1x3x37x51", torch.rand(1, 3, 37, 51))):
    ref64 = TP.imdn(sd64, x.double())
    d = TP.imdn(sd, x)
    wv = imdn_wino(sd, x)
    print(f"{name}: direct fp32 vs fp64 {float((d - ref64).abs().max()):.2e}   winograd fp32 vs fp64 {float((wv - ref64).abs().max()):.2e}"
          f"   wino vs direct {float((wv - d).abs().max()):.2e}")
g = np.load(os.path.join(root, "tests", "golden", "big_imdn_baseline_256x256.npz"))
print(sorted(g.files))
