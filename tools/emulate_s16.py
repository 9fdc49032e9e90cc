#!/usr/bin/env python3
"""CPU emulation of 16-bit activation STORAGE for RLFN / RFDN (design aid for the s16 kernels, not a test):
every tensor the engine would keep in a 16-bit NHWC buffer is rounded (RNE) where it is stored, weights of the
full-resolution convs are rounded, accumulation stays fp32, the ESA low-resolution branch stays fp32.  Prints the
PSNR shift against the fp32 graph on the stated-size fixtures, for a few storage policies."""
import os, sys
import numpy as np, torch, torch.nn.functional as F
from PIL import Image
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from safetensors.torch import load_file
from ntire2022_esr_amd import image_util as util
GOLD = os.path.join(REPO, "tests", "golden")
torch.set_num_threads(16)

class Q:
    def __init__(self, dt, policy): self.dt, self.policy = dt, policy
    def __call__(self, x, tag):
        if self.dt is None or tag in self.policy.get("keep32", ()): return x
        return x.to(self.dt).float()
    def w(self, sd, name):
        w = sd[name + ".weight"]
        return w if self.dt is None else w.to(self.dt).float()

def conv(q, sd, name, x, stride=1, padding=None, lowres=False):
    w = sd[name + ".weight"] if lowres else q.w(sd, name)
    if padding is None: padding = (w.shape[2] - 1) // 2
    return F.conv2d(x, w, sd[name + ".bias"], stride=stride, padding=padding)

def lrelu(x): return F.leaky_relu(x, 0.05)

def esa_tail(q, sd, p, x, c1_, c3):
    c3 = F.interpolate(c3, (x.size(2), x.size(3)), mode="bilinear", align_corners=False)
    cf = F.conv2d(c1_, sd[p + "conv_f.weight"], sd[p + "conv_f.bias"])
    c4 = F.conv2d(c3 + cf, sd[p + "conv4.weight"], sd[p + "conv4.bias"])
    return x * torch.sigmoid(c4)

def rlfn(q, sd, x):
    fea = q(conv(q, sd, "fea_conv", x, lowres=q.policy.get("head32", True)), "trunk")
    t = fea
    for i in range(1, 5):
        p = f"B{i}."
        o = q(lrelu(conv(q, sd, p + "c1_r", t)), "t")
        o = q(lrelu(conv(q, sd, p + "c2_r", o)), "t")
        u = lrelu(conv(q, sd, p + "c3_r", o)) + t
        if q.policy.get("fuse_c5"):
            # c5 and esa.conv1 as post-1x1s of the unrounded u inside c3_r's epilogue
            v32 = conv(q, sd, p + "c5", u)
            c1_ = q(conv(q, sd, p + "esa.conv1", v32), "c1")
            v = q(v32, "v")
        else:
            u = q(u, "u")
            v = q(conv(q, sd, p + "c5", u), "v")
            c1_ = q(conv(q, sd, p + "esa.conv1", v), "c1")
        c1 = conv(q, sd, p + "esa.conv2", c1_, stride=2, padding=0, lowres=True)
        vv = F.max_pool2d(c1, 7, 3)
        c3 = conv(q, sd, p + "esa.conv3", vv, lowres=True)
        t = q(esa_tail(q, sd, p + "esa.", v, c1_, c3), "trunk")
    out_lr = q(conv(q, sd, "LR_conv", t) + fea, "trunk")
    return F.pixel_shuffle(conv(q, sd, "upsampler.0", out_lr), 4)

def rfdn(q, sd, x):
    fea = q(conv(q, sd, "fea_conv", x, lowres=True), "trunk")
    outs, t = [], fea
    for i in range(1, 5):
        p = f"B{i}."
        d1 = q(lrelu(conv(q, sd, p + "c1_d", t)), "d")
        r1 = q(lrelu(conv(q, sd, p + "c1_r", t) + t), "r")
        d2 = q(lrelu(conv(q, sd, p + "c2_d", r1)), "d")
        r2 = q(lrelu(conv(q, sd, p + "c2_r", r1) + r1), "r")
        d3 = q(lrelu(conv(q, sd, p + "c3_d", r2)), "d")
        r3 = q(lrelu(conv(q, sd, p + "c3_r", r2) + r2), "r")
        r4 = q(lrelu(conv(q, sd, p + "c4", r3)), "d")
        v = q(conv(q, sd, p + "c5", torch.cat([d1, d2, d3, r4], 1)), "v")
        c1_ = q(conv(q, sd, p + "esa.conv1", v), "c1")
        c1 = conv(q, sd, p + "esa.conv2", c1_, stride=2, padding=0, lowres=True)
        vv = F.max_pool2d(c1, 7, 3)
        vv = F.relu(conv(q, sd, p + "esa.conv_max", vv, lowres=True))
        c3 = F.relu(conv(q, sd, p + "esa.conv3", vv, lowres=True))
        c3 = conv(q, sd, p + "esa.conv3_", c3, lowres=True)
        t = q(esa_tail(q, sd, p + "esa.", v, c1_, c3), "trunk")
        outs.append(t)
    ob = q(lrelu(conv(q, sd, "c.0", torch.cat(outs, 1))), "v")
    out_lr = q(conv(q, sd, "LR_conv", ob) + fea, "trunk")
    return F.pixel_shuffle(conv(q, sd, "upsampler.0", out_lr), 4)

def hr_of(h4, w4):
    img = np.array(Image.open(os.path.join(GOLD, "test.bmp")).convert("RGB"))
    return np.pad(img, ((0, h4 - img.shape[0]), (0, w4 - img.shape[1]), (0, 0)), mode="symmetric")

def main():
    nets = {"team04_rlfn": rlfn, "rfdn_baseline": rfdn}
    pols = {"all16": {}, "trunk32": {"keep32": ("trunk",)}, "fuse_c5": {"fuse_c5": True},
            "fuse_c5+trunk32": {"fuse_c5": True, "keep32": ("trunk",)}, "v32": {"keep32": ("v",)}, "u32": {"keep32": ("u",)},
            "t32": {"keep32": ("t", "r", "d")}}
    for name, fn in nets.items():
        sd = load_file(os.path.join(REPO, "weights", name + ".safetensors"))
        cases = []
        for h, w in ((256, 256), (339, 510)):
            g = np.load(os.path.join(GOLD, f"big_{name}_{h}x{w}.npz"))
            cases.append((g["lr"], hr_of(4 * h, 4 * w)))
        for i in range(3):
            cases.append((util.imread_uint(os.path.join(GOLD, "mini_div2k", "DIV2K_valid_LR", f"{801+i:04}x4.png")),
                          util.modcrop(util.imread_uint(os.path.join(GOLD, "mini_div2k", "DIV2K_valid_HR", f"{801+i:04}.png")), 4)))
        with torch.no_grad():
            base = [util.calculate_psnr(util.tensor2uint(fn(Q(None, {}), sd, util.uint2tensor4(lr, 255.0)), 255.0), hr, 4) for lr, hr in cases]
            for dt in (torch.bfloat16, torch.float16):
                for pn, pol in pols.items():
                    if name == "rfdn_baseline" and "c5" in pn: continue
                    d = [util.calculate_psnr(util.tensor2uint(fn(Q(dt, pol), sd, util.uint2tensor4(lr, 255.0)), 255.0), hr, 4) - b
                         for (lr, hr), b in zip(cases, base)]
                    print(f"{name:14s} {str(dt)[6:]:9s} {pn:16s} " + " ".join(f"{v:+.4f}" for v in d) + f"   mean|d| {np.mean(np.abs(d)):.4f}")

if __name__ == "__main__":
    main()
