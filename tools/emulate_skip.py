#!/usr/bin/env python3
"""CPU emulation of the bf16 long skip `head -> (+) -> upsampler` of RLFN / RFDN on the eight 339x510 textures (design aid for the hi + lo
storage of `fea` / `out_lr`, LAB_NOTES 9.4): which of the three roundings on the skip -- fea as the residual, out_lr as stored, the
upsampler's weights -- costs the near-detail-free tile its 0.09 dB?   usage: emulate_skip.py [image indices]"""
import os, sys
import numpy as np, torch, torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "tools"))
from safetensors.torch import load_file
from ntire2022_esr_amd import image_util as util
from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16
import emulate_s16 as E
import test_gpu_multi as TM
torch.set_num_threads(32)
BF = torch.bfloat16


def split(x, parts):
    """x as a sum of `parts` bf16 numbers (parts = 0: fp32)"""
    if parts == 0: return x
    acc = torch.zeros_like(x)
    for _ in range(parts):
        acc = acc + (x - acc).to(BF).float()
    return acc


class Q(E.Q):
    def w(self, sd, name):
        w = sd[name + ".weight"]
        if self.dt is None: return w
        if name == "upsampler.0":
            m = self.policy.get("up_w", "diff")
            if m == "f32": return w
            if m == "hilo": return split(w, 2)
        if w.shape[2] == 3:                                    # what the packer stores: error-diffused taps
            return unpack_conv_s16(pack_conv_s16(w, sd[name + ".bias"], "bf16"), w.shape[1], w.shape[0], 3, "bf16")[0]
        return split(w, 2)                                     # 1x1: hi + lo
    def __call__(self, x, tag):
        if self.dt is None: return x
        if tag in ("fea_skip", "out_lr"): return split(x, self.policy.get(tag, 1))
        return x.to(self.dt).float()


def rlfn(q, sd, x):
    fea32 = E.conv(q, sd, "fea_conv", x, lowres=True)
    t = q(fea32, "trunk")
    for i in range(1, 5):
        p = f"B{i}."
        o = q(E.lrelu(E.conv(q, sd, p + "c1_r", t)), "t")
        o = q(E.lrelu(E.conv(q, sd, p + "c2_r", o)), "t")
        u = E.lrelu(E.conv(q, sd, p + "c3_r", o)) + t
        v32 = E.conv(q, sd, p + "c5", u)
        c1_ = q(E.conv(q, sd, p + "esa.conv1", v32), "c1")
        v = q(v32, "v")
        c1 = E.conv(q, sd, p + "esa.conv2", c1_, stride=2, padding=0, lowres=True)
        c3 = E.conv(q, sd, p + "esa.conv3", F.max_pool2d(c1, 7, 3), lowres=True)
        t = q(E.esa_tail(q, sd, p + "esa.", v, c1_, c3), "trunk")
    out_lr = q(E.conv(q, sd, "LR_conv", t) + q(fea32, "fea_skip"), "out_lr")
    return F.pixel_shuffle(E.conv(q, sd, "upsampler.0", out_lr), 4)


def rfdn(q, sd, x):
    fea32 = E.conv(q, sd, "fea_conv", x, lowres=True)
    outs, t = [], q(fea32, "trunk")
    for i in range(1, 5):
        p = f"B{i}."
        d1 = q(E.lrelu(E.conv(q, sd, p + "c1_d", t)), "d")
        r1 = q(E.lrelu(E.conv(q, sd, p + "c1_r", t) + t), "r")
        d2 = q(E.lrelu(E.conv(q, sd, p + "c2_d", r1)), "d")
        r2 = q(E.lrelu(E.conv(q, sd, p + "c2_r", r1) + r1), "r")
        d3 = q(E.lrelu(E.conv(q, sd, p + "c3_d", r2)), "d")
        r3 = q(E.lrelu(E.conv(q, sd, p + "c3_r", r2) + r2), "r")
        r4 = q(E.lrelu(E.conv(q, sd, p + "c4", r3)), "d")
        v = q(E.conv(q, sd, p + "c5", torch.cat([d1, d2, d3, r4], 1)), "v")
        c1_ = q(E.conv(q, sd, p + "esa.conv1", v), "c1")
        c1 = E.conv(q, sd, p + "esa.conv2", c1_, stride=2, padding=0, lowres=True)
        vv = F.relu(E.conv(q, sd, p + "esa.conv_max", F.max_pool2d(c1, 7, 3), lowres=True))
        c3 = E.conv(q, sd, p + "esa.conv3_", F.relu(E.conv(q, sd, p + "esa.conv3", vv, lowres=True)), lowres=True)
        t = q(E.esa_tail(q, sd, p + "esa.", v, c1_, c3), "trunk")
        outs.append(t)
    ob = q(E.lrelu(E.conv(q, sd, "c.0", torch.cat(outs, 1))), "v")
    out_lr = q(E.conv(q, sd, "LR_conv", ob) + q(fea32, "fea_skip"), "out_lr")
    return F.pixel_shuffle(E.conv(q, sd, "upsampler.0", out_lr), 4)


def main():
    ks = [int(a) for a in sys.argv[1:]] or [6, 0, 3]
    pols = {"as built": {}, "fea_skip 2": {"fea_skip": 2}, "out_lr 2": {"out_lr": 2}, "fea 2 + out_lr 2": {"fea_skip": 2, "out_lr": 2},
            "fea 2 + out_lr 2 + w hilo": {"fea_skip": 2, "out_lr": 2, "up_w": "hilo"}, "out_lr 2 + w hilo": {"out_lr": 2, "up_w": "hilo"},
            "fea 0 + out_lr 0 + w f32": {"fea_skip": 0, "out_lr": 0, "up_w": "f32"}, "w hilo": {"up_w": "hilo"}}
    for name, fn in (("team04_rlfn", rlfn), ("rfdn_baseline", rfdn)):
        sd = load_file(os.path.join(REPO, "weights", name + ".safetensors"))
        cases = []
        for k in ks:
            g = np.load(os.path.join(REPO, "tests", "golden", "multi", f"multi_{k}.npz"))
            cases.append((g["lr"], TM.hr_source(k), float(g[f"{name}_psnr"])))
        with torch.no_grad():
            base = [util.calculate_psnr(util.tensor2uint(fn(Q(None, {}), sd, util.uint2tensor4(lr, 255.0)), 255.0), hr, 4) for lr, hr, _ in cases]
            print(name, "fp32 graph vs reference PSNR", [round(b - c[2], 5) for b, c in zip(base, cases)])
            for pn, pol in pols.items():
                d = [util.calculate_psnr(util.tensor2uint(fn(Q(BF, pol), sd, util.uint2tensor4(lr, 255.0)), 255.0), hr, 4) - b
                     for (lr, hr, _), b in zip(cases, base)]
                print(f"{name:14s} {pn:28s} " + " ".join(f"{v:+.4f}" for v in d), flush=True)


if __name__ == "__main__":
    main()
