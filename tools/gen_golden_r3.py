#!/usr/bin/env python3
"""Round-3 golden vectors: the stated-size parity evidence on MORE THAN ONE TEXTURE (authoring container only: imports the
reference; what travels is data).  Re-run:  python tools/gen_golden_r3.py

  tests/golden/multi/src_noise_f1.png, src_noise_f2.png   two seeded 256x256 noise tiles with natural-image spectra (1/f and 1/f^2
                                                          amplitude), generated here once and stored (sources, like utils/test.bmp)
  tests/golden/multi/multi_<k>.npz  (k = 0..7)            one DIV2K-val-shaped 339x510 LR image each.  HR (1356x2040, rebuilt by
                                                          the tests with hr_source(), never stored) = a small source image
                                                          mirror-tiled: the three mini_div2k HR images, test.bmp rolled /
                                                          flipped+transposed, the two noise tiles, test.bmp itself shifted.
                                                          LR = PIL-bicubic x4 reduction (stored).  Per network (imdn_baseline,
                                                          rfdn_baseline, team04_rlfn, team18_bsrn): the reference's fp32 SR strided
                                                          sample (::31), its mean, and the PSNR of its uint8 SR against HR by the
                                                          reference's own tensor2uint / calculate_psnr(border=4).
"""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import torch
from PIL import Image

import gen_golden as G

GOLD = G.GOLD
OUT = os.path.join(GOLD, "multi")
H, W = 339, 510
torch.set_num_threads(16)


def noise_tile(seed, power):
    """256x256x3 uint8 noise with amplitude spectrum 1/f^power per channel (correlated channels, like a photograph)"""
    rng = np.random.default_rng(seed)
    fy, fx = np.meshgrid(np.fft.fftfreq(256), np.fft.fftfreq(256), indexing="ij")
    f = np.sqrt(fx * fx + fy * fy)
    f[0, 0] = 1.0
    base = np.fft.ifft2(np.fft.fft2(rng.standard_normal((256, 256))) / f ** power).real
    out = []
    for c in range(3):
        ch = 0.8 * base + 0.2 * np.fft.ifft2(np.fft.fft2(rng.standard_normal((256, 256))) / f ** power).real
        ch = (ch - ch.mean()) / ch.std()
        out.append(np.clip(127.5 + 55.0 * ch, 0, 255))
    return np.round(np.stack(out, axis=2)).astype(np.uint8)


def tile_to(img, h4, w4):
    reps = (-(-h4 // img.shape[0]), -(-w4 // img.shape[1]))
    if reps[0] > 1 or reps[1] > 1:                    # mirror-tile: np.pad(symmetric) handles any size
        return np.pad(img, ((0, h4 - img.shape[0]), (0, w4 - img.shape[1]), (0, 0)), mode="symmetric")
    return img[:h4, :w4]


def hr_source(k, gold=GOLD):
    """the k-th HR image (4H x 4W uint8); tests/test_gpu_multi.py rebuilds it with this same function"""
    bmp = np.array(Image.open(os.path.join(gold, "test.bmp")).convert("RGB"))
    mini = lambda n: np.array(Image.open(os.path.join(gold, "mini_div2k", "DIV2K_valid_HR", n)).convert("RGB"))
    noise = lambda n: np.array(Image.open(os.path.join(gold, "multi", n)).convert("RGB"))
    src = [lambda: mini("0801.png"), lambda: mini("0802.png"), lambda: mini("0803.png"),
           lambda: np.roll(bmp, (97, 53), axis=(0, 1)), lambda: np.ascontiguousarray(bmp[:, ::-1].transpose(1, 0, 2)),
           lambda: noise("src_noise_f1.png"), lambda: noise("src_noise_f2.png"), lambda: np.roll(bmp, (128, 128), axis=(0, 1))][k]()
    return tile_to(src, 4 * H, 4 * W)


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, seed, power in (("src_noise_f1.png", 11, 1.0), ("src_noise_f2.png", 12, 2.0)):
        Image.fromarray(noise_tile(seed, power)).save(os.path.join(OUT, name))
    G._stub_cv2_torchvision()
    models = G.load_reference_models()
    import utils.utils_image as util
    names = ["imdn_baseline", "rfdn_baseline", "team04_rlfn", "team18_bsrn"]
    with torch.no_grad():
        for k in range(8):
            hr = hr_source(k)
            lr = np.array(Image.fromarray(hr).resize((W, H), Image.BICUBIC))
            rec = {"lr": lr}
            for name in names:
                m, _, dr = models[name]
                y = m(util.uint2tensor4(lr, dr))
                y8 = util.tensor2uint(y.clone(), dr)
                psnr = util.calculate_psnr(y8, util.modcrop(hr, 4), border=4)
                rec[f"{name}_sample"] = y[0, :, ::31, ::31].numpy().copy()
                rec[f"{name}_mean"] = np.float64(y.double().mean().item())
                rec[f"{name}_psnr"] = np.float64(psnr)
                rec[f"{name}_dr"] = np.float32(dr)
                print(k, name, "psnr", round(psnr, 4), flush=True)
            np.savez_compressed(os.path.join(OUT, f"multi_{k}.npz"), **rec)


if __name__ == "__main__":
    main()
