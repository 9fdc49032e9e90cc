#!/usr/bin/env python3
"""Round-5 golden vectors (authoring container only: imports the reference; what travels is data).  Re-run:  python tools/gen_golden_r5.py

  tests/golden/big_checks.npz     FULL-TENSOR checksums of the reference's fp32 SR output for every big_<model>_<H>x<W>.npz case (the
                                  strided ::9 sample covers 1/81 of the output): per case the fp64 sum, sum of squares and the sums of a 4 x 4 grid of
                                  tiles per channel -- a localised defect anywhere in the 1356 x 2040 image moves one of them.
  tests/golden/crops.npz          ten more natural test images for the 16-bit PSNR budgets: HR = utils/test.bmp (256 x 256) under rotations / flips /
                                  odd rolls, and its 2x box-reduced version mirror-tiled back to 256 x 256 (content at another scale); LR = PIL-bicubic
                                  x4 reduction (64 x 64, stored); per network the reference's PSNR (tensor2uint + calculate_psnr(border=4)) and the fp64
                                  sum of its fp32 SR.  The tests rebuild HR with hr_crop() below.
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
import gen_golden_r2 as G2

GOLD = G.GOLD
torch.set_num_threads(16)
NAMES = ["imdn_baseline", "rfdn_baseline", "team04_rlfn", "team18_bsrn"]
NCROPS = 10


def tile_sums(y):
    """y: [C, H, W] float64 -> [C, 4, 4] sums over a 4 x 4 grid of (nearly) equal tiles"""
    c, h, w = y.shape
    ys = [round(i * h / 4) for i in range(5)]
    xs = [round(i * w / 4) for i in range(5)]
    return np.array([[[y[k, ys[i]:ys[i + 1], xs[j]:xs[j + 1]].sum() for j in range(4)] for i in range(4)] for k in range(c)])


def hr_crop(k, bmp):
    """the k-th 256 x 256 HR image; tests/test_gpu_crops.py rebuilds it with this same function (copied there: tools/ does not travel as a module)"""
    half = bmp.reshape(128, 2, 128, 2, 3).astype(np.float64).mean(axis=(1, 3))
    half = np.round(half).astype(np.uint8)
    half = np.pad(half, ((0, 128), (0, 128), (0, 0)), mode="symmetric")
    src = [bmp, np.rot90(bmp, 1), bmp[::-1], np.rot90(bmp, 3), np.roll(bmp, (37, 91), axis=(0, 1)), np.roll(bmp[:, ::-1], (131, 17), axis=(0, 1)),
           half, np.rot90(half, 1), np.roll(half, (64, 64), axis=(0, 1)), np.roll(bmp.transpose(1, 0, 2), (5, 201), axis=(0, 1))][k]
    return np.ascontiguousarray(src)


def main():
    G._stub_cv2_torchvision()
    models = G.load_reference_models()
    import utils.utils_image as util
    bmp = np.array(Image.open(os.path.join(GOLD, "test.bmp")).convert("RGB"))
    cases = [("imdn_baseline", 339, 510), ("rfdn_baseline", 339, 510), ("team04_rlfn", 339, 510), ("team18_bsrn", 339, 510),
             ("team18_bsrn", 270, 480), ("team04_rlfn", 270, 480),
             ("imdn_baseline", 256, 256), ("rfdn_baseline", 256, 256), ("team04_rlfn", 256, 256), ("team18_bsrn", 256, 256)]
    out = {}
    with torch.no_grad():
        for name, h, w in cases:
            m, _, dr = models[name]
            g = np.load(os.path.join(GOLD, f"big_{name}_{h}x{w}.npz"))
            y = m(util.uint2tensor4(g["lr"], dr))[0].double().numpy()
            assert np.abs(y[:, ::9, ::9] - g["sr_sample"]).max() < 1e-4 * dr          # the same forward the stored sample came from
            key = f"{name}_{h}x{w}"
            out[key + "_sum"] = np.float64(y.sum())
            out[key + "_sumsq"] = np.float64((y * y).sum())
            out[key + "_tiles"] = tile_sums(y)
            print(key, out[key + "_sum"], out[key + "_sumsq"])
    np.savez_compressed(os.path.join(GOLD, "big_checks.npz"), **out)

    rec = {}
    with torch.no_grad():
        for k in range(NCROPS):
            hr = hr_crop(k, bmp)
            lr = np.array(Image.fromarray(hr).resize((64, 64), Image.BICUBIC))
            rec[f"lr_{k}"] = lr
            for name in NAMES:
                m, _, dr = models[name]
                y = m(util.uint2tensor4(lr, dr))
                y8 = util.tensor2uint(y.clone(), dr)
                rec[f"{name}_psnr_{k}"] = np.float64(util.calculate_psnr(y8, hr, border=4))
                rec[f"{name}_sum_{k}"] = np.float64(y.double().sum().item())
            print(k, [round(float(rec[f"{n}_psnr_{k}"]), 3) for n in NAMES])
    np.savez_compressed(os.path.join(GOLD, "crops.npz"), **rec)


if __name__ == "__main__":
    main()
