"""-m gpu: the 16-bit PSNR budgets on ten MORE natural test images (tests/golden/crops.npz, generated from the real reference by
tools/gen_golden_r5.py; VERDICT r04 #6: RLFN bf16's evidence was four images with one at -0.0185 dB): HR = utils/test.bmp under rotations /
flips / odd rolls and its 2x box-reduced version mirror-tiled back (content at another scale), LR = its bicubic x4 reduction (64 x 64).
Asserted: fp32 -- every image within 0.002 dB of the REFERENCE's PSNR and the full-tensor sum within 2e-6 of the range per value; bf16 / fp16 --
the MEAN |dPSNR| within the budget (0.01 / 0.005 dB), the worst image reported and bounded at twice the budget (a 64 x 64 LR image has
200 000 SR samples: one uint8 flip in a thousand moves its PSNR by ~0.001 dB)."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import GOLD

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
IDS = {"imdn_baseline": -1, "rfdn_baseline": 0, "team04_rlfn": 4, "team18_bsrn": 18}
NCROPS = 10
_models = {}


def _model(name, compute):
    from ntire2022_esr_amd.registry import select_model
    if name not in _models:
        _models[name] = select_model(IDS[name], torch.device(DEV))
    m, _, dr, _ = _models[name]
    m.set_compute(compute)
    return m, dr


def hr_crop(k, bmp):
    """the k-th HR image, rebuilt exactly as tools/gen_golden_r5.py built it"""
    half = bmp.reshape(128, 2, 128, 2, 3).astype(np.float64).mean(axis=(1, 3))
    half = np.round(half).astype(np.uint8)
    half = np.pad(half, ((0, 128), (0, 128), (0, 0)), mode="symmetric")
    src = [bmp, np.rot90(bmp, 1), bmp[::-1], np.rot90(bmp, 3), np.roll(bmp, (37, 91), axis=(0, 1)), np.roll(bmp[:, ::-1], (131, 17), axis=(0, 1)),
           half, np.rot90(half, 1), np.roll(half, (64, 64), axis=(0, 1)), np.roll(bmp.transpose(1, 0, 2), (5, 201), axis=(0, 1))][k]
    return np.ascontiguousarray(src)


def _run(name, compute):
    from ntire2022_esr_amd import image_util as util
    g = np.load(os.path.join(GOLD, "crops.npz"))
    bmp = np.array(Image.open(os.path.join(GOLD, "test.bmp")).convert("RGB"))
    m, dr = _model(name, compute)
    out = []
    try:
        for k in range(NCROPS):
            y = m(util.uint2tensor4(g[f"lr_{k}"], dr).to(DEV))
            psnr = util.calculate_psnr(util.tensor2uint(y, dr), hr_crop(k, bmp), border=4)
            out.append((psnr - float(g[f"{name}_psnr_{k}"]), float(y.double().sum()) - float(g[f"{name}_sum_{k}"])))
    finally:
        m.set_compute("f32")
    return out, dr


@pytest.mark.parametrize("name", sorted(IDS))
def test_fp32_ten_crops(name):
    r, dr = _run(name, "f32")
    print(name, "f32 dPSNR", [round(d, 5) for d, _ in r])
    for d, ds in r:
        assert abs(d) <= 0.002, d
        assert abs(ds) <= 2e-6 * dr * 3 * 256 * 256, ds


@pytest.mark.parametrize("name,compute,budget", [("team04_rlfn", "bf16", 0.01), ("rfdn_baseline", "bf16", 0.01), ("imdn_baseline", "bf16", 0.01),
                                                 ("team18_bsrn", "f16", 0.005), ("team04_rlfn", "f16", 0.005)])
def test_16bit_mean_psnr_budget_over_ten_crops(name, compute, budget):
    r, _ = _run(name, compute)
    ds = [d for d, _ in r]
    mean_abs, worst = float(np.mean(np.abs(ds))), max(ds, key=abs)
    print(f"{name} {compute}: dPSNR per crop {[round(d, 4) for d in ds]}  mean |d| {mean_abs:.5f}  worst {worst:+.5f}")
    assert mean_abs <= budget, ds
    assert abs(worst) <= 2 * budget, ds
