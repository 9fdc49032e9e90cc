"""-m gpu: the 16-bit-operand convolution (bf16 / fp16 MFMA operands, fp32 accumulate, fp32 storage).
Kernel logic is checked against ATen fp32 on the SAME rounded operands (so only accumulation order differs,
tolerance 2e-5 * scale); the rounding itself is then bounded at network level as a PSNR shift."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLD, load_sd_torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DT = {"bf16": torch.bfloat16, "f16": torch.float16}


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("compute", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout,hw", [(64, 64, (16, 16)), (48, 64, (23, 37)), (48, 16, (17, 15)), (64, 48, (40, 56)),
                                         (56, 50, (20, 36)), (8, 16, (5, 3))])
def test_h16_conv_matches_rounded_operand_reference(compute, cin, cout, hw):
    from ntire2022_esr_amd import ops
    g = torch.Generator().manual_seed(cin + cout + hw[0])
    x = torch.randn(2, cin, *hw, generator=g)
    r = torch.randn(2, cout, *hw, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    b = torch.randn(cout, generator=g)
    xr, wr = x.to(DT[compute]).double(), w.to(DT[compute]).double()
    ref = F.leaky_relu(F.conv2d(xr, wr, b.double(), padding=1) + r.double(), 0.05).float()
    rp = F.pad(_nhwc(r), (0, (-cout) % 4))
    y = ops.conv2d(_nhwc(x).to(DEV), w, b, act=1, res=rp.to(DEV), res_mode=1, compute=compute)
    got = y.cpu().permute(0, 3, 1, 2)[:, :cout]
    assert float((got - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


def test_h16_rejects_unsupported():
    from ntire2022_esr_amd import _lib as L, ops
    with pytest.raises(L.EsrError):
        ops.conv2d(torch.randn(1, 8, 8, 16, device=DEV), torch.randn(16, 16, 1, 1), torch.randn(16),
                   packed=torch.zeros(4096, device=DEV), compute="bf16")            # 1x1 has no 16-bit path


@pytest.mark.parametrize("mid,compute,max_dpsnr", [(-1, "f16", 0.005), (-1, "bf16", 0.01), (0, "bf16", 0.01), (4, "bf16", 0.01),
                                                   (4, "f16", 0.005), (18, "f16", 0.005), (0, "f16", 0.005), (18, "bf16", 0.01)])
def test_network_psnr_shift(mid, compute, max_dpsnr):
    """PSNR of the 16-bit-operand network vs the fp32 network's PSNR on the natural image (64x64 bicubic LR of
    utils/test.bmp -> 256x256 vs the original): |dPSNR| budget per SURVEY 8c / BASELINE.md section 4: fp16 <= 0.005 dB,
    bf16 <= 0.01 dB (the stated-size cases against the REFERENCE's PSNR are in test_gpu_big.py)."""
    from PIL import Image
    from ntire2022_esr_amd import image_util as util
    from ntire2022_esr_amd.registry import select_model
    model, name, dr, _ = select_model(mid, torch.device(DEV))
    hr = np.array(Image.open(os.path.join(GOLD, "test.bmp")).convert("RGB"))
    lr = np.array(Image.fromarray(hr).resize((64, 64), Image.BICUBIC))
    x = util.uint2tensor4(lr, dr).to(DEV)
    p32 = util.calculate_psnr(util.tensor2uint(model(x), dr), hr, border=4)
    y32 = model(x).clone()
    model.set_compute(compute)
    y16 = model(x)
    p16 = util.calculate_psnr(util.tensor2uint(y16, dr), hr, border=4)
    rel = float((y16 - y32).abs().max()) / dr
    print(f"{name} {compute}: PSNR {p32:.4f} -> {p16:.4f} dB (d = {p16 - p32:+.4f}), max|dy|/data_range = {rel:.2e}")
    assert abs(p16 - p32) < max_dpsnr
    model.set_compute("f32")
    assert torch.equal(model(x), y32)                       # switching back restores the exact fp32 path
