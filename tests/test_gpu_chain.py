"""-m gpu: esr_conv_chain_s16 (ABI v11) -- RLFB's c1_r -> c2_r -> c3_r (+ x) -> c5 -> esa.conv1 (team04_rlfn.py:109-122) as ONE launch
(rlfb_chain_kernel: a layer-per-SIMD pipeline with the intermediate rows in LDS) against the three esr_conv2d_f32 launches it replaces.
The chain keeps their packed weights, operation order and roundings, so the comparison is BIT-EXACT; the separate launches are pinned
to ATen elsewhere (test_gpu_h16.py) and the whole network to the reference's goldens (test_gpu_esa_models.py, test_gpu_big.py)."""
import numpy as np
import pytest
import torch

from conftest import load_sd_torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DT = {"bf16": torch.bfloat16, "f16": torch.float16}


def _weights(seed, nf=46, mf=48, f=16, scale=0.06):
    g = torch.Generator().manual_seed(seed)
    ws = [torch.randn(mf, nf, 3, 3, generator=g) * scale, torch.randn(mf, mf, 3, 3, generator=g) * scale, torch.randn(nf, mf, 3, 3, generator=g) * scale]
    bs = [torch.randn(mf, generator=g) * 0.1, torch.randn(mf, generator=g) * 0.1, torch.randn(nf, generator=g) * 0.1]
    w5, b5 = torch.randn(nf, nf, generator=g) * 0.15, torch.randn(nf, generator=g) * 0.1
    w1, b1 = torch.randn(f, nf, generator=g) * 0.15, torch.randn(f, generator=g) * 0.1
    return ws, bs, w5, b5, w1, b1


def _separate(x, ws, bs, w5, b5, w1, b1, nf):
    """the chain as the three launches of the unfused plan (conv48r / conv48rp / conv_s16 kernels, whichever the shape takes)"""
    from ntire2022_esr_amd import _lib as L, ops
    t1 = ops.conv2d(x, ws[0], bs[0], act=L.ACT_LRELU, cin=nf)
    t1 = torch.nn.functional.pad(t1, (0, 48 - t1.shape[-1])) if t1.shape[-1] < 48 else t1
    t2 = ops.conv2d(t1, ws[1], bs[1], act=L.ACT_LRELU)
    t2 = torch.nn.functional.pad(t2, (0, 48 - t2.shape[-1])) if t2.shape[-1] < 48 else t2
    _, v, c1 = ops.conv2d(t2, ws[2], bs[2], act=L.ACT_LRELU, res=x, res_mode=L.RES_POST_ACT, post_weight=w5, post_bias=b5,
                          post2_weight=w1, post2_bias=b1, store_main=False)
    return v, c1


@pytest.fixture(params=["2", "3"])
def strip_groups(request):
    """both strip widths of the kernel (G = 2: 28 columns, G = 3: 44), forced through the research switch ESR_CHAIN_G"""
    import os
    old = os.environ.get("ESR_CHAIN_G")
    os.environ["ESR_CHAIN_G"] = request.param
    yield request.param
    if old is None:
        del os.environ["ESR_CHAIN_G"]
    else:
        os.environ["ESR_CHAIN_G"] = old


@pytest.mark.parametrize("compute", ["bf16", "f16"])
@pytest.mark.parametrize("n,h,w", [(1, 40, 56), (2, 23, 37), (1, 17, 15), (1, 64, 28), (3, 33, 29), (1, 5, 90), (1, 96, 61), (2, 50, 44), (1, 30, 133)])
def test_chain_equals_separate_launches(compute, n, h, w, strip_groups):
    """ragged strips (w not a multiple of the strip width), one-strip and one-segment images, several jobs per block: v and c1 bit for bit"""
    from ntire2022_esr_amd import ops
    dt = DT[compute]
    nf = 46
    ws, bs, w5, b5, w1, b1 = _weights(n * 1000 + h + w)
    g = torch.Generator().manual_seed(h * w)
    x = torch.zeros(n, h, w, 48)
    x[..., :nf] = torch.randn(n, h, w, nf, generator=g)
    x = x.to(dt).to(DEV)
    v, c1 = ops.conv_chain(x, ws, bs, w5, b5, w1, b1, cin=nf)
    rv, rc1 = _separate(x, ws, bs, w5, b5, w1, b1, nf)
    torch.cuda.synchronize()
    assert v.shape == (n, h, w, 48) and c1.shape == (n, h, w, 16)
    dv = (v[..., :nf].float() - rv[..., :nf].float()).abs().max().item()
    dc = (c1.float() - rc1.float()).abs().max().item()
    assert torch.equal(v[..., :nf], rv[..., :nf]) and torch.equal(c1, rc1), (dv, dc)
    assert torch.isfinite(v.float()).all() and float(v[..., nf:].float().abs().max()) == 0.0        # pad channels stay zero


@pytest.mark.parametrize("compute", ["bf16", "f16"])
def test_chain_matches_fp64_reference(compute, strip_groups):
    """... and directly against ATen in fp64 on the blobs' effective weights (one rounding per stored tensor: t1, t2, v, c1)"""
    import torch.nn.functional as F
    from ntire2022_esr_amd import ops
    from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16
    dt = DT[compute]
    nf, n, h, w = 46, 1, 37, 45
    ws, bs, w5, b5, w1, b1 = _weights(7)
    g = torch.Generator().manual_seed(3)
    x = torch.zeros(n, h, w, 48)
    x[..., :nf] = torch.randn(n, h, w, nf, generator=g)
    x = x.to(dt)
    v, c1 = ops.conv_chain(x.to(DEV), ws, bs, w5, b5, w1, b1, cin=nf)
    eff = []
    for wt, b in zip(ws, bs):
        cp = (wt.shape[1] + 15) // 16 * 16
        we, _ = unpack_conv_s16(pack_conv_s16(wt, b, compute, cin_phys=cp), wt.shape[1], wt.shape[0], 3, compute, cin_phys=cp)
        eff.append(we.double())
    xn = x[..., :nf].permute(0, 3, 1, 2).double()
    t1 = F.leaky_relu(F.conv2d(xn, eff[0], bs[0].double(), padding=1), 0.05).to(dt).double()
    t2 = F.leaky_relu(F.conv2d(t1, eff[1], bs[1].double(), padding=1), 0.05).to(dt).double()
    u = F.leaky_relu(F.conv2d(t2, eff[2], bs[2].double(), padding=1), 0.05) + xn
    vr = F.conv2d(u, w5.double()[:, :, None, None], b5.double())
    cr = F.conv2d(vr, w1.double()[:, :, None, None], b1.double())
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    gv = v[..., :nf].float().cpu().permute(0, 3, 1, 2).double()
    gc = c1.float().cpu().permute(0, 3, 1, 2).double()
    # the post 1x1s see u / v as hi + lo 16-bit parts and hi + lo weights: ~2^-16 (bf16) relative on top of the stored rounding
    for got, ref in ((gv, vr), (gc, cr)):
        tol = ref.abs() * eps * 1.05 + 4e-4 * float(ref.abs().max())
        assert bool(((got - ref).abs() <= tol).all()), float(((got - ref).abs() - tol).max())


@pytest.mark.parametrize("compute,shape", [("bf16", (1, 3, 64, 80)), ("bf16", (2, 3, 40, 56)), ("f16", (1, 3, 45, 61))])
def test_rlfn_with_chain_equals_without(compute, shape, strip_groups):
    """the whole network: fuse_chain on / off give bit-identical outputs (the checkpoint's weights)"""
    from ntire2022_esr_amd import RLFN_cut
    sd = load_sd_torch("team04_rlfn")
    m = RLFN_cut()
    m.load_state_dict(sd, strict=True)
    m = m.eval().to(DEV)
    m.set_compute(compute)
    x = (torch.rand(*shape, generator=torch.Generator().manual_seed(1)) * 255.0).to(DEV)
    m.fuse_chain = True
    y1 = m(x).clone()
    plan = m._plans[(shape[0], shape[1], shape[2], shape[3], torch.device(DEV))].plan
    assert sum(o["kind"] == "chain" for o in plan.ops) == 4
    m.fuse_chain = False
    y0 = m(x).clone()
    torch.cuda.synchronize()
    assert torch.equal(y0, y1), float((y0 - y1).abs().max())
