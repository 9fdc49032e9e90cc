"""-m gpu: conv64m_kernel (csrc/esr_c64m.hip, round 6) -- the 64 -> 64 3x3 family on v_mfma_f32_32x32x16: RFDB's c{j}_r = lrelu(conv3x3(x) + x)
(rfdn_baseline/block.py:150-158) plain and with the next distillation 1x1 + LeakyReLU in its epilogue.  Checked against an fp64 reference on the
same 16-bit inputs and the blob's EFFECTIVE weights (tolerance: the one rounding of the stored value + fp32 accumulation noise), at shapes with
more 16 x 16 tiles than persistent blocks, ragged edges, several images, logical channel counts below the physical 64."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DT = {"bf16": torch.bfloat16, "f16": torch.float16}


def _ref_conv(x, weff, b, res_in, slope=0.05):
    """fp64 on the GPU: x [n, h, w, c] 16-bit values, weff [c, c, 3, 3] effective weights"""
    xd = x.permute(0, 3, 1, 2).double()
    conv = F.conv2d(xd, weff.double().to(DEV), b.double().to(DEV), padding=1)
    if res_in:
        conv = conv + xd
    return F.leaky_relu(conv, slope)


def _tol(ref, dt, extra=0.0):
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    return ref.abs() * eps * 1.01 + 3e-5 * max(1.0, float(ref.abs().max())) + extra


@pytest.mark.parametrize("compute", ["bf16", "f16"])
@pytest.mark.parametrize("n,c,hw,res_in", [(1, 64, (270, 480), True), (1, 50, (339, 510), True), (3, 64, (160, 144), False), (2, 50, (250, 203), True),
                                           (32, 50, (64, 64), True)])
def test_conv64m_plain_matches_fp64_reference(compute, n, c, hw, res_in):
    from ntire2022_esr_amd import ops, _lib as L
    from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16
    dt = DT[compute]
    g = torch.Generator().manual_seed(n + c + hw[0])
    x = F.pad(torch.randn(n, *hw, c, generator=g), (0, 64 - c)).to(dt).to(DEV)
    w, b = torch.randn(c, c, 3, 3, generator=g) * 0.1, torch.randn(c, generator=g)
    blob = pack_conv_s16(w, b, compute, cin_phys=64)
    weff, _ = unpack_conv_s16(blob, c, c, 3, compute, cin_phys=64)
    ref = _ref_conv(x[..., :c], weff, b, res_in)
    kw = dict(act=1, cin=c, packed=blob.to(DEV))
    if res_in:
        kw.update(res=x, res_mode=L.RES_PRE_ACT)
    for _ in range(3):                  # a race would not show every time
        y = ops.conv2d(x, w, b, **kw)
        got = y.permute(0, 3, 1, 2)[:, :c].double()
        bad = int(((got - ref).abs() > _tol(ref, dt)).sum())
        assert bad == 0, (bad, float((got - ref).abs().max()))
    assert torch.all(y[..., c:(c + 7) // 8 * 8] == 0)        # pad channels of the last 16-byte granule are zeros


@pytest.mark.parametrize("compute", ["bf16", "f16"])
@pytest.mark.parametrize("n,c,pc,hw,res_in", [(1, 50, 25, (339, 510), True), (8, 50, 25, (128, 128), True), (9, 64, 32, (100, 77), True), (5, 50, 25, (64, 250), False)])
def test_conv64m_post_matches_fp64_reference(compute, n, c, pc, hw, res_in):
    from ntire2022_esr_amd import ops, _lib as L
    from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16
    dt = DT[compute]
    g = torch.Generator().manual_seed(n * 10 + hw[0] + c)
    x = F.pad(torch.randn(n, *hw, c, generator=g), (0, 64 - c)).to(dt).to(DEV)
    w, b = torch.randn(c, c, 3, 3, generator=g) * 0.1, torch.randn(c, generator=g)
    wp, bp = torch.randn(pc, c, generator=g) * 0.2, torch.randn(pc, generator=g)
    blob = pack_conv_s16(w, b, compute, cin_phys=64)
    weff, _ = unpack_conv_s16(blob, c, c, 3, compute, cin_phys=64)
    ref = _ref_conv(x[..., :c], weff, b, res_in)
    kw = dict(act=1, cin=c, packed=blob.to(DEV), post_weight=wp, post_bias=bp, post_act=1)
    if res_in:
        kw.update(res=x, res_mode=L.RES_PRE_ACT)
    for _ in range(2):
        y, yp = ops.conv2d(x, w, b, **kw)
        got = y.permute(0, 3, 1, 2)[:, :c].double()
        bad = int(((got - ref).abs() > _tol(ref, dt)).sum())
        assert bad == 0, (bad, float((got - ref).abs().max()))
        # the post 1x1 + LeakyReLU.  bf16: it sees the fp32 activations as hi + lo (16 mantissa bits) and hi + lo weights; fp16: the rounded
        # activations (the stored y) and the weights' high parts
        if compute == "bf16":
            pin, pw = ref, wp.double().to(DEV)
            extra = 2e-4
        else:
            pin, pw = got, wp.to(dt).double().to(DEV)
            extra = 2e-4
        pref = F.leaky_relu(torch.einsum("oc,nchw->nohw", pw, pin) + bp.double().to(DEV)[None, :, None, None], 0.05)
        gp = yp.permute(0, 3, 1, 2)[:, :pc].double()
        badp = int(((gp - pref).abs() > _tol(pref, dt, extra)).sum())
        assert badp == 0, (badp, float((gp - pref).abs().max()))
    assert torch.all(yp[..., pc:] == 0) and torch.all(y[..., c:(c + 7) // 8 * 8] == 0)


def test_conv64m_is_the_kernel_that_runs():
    """the 64 -> 64 3x3s of an RFDN forward at >= 256 tiles take conv64m_kernel (named by the device symbol the library launched)"""
    from ntire2022_esr_amd.registry import select_model
    m = select_model(0, torch.device(DEV))[0]
    m.set_compute("bf16")
    x = (torch.rand(1, 3, 339, 510) * 255.0).to(DEV)
    m(x)
    m.enable_profiling(1)
    m(x)
    torch.cuda.synchronize()
    m.collect_profile()
    m(x)
    torch.cuda.synchronize()
    names = {o["kernel"] for o in m.collect_profile()}
    m.disable_profiling()
    assert any(k.startswith("conv64m_kernel<true, true>") for k in names), names
    assert any(k.startswith("conv64m_kernel<true, false>") for k in names), names
