"""-m gpu: wino8_tail_f32_kernel (csrc/esr_wino.hip, round 6, ABI v13) -- IMDBlock's conv4 -> cat -> conv1x1 -> + x (models/basicblock.py:263-265)
in one launch with conv4 as Winograd F(2x2, 3x3): against the three ATen ops in fp64 on the CPU, against imdb_tail_kernel (the direct form of
the same launch), and through the network (blocked input / residual / output) against the plan without it."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("n,hw,mid_act", [(4, (339, 510), 1), (9, (256, 256), 0), (2, (510, 1021), 1)])
def test_wino_tail_matches_fp64_reference_and_the_direct_form(n, hw, mid_act):
    """>= 8192 strips of 4 x 16 pixels (ragged right / bottom edges included): the Winograd form of the fused tail against
    conv4 -> act -> cat -> 1x1 -> + x in fp64, 2e-5 of the result's range (test_gpu_conv.py's bar for every fp32 kernel), and against
    imdb_tail_kernel on the same tensors (another rounding of conv4: a few 1e-6 of range)"""
    from ntire2022_esr_amd import ops
    from ntire2022_esr_amd import _lib as L
    g = torch.Generator().manual_seed(n + hw[0])
    x = torch.randn(n, 48, *hw, generator=g)
    cat = torch.randn(n, 64, *hw, generator=g)                    # channels [8, 56) are the 1x1's other inputs
    r = torch.randn(n, 64, *hw, generator=g)
    w3 = torch.randn(16, 48, 3, 3, generator=g) * 0.1
    b3 = torch.randn(16, generator=g)
    w1 = torch.randn(64, 64, 1, 1, generator=g) * 0.1
    b1 = torch.randn(64, generator=g)
    c4 = F.conv2d(x.double(), w3.double(), b3.double(), padding=1)
    if mid_act:
        c4 = F.leaky_relu(c4, 0.05)
    ref = F.conv2d(torch.cat([cat[:, 8:56].double(), c4], 1), w1.double(), b1.double()) + r.double()
    kw = dict(res=_nhwc(r).to(DEV), res_mode=1, slope=0.05, tail_weight=w1, tail_bias=b1, tail_cat=_nhwc(cat).to(DEV), tail_cat_coff=8,
              tail_mid_act=mid_act)
    xd = _nhwc(x).to(DEV)
    y_w = ops.conv2d(xd, w3, b3, wino=True, **kw)
    y_d = ops.conv2d(xd, w3, b3, **kw)
    assert not torch.equal(y_w, y_d)        # (ops.conv2d(wino=True) raises unless esr_wino_tail_supported; another arithmetic did run)
    scale = float(ref.abs().max())
    got = y_w.cpu().permute(0, 3, 1, 2).double()
    err = float((got - ref).abs().max()) / scale
    assert err < 2e-5, err
    d = float((y_w - y_d).abs().max()) / scale
    assert d < 1e-5, d
    assert bool(torch.isfinite(y_w).all())


def test_wino_tail_declines_other_shapes():
    """fewer than 8192 strips, a final activation, a post-activation residual: esr_wino_tail_supported says no and ops.conv2d(wino=True) raises;
    the descriptor without wino weights stays on imdb_tail_kernel"""
    from ntire2022_esr_amd import ops
    from ntire2022_esr_amd import _lib as L
    g = torch.Generator().manual_seed(1)
    n, hw = 1, (64, 64)
    x = torch.randn(n, *hw, 48, generator=g).to(DEV)
    cat = torch.randn(n, *hw, 48, generator=g).to(DEV)
    r = torch.randn(n, *hw, 64, generator=g).to(DEV)
    w3, b3 = torch.randn(16, 48, 3, 3, generator=g) * 0.1, torch.randn(16, generator=g)
    w1, b1 = torch.randn(64, 64, 1, 1, generator=g) * 0.1, torch.randn(64, generator=g)
    with pytest.raises(L.EsrError):
        ops.conv2d(x, w3, b3, wino=True, res=r, res_mode=1, tail_weight=w1, tail_bias=b1, tail_cat=cat)
    y = ops.conv2d(x, w3, b3, res=r, res_mode=1, tail_weight=w1, tail_bias=b1, tail_cat=cat)
    assert bool(torch.isfinite(y).all())


def test_imdn_batch_runs_the_wino_tail_and_agrees_with_the_direct_tail():
    """the network's own layouts (blocked conv3 remainder in, blocked block input as residual -- NHWC `fea` in block 0 --, blocked output): IMDN fp32
    at 8 x 256 x 256 with and without Plan.winograd_tail; the profiled forward names the kernel.  The x4 output agrees to 1e-5 of the data range
    (the bar of the network tests against the oracle is 2e-5)"""
    from ntire2022_esr_amd import engine
    from ntire2022_esr_amd.registry import select_model
    m, _, dr, _ = select_model(-1, torch.device(DEV))
    x = (torch.rand(8, 3, 256, 256, generator=torch.Generator().manual_seed(2)) * dr).to(DEV)
    try:
        engine.Plan.winograd_tail = False
        m._drop_plans()
        y0 = m(x).clone()
        engine.Plan.winograd_tail = True
        m._drop_plans()
        y1 = m(x).clone()
        m.enable_profiling(1)
        m(x)
        torch.cuda.synchronize()
        m.collect_profile()
        m(x)
        torch.cuda.synchronize()
        names = {o["kernel"] for o in m.collect_profile()}
        m.disable_profiling()
    finally:
        engine.Plan.winograd_tail = True
        m._drop_plans()
    assert any(k.startswith("wino8_tail_f32_kernel<true>") for k in names), names
    assert not any(k.startswith("imdb_tail_kernel") for k in names), names
    d = float((y1 - y0).abs().max())
    assert d < 1e-5 * dr, d
    assert not torch.equal(y0, y1)          # (another arithmetic did run)
    # images of the batch are independent: image 3 alone (imdb_tail_kernel: < 8192 strips) within the same bound
    y3 = m(x[3:4].contiguous())
    assert float((y3 - y1[3:4]).abs().max()) < 1e-5 * dr
