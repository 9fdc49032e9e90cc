"""-m gpu: the Winograd F(2x2,3x3) fp32 convolution (wino_f32_kernel, through the C ABI: esr_conv_desc.wino_wpacked) against the
ATen fp32 op it replaces (models/basicblock.py:61-98 `conv`), evaluated on the CPU.  Tolerance 2e-5 * scale, the
same bar as the direct kernel's tests in test_gpu_conv.py."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ACTS = {0: lambda v: v, 1: lambda v: F.leaky_relu(v, 0.05), 2: F.relu, 3: F.gelu}


def _dev():
    assert torch.cuda.is_available(), "gpu-marked test needs an MI355X"
    return torch.device("cuda:0")


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _check(y_gpu_nhwc, ref_nchw, tol=2e-5):
    y = y_gpu_nhwc.cpu().permute(0, 3, 1, 2)[:, :ref_nchw.shape[1]]
    s = max(1.0, float(ref_nchw.abs().max()))
    err = float((y - ref_nchw).abs().max()) / s
    assert err < tol, err


@pytest.mark.parametrize("cin,cout", [(64, 64), (48, 64), (32, 32), (64, 48), (48, 16), (64, 50)])
@pytest.mark.parametrize("hw", [(16, 16), (17, 15), (40, 56), (5, 3), (1, 1), (33, 64), (31, 49)])
def test_wino_plain(cin, cout, hw):
    from ntire2022_esr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(cin * 1000 + cout * 10 + hw[0])
    x = torch.randn(2, cin, *hw, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, w, b, padding=1)
    y = ops.conv2d(_nhwc(x).to(dev), w, b, wino=True)
    _check(y, ref)


def test_wino_equals_direct_kernel_closely():
    """same input through both kernels: they differ by rounding only (different summation, ~1e-6)"""
    from ntire2022_esr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    x = _nhwc(torch.randn(3, 64, 50, 70, generator=g)).to(dev)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    b = torch.randn(64, generator=g)
    yd = ops.conv2d(x, w, b, act=1)
    yw = ops.conv2d(x, w, b, act=1, wino=True)
    assert float((yd - yw).abs().max()) < 2e-5 * max(1.0, float(yd.abs().max()))
    assert not torch.equal(yd, yw)          # it IS the other kernel


@pytest.mark.parametrize("act", [0, 1, 2, 3])
@pytest.mark.parametrize("res_mode", [0, 1, 2])
def test_wino_epilogues(act, res_mode):
    from ntire2022_esr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(7 + act * 3 + res_mode)
    x = torch.randn(1, 48, 23, 37, generator=g)
    r = torch.randn(1, 64, 23, 37, generator=g)
    w = torch.randn(64, 48, 3, 3, generator=g) * 0.1
    b = torch.randn(64, generator=g)
    c = F.conv2d(x, w, b, padding=1)
    ref = {0: ACTS[act](c), 1: ACTS[act](c + r), 2: ACTS[act](c) + r}[res_mode]
    y = ops.conv2d(_nhwc(x).to(dev), w, b, act=act, slope=0.05,
                   res=_nhwc(r).to(dev) if res_mode else None, res_mode=res_mode, wino=True)
    _check(y, ref)


def test_wino_channel_slices_and_split_store():
    """IMDBlock conv2 (basicblock.py:261): reads 48 channels of a wider buffer, 16 / 48 split store into a concat slice"""
    from ntire2022_esr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    big = torch.randn(2, 64, 19, 21, generator=g)
    w = torch.randn(64, 48, 3, 3, generator=g) * 0.1
    b = torch.randn(64, generator=g)
    ref = F.leaky_relu(F.conv2d(big[:, 16:], w, b, padding=1), 0.05)
    cat = torch.full((2, 19, 21, 64), 7.0, device=dev)
    rem = torch.full((2, 19, 21, 48), 9.0, device=dev)
    ops.conv2d(_nhwc(big).to(dev), w, b, act=1, in_coff=16, cin=48, split=16,
               out=cat, out_coff=32, out1=rem, out1_coff=0, wino=True)
    cat_c, rem_c = cat.cpu(), rem.cpu()
    assert torch.all(cat_c[..., :32] == 7.0) and torch.all(cat_c[..., 48:] == 7.0)
    _check(cat_c[..., 32:48], ref[:, :16])
    _check(rem_c, ref[:, 16:])


def test_wino_blocked_split_store():
    """IMDBlock conv3: the 48 remaining channels go to a channel-blocked [N, 6, H, W, 8] tensor (esr_conv_desc.blocked8)"""
    from ntire2022_esr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 48, 37, 29, generator=g)
    w = torch.randn(64, 48, 3, 3, generator=g) * 0.1
    b = torch.randn(64, generator=g)
    ref = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.05)
    cat = torch.zeros((2, 37, 29, 48), device=dev)
    rem = torch.full((2, 6, 37, 29, 8), 9.0, device=dev)
    ops.conv2d(_nhwc(x).to(dev), w, b, act=1, split=16, out=cat, out_coff=16, out1=rem, blocked_out1=True, wino=True)
    _check(cat.cpu()[..., 16:32], ref[:, :16])
    r = rem.cpu().permute(0, 1, 4, 2, 3).reshape(2, 48, 37, 29)          # [n][c/8][8][h][w]
    s = max(1.0, float(ref.abs().max()))
    assert float((r - ref[:, 16:]).abs().max()) / s < 2e-5


def test_wino_more_items_than_blocks_and_batch():
    """> 512 work items: persistent blocks walk several tiles, the staging ring crosses item boundaries"""
    from ntire2022_esr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(13)
    x = torch.randn(5, 64, 130, 150, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    b = torch.randn(64, generator=g)
    ref = F.conv2d(x, w, b, padding=1)
    y = ops.conv2d(_nhwc(x).to(dev), w, b, wino=True)
    _check(y, ref)


def test_wino_unsupported_shapes_fall_back():
    """odd chunk counts / 1x1 / tiny cin: esr_wino_supported says no and ops.conv2d(wino=True) refuses; the engine keeps conv_f32_kernel"""
    from ntire2022_esr_amd import _lib as L, ops
    dev = _dev()
    x = torch.randn(1, 9, 9, 56, device=dev)
    w = torch.randn(50, 50, 3, 3) * 0.1
    with pytest.raises(L.EsrError):
        ops.conv2d(x, w, None, cin=50, wino=True)            # 7 chunks
    x = torch.randn(1, 9, 9, 16, device=dev)
    with pytest.raises(L.EsrError):
        ops.conv2d(x, torch.randn(16, 16, 3, 3), None, wino=True)      # 2 chunks


def test_wino_uneven_split_between_block_classes():
    """>= 8 items per block: the first-dispatched half of the grid takes 9/16 of the items (two walks over two ranges); every
    output pixel is still written exactly once"""
    from ntire2022_esr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    x = torch.randn(12, 64, 200, 216, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    b = torch.randn(64, generator=g)
    ref = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.05)
    y = ops.conv2d(_nhwc(x).to(dev), w, b, act=1, wino=True)
    _check(y, ref)


@pytest.mark.parametrize("hw", [(21, 30), (32, 32), (17, 15)])
def test_wino_pixelshuffle_output(hw):
    """the network's last convolution: conv + nn.PixelShuffle(4) fused into the store (basicblock.py:446-449, 84-85)"""
    from ntire2022_esr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5 + hw[0])
    t = torch.randn(2, 64, *hw, generator=g)
    w2 = torch.randn(48, 64, 3, 3, generator=g) * 0.05
    b2 = torch.randn(48, generator=g)
    ref = F.pixel_shuffle(F.conv2d(t, w2, b2, padding=1), 4)
    out = ops.conv2d(_nhwc(t).to(dev), w2, b2, shuffle_out=True, wino=True).cpu()
    assert out.shape == ref.shape
    assert float((out - ref).abs().max()) / max(1.0, float(ref.abs().max())) < 2e-5
