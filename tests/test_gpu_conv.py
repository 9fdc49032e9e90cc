"""-m gpu: the HIP convolution (through the C ABI) against the ATen fp32 op it replaces,
evaluated on the CPU -- floating-point kernel, so a torch fp32 reference is the per-op
oracle; tolerance 2e-5 * scale (fp32 noise floor 2e-6..5e-6)."""
import numpy as np
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


def _check(y_gpu_nhwc, ref_nchw, scale=None):
    y = y_gpu_nhwc.cpu().permute(0, 3, 1, 2)[:, :ref_nchw.shape[1]]
    s = max(1.0, float(ref_nchw.abs().max())) if scale is None else scale
    err = float((y - ref_nchw).abs().max()) / s
    assert err < 2e-5, err


@pytest.mark.parametrize("cin,cout,k", [(64, 64, 3), (48, 64, 3), (48, 16, 3), (64, 48, 3), (64, 64, 1),
                                        (16, 32, 3), (8, 16, 1), (40, 48, 3), (128, 64, 1)])
@pytest.mark.parametrize("hw", [(16, 16), (17, 15), (40, 56), (5, 3), (1, 1), (33, 64)])
def test_conv_plain(cin, cout, k, hw):
    from ntire2022_esr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(cin * 1000 + cout * 10 + k + hw[0])
    x = torch.randn(2, cin, *hw, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * 0.1
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, w, b, padding=k // 2)
    y = ops.conv2d(_nhwc(x).to(dev), w, b)
    _check(y, ref)


@pytest.mark.parametrize("act", [0, 1, 2, 3])
@pytest.mark.parametrize("res_mode", [0, 1, 2])
def test_conv_epilogues(act, res_mode):
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
                   res=_nhwc(r).to(dev) if res_mode else None, res_mode=res_mode)
    _check(y, ref)


def test_conv_channel_slices_and_split_store():
    """torch.split / torch.cat become pitch+offset views: IMDBlock conv1 (basicblock.py:260)."""
    from ntire2022_esr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    big = torch.randn(2, 64, 19, 21, generator=g)          # read channels 16..63 of a 64-pitch buffer
    w = torch.randn(64, 48, 3, 3, generator=g) * 0.1
    b = torch.randn(64, generator=g)
    ref = F.leaky_relu(F.conv2d(big[:, 16:], w, b, padding=1), 0.05)
    cat = torch.full((2, 19, 21, 64), 7.0, device=dev)
    rem = torch.full((2, 19, 21, 48), 9.0, device=dev)
    ops.conv2d(_nhwc(big).to(dev), w, b, act=1, in_coff=16, cin=48, split=16,
               out=cat, out_coff=32, out1=rem, out1_coff=0)
    cat_c, rem_c = cat.cpu(), rem.cpu()
    assert torch.all(cat_c[..., :32] == 7.0) and torch.all(cat_c[..., 48:] == 7.0)   # untouched slices
    _check(cat_c[..., 32:48], ref[:, :16])
    _check(rem_c, ref[:, 16:])


def test_conv_head_nchw_and_tail_pixelshuffle():
    """head reads the NCHW network input (uint2tensor4 layout); tail fuses nn.PixelShuffle(4)."""
    from ntire2022_esr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 21, 30, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) * 0.2
    b = torch.randn(64, generator=g)
    y = ops.conv2d(x.to(dev), w, b, in_nchw=True)
    _check(y, F.conv2d(x, w, b, padding=1))
    t = torch.randn(2, 64, 21, 30, generator=g)
    w2 = torch.randn(48, 64, 3, 3, generator=g) * 0.05
    b2 = torch.randn(48, generator=g)
    ref = F.pixel_shuffle(F.conv2d(t, w2, b2, padding=1), 4)
    out = ops.conv2d(_nhwc(t).to(dev), w2, b2, shuffle_out=True).cpu()
    assert out.shape == ref.shape
    assert float((out - ref).abs().max()) / max(1.0, float(ref.abs().max())) < 2e-5


def test_conv_padded_concat_map():
    """cin_map: physical slots carrying logical channels / zero pads (padded concat buffers)."""
    from ntire2022_esr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 20, 18, 18, generator=g)
    w = torch.randn(32, 20, 1, 1, generator=g)
    b = torch.randn(32, generator=g)
    # physical layout: [0..9] <- ch 0..9, [10,11] pad, [12..21] <- ch 10..19, [22,23] pad (NaN-free garbage = 0)
    cmap = list(range(10)) + [-1, -1] + list(range(10, 20)) + [-1, -1]
    xp = torch.zeros(1, 24, 18, 18)
    xp[:, 0:10], xp[:, 12:22] = x[:, :10], x[:, 10:]
    y = ops.conv2d(_nhwc(xp).to(dev), w, b, cin=24, cin_map=cmap)
    _check(y, F.conv2d(x, w, b))


def test_bad_arguments_are_rejected():
    from ntire2022_esr_amd import _lib as L, ops
    dev = _dev()
    x = torch.randn(1, 8, 8, 62, device=dev)             # pitch not a multiple of 4
    with pytest.raises(L.EsrError):
        ops.conv2d(x, torch.randn(16, 62, 3, 3), torch.randn(16))
    with pytest.raises(L.EsrError):
        ops.conv2d(torch.randn(1, 8, 8, 64, device=dev), torch.randn(80, 64, 3, 3), torch.randn(80))  # cout > 64
    with pytest.raises(L.EsrError):
        ops.conv2d(torch.randn(1, 8, 8, 64), torch.randn(16, 64, 3, 3), torch.randn(16))            # CPU tensor


def _block_waves(n, h, w, cin, cout, k=3):
    """esr_conv_block_waves for an NHWC fp32 conv of that shape."""
    import ctypes
    from ntire2022_esr_amd import _lib as L
    d = L.ConvDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.ksize = n, h, w, cin, cout, k
    d.in_layout = L.NHWC
    return L.lib().esr_conv_block_waves(ctypes.byref(d))


@pytest.mark.parametrize("n,hw,cin,cout", [(5, (200, 136), 64, 64), (3, (250, 250), 48, 64), (9, (97, 130), 40, 48),
                                           (2, (509, 340), 64, 64)])
@pytest.mark.parametrize("act,res_mode", [(1, 0), (0, 2), (1, 1)])
def test_conv_8wave_tall_tiles(n, hw, cin, cout, act, res_mode):
    """Launches large enough for the 8-wave / 16x32-tile variant (ragged right and bottom edges, rows that are not a
    multiple of 32), all epilogue families, against ATen."""
    from ntire2022_esr_amd import ops
    dev = _dev()
    assert _block_waves(n, *hw, cin, cout) == 8 and _block_waves(1, 64, 64, cin, cout) == 4
    g = torch.Generator().manual_seed(n * 7 + hw[0] + act + 3 * res_mode)
    x = torch.randn(n, cin, *hw, generator=g)
    r = torch.randn(n, cout, *hw, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    b = torch.randn(cout, generator=g)
    c = F.conv2d(x, w, b, padding=1)
    ref = {0: ACTS[act](c), 1: ACTS[act](c + r), 2: ACTS[act](c) + r}[res_mode]
    y = ops.conv2d(_nhwc(x).to(dev), w, b, act=act, slope=0.05,
                   res=_nhwc(r).to(dev) if res_mode else None, res_mode=res_mode)
    _check(y, ref)


def test_conv_8wave_split_store_and_shuffle():
    """the 8-wave variant through the IMDBlock split store and the PixelShuffle(4) tail"""
    from ntire2022_esr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(23)
    n, h, wd = 6, 160, 144
    assert _block_waves(n, h, wd, 48, 64) == 8 and _block_waves(n, h, wd, 64, 48) == 8
    big = torch.randn(n, 64, h, wd, generator=g)
    w = torch.randn(64, 48, 3, 3, generator=g) * 0.1
    b = torch.randn(64, generator=g)
    ref = F.leaky_relu(F.conv2d(big[:, 16:], w, b, padding=1), 0.05)
    cat = torch.full((n, h, wd, 64), 7.0, device=dev)
    rem = torch.full((n, h, wd, 48), 9.0, device=dev)
    ops.conv2d(_nhwc(big).to(dev), w, b, act=1, in_coff=16, cin=48, split=16, out=cat, out_coff=32, out1=rem, out1_coff=0)
    cat_c, rem_c = cat.cpu(), rem.cpu()
    assert torch.all(cat_c[..., :32] == 7.0) and torch.all(cat_c[..., 48:] == 7.0)
    _check(cat_c[..., 32:48], ref[:, :16])
    _check(rem_c, ref[:, 16:])
    wt = torch.randn(48, 64, 3, 3, generator=g) * 0.1
    bt = torch.randn(48, generator=g)
    ref = F.pixel_shuffle(F.conv2d(big, wt, bt, padding=1), 4)
    y = ops.conv2d(_nhwc(big).to(dev), wt, bt, shuffle_out=True).cpu()
    assert float((y - ref).abs().max()) / max(1.0, float(ref.abs().max())) < 2e-5


@pytest.mark.parametrize("n,hw", [(1, (16, 16)), (2, (37, 45)), (3, (130, 96)), (9, (150, 171))])      # the last: 990 tiles on <= 512 blocks
@pytest.mark.parametrize("mid_act,res_mode,cat_c", [(0, 1, 48), (1, 0, 48), (0, 2, 32)])
def test_conv3x3_with_fused_1x1_tail(n, hw, mid_act, res_mode, cat_c):
    """esr_conv_desc.tail_*: IMDBlock's conv4 -> cat -> conv1x1 -> + x (basicblock.py:263-265) in one launch, against
    the three ATen ops; the concat part is a channel slice of a wider buffer.  cat_c = 48 with no / a pre-activation residual is
    the network's own shape and runs on imdb_tail_kernel (csrc/imdb_tail.inc), the rest on conv_f32_kernel's TAIL variant."""
    from ntire2022_esr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(n + hw[0] + 5 * mid_act + res_mode + cat_c)
    x = torch.randn(n, 48, *hw, generator=g)
    cat = torch.randn(n, 64, *hw, generator=g)                    # channels [8, 8 + cat_c) are the 1x1's other inputs
    r = torch.randn(n, 64, *hw, generator=g)
    w3 = torch.randn(16, 48, 3, 3, generator=g) * 0.1
    b3 = torch.randn(16, generator=g)
    w1 = torch.randn(64, cat_c + 16, 1, 1, generator=g) * 0.1
    b1 = torch.randn(64, generator=g)
    c4 = ACTS[mid_act](F.conv2d(x, w3, b3, padding=1))
    y1 = F.conv2d(torch.cat([cat[:, 8:8 + cat_c], c4], 1), w1, b1)
    ref = {0: y1, 1: y1 + r, 2: F.leaky_relu(y1, 0.05) + r}[res_mode]
    y = ops.conv2d(_nhwc(x).to(dev), w3, b3, act=1 if res_mode == 2 else 0, slope=0.05,
                   res=_nhwc(r).to(dev) if res_mode else None, res_mode=res_mode,
                   tail_weight=w1, tail_bias=b1, tail_cat=_nhwc(cat).to(dev), tail_cat_coff=8, tail_mid_act=mid_act)
    assert y.shape[-1] == 64
    _check(y, ref)


@pytest.mark.parametrize("n,h,wd", [(2, 19, 21), (6, 160, 144), (3, 37, 130)])
def test_split_store_into_blocked_out1_and_blocked_tail_input(n, h, wd):
    """esr_conv_desc.blocked8 (ABI v6): IMDBlock's conv3 stores its remaining 48 channels channel-blocked [N, 6, H, W, 8]
    (4-wave and 8-wave epilogues, edge tiles), pad planes stay untouched; the fused tail reads that layout (imdb_tail_kernel) and
    gives the NHWC result bit for bit."""
    from ntire2022_esr_amd import _lib as L
    from ntire2022_esr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(n + h)
    x = torch.randn(n, 48, h, wd, generator=g)
    w = torch.randn(64, 48, 3, 3, generator=g) * 0.1
    b = torch.randn(64, generator=g)
    ref = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.05)
    cat = torch.full((n, h, wd, 48), 7.0, device=dev)
    rem = torch.full((n, 7, h, wd, 8), 9.0, device=dev)             # 56-channel blocked tensor, channels 8..55 are written
    ops.conv2d(_nhwc(x).to(dev), w, b, act=1, split=16, out=cat, out_coff=32, out1=rem, out1_coff=8, blocked_out1=True)
    rem_c = rem.cpu()
    assert torch.all(rem_c[:, 0] == 9.0) and torch.all(cat.cpu()[..., :32] == 7.0)
    _check(cat.cpu()[..., 32:48], ref[:, :16])
    got = rem_c[:, 1:].permute(0, 2, 3, 1, 4).reshape(n, h, wd, 48)            # [N, 6, H, W, 8] -> NHWC
    _check(got, ref[:, 16:])
    with pytest.raises(L.EsrError):                                # blocked out1 without a split store
        ops.conv2d(_nhwc(x).to(dev), w, b, out1=rem, blocked_out1=True)
    # the fused tail on the blocked tensor == on its NHWC copy
    r = torch.randn(n, h, wd, 64, generator=g).to(dev)
    w3, b3 = torch.randn(16, 48, 3, 3, generator=g) * 0.1, torch.randn(16, generator=g)
    w1, b1 = torch.randn(64, 64, 1, 1, generator=g) * 0.1, torch.randn(64, generator=g)
    nhwc = got.contiguous().to(dev)
    kw = dict(res=r, res_mode=1, tail_weight=w1, tail_bias=b1, tail_cat=cat, tail_cat_coff=0)
    y_a = ops.conv2d(nhwc, w3, b3, **kw)
    y_b = ops.conv2d(rem, w3, b3, in_coff=8, cin=48, blocked_in=True, **kw)
    assert torch.equal(y_a, y_b)
    with pytest.raises(L.EsrError):                                # a blocked input anywhere else
        ops.conv2d(rem, w, b, in_coff=8, cin=48, blocked_in=True)


def test_fused_tail_argument_checks():
    import ctypes
    from ntire2022_esr_amd import _lib as L
    from ntire2022_esr_amd import ops
    dev = _dev()
    x = torch.randn(1, 8, 8, 48, device=dev)
    cat = torch.randn(1, 8, 8, 64, device=dev)
    w3, b3 = torch.randn(32, 48, 3, 3), torch.randn(32)            # 3x3 with more than 16 outputs: unsupported
    with pytest.raises(L.EsrError):
        ops.conv2d(x, w3, b3, tail_weight=torch.randn(64, 64), tail_bias=torch.randn(64), tail_cat=cat)
    w3, b3 = torch.randn(16, 48, 3, 3), torch.randn(16)
    with pytest.raises(L.EsrError):                                 # 1x1 with too few outputs for the 4-tile tail
        ops.conv2d(x, w3, b3, tail_weight=torch.randn(32, 64), tail_bias=torch.randn(32), tail_cat=cat)


def test_conv_randomised_shapes():
    """Seeded sweep over shapes / channel counts / epilogues, sized so that both block shapes (4-wave 16x16 tiles and
    8-wave 16x32 tiles) and every edge-tile combination occur; each case against ATen."""
    import random
    from ntire2022_esr_amd import ops
    dev = _dev()
    rng = random.Random(1234)
    seen = set()
    for case in range(36):
        n = rng.choice([1, 2, 3, 5, 8])
        h, w = rng.randint(3, 300), rng.randint(3, 300)
        if n * h * w > 400_000:
            h, w = h // 2 + 2, w // 2 + 2
        cin = rng.choice([8, 16, 24, 40, 48, 56, 64])
        cout = rng.choice([16, 24, 48, 52, 64])
        k = rng.choice([1, 3, 3, 3])
        act, res_mode = rng.choice([0, 1, 2, 3]), rng.choice([0, 0, 1, 2])
        seen.add(_block_waves(n, h, w, cin, cout, k))
        g = torch.Generator().manual_seed(case)
        x = torch.randn(n, cin, h, w, generator=g)
        r = torch.randn(n, cout, h, w, generator=g)
        wt = torch.randn(cout, cin, k, k, generator=g) * 0.1
        b = torch.randn(cout, generator=g)
        c = F.conv2d(x, wt, b, padding=k // 2)
        ref = {0: ACTS[act](c), 1: ACTS[act](c + r), 2: ACTS[act](c) + r}[res_mode]
        y = ops.conv2d(_nhwc(x).to(dev), wt, b, act=act, slope=0.05,
                       res=_nhwc(r).to(dev) if res_mode else None, res_mode=res_mode)
        y = y.cpu().permute(0, 3, 1, 2)[:, :cout]
        err = float((y - ref).abs().max()) / max(1.0, float(ref.abs().max()))
        assert err < 2e-5, (case, n, h, w, cin, cout, k, act, res_mode, err)
    assert seen == {4, 8}


@pytest.mark.parametrize("n,hw,c", [(1, (19, 23), 64), (2, (40, 56), 48), (4, (150, 140), 40), (6, (160, 144), 56)])
@pytest.mark.parametrize("act", [1, 0])
def test_conv_residual_is_the_input(n, hw, c, act):
    """act(conv(x) + x) with the residual view identical to the input view (RFDB's refinement convs): the kernel takes
    the residual from the staged input tile instead of loading it; both block shapes."""
    from ntire2022_esr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(n + hw[0] + c + act)
    x = torch.randn(n, c, *hw, generator=g)
    w = torch.randn(c, c, 3, 3, generator=g) * 0.1
    b = torch.randn(c, generator=g)
    ref = ACTS[act](F.conv2d(x, w, b, padding=1) + x)
    xg = torch.zeros(n, *hw, (c + 7) // 8 * 8)
    xg[..., :c] = x.permute(0, 2, 3, 1)
    xg = xg.to(dev)
    y = ops.conv2d(xg, w, b, act=act, slope=0.05, res=xg, res_mode=1, cin=c)
    _check(y, ref)


@pytest.mark.parametrize("n,hw", [(1, (20, 24)), (5, (200, 136)), (3, (250, 250))])
@pytest.mark.parametrize("c,pc,res_in", [(50, 25, True), (64, 32, False), (56, 20, True)])
def test_conv_with_post_1x1(n, hw, c, pc, res_in):
    """esr_conv_desc.post_*: r = lrelu(conv3x3(x) [+ x]) stored, and d = lrelu(W_p . r + b_p) stored by the same launch
    (RFDB: the next distillation conv rides in the producer's epilogue); small shapes take the two-launch fallback."""
    from ntire2022_esr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(n + hw[0] + c + pc)
    x = torch.randn(n, c, *hw, generator=g)
    w = torch.randn(c, c, 3, 3, generator=g) * 0.1
    b = torch.randn(c, generator=g)
    wp = torch.randn(pc, c, generator=g) * 0.2
    bp = torch.randn(pc, generator=g)
    r = F.leaky_relu(F.conv2d(x, w, b, padding=1) + (x if res_in else 0), 0.05)
    dref = F.leaky_relu(F.conv2d(r, wp[:, :, None, None], bp), 0.05)
    pitch = (c + 7) // 8 * 8
    xg = torch.zeros(n, *hw, pitch)
    xg[..., :c] = x.permute(0, 2, 3, 1)
    xg = xg.to(dev)
    out = torch.zeros(n, *hw, pitch, device=dev)
    y, yd = ops.conv2d(xg, w, b, act=1, slope=0.05, cin=c, out=out, res=xg if res_in else None, res_mode=1 if res_in else 0,
                       post_weight=wp, post_bias=bp, post_act=1)
    _check(y, r)
    _check(yd, dref)


def test_kernel_level_custom_ops_run_the_hip_kernels():
    """torch.ops.esr.conv2d / esa_apply (ops.py: torch.library operators over the C ABI) against ATen"""
    from ntire2022_esr_amd import ops  # noqa: F401
    dev = _dev()
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 64, 33, 47, generator=g)
    r = torch.randn(2, 64, 33, 47, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    b = torch.randn(64, generator=g)
    for wino in (False, True):
        y = torch.ops.esr.conv2d(_nhwc(x).to(dev), w, b, 1, 0.05, _nhwc(r).to(dev), 1, wino)
        _check(y, F.leaky_relu(F.conv2d(x, w, b, padding=1) + r, 0.05))
    y1 = torch.ops.esr.conv2d(_nhwc(x).to(dev), w[:, :, 1:2, 1:2].contiguous(), None, 0, 0.05, None, 0, True)     # 1x1: the direct kernel
    _check(y1, F.conv2d(x, w[:, :, 1:2, 1:2]))
    c, f, hw, lo = 50, 12, (40, 56), (5, 8)
    xx = torch.randn(2, c, *hw, generator=g) * 30
    c1 = torch.randn(2, f, *hw, generator=g)
    c3 = torch.randn(2, f, *lo, generator=g)
    wf, bf = torch.randn(f, f, generator=g) * 0.3, torch.randn(f, generator=g)
    w4, b4 = torch.randn(c, f, generator=g) * 0.3, torch.randn(c, generator=g)
    ref = xx * torch.sigmoid(F.conv2d(F.interpolate(c3, hw, mode="bilinear", align_corners=False) + F.conv2d(c1, wf[:, :, None, None], bf),
                                      w4[:, :, None, None], b4))

    def pad16(t, p):
        o = torch.zeros(t.shape[0], t.shape[2], t.shape[3], p)
        o[..., :t.shape[1]] = t.permute(0, 2, 3, 1)
        return o.to(dev)
    ye = torch.ops.esr.esa_apply(pad16(xx, 56), pad16(c1, 16), pad16(c3, 16), wf, bf, w4, b4)
    _check(ye[..., :c], ref)
