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
    assert any(k.startswith("conv64m_kernel<true, true, false, 4, false>") for k in names), names          # c1_r / c2_r with the next distillation 1x1
    assert any(k.startswith("conv64m_kernel<true, false, false, 4, false>") for k in names), names         # c3_r
    assert any(k.startswith("conv64m_kernel<true, false, true, 4, false>") for k in names), names          # LR_conv on hi + lo pairs


def _hilo(w, dt):
    """the 16-bit hi + lo value of an fp32 weight (what the 1x1 images hold)"""
    hi = w.to(dt).float()
    return (hi + (w - hi).to(dt).float()).double()


@pytest.mark.parametrize("compute", ["bf16", "f16"])
@pytest.mark.parametrize("n,hw,nf,dc,f", [(1, (339, 510), 50, 25, 16), (4, (128, 144), 50, 25, 16), (2, (250, 203), 64, 32, 16), (32, (64, 64), 50, 25, 12)])
def test_rfdb_tail_matches_fp64_reference(compute, n, hw, nf, dc, f):
    """rfdb_tail_kernel (ABI v12, esr_conv_desc.tail_* in 16-bit storage): r4 = round(lrelu(c4(r3))), v = c5 . [d1 d2 d3 r4], c1 = conv1 . v in one
    launch (rfdn_baseline/block.py:161-164, :117), against fp64 on the same 16-bit inputs and the blobs' effective weights."""
    from ntire2022_esr_amd import _lib as L
    from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16, pack_tail_s16, pack_post_s16
    dt = DT[compute]
    g = torch.Generator().manual_seed(n + hw[0] + nf)
    r3 = F.pad(torch.randn(n, *hw, nf, generator=g), (0, 64 - nf)).to(dt).to(DEV)
    ds = F.pad(torch.randn(3, n, *hw, dc, generator=g), (0, 32 - dc)).to(dt).to(DEV)
    w4, b4 = torch.randn(dc, nf, 3, 3, generator=g) * 0.1, torch.randn(dc, generator=g)
    w5, b5 = torch.randn(nf, 4 * dc, generator=g) * 0.15, torch.randn(nf, generator=g)
    wc, bc = torch.randn(f, nf, generator=g) * 0.2, torch.randn(f, generator=g)
    blob4 = pack_conv_s16(w4, b4, compute, cin_phys=64)
    w4e, _ = unpack_conv_s16(blob4, nf, dc, 3, compute, cin_phys=64)
    blob5 = pack_tail_s16(w5, b5, 3, dc, dc, compute).to(DEV)
    blobc = pack_post_s16(wc, bc, compute).to(DEV)
    blob4 = blob4.to(DEV)
    v = torch.full((n, *hw, 64), 7.0, dtype=dt, device=DEV)
    c1 = torch.full((n, *hw, 16), 7.0, dtype=dt, device=DEV)
    d = L.ConvDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.ksize = n, hw[0], hw[1], nf, dc, 3
    d.in_layout = d.out_layout = L.NHWC
    d.storage = d.compute = L.STORE[compute]
    d.act, d.slope = L.ACT_NONE, 0.05
    d.inp = L.View(ctypes.c_void_p(r3.data_ptr()), 64, 0)
    d.out0 = L.View(ctypes.c_void_p(v.data_ptr()), 64, 0)
    d.wpacked = ctypes.c_void_p(blob4.data_ptr())
    d.tail_wpacked = ctypes.c_void_p(blob5.data_ptr())
    d.tail_cat = L.View(ctypes.c_void_p(ds.data_ptr()), 32, 0)
    d.tail_cat_c, d.tail_cout, d.tail_mid_act = 96, nf, L.ACT_LRELU
    d.tail_seg_stride16 = ds[0].numel() * 2 // 16
    d.post_wpacked = ctypes.c_void_p(blobc.data_ptr())
    d.post_out = L.View(ctypes.c_void_p(c1.data_ptr()), 16, 0)
    d.post_cout, d.post_act = f, L.ACT_NONE
    assert L.lib().esr_conv_tail_supported(ctypes.byref(d)) == 1
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    # fp64 reference
    x = r3[..., :nf].permute(0, 3, 1, 2).double()
    r4 = F.leaky_relu(F.conv2d(x, w4e.double().to(DEV), b4.double().to(DEV), padding=1), 0.05)
    r4q = r4.to(dt).double()                                                       # the tensor the separate launches store
    cat = torch.cat([ds[j, ..., :dc].permute(0, 3, 1, 2).double() for j in range(3)] + [r4q], 1)
    vref = torch.einsum("oc,nchw->nohw", _hilo(w5, dt).to(DEV), cat) + b5.double().to(DEV)[None, :, None, None]
    for _ in range(3):
        L.check(L.lib().esr_conv2d_f32(ctypes.byref(d), st), "tail")
        torch.cuda.synchronize()
        got = v.permute(0, 3, 1, 2)[:, :nf].double()
        # r4 sits on rounding boundaries now and then: a flipped r4 value moves v by |w5| x one step of r4
        extra = 1.2 * float(w5.abs().max()) * (2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11) * float(r4.abs().max())
        bad = int(((got - vref).abs() > _tol(vref, dt, extra)).sum())
        assert bad == 0, (bad, float((got - vref).abs().max()))
        if compute == "bf16":
            cref = torch.einsum("oc,nchw->nohw", _hilo(wc, dt).to(DEV), vref) + bc.double().to(DEV)[None, :, None, None]
        else:
            cref = torch.einsum("oc,nchw->nohw", wc.to(dt).double().to(DEV), got) + bc.double().to(DEV)[None, :, None, None]
        gc = c1.permute(0, 3, 1, 2)[:, :f].double()
        badc = int(((gc - cref).abs() > _tol(cref, dt, 2e-4 + extra * float(wc.abs().max()) * 8)).sum())
        assert badc == 0, (badc, float((gc - cref).abs().max()))
    assert torch.all(v[..., (nf + 7) // 8 * 8:] == 7.0) and torch.all(v[..., nf:(nf + 7) // 8 * 8] == 0)
    assert torch.all(c1[..., f:(f + 7) // 8 * 8] == 0)


@pytest.mark.parametrize("compute", ["bf16", "f16"])
def test_rfdn_with_and_without_the_fused_block_tail(compute):
    """model.fuse_tail: the same network with RFDB's c4 / c5 + esa.conv1 as two launches or as rfdb_tail_kernel.  The fused form keeps c5's
    weights as hi + lo and v's low part for conv1 (the launches: one 16-bit value each), so the two differ by storage noise; measured against the
    fp32 engine the fused network must not be the worse one"""
    from ntire2022_esr_amd.registry import select_model
    m = select_model(0, torch.device(DEV))[0]
    x = (torch.rand(1, 3, 339, 510, generator=torch.Generator().manual_seed(3)) * 255.0).to(DEV)
    m.set_compute("f32")
    y32 = m(x).clone()
    m.set_compute(compute)
    y1 = m(x).clone()
    m.enable_profiling(1); m(x); torch.cuda.synchronize(); m.collect_profile(); m(x); torch.cuda.synchronize()
    names = {o["kernel"] for o in m.collect_profile()}
    m.disable_profiling()
    assert any(k.startswith("rfdb_tail_kernel") for k in names), names
    m.fuse_tail = False
    y0 = m(x).clone()
    m.fuse_tail = True

    def psnr(a, b):
        return float(10.0 * torch.log10(255.0 ** 2 / ((a - b) ** 2).mean().clamp_min(1e-12)))
    fused, separate, between = psnr(y1, y32), psnr(y0, y32), psnr(y1, y0)
    assert fused > separate - 0.5, (fused, separate, between)
    assert between > (50.0 if compute == "bf16" else 64.0), (fused, separate, between)
    assert torch.equal(m(x), y1)


@pytest.mark.parametrize("n,c,hw,act", [(1, 50, (339, 510), 0), (2, 64, (250, 203), 1), (32, 50, (64, 64), 0), (5, 50, (100, 177), 0)])
def test_conv64m_hilo_lr_conv_matches_fp64_reference(n, c, hw, act):
    """conv64m_kernel<bf16, plain, HL>: the LR conv behind the long skip on hi + lo pairs (RFDN: LR_conv(out_B) + out_fea, rfdn_baseline/RFDN.py:50-52)
    -- the residual pair of ANOTHER tensor as eight selection MFMAs in front of the next row pair's stream, the fp32 result stored as a hi + lo
    pair.  hi + lo is the fp64 result to two bf16 numbers; every image alone (fewer than 256 tiles: conv_s16_kernel's HILO instantiation) agrees
    to the same bound; rows / columns beyond the image and pad channels stay untouched / zero."""
    from ntire2022_esr_amd import _lib as L, ops
    from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16
    g = torch.Generator().manual_seed(n * 100 + c + hw[0])
    cp = 64
    x = torch.randn(n, c, *hw, generator=g).to(torch.bfloat16)
    r32 = torch.randn(n, c, *hw, generator=g) * 3.0
    w, b = torch.randn(c, c, 3, 3, generator=g) * 0.1, torch.randn(c, generator=g)
    blob = pack_conv_s16(w, b, "bf16", cin_phys=cp).to(DEV)
    weff, _ = unpack_conv_s16(blob.cpu(), c, c, 3, "bf16", cin_phys=cp)
    xin = F.pad(x.permute(0, 2, 3, 1), (0, cp - c)).contiguous().to(DEV)
    t = F.pad(r32.permute(0, 2, 3, 1), (0, cp - c))
    hi = t.to(torch.bfloat16)
    rin = torch.stack([hi, (t - hi.float()).to(torch.bfloat16)]).contiguous().to(DEV)
    d = L.ConvDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.ksize = n, hw[0], hw[1], c, c, 3
    d.in_layout = d.out_layout = L.NHWC
    d.storage, d.act, d.res_mode, d.hilo, d.hilo_stride = L.STORE["bf16"], act, L.RES_PRE_ACT, L.HILO_RES | L.HILO_OUT, 4096
    d.res = L.View(ctypes.c_void_p(rin.data_ptr()), cp, 0)
    assert L.lib().esr_conv_block_waves(ctypes.byref(d)) == 1
    kw = dict(cin=c, packed=blob, act=act, res_mode=L.RES_PRE_ACT, hilo=L.HILO_RES | L.HILO_OUT)
    rsum = (rin[0].double() + rin[1].double()).permute(0, 3, 1, 2)[:, :c]
    conv = F.conv2d(x.double().to(DEV), weff.double().to(DEV), b.double().to(DEV), padding=1) + rsum
    ref = F.leaky_relu(conv, 0.05) if act == 1 else conv
    tol = ref.abs() * 2.0 ** -15 + 3e-5 * max(1.0, float(ref.abs().max()))
    for _ in range(2):
        y = ops.conv2d(xin, w, b, res=rin, **kw)
        torch.cuda.synchronize()
        assert tuple(y.shape) == (2, n, *hw, cp)
        got = (y[0].double() + y[1].double()).permute(0, 3, 1, 2)[:, :c]
        assert int(((got - ref).abs() > tol).sum()) == 0, float(((got - ref).abs() - tol).max())
        assert bool((y[1].float().abs() <= y[0].float().abs() * 2.0 ** -7 + 1e-30).all())          # the low parts are remainders of the high ones
        assert torch.all(y[..., (c + 7) // 8 * 8:] == 0) or c == 64
    if n > 1 and n <= 5:
        y1 = ops.conv2d(xin[:1].contiguous(), w, b, res=rin[:, :1].contiguous(), **kw)             # the general kernel
        g1 = (y1[0].double() + y1[1].double()).permute(0, 3, 1, 2)[:, :c]
        assert int(((g1 - ref[:1]).abs() > tol[:1]).sum()) == 0
        assert float((g1 - got[:1]).abs().max()) <= 2.0 * float(tol[:1].max())


@pytest.mark.parametrize("compute", ["f16", "bf16"])
@pytest.mark.parametrize("n,hw,C,dc,f", [(1, (339, 510), 48, 24, 12), (4, (128, 160), 48, 24, 16), (2, (250, 203), 40, 20, 10)])
def test_esdb_tail_matches_fp64_reference(compute, n, hw, C, dc, f):
    """rfdb_tail_kernel<.., 3, true>: ESDB's tail (team18_bsrn.py:165-171, :109) -- c4 = BSConvU as a dense 3x3 over 48 physical channels with
    the merged pointwise bias's border table and GELU, r4 never stored, v = c5 . [d1 d2 d3 r4], c1_ = esa.conv1 . v -- against fp64 on the
    same 16-bit inputs, the blobs' effective weights and the exact GELU (the kernel's polynomial: |error| <= 1.3e-4, esr_s16_dev.h)."""
    from ntire2022_esr_amd import _lib as L, BSRN
    from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16, pack_tail_s16, pack_post_s16
    dt = DT[compute]
    g = torch.Generator().manual_seed(n + hw[0] + C)
    pw, dw = torch.nn.Linear(C, dc), torch.nn.Conv2d(dc, dc, 3, padding=1, groups=dc)
    with torch.no_grad():
        pw.weight.copy_(torch.randn(dc, C, generator=g) * 0.2); pw.bias.copy_(torch.randn(dc, generator=g))
        dw.weight.copy_(torch.randn(dc, 1, 3, 3, generator=g) * 0.4); dw.bias.copy_(torch.randn(dc, generator=g) * 0.3)
    w4, b4, table = BSRN._merged_bsconv(pw, dw)
    r3 = F.pad(torch.randn(n, *hw, C, generator=g), (0, 48 - C)).to(dt).to(DEV)
    ds = F.pad(torch.randn(3, n, *hw, dc, generator=g), (0, 32 - dc)).to(dt).to(DEV)
    w5, b5 = torch.randn(C, 4 * dc, generator=g) * 0.15, torch.randn(C, generator=g)
    wc, bc = torch.randn(f, C, generator=g) * 0.2, torch.randn(f, generator=g)
    blob4 = pack_conv_s16(w4, b4, compute, cin_phys=48)
    w4e, _ = unpack_conv_s16(blob4, C, dc, 3, compute, cin_phys=48)
    blob5 = pack_tail_s16(w5, b5, 3, dc, dc, compute).to(DEV)
    blobc = pack_post_s16(wc, bc, compute).to(DEV)
    blob4, tab = blob4.to(DEV), table.to(DEV)
    assert tuple(tab.shape) == (16, 32)
    v = torch.full((n, *hw, 48), 7.0, dtype=dt, device=DEV)
    c1 = torch.full((n, *hw, 16), 7.0, dtype=dt, device=DEV)
    d = L.ConvDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.ksize = n, hw[0], hw[1], C, dc, 3
    d.in_layout = d.out_layout = L.NHWC
    d.storage = d.compute = L.STORE[compute]
    d.act = L.ACT_NONE
    d.inp = L.View(ctypes.c_void_p(r3.data_ptr()), 48, 0)
    d.out0 = L.View(ctypes.c_void_p(v.data_ptr()), 48, 0)
    d.wpacked = ctypes.c_void_p(blob4.data_ptr())
    d.border_bias = ctypes.c_void_p(tab.data_ptr())
    d.tail_wpacked = ctypes.c_void_p(blob5.data_ptr())
    d.tail_cat = L.View(ctypes.c_void_p(ds.data_ptr()), 32, 0)
    d.tail_cat_c, d.tail_cout, d.tail_mid_act = 96, C, L.ACT_GELU
    d.tail_seg_stride16 = ds[0].numel() * 2 // 16
    d.post_wpacked = ctypes.c_void_p(blobc.data_ptr())
    d.post_out = L.View(ctypes.c_void_p(c1.data_ptr()), 16, 0)
    d.post_cout, d.post_act = f, L.ACT_NONE
    assert L.lib().esr_conv_tail_supported(ctypes.byref(d)) == 1
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    # fp64 reference: the dense 3x3 + the table row of the pixel's outside sides, exact GELU
    x = r3[..., :C].permute(0, 3, 1, 2).double()
    pre = F.conv2d(x, w4e.double().to(DEV), b4.double().to(DEV), padding=1)
    ys, xs = torch.arange(hw[0], device=DEV), torch.arange(hw[1], device=DEV)
    mask = ((xs == 0).long() + 2 * (xs == hw[1] - 1).long())[None, :] + (4 * (ys == 0).long() + 8 * (ys == hw[0] - 1).long())[:, None]
    pre = pre + tab.double()[mask][..., :dc].permute(2, 0, 1)[None]
    r4 = F.gelu(pre)
    r4q = r4.to(dt).double()
    cat = torch.cat([ds[j, ..., :dc].permute(0, 3, 1, 2).double() for j in range(3)] + [r4q], 1)
    vref = torch.einsum("oc,nchw->nohw", _hilo(w5, dt).to(DEV), cat) + b5.double().to(DEV)[None, :, None, None]
    step = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    # a flipped rounding of r4 / the polynomial's 1.3e-4 move v by |w5| x that much
    extra = float(w5.abs().max()) * (1.2 * step * float(r4.abs().max()) + 4 * 1.3e-4)
    for _ in range(2):
        L.check(L.lib().esr_conv2d_f32(ctypes.byref(d), st), "tail")
        torch.cuda.synchronize()
        got = v.permute(0, 3, 1, 2)[:, :C].double()
        bad = int(((got - vref).abs() > _tol(vref, dt, extra)).sum())
        assert bad == 0, (bad, float((got - vref).abs().max()))
        cref = torch.einsum("oc,nchw->nohw", (_hilo(wc, dt) if compute == "bf16" else wc.to(dt).double()).to(DEV),
                            vref if compute == "bf16" else got) + bc.double().to(DEV)[None, :, None, None]
        gc = c1.permute(0, 3, 1, 2)[:, :f].double()
        badc = int(((gc - cref).abs() > _tol(cref, dt, 2e-4 + extra * float(wc.abs().max()) * 8)).sum())
        assert badc == 0, (badc, float((gc - cref).abs().max()))
    assert torch.all(v[..., (C + 7) // 8 * 8:] == 7.0) and torch.all(v[..., C:(C + 7) // 8 * 8] == 0)
    assert torch.all(c1[..., f:(f + 7) // 8 * 8] == 0)


def test_bsrn_with_and_without_the_fused_block_tail():
    """model.fuse_tail on BSRN fp16 (BASELINE config [4]): ESDB's c4 / c5 + esa.conv1 as two launches or as rfdb_tail_kernel<false, 3, true>;
    against the fp32 engine the fused network must not be the worse one."""
    from ntire2022_esr_amd.registry import select_model
    m = select_model(18, torch.device(DEV))[0]
    x = (torch.rand(1, 3, 270, 480, generator=torch.Generator().manual_seed(3)) * 255.0).to(DEV)
    m.set_compute("f32")
    y32 = m(x).clone()
    m.set_compute("f16")
    y1 = m(x).clone()
    m.enable_profiling(1); m(x); torch.cuda.synchronize(); m.collect_profile(); m(x); torch.cuda.synchronize()
    names = {o["kernel"] for o in m.collect_profile()}
    m.disable_profiling()
    assert any(k.startswith("rfdb_tail_kernel<false, 3, true>") for k in names), names
    assert any(k.startswith("conv64m_kernel<false, true, false, 3, true>") for k in names), names         # c1_r / c2_r + the next distillation Linear
    assert any(k.startswith("conv64m_kernel<false, false, false, 3, true>") for k in names), names        # c3_r
    m.fuse_tail = False
    y0 = m(x).clone()
    m.fuse_tail = True

    def psnr(a, b):
        return float(10.0 * torch.log10(255.0 ** 2 / ((a - b) ** 2).mean().clamp_min(1e-12)))
    fused, separate, between = psnr(y1, y32), psnr(y0, y32), psnr(y1, y0)
    assert fused > separate - 0.5, (fused, separate, between)
    assert between > 60.0, (fused, separate, between)
    assert torch.equal(m(x), y1)


@pytest.mark.parametrize("compute,post", [("f16", True), ("f16", False), ("bf16", False)])
@pytest.mark.parametrize("n,c,hw", [(1, 48, (270, 480)), (6, 40, (144, 160)), (3, 48, (250, 203))])
def test_esdb_r_on_conv64m_matches_fp64_reference(compute, post, n, c, hw):
    """conv64m_kernel<.., 3, true>: ESDB's c{j}_r (team18_bsrn.py:150-163) -- gelu(dense BSConvU(x) + table row + x) over 48 physical channels,
    plain or (fp16) with the next distillation Linear + GELU behind it -- against fp64 on the same 16-bit inputs, the blob's effective weights and
    the exact GELU (the kernel's polynomial: |error| <= 1.3e-4)."""
    from ntire2022_esr_amd import ops, _lib as L
    from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16
    dt = DT[compute]
    g = torch.Generator().manual_seed(n + c + hw[0] + post)
    pc = c // 2
    x = F.pad(torch.randn(n, *hw, c, generator=g), (0, 48 - c)).to(dt).to(DEV)
    w, b = torch.randn(c, c, 3, 3, generator=g) * 0.1, torch.randn(c, generator=g)
    wp, bp = torch.randn(pc, c, generator=g) * 0.2, torch.randn(pc, generator=g)
    table = torch.randn(16, 48, generator=g) * 0.2
    table[0] = 0
    table[:, c:] = 0
    table = table.to(DEV)
    blob = pack_conv_s16(w, b, compute, cin_phys=48)
    weff, _ = unpack_conv_s16(blob, c, c, 3, compute, cin_phys=48)
    kw = dict(act=L.ACT_GELU, cin=c, packed=blob.to(DEV), border=table, res=x, res_mode=L.RES_PRE_ACT)
    if post:
        kw.update(post_weight=wp, post_bias=bp, post_act=L.ACT_GELU)
    d = L.ConvDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.ksize = n, hw[0], hw[1], c, c, 3
    d.in_layout = d.out_layout = L.NHWC
    d.storage = L.STORE[compute]
    assert L.lib().esr_conv_block_waves(ctypes.byref(d)) == 1
    xd = x[..., :c].permute(0, 3, 1, 2).double()
    ys, xs = torch.arange(hw[0], device=DEV), torch.arange(hw[1], device=DEV)
    mask = ((xs == 0).long() + 2 * (xs == hw[1] - 1).long())[None, :] + (4 * (ys == 0).long() + 8 * (ys == hw[0] - 1).long())[:, None]
    pre = F.conv2d(xd, weff.double().to(DEV), b.double().to(DEV), padding=1) + table.double()[mask][..., :c].permute(2, 0, 1)[None] + xd
    ref = F.gelu(pre)
    for _ in range(2):
        out = ops.conv2d(x, w, b, **kw)
        torch.cuda.synchronize()
        y, yp = out if post else (out, None)
        got = y.permute(0, 3, 1, 2)[:, :c].double()
        bad = int(((got - ref).abs() > _tol(ref, dt, 2.6e-4)).sum())
        assert bad == 0, (bad, float((got - ref).abs().max()))
        assert torch.all(y[..., c:] == 0)
        if post:
            pref = F.gelu(torch.einsum("oc,nchw->nohw", wp.to(dt).double().to(DEV), got) + bp.double().to(DEV)[None, :, None, None])
            gp = yp.permute(0, 3, 1, 2)[:, :pc].double()
            badp = int(((gp - pref).abs() > _tol(pref, dt, 4e-4)).sum())
            assert badp == 0, (badp, float((gp - pref).abs().max()))
            assert torch.all(yp[..., pc:] == 0)
