"""-m gpu: the 16-bit-STORAGE path (bf16 / fp16 activations in HBM = MFMA operands, fp32 accumulate, one rounding at
the store).  Kernel logic is checked against an fp64 reference on the same 16-bit inputs and the blob's effective
weights (tolerance: that one rounding); the rounding itself is then bounded at network level as a PSNR shift."""
import ctypes
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


def _ulp_ok(got, ref, dt):
    """got (16-bit, as float) vs the fp64 reference: one rounding (half an ulp) + the fp32 accumulation noise"""
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    tol = ref.abs() * eps * 1.01 + 3e-5 * max(1.0, float(ref.abs().max()))
    return bool(((got.double() - ref).abs() <= tol).all()), float(((got.double() - ref).abs() - tol).max())


def _same_or_one_step(a, b, dt, frac=0.03, scale=1.0):
    """two 16-bit results of the same arithmetic in another fp32 accumulation order (conv64m_kernel against conv_s16_kernel, round 6): equal but for a
    few values that sat on a rounding boundary, and those differ by one step of the storage type"""
    af, bf = a.float(), b.float()
    eps = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    step = torch.maximum(af.abs(), bf.abs()) * eps + 3e-5 * max(1.0, float(af.abs().max()))      # (+ the accumulation noise of values that cancel to ~0)
    step = step * scale
    diff = (af - bf).abs()
    ok = bool((diff <= step).all()) and float((diff > 0).float().mean()) < frac
    if not ok:
        print(f"_same_or_one_step: worst diff / step = {float((diff / step).max()):.3f}, differing fraction = {float((diff > 0).float().mean()):.4f} (limit {frac})")
    return ok


ACTS = {0: lambda t: t, 1: lambda t: F.leaky_relu(t, 0.05), 2: F.relu, 3: lambda t: F.gelu(t)}


@pytest.mark.parametrize("compute", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout,k,hw,act,res_mode", [
    (64, 64, 3, (16, 32), 1, 1), (48, 48, 3, (23, 37), 1, 2), (48, 16, 3, (17, 15), 0, 0), (64, 48, 3, (40, 56), 1, 0),
    (50, 50, 3, (20, 36), 1, 1), (16, 16, 3, (5, 3), 2, 0), (46, 46, 1, (33, 18), 0, 0), (50, 25, 1, (40, 40), 1, 0),
    (128, 50, 1, (19, 70), 0, 0), (256, 50, 1, (35, 33), 1, 0), (48, 48, 1, (64, 64), 3, 2), (32, 64, 3, (70, 50), 1, 1)])
def test_s16_conv_matches_fp64_reference(compute, cin, cout, k, hw, act, res_mode):
    """16-bit storage conv (esr_conv2d_f32 with storage = bf16 / f16): inputs are exact 16-bit values, the reference uses
    the EFFECTIVE weights of the packed blob (error-diffused 3x3 taps / hi + lo 1x1) in fp64, so the only differences
    are fp32 accumulation order and the single rounding of the stored result."""
    from ntire2022_esr_amd import ops
    from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16
    dt = DT[compute]
    g = torch.Generator().manual_seed(cin + cout + hw[0] + k)
    cp = (cin + 15) // 16 * 16
    x = torch.randn(2, cin, *hw, generator=g).to(dt)
    r = torch.randn(2, cout, *hw, generator=g).to(dt)
    w = torch.randn(cout, cin, k, k, generator=g) * (0.1 if k == 3 else 0.2)
    b = torch.randn(cout, generator=g)
    blob = pack_conv_s16(w, b, compute, cin_phys=cp)
    weff, _ = unpack_conv_s16(blob, cin, cout, k, compute, cin_phys=cp)
    conv = F.conv2d(x.double(), weff.double(), b.double(), padding=k // 2)
    ref = ACTS[act](conv + r.double()) if res_mode == 1 else (ACTS[act](conv) + r.double() if res_mode == 2 else ACTS[act](conv))
    xin = F.pad(_nhwc(x), (0, cp - cin)).to(DEV)
    rp = F.pad(_nhwc(r), (0, (-cout) % 8)).to(DEV) if res_mode else None
    y = ops.conv2d(xin, w, b, act=act, res=rp, res_mode=res_mode, cin=cin, packed=blob.to(DEV))
    assert y.dtype == dt and y.shape[-1] == (cout + 7) // 8 * 8
    got = y.float().cpu().permute(0, 3, 1, 2)
    if act == 3:            # GELU in the 16-bit modes is gelu16(): |error| <= 1.3e-4 (tools/fit_gelu.py) on top of the rounding
        assert bool(((got[:, :cout].double() - ref).abs() <= ref.abs() * 2.0 ** (-8 if dt == torch.bfloat16 else -11) * 1.01 + 2.5e-4).all())
    else:
        ok, worst = _ulp_ok(got[:, :cout], ref, dt)
        assert ok, worst
    assert torch.all(got[:, cout:] == 0)                    # pad channels of the 16-byte granule are written as zeros


@pytest.mark.parametrize("compute", ["bf16", "f16"])
@pytest.mark.parametrize("n,cin,cout,k,hw,res_mode", [
    (1, 64, 64, 3, (270, 480), 0), (1, 64, 64, 3, (270, 480), 2), (1, 64, 64, 1, (270, 480), 2), (1, 48, 48, 3, (339, 510), 2),
    (1, 64, 64, 3, (339, 510), 1), (3, 48, 48, 3, (256, 256), 0), (1, 16, 16, 3, (270, 480), 1), (5, 50, 50, 1, (200, 200), 0),
    (2, 48, 48, 3, (339, 510), 0), (1, 40, 44, 3, (339, 510), 0)])      # three output tiles, no HBM residual, >= 256 tiles: conv48r_kernel (48 channels)
def test_s16_conv_more_tiles_than_blocks(compute, n, cin, cout, k, hw, res_mode):
    """Shapes with MORE 16x32 tiles than the 256 persistent blocks and partial tiles at both edges: the tile-to-tile path of a
    block (the epilogue of tile k inside the first MFMA group of tile k+1, residual registers reused across tiles, unequal
    tile counts per block, the drain iteration) -- the small shapes above give every block at most one tile.  The 3x3s over 48
    physical channels without a residual from HBM take conv48r_kernel (esr_conv_block_waves == 1)."""
    from ntire2022_esr_amd import ops
    from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16
    dt = DT[compute]
    g = torch.Generator().manual_seed(n + cin + cout + hw[0] + k)
    cp = (cin + 15) // 16 * 16
    x = torch.randn(n, cin, *hw, generator=g).to(dt)
    r = torch.randn(n, cout, *hw, generator=g).to(dt)
    w = torch.randn(cout, cin, k, k, generator=g) * (0.1 if k == 3 else 0.2)
    b = torch.randn(cout, generator=g)
    blob = pack_conv_s16(w, b, compute, cin_phys=cp)
    weff, _ = unpack_conv_s16(blob, cin, cout, k, compute, cin_phys=cp)
    conv = F.conv2d(x.double().to(DEV), weff.double().to(DEV), b.double().to(DEV), padding=k // 2)
    rd = r.double().to(DEV)
    ref = ACTS[1](conv + rd) if res_mode == 1 else (ACTS[1](conv) + rd if res_mode == 2 else ACTS[1](conv))
    xin = F.pad(_nhwc(x), (0, cp - cin)).to(DEV)
    rp = F.pad(_nhwc(r), (0, (-cout) % 8)).to(DEV) if res_mode else None
    for _ in range(3):                  # a race would not show every time
        y = ops.conv2d(xin, w, b, act=1, res=rp, res_mode=res_mode, cin=cin, packed=blob.to(DEV))
        got = y.permute(0, 3, 1, 2)[:, :cout].double()
        eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
        tol = ref.abs() * eps * 1.01 + 3e-5 * max(1.0, float(ref.abs().max()))
        bad = int(((got - ref).abs() > tol).sum())
        assert bad == 0, bad


@pytest.mark.parametrize("compute", ["bf16", "f16"])
@pytest.mark.parametrize("nseg,pp,cout,hw,res_mode", [(4, 32, 48, (40, 56), 0), (4, 16, 64, (33, 70), 1), (2, 48, 50, (270, 480), 0), (5, 16, 16, (20, 20), 2)])
def test_s16_segmented_input_equals_dense_concat(compute, nseg, pp, cout, hw, res_mode):
    """esr_conv_desc.in_seg_stride / in_seg_chunks: the 1x1 over a concat kept as `nseg` dense tensors must give exactly what it
    gives on the dense [.., nseg * pp] buffer holding the same channels (same blob, same chunk order, same arithmetic)."""
    from ntire2022_esr_amd import ops
    dt = DT[compute]
    g = torch.Generator().manual_seed(nseg + pp + cout)
    xs = torch.randn(nseg, 2, *hw, pp, generator=g).to(dt).to(DEV)                # planar
    xd = torch.cat([xs[j] for j in range(nseg)], dim=-1).contiguous()             # dense concat
    w = torch.randn(cout, nseg * pp, generator=g) * 0.1
    b = torch.randn(cout, generator=g)
    r = torch.randn(2, *hw, (cout + 7) // 8 * 8, generator=g).to(dt).to(DEV) if res_mode else None
    yd = ops.conv2d(xd, w, b, act=1, res=r, res_mode=res_mode)
    ys = ops.conv2d(xs, w, b, act=1, res=r, res_mode=res_mode)
    assert torch.equal(yd, ys)


@pytest.mark.parametrize("compute", ["bf16", "f16"])
def test_s16_split_store_slices_and_shuffle(compute):
    """channel-split store (IMDBlock: 16 -> concat slice, 48 -> next conv), reads from / writes into slices of wider
    buffers, and the PixelShuffle(4) tail writing the fp32 NCHW network output"""
    from ntire2022_esr_amd import ops
    from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16
    dt = DT[compute]
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 64, 37, 21, generator=g).to(dt)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.1
    b = torch.randn(64, generator=g)
    weff, _ = unpack_conv_s16(pack_conv_s16(w, b, compute), 64, 64, 3, compute)
    ref = F.leaky_relu(F.conv2d(x.double(), weff.double(), b.double(), padding=1), 0.05)
    cat = torch.zeros(1, 37, 21, 64, dtype=dt, device=DEV)
    nxt = torch.zeros(1, 37, 21, 48, dtype=dt, device=DEV)
    wide = torch.zeros(1, 37, 21, 128, dtype=dt, device=DEV)
    wide[..., 64:] = _nhwc(x).to(DEV)
    ops.conv2d(wide, w, b, act=1, in_coff=64, cin=64, split=16, out=cat, out_coff=32, out1=nxt)
    ok, worst = _ulp_ok(cat[..., 32:48].float().cpu().permute(0, 3, 1, 2), ref[:, :16], dt)
    assert ok, worst
    ok, worst = _ulp_ok(nxt.float().cpu().permute(0, 3, 1, 2), ref[:, 16:], dt)
    assert ok, worst
    assert torch.all(cat[..., :32] == 0) and torch.all(cat[..., 48:] == 0)
    # PixelShuffle tail: 16-bit NHWC in, fp32 NCHW out (no rounding of the result)
    w2 = torch.randn(48, 64, 3, 3, generator=g) * 0.1
    b2 = torch.randn(48, generator=g)
    w2e, _ = unpack_conv_s16(pack_conv_s16(w2, b2, compute), 64, 48, 3, compute)
    y = ops.conv2d(_nhwc(x).to(DEV), w2, b2, shuffle_out=True)
    ref2 = F.pixel_shuffle(F.conv2d(x.double(), w2e.double(), b2.double(), padding=1), 4)
    assert y.dtype == torch.float32 and tuple(y.shape) == (1, 3, 148, 84)
    assert float((y.cpu().double() - ref2).abs().max()) < 3e-5 * float(ref2.abs().max())


@pytest.mark.parametrize("compute", ["bf16", "f16"])
def test_s16_head_from_nchw_fp32(compute):
    """network head of a 16-bit network: fp32 NCHW input (exact), fp32 MFMA, NHWC result stored as 16-bit"""
    from ntire2022_esr_amd import ops
    dt = DT[compute]
    g = torch.Generator().manual_seed(9)
    x = torch.rand(2, 3, 33, 29, generator=g) * 255
    w = torch.randn(46, 3, 3, 3, generator=g) * 0.05
    b = torch.randn(46, generator=g)
    y = ops.conv2d(x.to(DEV), w, b, in_nchw=True, store=compute)
    assert y.dtype == dt and y.shape[-1] == 48
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    ok, worst = _ulp_ok(y.float().cpu().permute(0, 3, 1, 2)[:, :46], ref, dt)
    assert ok, worst
    assert torch.all(y[..., 46:] == 0)


@pytest.mark.parametrize("compute", ["bf16", "f16"])
@pytest.mark.parametrize("n,hw", [(2, (37, 29)), (1, (256, 250)), (2, (339, 510)), (3, (200, 123))])
def test_s16_post_chain_rlfb(compute, n, hw):
    """RLFB tail in one launch: u = lrelu(c3_r(x)) + r (never stored), v = c5(u), c1 = esa.conv1(v): the 1x1s run on the fp32
    tile (hi + lo operands), only v and c1 are rounded -- against fp64 with the blob's effective 3x3 weights.  The small shape runs
    on conv_s16_kernel, the others (>= 256 tiles of 16 x 16, < 1024 of 16 x 32) on conv48rp_kernel: weights in registers, the residual
    staged by each wave for its own rows, ragged edges in both directions."""
    from ntire2022_esr_amd import ops
    from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16
    dt = DT[compute]
    g = torch.Generator().manual_seed(21 + n + hw[0])
    x = torch.randn(n, 48, *hw, generator=g).to(dt)
    r = torch.randn(n, 46, *hw, generator=g).to(dt)
    w, b = torch.randn(46, 48, 3, 3, generator=g) * 0.1, torch.randn(46, generator=g)
    w5, b5 = torch.randn(46, 46, generator=g) * 0.2, torch.randn(46, generator=g)
    w1, b1 = torch.randn(16, 46, generator=g) * 0.2, torch.randn(16, generator=g)
    weff, _ = unpack_conv_s16(pack_conv_s16(w, b, compute), 48, 46, 3, compute)
    u = F.leaky_relu(F.conv2d(x.double(), weff.double(), b.double(), padding=1), 0.05) + r.double()
    # bf16: hi + lo operands (the chain sees ~fp32 values and weights); fp16: the 11-bit high parts only -- operands of the
    # chain's MFMAs are the fp16 roundings of the fp32 tile and of the weights (the network's own storage precision)
    q = (lambda t: t) if compute == "bf16" else (lambda t: t.to(torch.float16).double())
    v = F.conv2d(q(u), q(w5.double())[:, :, None, None], b5.double())
    c1 = F.conv2d(q(v), q(w1.double())[:, :, None, None], b1.double())
    rp = F.pad(_nhwc(r), (0, 2)).to(DEV)
    y, yv, yc = ops.conv2d(_nhwc(x).to(DEV), w, b, act=1, res=rp, res_mode=2, post_weight=w5, post_bias=b5,
                           post2_weight=w1, post2_bias=b1, store_main=False)
    assert y is None and yv.dtype == dt and yv.shape[-1] == 48 and yc.shape[-1] == 16
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11

    def close(got, want):
        # one rounding of the stored value + the dropped lo x lo terms of the hi/lo split (2^-16 bf16 / 2^-22 fp16 relative)
        # fp16: + the chain operands that sit on a rounding boundary in fp32 and round the other way in this fp64 reference
        # (|w| * ulp each, a handful per output)
        slack = (1e-3 * 32 if compute == "bf16" else 0.5) * eps * max(1.0, float(want.abs().max()))
        tol = want.abs() * eps * 1.01 + slack
        err = (got.double() - want).abs()
        if compute == "f16" and want.numel() > 1e6:
            # large images: the boundary cases are a fixed small FRACTION of the outputs, and a few of them stack up
            print(f"fraction beyond tol {float((err > tol).double().mean()):.2e}, worst {float((err / tol).max()):.2f} x tol")
            return float((err > tol).double().mean()) <= 2e-5 and bool((err <= 3 * tol).all())
        return bool((err <= tol).all())

    assert close(yv.float().cpu().permute(0, 3, 1, 2)[:, :46], v)
    assert close(yc.float().cpu().permute(0, 3, 1, 2)[:, :16], c1)
    assert torch.all(yv[..., 46:] == 0)


@pytest.mark.parametrize("compute", ["bf16", "f16"])
@pytest.mark.parametrize("nf", [50, 40])
def test_s16_post_rfdb(compute, nf):
    """RFDB: r = lrelu(c_r(x) + x) stored, d = lrelu(c_d(r)) from the same launch (rfdn_baseline/block.py:150-160)"""
    from ntire2022_esr_amd import ops
    from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16
    dt = DT[compute]
    dc = nf // 2
    g = torch.Generator().manual_seed(nf)
    P = (nf + 15) // 16 * 16
    x = torch.randn(1, nf, 45, 33, generator=g).to(dt)
    w, b = torch.randn(nf, nf, 3, 3, generator=g) * 0.08, torch.randn(nf, generator=g)
    wd, bd = torch.randn(dc, nf, generator=g) * 0.2, torch.randn(dc, generator=g)
    weff, _ = unpack_conv_s16(pack_conv_s16(w, b, compute, cin_phys=P), nf, nf, 3, compute, cin_phys=P)
    rr = F.leaky_relu(F.conv2d(x.double(), weff.double(), b.double(), padding=1) + x.double(), 0.05)
    # hi + lo post weights for both shapes (the epilogue needs no LDS scratch any more: the low-part images fit next to nf = 50's
    # 80 KB of 3x3 weights too)
    # (fp16: the chain multiplies the fp16 roundings of the fp32 tile and of the weights, see test_s16_post_chain_rlfb)
    q = (lambda t: t) if compute == "bf16" else (lambda t: t.to(torch.float16).double())
    wd_eff = q(wd.double())
    dd = F.leaky_relu(F.conv2d(q(rr), wd_eff[:, :, None, None], bd.double()), 0.05)
    xin = F.pad(_nhwc(x), (0, P - nf)).to(DEV)
    y, yd = ops.conv2d(xin, w, b, act=1, res=xin, res_mode=1, cin=nf, post_weight=wd, post_bias=bd, post_act=1)
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11

    def close(got, want):
        slack = (2e-2 if compute == "bf16" else 0.5) * eps * max(1.0, float(want.abs().max()))      # fp16: see test_s16_post_chain_rlfb
        tol = want.abs() * eps * 1.01 + slack
        return bool(((got.double() - want).abs() <= tol).all())

    assert close(y.float().cpu().permute(0, 3, 1, 2)[:, :nf], rr)
    assert close(yd.float().cpu().permute(0, 3, 1, 2)[:, :dc], dd)


def test_s16_rejects_bad_descriptors():
    from ntire2022_esr_amd import _lib as L, ops
    x = torch.zeros(1, 8, 8, 16, dtype=torch.bfloat16, device=DEV)            # pitch 16 < round_up(cin, 8) = 24: the pixel does not hold the channels
    with pytest.raises(L.EsrError):
        ops.conv2d(x, torch.randn(16, 24, 3, 3), torch.randn(16))
    x = torch.zeros(1, 8, 8, 80, dtype=torch.float16, device=DEV)             # 3x3 with 80 input channels: weights do not fit LDS
    with pytest.raises(L.EsrError):
        ops.conv2d(x, torch.randn(64, 80, 3, 3), torch.randn(64))


@pytest.mark.parametrize("compute", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout,k,hw,n", [(24, 16, 3, (19, 23), 2), (50, 50, 3, (40, 37), 1), (50, 25, 1, (33, 18), 2), (50, 50, 3, (256, 256), 1), (40, 64, 3, (256, 272), 1)])
def test_tight_pitch_equals_whole_chunk_pitch(compute, cin, cout, k, hw, n):
    """TIGHT PITCH (round 6, esr_conv2d_s16): an input whose pixels hold round_up(cin, 8) channels -- 56 for RFDN's nf = 50, 112-byte pixels --
    instead of whole 16-channel K chunks.  The last chunk's second half is then the next pixel's first 16 bytes (NON-zero here: random data) and
    meets zero weight rows: the result is bit-identical to the same convolution on the zero-padded whole-chunk tensor, on the general kernel and
    (>= 256 tiles, 64 -> 64) on conv64m_kernel."""
    from ntire2022_esr_amd import ops
    dt = torch.bfloat16 if compute == "bf16" else torch.float16
    g = torch.Generator().manual_seed(cin + cout + hw[0])
    p8, p16 = (cin + 7) // 8 * 8, (cin + 15) // 16 * 16
    xt = torch.randn(n, *hw, p8, generator=g).to(dt)
    xt[..., cin:] = 0                                   # the pad channels inside the pixel are zeros, as every producer leaves them
    xp = torch.zeros(n, *hw, p16, dtype=dt)
    xp[..., :p8] = xt
    w, b = torch.randn(cout, cin, k, k, generator=g) * 0.1, torch.randn(cout, generator=g)
    xt_d, xp_d = xt.to(DEV), xp.to(DEV)
    y_t = ops.conv2d(xt_d, w, b, act=1)
    y_p = ops.conv2d(xp_d, w, b, act=1)
    assert y_t.shape[-1] >= cout and bool(torch.isfinite(y_t.float()).all())
    assert torch.equal(y_t[..., :cout], y_p[..., :cout])
    if cin == cout and k == 3:
        # RFDB's c{j}_r: + the input itself, before the activation (taken from the staged tile: the neighbour pixel's bytes land in channels that
        # are not stored)
        r_t = ops.conv2d(xt_d, w, b, act=1, res=xt_d, res_mode=1)
        r_p = ops.conv2d(xp_d, w, b, act=1, res=xp_d, res_mode=1)
        assert torch.equal(r_t[..., :cout], r_p[..., :cout]) and not torch.equal(r_t[..., :cout], y_t[..., :cout])


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
    # (rounds 2-3 allowed this single 4 k-pixel image 1.5 x the dataset-mean budget: RLFN bf16 sat at -0.011 dB; with the long skip in
    # hi + lo pairs -- round 4 -- it is -0.005 dB and the budget itself holds)
    assert abs(p16 - p32) < max_dpsnr
    model.set_compute("f32")
    assert torch.equal(model(x), y32)                       # switching back restores the exact fp32 path


@pytest.mark.parametrize("compute", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout,act,res_in,border,hw,n", [
    (48, 48, 1, False, False, (128, 128), 8),      # RLFB c1_r / c2_r: plain, LeakyReLU
    (46, 46, 1, False, False, (150, 97), 9),       # logical 46 channels, ragged edges, tiles not a multiple of the grid
    (48, 48, 3, True, True, (128, 128), 8),        # ESDB c{j}_r as a dense BSConvU: + input, border table, GELU
    (48, 48, 3, True, True, (90, 130), 12),
    (48, 24, 3, False, True, (128, 128), 8),       # ESDB c4: two output tiles
    (48, 48, 0, True, False, (128, 128), 8),       # residual == input without activation
    (48, 48, 1, False, False, (128, 128), 32),     # >= 1024 tiles of 16 x 32: the 8-rows-per-wave shape (the others: 16 x 16 tiles)
    (48, 48, 3, True, True, (128, 120), 33)])
def test_conv48r_equals_conv_s16(compute, cin, cout, act, res_in, border, hw, n):
    """conv48r_kernel (3x3 over 48 physical input channels, >= 256 tiles of 16 x 32: weights in registers, one wave per SIMD, row pairs
    as the outer loop) against conv_s16_kernel: the batch takes the new kernel (esr_conv_block_waves == 1), each image alone the old
    one (<= 72 tiles), same packed weights -- bit-identical results, and both within storage precision of the fp64 convolution."""
    from ntire2022_esr_amd import ops, _lib as L
    from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16
    dt = DT[compute]
    g = torch.Generator().manual_seed(cin + cout + act + hw[0] + n)
    cp = 48
    x = torch.randn(n, cin, *hw, generator=g).to(dt)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    b = torch.randn(cout, generator=g)
    table = (torch.randn(16, (cout + 15) // 16 * 16, generator=g) * 0.2) if border else None
    if table is not None:
        table[0] = 0                                    # row 0 = interior pixels of a border tile
    blob = pack_conv_s16(w, b, compute, cin_phys=cp).to(DEV)
    xin = F.pad(_nhwc(x), (0, cp - cin)).to(DEV)
    kw = dict(act=act, cin=cin, packed=blob, border=None if table is None else table.to(DEV))
    if res_in:
        kw.update(res=xin, res_mode=L.RES_PRE_ACT)
    d = L.ConvDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.ksize = n, hw[0], hw[1], cin, cout, 3
    d.in_layout = d.out_layout = L.NHWC
    d.storage = L.STORE[compute]
    assert L.lib().esr_conv_block_waves(ctypes.byref(d)) == 1
    d.n = 1
    assert L.lib().esr_conv_block_waves(ctypes.byref(d)) != 1
    y = ops.conv2d(xin, w, b, **kw)
    for i in range(n):
        kw1 = dict(kw)
        if res_in:
            kw1["res"] = xin[i:i + 1]
        y1 = ops.conv2d(xin[i:i + 1].contiguous(), w, b, **kw1)
        if act == 3 and res_in and border and cout > 32:
            # ESDB's c{j}_r shape: since round 6 the batch runs on conv64m_kernel<.., 3, true> (v_mfma_f32_32x32x16, another accumulation order;
            # its fp64-reference check: test_gpu_c64m.py) -- equal but for values on a rounding boundary, those one storage step apart
            assert _same_or_one_step(y[i:i + 1], y1, dt), i
        else:
            assert torch.equal(y[i:i + 1], y1), i
    if not border:
        weff, _ = unpack_conv_s16(blob.cpu(), cin, cout, 3, compute, cin_phys=cp)
        conv = F.conv2d(x.double().to(DEV), weff.double().to(DEV), b.double().to(DEV), padding=1)
        if res_in:
            conv = conv + x.double().to(DEV)[:, :cout]
        ref = ACTS[act](conv)
        got = y.permute(0, 3, 1, 2)[:, :cout].double()
        eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
        tol = ref.abs() * eps * 1.01 + (3e-4 if act == 3 else 5e-5) * max(1.0, float(ref.abs().max()))
        assert int(((got - ref).abs() > tol).sum()) == 0


@pytest.mark.parametrize("compute", ["bf16", "f16"])
@pytest.mark.parametrize("hw,n", [((128, 128), 8), ((100, 77), 9)])
def test_conv48rp_equals_conv_s16(compute, hw, n):
    """conv48rp_kernel (RLFB c3_r + block input -> c5 -> esa.conv1: weights in registers, 16 x 16 tiles, the residual staged by each wave
    for its own rows, post images in LDS, the chain on both rows of a pair) against conv_s16_kernel: the batch takes the new kernel
    (>= 256 tiles of 16 x 16), each image alone the old one -- both outputs bit-identical, ragged edges included."""
    from ntire2022_esr_amd import ops, _lib as L
    dt = DT[compute]
    g = torch.Generator().manual_seed(hw[0] + n)
    x = _nhwc(torch.randn(n, 48, *hw, generator=g).to(dt)).to(DEV)
    r = F.pad(_nhwc(torch.randn(n, 46, *hw, generator=g).to(dt)), (0, 2)).to(DEV)
    w, b = torch.randn(46, 48, 3, 3, generator=g) * 0.1, torch.randn(46, generator=g)
    w5, b5 = torch.randn(46, 46, generator=g) * 0.2, torch.randn(46, generator=g)
    w1, b1 = torch.randn(16, 46, generator=g) * 0.2, torch.randn(16, generator=g)
    kw = dict(act=1, res_mode=2, post_weight=w5, post_bias=b5, post2_weight=w1, post2_bias=b1, store_main=False)
    tiles16 = lambda nn: nn * ((hw[1] + 15) // 16) * ((hw[0] + 15) // 16)
    assert tiles16(n) >= 256 and tiles16(1) < 256
    y, yv, yc = ops.conv2d(x, w, b, res=r, **kw)
    assert y is None
    for i in range(n):
        _, v1, c1 = ops.conv2d(x[i:i + 1].contiguous(), w, b, res=r[i:i + 1].contiguous(), **kw)
        assert torch.equal(yv[i:i + 1], v1) and torch.equal(yc[i:i + 1], c1), i


def _hilo(t, cp):
    """NCHW fp32 -> bf16 pair [2, N, H, W, cp]: high parts, low parts (value = hi + lo; pad channels zero)"""
    t = F.pad(_nhwc(t), (0, cp - t.shape[1]))
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo]).contiguous()


@pytest.mark.parametrize("n,c,hw", [(2, 46, (37, 29)), (1, 50, (64, 80)), (3, 48, (100, 77)), (1, 64, (339, 510))])
def test_s16_hilo_skip_convs(n, c, hw):
    """esr_conv_desc.hilo (ABI v10): the three convolutions of the long skip on hi + lo tensors.
    head:       y = conv(x16)              -> hi + lo out: the high parts ARE the plain kernel's output (bit-identical), hi + lo is
                                              the fp64 result to 2^-16 relative (two bf16 numbers) instead of 2^-8
    LR_conv:    y = conv(x16) + (r_hi + r_lo)  -> hi + lo out, same bound
    upsampler:  y = shuffle(conv(x_hi + x_lo))  against the fp64 convolution of the SUM with the blob's effective weights"""
    from ntire2022_esr_amd import _lib as L, ops
    from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16
    g = torch.Generator().manual_seed(n * 1000 + c + hw[0])
    cp = (c + 15) // 16 * 16
    x = torch.randn(n, c, *hw, generator=g).to(torch.bfloat16)
    r32 = torch.randn(n, c, *hw, generator=g) * 3.0
    w = torch.randn(c, c, 3, 3, generator=g) * 0.1
    b = torch.randn(c, generator=g)
    blob = pack_conv_s16(w, b, "bf16", cin_phys=cp)
    weff, _ = unpack_conv_s16(blob, c, c, 3, "bf16", cin_phys=cp)
    xin = F.pad(_nhwc(x), (0, cp - c)).to(DEV)
    conv = F.conv2d(x.double(), weff.double(), b.double(), padding=1)

    def check(y, ref, what):
        assert y.dtype == torch.bfloat16 and tuple(y.shape) == (2, n, *hw, cp)
        hi, lo = y[0].float().cpu(), y[1].float().cpu()
        got = (hi.double() + lo.double()).permute(0, 3, 1, 2)[:, :c]
        tol = ref.abs() * 2.0 ** -15 + 3e-5 * max(1.0, float(ref.abs().max()))           # fp32 accumulation noise dominates
        assert bool(((got - ref).abs() <= tol).all()), (what, float(((got - ref).abs() - tol).max()))
        assert float((lo.abs() > hi.abs() * 2.0 ** -7 + 1e-30).float().mean()) < 1e-3, what         # the low parts are remainders of the high ones
        return hi

    plain = ops.conv2d(xin, w, b, cin=c, packed=blob.to(DEV))
    y = ops.conv2d(xin, w, b, cin=c, packed=blob.to(DEV), hilo=L.HILO_OUT)
    hi = check(y, conv, "head")
    assert torch.equal(hi[..., :plain.shape[-1]], plain.float().cpu()), "the high parts are the plain kernel's output"
    rin = _hilo(r32, cp).to(DEV)
    rsum = (rin[0].double() + rin[1].double()).cpu().permute(0, 3, 1, 2)[:, :c]
    y = ops.conv2d(xin, w, b, cin=c, packed=blob.to(DEV), res=rin, res_mode=1, hilo=L.HILO_RES | L.HILO_OUT)
    check(y, conv + rsum, "LR_conv")
    # upsampler: 48 output channels, pixel shuffle, hi + lo input
    wu = torch.randn(48, c, 3, 3, generator=g) * 0.1
    bu = torch.randn(48, generator=g)
    blobu = pack_conv_s16(wu, bu, "bf16", cin_phys=cp)
    weffu, _ = unpack_conv_s16(blobu, c, 48, 3, "bf16", cin_phys=cp)
    x32 = torch.randn(n, c, *hw, generator=g) * 2.0
    xh = _hilo(x32, cp).to(DEV)
    xsum = (xh[0].double() + xh[1].double()).cpu().permute(0, 3, 1, 2)[:, :c]
    ref = F.pixel_shuffle(F.conv2d(xsum, weffu.double(), bu.double(), padding=1), 4)
    yu = ops.conv2d(xh, wu, bu, cin=c, packed=blobu.to(DEV), shuffle_out=True, hilo=L.HILO_IN)
    assert yu.dtype == torch.float32 and tuple(yu.shape) == (n, 3, 4 * hw[0], 4 * hw[1])
    err = float((yu.double().cpu() - ref).abs().max())
    assert err <= 3e-5 * max(1.0, float(ref.abs().max())), err
    # and the same input WITHOUT its low parts is off by the bf16 rounding of x: the low half is really read
    y0 = ops.conv2d(xh[0], wu, bu, cin=c, packed=blobu.to(DEV), shuffle_out=True)
    assert float((y0.double().cpu() - ref).abs().max()) > 20 * err


def test_s16_hilo_rejects_what_it_does_not_cover():
    from ntire2022_esr_amd import _lib as L, ops
    x = torch.zeros(1, 20, 20, 32, dtype=torch.bfloat16, device=DEV)
    w, b = torch.zeros(32, 32, 3, 3), torch.zeros(32)
    with pytest.raises(L.EsrError):                      # two output tiles: no hi + lo kernel
        ops.conv2d(x, w, b, hilo=L.HILO_OUT)
    x = torch.zeros(1, 20, 20, 48, dtype=torch.float16, device=DEV)
    w, b = torch.zeros(48, 48, 3, 3), torch.zeros(48)
    with pytest.raises(L.EsrError):                      # fp16 storage: 11 bits already
        ops.conv2d(x, w, b, hilo=L.HILO_OUT)
    x = torch.zeros(1, 20, 20, 48, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(L.EsrError):                      # the input is not a pair
        ops.conv2d(x, w, b, hilo=L.HILO_IN)
    x = torch.zeros(2, 1, 20, 20, 48, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(L.EsrError):                      # 1x1: not covered
        ops.conv2d(x, torch.zeros(48, 48, 1, 1), b, hilo=L.HILO_IN)


@pytest.mark.parametrize("compute", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout,act,res_in,hw,n", [
    (64, 64, 1, False, (128, 128), 4),       # plain, LeakyReLU
    (50, 50, 1, True, (100, 77), 9),         # RFDB c{j}_r: lrelu(conv(x) + x), 50 logical channels in 64, ragged edges
    (64, 32, 1, False, (128, 128), 5),       # RFDB c4: two output tiles
    (50, 25, 1, False, (64, 250), 7),
    (64, 64, 0, True, (128, 128), 4)])       # residual == input without activation
def test_conv64r_equals_conv_s16(compute, cin, cout, act, res_in, hw, n):
    """The 3x3s over 64 physical input channels at >= 256 tiles of 16 x 16 (one wave per SIMD, weights in registers, row pairs as the outer
    loop, 160-byte LDS pixels) against conv_s16_kernel: the batch takes the register-resident kernel (esr_conv_block_waves == 1), each image
    alone the general one (< 256 tiles), same packed weights.  Two output tiles: conv64r_kernel, bit-identical results; four: conv64m_kernel
    (round 6, 32x32x16 MFMAs and nine taps: another accumulation order), equal to one storage step.  Both within storage precision of the
    fp64 convolution."""
    from ntire2022_esr_amd import ops, _lib as L
    from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16
    dt = DT[compute]
    g = torch.Generator().manual_seed(cin + cout + act + hw[0] + n)
    cp = 64
    x = torch.randn(n, cin, *hw, generator=g).to(dt)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    b = torch.randn(cout, generator=g)
    blob = pack_conv_s16(w, b, compute, cin_phys=cp).to(DEV)
    xin = F.pad(_nhwc(x), (0, cp - cin)).to(DEV)
    kw = dict(act=act, cin=cin, packed=blob)
    if res_in:
        kw.update(res=xin, res_mode=L.RES_PRE_ACT)
    d = L.ConvDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.ksize = n, hw[0], hw[1], cin, cout, 3
    d.in_layout = d.out_layout = L.NHWC
    d.storage = L.STORE[compute]
    assert L.lib().esr_conv_block_waves(ctypes.byref(d)) == 1
    d.n = 1
    assert L.lib().esr_conv_block_waves(ctypes.byref(d)) != 1
    y = ops.conv2d(xin, w, b, **kw)
    for i in range(n):
        kw1 = dict(kw)
        if res_in:
            kw1["res"] = xin[i:i + 1]
        y1 = ops.conv2d(xin[i:i + 1].contiguous(), w, b, **kw1)
        if cout > 48:      # 64 outputs: the batch runs on conv64m_kernel (32x32x16 MFMAs, nine taps, another accumulation order) since round 6
            assert _same_or_one_step(y[i:i + 1], y1, dt), i
        else:
            assert torch.equal(y[i:i + 1], y1), i
    weff, _ = unpack_conv_s16(blob.cpu(), cin, cout, 3, compute, cin_phys=cp)
    conv = F.conv2d(x.double().to(DEV), weff.double().to(DEV), b.double().to(DEV), padding=1)
    if res_in:
        conv = conv + x.double().to(DEV)[:, :cout]
    ref = ACTS[act](conv)
    got = y.permute(0, 3, 1, 2)[:, :cout].double()
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    tol = ref.abs() * eps * 1.01 + (3e-4 if act == 3 else 5e-5) * max(1.0, float(ref.abs().max()))
    assert int(((got - ref).abs() > tol).sum()) == 0


@pytest.mark.parametrize("hw,n,c,act", [((128, 128), 8, 48, 0), ((100, 77), 9, 46, 0), ((64, 250), 5, 46, 1)])
def test_conv48rl_equals_conv_s16(hw, n, c, act):
    """conv48rp_kernel<bf16, LRS> (the LR conv of a 48-channel network on hi + lo pairs: residual pair staged by each wave for its own rows,
    (conv + hi) + lo, hi / lo stores) against conv_s16_kernel's HILO instantiation: the batch takes the new kernel (>= 256 tiles of
    16 x 16, esr_conv_block_waves == 1), each image alone the old one -- both tensors of the output pair bit-identical, ragged edges
    included; and hi + lo is the fp64 result to two bf16 numbers."""
    from ntire2022_esr_amd import _lib as L, ops
    from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16
    g = torch.Generator().manual_seed(n * 100 + c + hw[0])
    cp = 48
    x = torch.randn(n, c, *hw, generator=g).to(torch.bfloat16)
    r32 = torch.randn(n, c, *hw, generator=g) * 3.0
    w = torch.randn(c, c, 3, 3, generator=g) * 0.1
    b = torch.randn(c, generator=g)
    blob = pack_conv_s16(w, b, "bf16", cin_phys=cp).to(DEV)
    weff, _ = unpack_conv_s16(blob.cpu(), c, c, 3, "bf16", cin_phys=cp)
    xin = F.pad(_nhwc(x), (0, cp - c)).to(DEV)
    rin = _hilo(r32, cp).to(DEV)
    d = L.ConvDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.ksize = n, hw[0], hw[1], c, c, 3
    d.in_layout = d.out_layout = L.NHWC
    d.storage, d.act, d.res_mode, d.hilo, d.hilo_stride = L.STORE["bf16"], act, L.RES_PRE_ACT, L.HILO_RES | L.HILO_OUT, 4096
    d.res = L.View(ctypes.c_void_p(rin.data_ptr()), cp, 0)
    assert L.lib().esr_conv_block_waves(ctypes.byref(d)) == 1
    d.n = 1
    assert L.lib().esr_conv_block_waves(ctypes.byref(d)) == 8
    kw = dict(cin=c, packed=blob, act=act, res_mode=L.RES_PRE_ACT, hilo=L.HILO_RES | L.HILO_OUT)
    y = ops.conv2d(xin, w, b, res=rin, **kw)
    assert tuple(y.shape) == (2, n, *hw, cp)
    for i in range(n):
        y1 = ops.conv2d(xin[i:i + 1].contiguous(), w, b, res=rin[:, i:i + 1].contiguous(), **kw)
        assert torch.equal(y[:, i:i + 1], y1), i
    rsum = (rin[0].double() + rin[1].double()).permute(0, 3, 1, 2)[:, :c]
    ref = ACTS[act](F.conv2d(x.double().to(DEV), weff.double().to(DEV), b.double().to(DEV), padding=1) + rsum)
    got = (y[0].double() + y[1].double()).permute(0, 3, 1, 2)[:, :c]
    tol = ref.abs() * 2.0 ** -15 + 3e-5 * max(1.0, float(ref.abs().max()))
    assert int(((got - ref).abs() > tol).sum()) == 0


@pytest.mark.parametrize("c,pc,hw", [(50, 25, (40, 56)), (48, 24, (64, 80))])
def test_s16_hilo_head_with_post(c, pc, hw):
    """ESR_HILO_OUT + one post 1x1 (the head of RFDN / BSRN in bf16: block 1's first distillation conv in its epilogue): the pair's high
    parts and the post output are the plain post-chain kernel's, bit for bit; hi + lo carries the fp32 result."""
    from ntire2022_esr_amd import _lib as L, ops
    g = torch.Generator().manual_seed(c + hw[0])
    x = torch.randn(2, hw[0], hw[1], 16, generator=g).to(torch.bfloat16).to(DEV)
    w, b = torch.randn(c, 9, 3, 3, generator=g) * 0.2, torch.randn(c, generator=g)
    wp, bp = torch.randn(pc, c, generator=g) * 0.2, torch.randn(pc, generator=g)
    y0, p0 = ops.conv2d(x, w, b, cin=9, post_weight=wp, post_bias=bp, post_act=1)
    y1, p1 = ops.conv2d(x, w, b, cin=9, post_weight=wp, post_bias=bp, post_act=1, hilo=L.HILO_OUT)
    assert y1.dim() == 5 and torch.equal(y1[0][..., :y0.shape[-1]], y0) and torch.equal(p1, p0)
    lo = y1[1].float()
    assert float(lo.abs().max()) > 0 and bool((lo.abs() <= y1[0].float().abs() * 2.0 ** -7 + 1e-30).all())


@pytest.mark.parametrize("hw,n,border,act,pact", [((128, 128), 8, True, 3, 3), ((100, 77), 9, False, 3, 3), ((64, 250), 5, True, 1, 1)])
def test_conv48rq_equals_conv_s16(hw, n, border, act, pact):
    """conv48rq_kernel (fp16: ESDB c{j}_r as a dense BSConvU + input + GELU, stored, with the next distillation 1x1 + GELU as 87 micro-operations
    behind the next row pair's MFMAs) against conv_s16_kernel<3, 3, 8, .., 2, 0>: the batch takes the new kernel (>= 256 tiles of 16 x 16,
    esr_conv_block_waves == 1), each image alone the old one -- both outputs bit-identical, ragged edges and the border table included."""
    from ntire2022_esr_amd import ops, _lib as L
    from ntire2022_esr_amd.engine import pack_conv_s16
    g = torch.Generator().manual_seed(n * 10 + hw[0])
    c, pc = 48, 24
    x = torch.randn(n, *hw, c, generator=g).to(torch.float16).to(DEV)
    w, b = torch.randn(c, c, 3, 3, generator=g) * 0.1, torch.randn(c, generator=g)
    wp, bp = torch.randn(pc, c, generator=g) * 0.2, torch.randn(pc, generator=g)
    table = None
    if border:
        table = torch.randn(16, 48, generator=g) * 0.2
        table[0] = 0
        table = table.to(DEV)
    blob = pack_conv_s16(w, b, "f16").to(DEV)
    kw = dict(act=act, packed=blob, border=table, res_mode=L.RES_PRE_ACT, post_weight=wp, post_bias=bp, post_act=pact)
    y, yp = ops.conv2d(x, w, b, res=x, **kw)
    d = L.ConvDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.ksize = n, hw[0], hw[1], c, c, 3
    d.in_layout = d.out_layout = L.NHWC
    d.storage, d.act, d.res_mode, d.post_cout = L.STORE["f16"], act, L.RES_PRE_ACT, pc
    d.inp = d.res = L.View(ctypes.c_void_p(x.data_ptr()), c, 0)
    d.out0 = L.View(ctypes.c_void_p(y.data_ptr()), y.shape[-1], 0)
    d.post_wpacked = ctypes.c_void_p(blob.data_ptr())                  # (any non-null pointer: the query does not read it)
    assert L.lib().esr_conv_block_waves(ctypes.byref(d)) == 1
    d.n = 1
    assert L.lib().esr_conv_block_waves(ctypes.byref(d)) == 8
    for i in range(n):
        xi = x[i:i + 1].contiguous()
        y1, p1 = ops.conv2d(xi, w, b, res=xi, **kw)
        if border and act == 3 and pact == 3:
            # (the batch: conv64m_kernel<false, true, false, 3, true> since round 6 -- another accumulation order, see test_conv48r_equals_conv_s16)
            assert _same_or_one_step(y[i:i + 1], y1, torch.float16) and _same_or_one_step(yp[i:i + 1], p1, torch.float16, frac=0.2, scale=10.0), i
        else:
            assert torch.equal(y[i:i + 1], y1) and torch.equal(yp[i:i + 1], p1), i
    assert bool(torch.isfinite(y.float()).all()) and float(yp.float().abs().max()) > 0


@pytest.mark.parametrize("compute", ["bf16", "f16"])
@pytest.mark.parametrize("hw,n,c,pc,res_in", [((128, 128), 8, 50, 25, True), ((100, 77), 9, 64, 32, True), ((64, 250), 5, 50, 25, False)])
def test_conv64m_post_agrees_with_conv_s16(compute, hw, n, c, pc, res_in):
    """conv64m_kernel<.., POST> (esr_c64m.hip, round 6: RFDB c{j}_r = lrelu(conv(x) + x), stored, with c{j+1}_d + LeakyReLU in its epilogue, on
    v_mfma_f32_32x32x16; rounds 4 / 5: conv64rq_kernel) against conv_s16_kernel<4, 3, 8, .., 2, 0>: the batch takes the register-resident kernel
    (>= 256 tiles of 16 x 16), each image alone the general one -- same weights, same arithmetic in another fp32 accumulation order: the outputs
    agree to one step of the storage type, ragged edges included (the fp64-reference check of the new kernel: test_gpu_c64m.py)."""
    from ntire2022_esr_amd import ops, _lib as L
    from ntire2022_esr_amd.engine import pack_conv_s16
    dt = DT[compute]
    g = torch.Generator().manual_seed(n * 10 + hw[0] + c)
    x = F.pad(torch.randn(n, *hw, c, generator=g), (0, 64 - c)).to(dt).to(DEV)
    w, b = torch.randn(c, c, 3, 3, generator=g) * 0.1, torch.randn(c, generator=g)
    wp, bp = torch.randn(pc, c, generator=g) * 0.2, torch.randn(pc, generator=g)
    blob = pack_conv_s16(w, b, compute, cin_phys=64).to(DEV)
    kw = dict(act=1, cin=c, packed=blob, post_weight=wp, post_bias=bp, post_act=1)
    if res_in:
        kw.update(res_mode=L.RES_PRE_ACT)
    y, yp = ops.conv2d(x, w, b, **(dict(res=x) if res_in else {}), **kw)
    d = L.ConvDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.ksize = n, hw[0], hw[1], c, c, 3
    d.in_layout = d.out_layout = L.NHWC
    d.storage, d.act, d.post_cout, d.post_act = L.STORE[compute], 1, pc, 1
    d.inp = L.View(ctypes.c_void_p(x.data_ptr()), 64, 0)
    if res_in:
        d.res_mode, d.res = L.RES_PRE_ACT, d.inp
    d.out0 = L.View(ctypes.c_void_p(y.data_ptr()), y.shape[-1], 0)
    d.post_wpacked = ctypes.c_void_p(blob.data_ptr())
    assert L.lib().esr_conv_block_waves(ctypes.byref(d)) == 1
    d.n = 1
    assert L.lib().esr_conv_block_waves(ctypes.byref(d)) == 8
    for i in range(n):
        xi = x[i:i + 1].contiguous()
        y1, p1 = ops.conv2d(xi, w, b, **(dict(res=xi) if res_in else {}), **kw)
        # (round 6: the batch runs on conv64m_kernel -- same weights and arithmetic in another accumulation order)
        # (fp16: the 1x1 reads the ROUNDED activations, so a value of y that flipped by one step -- 1e-3 at |y| ~ 1 -- moves every post output it
        # feeds by up to |w| x that: a few steps of a small output)
        assert _same_or_one_step(y[i:i + 1], y1, dt) and _same_or_one_step(yp[i:i + 1], p1, dt, 0.03, 1.0 if compute == "bf16" else 10.0), i
    assert bool(torch.isfinite(y.float()).all()) and float(yp.float().abs().max()) > 0
