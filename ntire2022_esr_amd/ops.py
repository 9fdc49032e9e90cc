"""Single-op entry points over the C ABI (used by the parity tests and by callers that
want one fused convolution rather than a whole network).  Tensors are torch CUDA tensors;
only their data_ptr()/stream cross the boundary."""
import ctypes

import torch

from . import _lib as L
from .engine import pack_conv, pack_conv_s16, pack_post_s16, pack_wino


def _view(t, coff=0):
    """NHWC tensor [N,H,W,pitch] -> esr_view at channel offset coff."""
    return L.View(ctypes.c_void_p(t.data_ptr()), t.shape[-1], coff)


_STORE_OF = {torch.float32: "f32", torch.bfloat16: "bf16", torch.float16: "f16"}


def conv2d(x, weight, bias, *, act=L.ACT_NONE, slope=0.05, res=None, res_mode=L.RES_NONE,
           in_nchw=False, shuffle_out=False, out=None, out_coff=0, in_coff=0, cin=None,
           split=0, out1=None, out1_coff=0, res_coff=0, packed=None, cin_map=None, store=None,
           tail_weight=None, tail_bias=None, tail_cat=None, tail_cat_coff=0, tail_mid_act=L.ACT_NONE,
           post_weight=None, post_bias=None, post_act=L.ACT_NONE, post2_weight=None, post2_bias=None, store_main=True,
           border=None, blocked_in=False, blocked_out1=False, wino=False, hilo=0):
    """Fused conv (k=1|3, stride 1, same padding) on the current stream.

    x       NHWC [N,H,W,pitch] (channels [in_coff, in_coff+cin) are read) or NCHW fp32 if in_nchw.  The dtype of an NHWC
            x is the storage type of the op: float32 -> exact fp32 MFMA; bfloat16 / float16 -> conv_s16_kernel (16-bit
            operands as stored, fp32 accumulate, one rounding at the store); res / out / out1 have the same dtype.
    store   NCHW input only: "bf16" | "f16" makes the NHWC output 16-bit (the head of a 16-bit network)
    returns NHWC [N,H,W,cout] (or `out`), or NCHW fp32 [N,cout/16,4H,4W] if shuffle_out
    post_*  esr_conv_desc.post_*: post_weight [pc, cout(, 1, 1)] applied to this conv's activated output; returns (y, post).
            16-bit storage: applied to the finished fp32 result (residual included); post2_weight [pc2, pc] chains a second 1x1
            on the first (returns (y, post, post2)); store_main=False does not store y (returns None in its place)
    border  esr_conv_desc.border_bias: fp32 [16, round_up(cout, 16)] table added by outside-mask (16-bit storage only)
    blocked_in / blocked_out1   esr_conv_desc.blocked8: x / out1 is a channel-blocked fp32 tensor [N, C/8, H, W, 8]
    hilo    esr_conv_desc.hilo (bf16, 3x3, 33..64 output channels): L.HILO_IN -- x is a contiguous [2, N, H, W, P] pair (value = x[0] + x[1]:
            high parts, low parts); L.HILO_RES -- so is res; L.HILO_OUT -- so will y be ([2, N, H, W, round_up(cout, 16)]); all pairs of one
            call must have the same x.stride(0)
    wino    fp32 3x3: also pass Winograd F(2x2, 3x3) weights (esr_conv_desc.wino_wpacked); raises if the shape does not qualify
    tail_*  fused 1x1 (esr_conv_desc.tail_*): tail_weight [cout1, cat_c + 16(, 1, 1)] over concat(tail_cat slice,
            mid_act(this 3x3 conv)); act / res / out then apply to the 1x1, whose output is returned
    """
    if not x.is_cuda:
        raise L.EsrError("conv2d: tensors must live on the GPU; there is no CPU fallback")
    lib = L.lib()
    w4 = weight if weight.dim() == 4 else weight[:, :, None, None]
    cout, wcin, k, _ = w4.shape
    st = _STORE_OF[x.dtype] if not in_nchw else (store or "f32")
    s16 = st != "f32" and not in_nchw
    if packed is None:
        packed = (pack_conv_s16(weight, bias, st, cin_map=cin_map) if s16 else pack_conv(weight, bias, cin_map=cin_map)).to(x.device)
    odt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[st]
    gran = 4 if st == "f32" else 8
    d = L.ConvDesc()
    d.storage = L.STORE[st]
    d.compute = L.COMPUTE[st] if s16 else 0
    hilo_strides = []
    if in_nchw:
        n, c, h, w = x.shape
        d.in_layout, cin = L.NCHW_IN, c
        d.inp = L.View(ctypes.c_void_p(x.data_ptr()), 0, 0)
    elif blocked_in:
        if x.dim() != 5 or x.shape[-1] != 8 or x.dtype != torch.float32 or not x.is_contiguous():
            raise L.EsrError("conv2d: a blocked input is a contiguous fp32 [N, C/8, H, W, 8] tensor")
        n, pl, h, w, _ = x.shape
        cin = wcin if cin is None else cin
        d.in_layout = L.NHWC
        d.inp = L.View(ctypes.c_void_p(x.data_ptr()), pl * 8, in_coff)
        d.blocked8 |= L.BLOCKED_IN
    elif hilo & L.HILO_IN:
        if x.dim() != 5 or x.shape[0] != 2 or not x.is_contiguous():
            raise L.EsrError("conv2d: a hi + lo input is a contiguous [2, N, H, W, P] pair")
        _, n, h, w, _ = x.shape
        cin = wcin if cin is None else cin
        d.in_layout = L.NHWC
        d.inp = _view(x[0], in_coff)
        _hilo_pair(x, "input", hilo_strides)
    elif x.dim() == 5:
        # planar concat [S, N, H, W, P]: S dense tensors one stride apart (esr_conv_desc.in_seg_stride / in_seg_chunks)
        if not s16 or not x.is_contiguous() or x.shape[-1] % 16:
            raise L.EsrError("conv2d: a segmented input is a contiguous 16-bit [S, N, H, W, P] tensor with P a multiple of 16")
        nseg, n, h, w, pp = x.shape
        cin = wcin if cin is None else cin
        d.in_layout = L.NHWC
        d.inp = L.View(ctypes.c_void_p(x.data_ptr()), pp, 0)
        d.in_seg_stride, d.in_seg_chunks = x.stride(0) * x.element_size(), pp // 16
    else:
        n, h, w, _ = x.shape
        cin = wcin if cin is None else cin
        d.in_layout = L.NHWC
        d.inp = _view(x, in_coff)
    d.n, d.h, d.w, d.cin, d.cout, d.ksize = n, h, w, cin, cout, k
    keep = None
    if tail_weight is not None:
        tw = tail_weight if tail_weight.dim() == 4 else tail_weight[:, :, None, None]
        keep = pack_conv(tw, tail_bias).to(x.device)
        d.tail_wpacked = ctypes.c_void_p(keep.data_ptr())
        d.tail_cat = _view(tail_cat, tail_cat_coff)
        d.tail_cat_c, d.tail_cout, d.tail_mid_act = tw.shape[1] - 16, tw.shape[0], tail_mid_act
        cout = tw.shape[0]                      # what the epilogue stores
    d.act, d.slope, d.res_mode, d.split = act, slope, res_mode, split
    if border is not None:
        if border.dtype != torch.float32 or not border.is_contiguous() or tuple(border.shape) != (16, (cout + 15) // 16 * 16):
            raise L.EsrError("conv2d: border must be a contiguous fp32 [16, round_up(cout, 16)] tensor")
        d.border_bias = ctypes.c_void_p(border.data_ptr())
    if shuffle_out:
        y = torch.empty((n, cout // 16, 4 * h, 4 * w), dtype=torch.float32, device=x.device) if out is None else out
        d.out_layout = L.NCHW_SHUFFLE4
        d.out0 = L.View(ctypes.c_void_p(y.data_ptr()), 0, 0)
    else:
        d.out_layout = L.NHWC
        if not store_main:
            y = None
        else:
            if out is None:
                c_store = (min(split, cout) if split else cout)
                y = torch.zeros(((2,) if hilo & L.HILO_OUT else ()) + (n, h, w, (c_store + gran - 1) // gran * gran if not hilo & L.HILO_OUT else (cout + 15) // 16 * 16),
                                dtype=odt, device=x.device)
            else:
                y = out
            if hilo & L.HILO_OUT:
                if y.dim() != 5 or y.shape[0] != 2 or not y.is_contiguous():
                    raise L.EsrError("conv2d: a hi + lo output is a contiguous [2, N, H, W, P] pair")
                d.out0 = _view(y[0], out_coff)
                _hilo_pair(y, "output", hilo_strides)
            else:
                d.out0 = _view(y, out_coff)
        if out1 is not None:
            if blocked_out1:
                if out1.dim() != 5 or out1.shape[-1] != 8 or out1.dtype != torch.float32 or not out1.is_contiguous():
                    raise L.EsrError("conv2d: a blocked out1 is a contiguous fp32 [N, C/8, H, W, 8] tensor")
                d.out1 = L.View(ctypes.c_void_p(out1.data_ptr()), out1.shape[1] * 8, out1_coff)
                d.blocked8 |= L.BLOCKED_OUT1
            else:
                d.out1 = _view(out1, out1_coff)
    if res is not None and hilo & L.HILO_RES:
        if res.dim() != 5 or res.shape[0] != 2 or not res.is_contiguous():
            raise L.EsrError("conv2d: a hi + lo residual is a contiguous [2, N, H, W, P] pair")
        d.res = _view(res[0], res_coff)
        _hilo_pair(res, "residual", hilo_strides)
    elif res is not None:
        d.res = _view(res, res_coff)
    d.hilo = hilo
    if hilo_strides:
        # ONE stride field serves every pair of the descriptor (esr_conv_desc.hilo_stride): pairs of different geometry would make the
        # kernel address a low tensor at the wrong place
        if len(set(hilo_strides)) != 1:
            raise L.EsrError(f"conv2d: the hi + lo pairs of one call must have the same stride between their halves, got {sorted(set(hilo_strides))}")
        d.hilo_stride = hilo_strides[0]
    d.wpacked = ctypes.c_void_p(packed.data_ptr())
    if wino:
        keepw = pack_wino(w4, bias, cin_map=cin_map).to(x.device)
        if not lib.esr_wino_supported(ctypes.byref(d)):
            raise L.EsrError("conv2d: this descriptor does not qualify for the Winograd kernel (esr_wino_supported)")
        d.wino_wpacked = ctypes.c_void_p(keepw.data_ptr())
    yp = yp2 = None
    if post_weight is not None:
        pw = post_weight if post_weight.dim() == 4 else post_weight[:, :, None, None]
        keep2 = (pack_post_s16(pw, post_bias, st) if s16 else pack_conv(pw, post_bias)).to(x.device)
        yp = torch.zeros((n, h, w, (pw.shape[0] + gran - 1) // gran * gran), dtype=odt, device=x.device)
        d.post_wpacked, d.post_out = ctypes.c_void_p(keep2.data_ptr()), _view(yp)
        d.post_cout, d.post_act = pw.shape[0], post_act
        if post2_weight is not None:
            keep3 = pack_post_s16(post2_weight, post2_bias, st).to(x.device)
            yp2 = torch.zeros((n, h, w, (post2_weight.shape[0] + 7) // 8 * 8), dtype=odt, device=x.device)
            d.post2_wpacked, d.post2_out, d.post2_cout = ctypes.c_void_p(keep3.data_ptr()), _view(yp2), post2_weight.shape[0]
    stream = torch.cuda.current_stream(x.device).cuda_stream
    L.check(lib.esr_conv2d_f32(ctypes.byref(d), ctypes.c_void_p(stream)), "esr_conv2d_f32")
    if yp is None:
        return y
    return (y, yp) if yp2 is None else (y, yp, yp2)


def conv_chain(x, weights, biases, post_weight, post_bias, post2_weight, post2_bias, *, act=L.ACT_LRELU, slope=0.05,
               res_mode=L.RES_POST_ACT, post_act=L.ACT_NONE, cin=None):
    """esr_conv_chain_s16 (ABI v11): a residual block's chain of 3x3 convolutions in ONE launch on a 16-bit NHWC tensor x [N, H, W, P]
    (RLFB.forward, team04_rlfn.py:109-122): t = x; t = act(conv_i(t)) for all but the last 3x3; u = act(conv_n(t)) + x;
    v = post_act(post_weight . u + post_bias) -> stored; c1 = post2_weight . v_fp32 + post2_bias -> stored.  Returns (v, c1).
    weights: list of OIHW fp32 3x3 weights, biases: list of fp32 biases (or None)."""
    if not x.is_cuda:
        raise L.EsrError("conv_chain: tensors must live on the GPU; there is no CPU fallback")
    st = _STORE_OF[x.dtype]
    if st == "f32":
        raise L.EsrError("conv_chain: 16-bit storage only")
    lib = L.lib()
    n, h, w, _ = x.shape
    d = L.ChainDesc()
    d.n, d.h, d.w, d.n_layers = n, h, w, len(weights)
    d.cin = weights[0].shape[1] if cin is None else cin
    d.cmid, d.cout = weights[0].shape[0], weights[-1].shape[0]
    d.act, d.slope, d.res_mode = act, slope, res_mode
    d.storage = d.compute = L.STORE[st]
    d.inp = _view(x)
    keep = []
    for i, (wt, b) in enumerate(zip(weights, biases)):
        blob = pack_conv_s16(wt, b, st, cin_phys=(wt.shape[1] + 15) // 16 * 16).to(x.device)
        keep.append(blob)
        d.wpacked[i] = blob.data_ptr()
    pw = post_weight if post_weight.dim() == 4 else post_weight[:, :, None, None]
    p1 = pack_post_s16(pw, post_bias, st).to(x.device)
    p2 = pack_post_s16(post2_weight, post2_bias, st).to(x.device)
    v = torch.zeros((n, h, w, (pw.shape[0] + 15) // 16 * 16), dtype=x.dtype, device=x.device)
    c1 = torch.zeros((n, h, w, (post2_weight.shape[0] + 7) // 8 * 8), dtype=x.dtype, device=x.device)
    d.post_wpacked, d.post_out, d.post_cout, d.post_act = ctypes.c_void_p(p1.data_ptr()), _view(v), pw.shape[0], post_act
    d.post2_wpacked, d.post2_out, d.post2_cout = ctypes.c_void_p(p2.data_ptr()), _view(c1), post2_weight.shape[0]
    if not lib.esr_conv_chain_supported(ctypes.byref(d)):
        raise L.EsrError("conv_chain: no kernel for this shape (esr_conv_chain_supported)")
    stream = torch.cuda.current_stream(x.device).cuda_stream
    L.check(lib.esr_conv_chain_s16(ctypes.byref(d), ctypes.c_void_p(stream)), "esr_conv_chain_s16")
    return v, c1


def _hilo_pair(t, what, strides):
    """checks a hi + lo pair [2, N, H, W, P] (bf16, P a multiple of 16) and records the byte stride between its halves"""
    if t.dtype != torch.bfloat16 or t.shape[-1] % 16:
        raise L.EsrError(f"conv2d: a hi + lo {what} is a bf16 pair with a pixel pitch that is a multiple of 16 channels")
    strides.append(t.stride(0) * t.element_size())


def tensor2uint_device(img_sr, data_range, nonfinite=None):
    """utils_image.tensor2uint on the GPU: [1,C,H,W] (or [C,H,W]) fp32 -> HWC uint8 tensor on the same device.
    nonfinite: optional 1-element int32 DEVICE tensor (caller-zeroed) that the kernel ORs 1 into when the image holds an Inf / NaN
    (esr_tensor2uint_u8_chk) -- the harness's overflow check without a full-size isfinite pass."""
    if not img_sr.is_cuda:
        raise L.EsrError("tensor2uint_device: tensor must live on the GPU")
    t = img_sr.detach()
    if t.dim() == 4:
        assert t.shape[0] == 1, "one image at a time, like the reference's run()"
        t = t[0]
    t = t.contiguous().float()
    c, h, w = t.shape
    out = torch.empty((h, w, c), dtype=torch.uint8, device=t.device)
    stream = torch.cuda.current_stream(t.device).cuda_stream
    if nonfinite is not None:
        assert nonfinite.is_cuda and nonfinite.dtype == torch.int32 and nonfinite.numel() == 1
        L.check(L.lib().esr_tensor2uint_u8_chk(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(out.data_ptr()), c, h, w, ctypes.c_float(data_range),
                                               ctypes.c_void_p(nonfinite.data_ptr()), ctypes.c_void_p(stream)), "esr_tensor2uint_u8_chk")
        return out
    L.check(L.lib().esr_tensor2uint_u8(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(out.data_ptr()), c, h, w,
                                       ctypes.c_float(data_range), ctypes.c_void_p(stream)), "esr_tensor2uint_u8")
    return out


def ssim_sum_device(a_u8, b_u8, border=0):
    """calculate_ssim's numerator on the GPU (utils/utils_image.py:509-554, esr_ssim_u8): the SSIM map of two HWC (or HW) uint8 CUDA
    tensors summed over the 'valid' region of the border-cropped image and all channels, as a 0-dim float64 DEVICE tensor (no host
    synchronisation), and the number of map elements it was summed over.  ssim = sum / count."""
    if a_u8.shape != b_u8.shape:
        raise ValueError('Input images must have the same dimensions.')
    a, b = a_u8.contiguous(), b_u8.contiguous()
    h, w = a.shape[:2]
    c = a.shape[2] if a.dim() == 3 else 1
    if a.dim() not in (2, 3) or c not in (1, 3):
        raise ValueError('Wrong input image dimensions.')
    n = int(L.lib().esr_ssim_partials(h, w, c, border))
    if n == 0:
        raise L.EsrError(f"ssim_device: a {h}x{w} image cropped by {border} is smaller than the 11x11 window")
    partials = torch.empty(n, dtype=torch.float64, device=a.device)
    stream = torch.cuda.current_stream(a.device).cuda_stream
    L.check(L.lib().esr_ssim_u8(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), h, w, c, border,
                                ctypes.c_void_p(partials.data_ptr()), n, ctypes.c_void_p(stream)), "esr_ssim_u8")
    return partials.sum(), (h - 2 * border - 10) * (w - 2 * border - 10) * c


def ssim_device(a_u8, b_u8, border=0):
    """calculate_ssim for two uint8 CUDA tensors: one scalar D2H.  Parity: pinned to image_util.calculate_ssim (the reference's needs
    cv2, absent in the authoring container: PARITY-UNPINNED against the reference itself)."""
    s, count = ssim_sum_device(a_u8, b_u8, border)
    return float(s.item()) / count


def sqerr_device(a_u8, b_u8, border=0):
    """sum over the border-cropped region of (a - b)^2 for two HWC uint8 CUDA tensors as a 1-element int64 DEVICE tensor:
    no host synchronisation (the harness pipeline reads it when the image retires)."""
    a, b = a_u8.contiguous(), b_u8.contiguous()
    h, w = a.shape[:2]
    c = a.shape[2] if a.dim() == 3 else 1
    acc = torch.empty(1, dtype=torch.int64, device=a.device)
    stream = torch.cuda.current_stream(a.device).cuda_stream
    L.check(L.lib().esr_sqerr_u8(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), h, w, c, border,
                                 ctypes.c_void_p(acc.data_ptr()), ctypes.c_void_p(stream)), "esr_sqerr_u8")
    return acc


def psnr_device(a_u8, b_u8, border=0):
    """calculate_psnr for two HWC uint8 CUDA tensors: exact integer squared-error sum on the device, one scalar D2H."""
    import math
    if a_u8.shape != b_u8.shape:
        raise ValueError('Input images must have the same dimensions.')
    a, b = a_u8.contiguous(), b_u8.contiguous()
    h, w = a.shape[:2]
    c = a.shape[2] if a.dim() == 3 else 1
    acc = torch.empty(1, dtype=torch.int64, device=a.device)
    stream = torch.cuda.current_stream(a.device).cuda_stream
    L.check(L.lib().esr_sqerr_u8(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), h, w, c, border,
                                 ctypes.c_void_p(acc.data_ptr()), ctypes.c_void_p(stream)), "esr_sqerr_u8")
    se = int(acc.item())
    count = (h - 2 * border) * (w - 2 * border) * c
    if se == 0:
        return float('inf')
    return 20 * math.log10(255.0 / math.sqrt(se / count))


def bsconv(x, pw_weight, pw_bias, dw_weight, dw_bias, *, act=L.ACT_NONE, slope=0.05, res=None, res_mode=L.RES_NONE,
           in_coff=0, cin=None, d_weight=None, d_bias=None, d_act=L.ACT_NONE):
    """BSConvU in one launch (esr_bsconv_f32): act(dw3x3(pw1x1(x)) [+ res]); returns y, or (y, distilled) when the
    distillation 1x1 `d_weight` [d_cout, cin] is given.  x: NHWC [N,H,W,pitch] fp32 / bfloat16 / float16 (the storage
    type of the op: res and the outputs have the same dtype); pw_weight [c, cin]; dw_weight [c,1,3,3]."""
    from .engine import pack_dw
    if not x.is_cuda:
        raise L.EsrError("bsconv: tensors must live on the GPU; there is no CPU fallback")
    lib = L.lib()
    st = _STORE_OF[x.dtype]
    n, h, w, _ = x.shape
    c, wcin = pw_weight.shape[0], pw_weight.shape[1]
    cin = wcin if cin is None else cin

    def pk(wt, b):
        w4 = wt.reshape(wt.shape[0], wcin, 1, 1)
        return (pack_conv(w4, b) if st == "f32" else pack_conv_s16(w4, b, st)).to(x.device)

    keep = [pk(pw_weight, pw_bias), pack_dw(dw_weight, dw_bias).to(x.device)]
    d = L.BsDesc()
    d.storage = L.STORE[st]
    d.n, d.h, d.w, d.cin, d.c = n, h, w, cin, c
    d.act, d.slope, d.res_mode = act, slope, res_mode
    d.inp = _view(x, in_coff)
    y = torch.zeros((n, h, w, (c + 3) // 4 * 4), dtype=x.dtype, device=x.device)
    d.out = _view(y)
    if res is not None:
        d.res = _view(res)
    d.pw_packed, d.dw_packed = ctypes.c_void_p(keep[0].data_ptr()), ctypes.c_void_p(keep[1].data_ptr())
    yd = None
    if d_weight is not None:
        dco = d_weight.shape[0]
        keep.append(pk(d_weight, d_bias))
        yd = torch.zeros((n, h, w, (dco + 3) // 4 * 4), dtype=x.dtype, device=x.device)
        d.d_packed, d.d_cout, d.d_act, d.d_out = ctypes.c_void_p(keep[2].data_ptr()), dco, d_act, _view(yd)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    L.check(lib.esr_bsconv_f32(ctypes.byref(d), ctypes.c_void_p(stream)), "esr_bsconv_f32")
    return y if yd is None else (y, yd)


def channel_attention(x, w1, b1, w2, b2, *, contrast=False, nchw=False, out=None):
    """CALayer / CCALayer (esr_channel_attention_f32): y = x * sigmoid(W2 . relu(W1 . s + b1) + b2) with s = mean over H, W
    (contrast=False, models/basicblock.py:333-348) or std + mean (contrast=True, models/team05_efdn/plainblock.py:106-122).
    x: NHWC [N,H,W,pitch] (fp32 / bf16 / fp16) or, with nchw=True, NCHW fp32 [N,C,H,W]; w1 [cr, c(,1,1)], w2 [c, cr(,1,1)]."""
    from .engine import pack_dense
    if not x.is_cuda:
        raise L.EsrError("channel_attention: tensors must live on the GPU; there is no CPU fallback")
    cr, c = w1.shape[0], w1.shape[1]
    c4 = (c + 3) // 4 * 4
    if nchw:
        n, _, h, w = x.shape
    else:
        n, h, w, _ = x.shape
    keep = [pack_dense(w1.reshape(cr, c, 1, 1), b1, c, cr).to(x.device), pack_dense(w2.reshape(c, cr, 1, 1), b2, cr, c4).to(x.device)]
    y = torch.zeros_like(x) if out is None else out
    stats = torch.empty(n * 2 * c4, dtype=torch.float64, device=x.device)
    d = L.CaDesc()
    d.n, d.h, d.w, d.c, d.cr, d.contrast = n, h, w, c, cr, int(contrast)
    d.layout = L.NCHW_IN if nchw else L.NHWC
    d.storage = L.STORE[_STORE_OF[x.dtype]]
    d.x = L.View(ctypes.c_void_p(x.data_ptr()), 0 if nchw else x.shape[-1], 0)
    d.y = L.View(ctypes.c_void_p(y.data_ptr()), 0 if nchw else y.shape[-1], 0)
    d.w1, d.w2, d.stats = keep[0].data_ptr(), keep[1].data_ptr(), stats.data_ptr()
    stream = torch.cuda.current_stream(x.device).cuda_stream
    L.check(L.lib().esr_channel_attention_f32(ctypes.byref(d), ctypes.c_void_p(stream)), "esr_channel_attention_f32")
    return y


def esa_apply(x, c1, c3, wf, bf, w4, b4, *, out=None, post=None, skip_y=False):
    """ESA's full-resolution tail in one launch (esr_esa_apply_f32): y = x * sigmoid(conv4(bilinear(c3 -> HxW) + conv_f(c1)))
    (models/rfdn_baseline/block.py:124-129).  x: NHWC [N,H,W,pitch] (fp32 / bf16 / fp16 storage), c channels = w4.shape[0];
    c1: NHWC [N,H,W,16] of the same dtype (conv1's output, f = wf.shape[0] <= 16 channels, pads zero); c3: fp32 NHWC [N,h_lo,w_lo,16];
    wf [f,f(,1,1)], w4 [c,f(,1,1)].
    post (16-bit storage): one or two dicts(weight [cout, cin(,1,1)], bias, act, slope, res) -- 1x1 convolutions evaluated in the same
    launch (esr_esa_desc.post[]): the first on y as stored (+ res, NHWC of x's dtype), the second on the first's fp32 result.  Returns
    (y, [out0(, out1)]) then; skip_y: y is not stored."""
    from .engine import pack_apply_post, pack_dense
    if not x.is_cuda:
        raise L.EsrError("esa_apply: tensors must live on the GPU; there is no CPU fallback")
    n, h, w, pitch = x.shape
    c, f = w4.shape[0], wf.shape[0]
    cp4 = (c + 3) // 4 * 4
    keep = [pack_dense(wf.reshape(f, f, 1, 1), bf, 16, 16).to(x.device), pack_dense(w4.reshape(c, f, 1, 1), b4, 16, cp4).to(x.device)]
    y = torch.zeros_like(x) if out is None else out
    d = L.EsaDesc()
    d.n, d.h, d.w, d.c, d.f, d.h_lo, d.w_lo = n, h, w, c, f, c3.shape[1], c3.shape[2]
    d.storage = L.STORE[_STORE_OF[x.dtype]]
    d.x, d.y = _view(x), _view(y)
    d.c1, d.c3, d.w0, d.w1 = c1.data_ptr(), c3.data_ptr(), keep[0].data_ptr(), keep[1].data_ptr()
    outs = []
    if post:
        p0, p1 = post[0], (post[1] if len(post) > 1 else None)
        keep.append(pack_apply_post(p0["weight"], p0.get("bias"), None if p1 is None else p1["weight"], None if p1 is None else p1.get("bias"),
                                    _STORE_OF[x.dtype]).to(x.device))
        d.post_w = keep[-1].data_ptr()
        d.skip_y = 1 if skip_y else 0
        for k, t in enumerate(post):
            co = t["weight"].shape[0]
            o = torch.zeros(n, h, w, (co + 7) // 8 * 8, dtype=x.dtype, device=x.device)
            outs.append(o)
            pp = d.post[k]
            pp.cout, pp.act, pp.slope = co, t.get("act", L.ACT_NONE), t.get("slope", 0.05)
            pp.out = _view(o)
            if t.get("res") is not None:
                pp.res_mode, pp.res = L.RES_PRE_ACT, _view(t["res"])
    stream = torch.cuda.current_stream(x.device).cuda_stream
    L.check(L.lib().esr_esa_apply_f32(ctypes.byref(d), ctypes.c_void_p(stream)), "esr_esa_apply_f32")
    return (y, outs) if post else y


# ---- torch.library operators over the kernels --------------------------------------------------------------------------------
# The "thin PyTorch-ROCm custom-op layer" of BASELINE.json's north star at KERNEL granularity (engine.py registers the whole-network
# op esr::sr_forward): every hot kernel family is a registered operator with a fake (shape) implementation, so FakeTensor tracing /
# torch.compile / export see opaque ops instead of ctypes calls.  Real implementations = the functions above; tensors are NHWC views
# as the C ABI sees them.  These replace nn.Conv2d + activation (+ residual) (models/basicblock.py:61-98), BSConvU
# (models/team18_bsrn.py:44-88), the ESA tail (models/rfdn_baseline/block.py:124-129) and CALayer / CCALayer
# (models/basicblock.py:333-348, models/team05_efdn/plainblock.py:106-122).
from typing import Optional  # noqa: E402

Tensor = torch.Tensor


def _pad_c(c, dtype):
    g = 4 if dtype == torch.float32 else 8
    return (c + g - 1) // g * g


@torch.library.custom_op("esr::conv2d", mutates_args=())
def conv2d_op(x: Tensor, weight: Tensor, bias: Optional[Tensor], act: int, slope: float, res: Optional[Tensor], res_mode: int,
              winograd: bool) -> Tensor:
    """act(conv(x) [+ res]) on NHWC x [N,H,W,pitch >= cin]; weight OIHW (k = 1 | 3); returns NHWC [N,H,W,pad(cout)].
    winograd: fp32 3x3 as Winograd F(2x2,3x3) when the shape qualifies (esr_wino_supported), else the direct kernel."""
    d = L.ConvDesc()
    d.ksize, d.in_layout, d.out_layout, d.cin, d.cout = weight.shape[2], L.NHWC, L.NHWC, weight.shape[1], weight.shape[0]
    d.inp = L.View(None, x.shape[-1], 0)
    d.res_mode = res_mode
    use_wino = bool(winograd) and x.dtype == torch.float32 and bool(L.lib().esr_wino_supported(ctypes.byref(d)))
    return conv2d(x, weight, bias, act=act, slope=slope, res=res, res_mode=res_mode, wino=use_wino)


@conv2d_op.register_fake
def _conv2d_fake(x, weight, bias, act, slope, res, res_mode, winograd):
    if x.dim() != 4 or weight.dim() != 4 or weight.shape[2] not in (1, 3) or weight.shape[1] > x.shape[-1]:
        raise L.EsrError("esr::conv2d: NHWC x [N,H,W,pitch >= cin], OIHW weight with k in {1, 3}")
    return x.new_empty((x.shape[0], x.shape[1], x.shape[2], _pad_c(weight.shape[0], x.dtype)))


@torch.library.custom_op("esr::bsconv", mutates_args=())
def bsconv_op(x: Tensor, pw_weight: Tensor, pw_bias: Optional[Tensor], dw_weight: Tensor, dw_bias: Optional[Tensor], act: int,
              slope: float, res: Optional[Tensor], res_mode: int) -> Tensor:
    """BSConvU in one launch: act(dw3x3(pw1x1(x)) [+ res]); NHWC in / out"""
    return bsconv(x, pw_weight, pw_bias, dw_weight, dw_bias, act=act, slope=slope, res=res, res_mode=res_mode)


@bsconv_op.register_fake
def _bsconv_fake(x, pw_weight, pw_bias, dw_weight, dw_bias, act, slope, res, res_mode):
    return x.new_empty((x.shape[0], x.shape[1], x.shape[2], (pw_weight.shape[0] + 3) // 4 * 4))


@torch.library.custom_op("esr::esa_apply", mutates_args=())
def esa_apply_op(x: Tensor, c1: Tensor, c3: Tensor, wf: Tensor, bf: Optional[Tensor], w4: Tensor, b4: Optional[Tensor]) -> Tensor:
    """x * sigmoid(conv4(bilinear(c3) + conv_f(c1))): ESA's full-resolution tail"""
    return esa_apply(x, c1, c3, wf, bf, w4, b4)


@esa_apply_op.register_fake
def _esa_apply_fake(x, c1, c3, wf, bf, w4, b4):
    return torch.empty_like(x)


@torch.library.custom_op("esr::channel_attention", mutates_args=())
def channel_attention_op(x: Tensor, w1: Tensor, b1: Optional[Tensor], w2: Tensor, b2: Optional[Tensor], contrast: bool,
                         nchw: bool) -> Tensor:
    """CALayer (contrast = False) / CCALayer (True): x * sigmoid(W2 . relu(W1 . s + b1) + b2)"""
    return channel_attention(x, w1, b1, w2, b2, contrast=contrast, nchw=nchw)


@channel_attention_op.register_fake
def _channel_attention_fake(x, w1, b1, w2, b2, contrast, nchw):
    return torch.empty_like(x)
