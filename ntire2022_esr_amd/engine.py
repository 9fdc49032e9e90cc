"""Host-side engine shared by the four networks: weight packing, workspace, op lists.

A network's forward is a flat `esr_op` list (include/esr_hip.h) built once per
(N, H, W, device) and replayed by ONE C call (`esr_run_ops`) on the caller's
current HIP stream -- so the reference's `start.record(); forward(); end.record()`
bracket (test_demo.py:429-432) times exactly the device work, and nothing here
synchronises the host.  PyTorch only provides device memory (caching allocator)
and the stream handle.
"""
import collections
import ctypes
import itertools
import threading
import weakref

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def pack_conv(weight, bias, cin_map=None, cin_phys=None):
    """OIHW (or [out,in] for nn.Linear) fp32 weights + bias -> packed blob (CPU float32 tensor).

    Host-side, no GPU needed.  `cin_map[s]` = logical input channel carried by physical
    slot s, or -1 for a zero pad slot (padded concat buffers)."""
    lib = L.lib()
    w = weight.detach().to("cpu", torch.float32).contiguous()
    if w.dim() == 2:
        w = w[:, :, None, None].contiguous()
    cout, cin, k, k2 = w.shape
    assert k == k2 and k in (1, 3), "only 1x1 / 3x3 kernels are on the hot path"
    b = None if bias is None else bias.detach().to("cpu", torch.float32).contiguous()
    if cin_map is not None:
        cm = np.ascontiguousarray(np.asarray(cin_map, dtype=np.int32))
        cin_phys = len(cm)
        cm_p = cm.ctypes.data_as(ctypes.c_void_p)
    else:
        cm, cm_p = None, None
        cin_phys = cin if cin_phys is None else cin_phys
    nbytes = lib.esr_packed_conv_bytes(cin_phys, cout, k)
    if nbytes == 0:
        raise L.EsrError(f"esr_packed_conv_bytes rejected cin_phys={cin_phys} cout={cout} k={k}")
    out = torch.empty(nbytes // 4, dtype=torch.float32)
    rc = lib.esr_pack_conv_f32(_ptr(w), _ptr(b) if b is not None else None, cin, cout, k,
                               cm_p, cin_phys, _ptr(out), nbytes)
    L.check(rc, "esr_pack_conv_f32")
    return out


def pack_wino(weight, bias, cin_map=None, cin_phys=None):
    """OIHW fp32 3x3 weights + bias -> the Winograd F(2x2, 3x3) blob of esr_pack_wino_f32 (U = G g G^T, computed in fp64 and
    rounded once, in wino_f32_kernel's A-operand order).  Host-side, no GPU needed."""
    lib = L.lib()
    w = weight.detach().to("cpu", torch.float32).contiguous()
    cout, cin, k, k2 = w.shape
    assert k == 3 and k2 == 3
    b = None if bias is None else bias.detach().to("cpu", torch.float32).contiguous()
    if cin_map is not None:
        cm = np.ascontiguousarray(np.asarray(cin_map, dtype=np.int32))
        cin_phys, cm_p = len(cm), cm.ctypes.data_as(ctypes.c_void_p)
    else:
        cm_p = None
        cin_phys = cin if cin_phys is None else cin_phys
    nbytes = lib.esr_packed_wino_bytes(cin_phys, cout)
    out = torch.empty(nbytes // 4, dtype=torch.float32)
    L.check(lib.esr_pack_wino_f32(_ptr(w), _ptr(b) if b is not None else None, cin, cout, cm_p, cin_phys, _ptr(out), nbytes),
            "esr_pack_wino_f32")
    return out


def unpack_wino(blob, cin, cout, cin_map=None, cin_phys=None):
    """U [cout, cin, 16] and bias of a pack_wino blob (tests)."""
    lib = L.lib()
    blob = blob.detach().to("cpu", torch.float32).contiguous()
    if cin_map is not None:
        cm = np.ascontiguousarray(np.asarray(cin_map, dtype=np.int32))
        cin_phys, cm_p = len(cm), cm.ctypes.data_as(ctypes.c_void_p)
    else:
        cm_p = None
        cin_phys = cin if cin_phys is None else cin_phys
    u = torch.empty(cout, cin, 16)
    b = torch.empty(cout)
    L.check(lib.esr_unpack_wino_f32(_ptr(blob), blob.numel() * 4, cin, cout, cm_p, cin_phys, _ptr(u), _ptr(b)), "esr_unpack_wino_f32")
    return u, b


def pack_conv_s16(weight, bias, compute, cin_map=None, cin_phys=None):
    """OIHW (or [out,in]) fp32 weights -> the 16-bit-storage blob of esr_pack_conv_s16 (bf16 or fp16; 3x3 taps rounded
    with error diffusion, 1x1 as hi + lo) + fp32 bias."""
    lib = L.lib()
    w = weight.detach().to("cpu", torch.float32).contiguous()
    if w.dim() == 2:
        w = w[:, :, None, None].contiguous()
    cout, cin, k, _ = w.shape
    assert k in (1, 3)
    b = None if bias is None else bias.detach().to("cpu", torch.float32).contiguous()
    if cin_map is not None:
        cm = np.ascontiguousarray(np.asarray(cin_map, dtype=np.int32))
        cin_phys, cm_p = len(cm), cm.ctypes.data_as(ctypes.c_void_p)
    else:
        cm_p = None
        cin_phys = cin if cin_phys is None else cin_phys
    nbytes = lib.esr_packed_conv_s16_bytes(cin_phys, cout, k)
    out = torch.empty((nbytes + 3) // 4, dtype=torch.float32)
    L.check(lib.esr_pack_conv_s16(_ptr(w), _ptr(b) if b is not None else None, cin, cout, k, cm_p, cin_phys,
                                  L.COMPUTE[compute] if isinstance(compute, str) else compute, _ptr(out), nbytes),
            "esr_pack_conv_s16")
    return out


def pack_head_s16(weight, bias, compute):
    """Weights of the network's first 3x3 (NCHW fp32 input, cin <= 4) for the 16-bit head: the input arrives as 16-bit slots
    [x_hi | x_lo | x_hi] (esr_pack_input_s16), the weights as [w_hi | w_hi | w_lo] with w = w_hi + w_lo in the storage type, so
    the products w_hi x_hi + w_hi x_lo + w_lo x_hi keep ~fp32 accuracy on the 16-bit matrix cores."""
    dt = torch.bfloat16 if compute == "bf16" else torch.float16
    w = weight.detach().float().cpu()
    hi = w.to(dt).float()
    lo = w - hi
    return pack_conv_s16(torch.cat([hi, hi, lo], dim=1), bias, compute)


def pack_post_s16(weight, bias, compute):
    """[cout, cin(, 1, 1)] fp32 weights of a 1x1 evaluated in a 16-bit conv's epilogue (esr_conv_desc.post_* / post2_*) ->
    esr_pack_post_s16 blob (MFMA images of the weights' 16-bit high and low parts + fp32 bias)."""
    lib = L.lib()
    w = weight.detach().to("cpu", torch.float32).reshape(weight.shape[0], -1).contiguous()
    cout, cin = w.shape
    b = None if bias is None else bias.detach().to("cpu", torch.float32).contiguous()
    nbytes = lib.esr_packed_post_s16_bytes(cin, cout)
    out = torch.empty((nbytes + 3) // 4, dtype=torch.float32)
    L.check(lib.esr_pack_post_s16(_ptr(w), _ptr(b) if b is not None else None, cin, cout,
                                  L.COMPUTE[compute] if isinstance(compute, str) else compute, _ptr(out), nbytes), "esr_pack_post_s16")
    return out


def pack_tail_s16(weight, bias, nseg, seg_c, mid_c, compute):
    """[cout, nseg * seg_c + mid_c(, 1, 1)] fp32 weights of the 1x1 of a 16-bit tail (esr_conv_desc.tail_*, ABI v12: RFDB's c5 over
    cat(d1, d2, d3, r4)) -> esr_pack_tail_s16 blob (hi / lo fragment images for v_mfma_f32_32x32x16 + fp32 bias)."""
    lib = L.lib()
    w = weight.detach().to("cpu", torch.float32).reshape(weight.shape[0], -1).contiguous()
    cout, kin = w.shape
    if kin != nseg * seg_c + mid_c:
        raise L.EsrError("pack_tail_s16: the 1x1 takes nseg * seg_c + mid_c inputs")
    b = None if bias is None else bias.detach().to("cpu", torch.float32).contiguous()
    nbytes = lib.esr_packed_tail_s16_bytes(nseg, seg_c, mid_c, cout)
    if not nbytes:
        raise L.EsrError("pack_tail_s16: unsupported shape")
    out = torch.empty((nbytes + 3) // 4, dtype=torch.float32)
    L.check(lib.esr_pack_tail_s16(_ptr(w), _ptr(b) if b is not None else None, nseg, seg_c, mid_c, cout,
                                  L.COMPUTE[compute] if isinstance(compute, str) else compute, _ptr(out), nbytes), "esr_pack_tail_s16")
    return out


def pack_apply_post(w0, b0, w1, b1, store):
    """Weights of the 1x1 chain riding in esr_esa_apply_f32's launch (esr_esa_desc.post_w): w0 [cout0, cin(, 1, 1)] applied to the
    apply result, w1 [cout1, cout0(, 1, 1)] (or None) applied to w0's result -> esr_pack_apply_post blob."""
    lib = L.lib()
    w0 = w0.detach().to("cpu", torch.float32).reshape(w0.shape[0], -1).contiguous()
    cout0, cin = w0.shape
    b0 = None if b0 is None else b0.detach().to("cpu", torch.float32).contiguous()
    cout1 = 0
    if w1 is not None:
        w1 = w1.detach().to("cpu", torch.float32).reshape(w1.shape[0], -1).contiguous()
        cout1 = w1.shape[0]
        if w1.shape[1] != cout0:
            raise L.EsrError("pack_apply_post: w1 must take w0's outputs")
        b1 = None if b1 is None else b1.detach().to("cpu", torch.float32).contiguous()
    st = L.STORE[store]
    nbytes = lib.esr_packed_apply_post_bytes(cin, cout0, cout1, st)
    if not nbytes:
        raise L.EsrError("pack_apply_post: unsupported shape / storage")
    out = torch.empty((nbytes + 3) // 4, dtype=torch.float32)
    L.check(lib.esr_pack_apply_post(_ptr(w0), _ptr(b0) if b0 is not None else None, _ptr(w1) if w1 is not None else None,
                                    _ptr(b1) if (w1 is not None and b1 is not None) else None, cin, cout0, cout1, st, _ptr(out), nbytes),
            "esr_pack_apply_post")
    return out


def unpack_conv_s16(blob, cin, cout, k, compute, cin_map=None, cin_phys=None):
    """EFFECTIVE fp32 weights (what the 16-bit kernel multiplies by) + bias of a pack_conv_s16 blob."""
    lib = L.lib()
    blob = blob.detach().to("cpu").contiguous()
    if cin_map is not None:
        cm = np.ascontiguousarray(np.asarray(cin_map, dtype=np.int32))
        cin_phys, cm_p = len(cm), cm.ctypes.data_as(ctypes.c_void_p)
    else:
        cm_p = None
        cin_phys = cin if cin_phys is None else cin_phys
    w = torch.empty(cout, cin, k, k)
    b = torch.empty(cout)
    L.check(lib.esr_unpack_conv_s16(_ptr(blob), blob.numel() * blob.element_size(), cin, cout, k, cm_p, cin_phys,
                                    L.COMPUTE[compute] if isinstance(compute, str) else compute, _ptr(w), _ptr(b)),
            "esr_unpack_conv_s16")
    return w, b


def pack_dense(weight, bias, cin_p, cout_p):
    """Plain [tap][cin_p][cout_p] + bias[cout_p] layout of the small ESA kernels (esr_pack_dense_f32)."""
    lib = L.lib()
    w = weight.detach().to("cpu", torch.float32).contiguous()
    if w.dim() == 2:
        w = w[:, :, None, None].contiguous()
    cout, cin, k, _ = w.shape
    b = None if bias is None else bias.detach().to("cpu", torch.float32).contiguous()
    nbytes = lib.esr_packed_dense_bytes(cin_p, cout_p, k)
    out = torch.empty(nbytes // 4, dtype=torch.float32)
    L.check(lib.esr_pack_dense_f32(_ptr(w), _ptr(b) if b is not None else None, cin, cout, k, cin_p, cout_p,
                                   _ptr(out), nbytes), "esr_pack_dense_f32")
    return out


def pack_dw(weight, bias):
    """Depthwise [C,1,3,3] weights + bias -> [tap][cp] + bias[cp] (esr_pack_dw_f32)."""
    lib = L.lib()
    w = weight.detach().to("cpu", torch.float32).contiguous()
    c = w.shape[0]
    b = None if bias is None else bias.detach().to("cpu", torch.float32).contiguous()
    nbytes = lib.esr_packed_dw_bytes(c)
    out = torch.empty(nbytes // 4, dtype=torch.float32)
    L.check(lib.esr_pack_dw_f32(_ptr(w), _ptr(b) if b is not None else None, c, _ptr(out), nbytes), "esr_pack_dw_f32")
    return out


def unpack_conv(blob, cin, cout, k, cin_map=None, cin_phys=None):
    lib = L.lib()
    blob = blob.detach().to("cpu", torch.float32).contiguous()
    if cin_map is not None:
        cm = np.ascontiguousarray(np.asarray(cin_map, dtype=np.int32))
        cin_phys, cm_p = len(cm), cm.ctypes.data_as(ctypes.c_void_p)
    else:
        cm_p = None
        cin_phys = cin if cin_phys is None else cin_phys
    w = torch.empty(cout, cin, k, k)
    b = torch.empty(cout)
    rc = lib.esr_unpack_conv_f32(_ptr(blob), blob.numel() * 4, cin, cout, k, cm_p, cin_phys, _ptr(w), _ptr(b))
    L.check(rc, "esr_unpack_conv_f32")
    return w, b


def _flat_ops(ops):
    """the op dicts with every `chain` op replaced by the convolutions it stands for (their weights are the chain's)"""
    for o in ops:
        if o["kind"] == "chain":
            yield from o["replaces"]
        else:
            yield o


class Buffer:
    """An NHWC activation buffer inside the workspace: [n][h][w][pitch] elements of `esize` bytes
    (4 = fp32; 2 = bf16 / fp16 for the full-resolution buffers of a 16-bit-storage plan)."""

    def __init__(self, name, pitch, offset_bytes, h=None, w=None, esize=4, arena=0, blocked=False):
        self.name, self.pitch, self.offset, self.h, self.w, self.esize = name, pitch, offset_bytes, h, w, esize
        self.arena = arena     # 0: full-resolution buffers (the plan's storage type), 1: low-resolution fp32 maps -- kept apart, see Plan
        self.blocked = blocked # fp32 only: stored [n][pitch / 8][h][w][8] (esr_conv_desc.blocked8) -- same bytes, other order

    def __getitem__(self, sl):
        """buf[a:b] -> channel slice view (coff=a, channels=b-a)."""
        a = 0 if sl.start is None else sl.start
        b = self.pitch if sl.stop is None else sl.stop
        return (self, a, b - a)


class Planar:
    """A channel concat kept as `nseg` dense tensors [n][h][w][seg_pitch] that lie `stride` bytes apart (16-bit plans): every
    producer writes whole lines into its own tensor (`seg(j)`), the consuming 1x1 walks the segments
    (esr_conv_desc.in_seg_stride / in_seg_chunks).  Physical channel slot s of the consumer = slot s % seg_pitch of segment
    s // seg_pitch -- the same slot order as a dense [.., nseg * seg_pitch] buffer, so the weight blobs do not change."""

    def __init__(self, segs):
        self.segs = segs
        self.pitch = segs[0].pitch
        self.stride = segs[1].offset - segs[0].offset if len(segs) > 1 else 0

    def seg(self, j):
        return self.segs[j]


INPUT = "__input__"
OUTPUT = "__output__"


def _same_view(a, b):
    """True when two op operands (Buffer | (Buffer, coff, channels) | Planar | INPUT / OUTPUT | None) name the same bytes."""
    if a is b:
        return True
    if a is None or b is None or isinstance(a, str) or isinstance(b, str) or isinstance(a, Planar) or isinstance(b, Planar):
        return False
    a = (a, 0, a.pitch) if isinstance(a, Buffer) else a
    b = (b, 0, b.pitch) if isinstance(b, Buffer) else b
    return a[0] is b[0] and a[1] == b[1] and a[2] == b[2]


def _stored_channels(v, logical, chunk):
    """channels of operand `v` a kernel actually moves: the whole pitch of a Buffer (pad slots included), the chunk-rounded width
    of a channel slice, `logical` for the network input / output"""
    if isinstance(v, Buffer):
        return v.pitch
    if isinstance(v, Planar):
        return v.pitch * len(v.segs)
    if isinstance(v, tuple):
        return (v[2] + chunk - 1) // chunk * chunk
    return logical


class Plan:
    """Builds the op list for one (N, H, W); see HipSRModel._build_plan in each network.

    `store` is the element type of the FULL-RESOLUTION activation buffers ("f32" | "bf16" | "f16"); the ESA
    low-resolution maps (a few hundred KB) are always fp32.  Offsets are bytes into one workspace."""

    def __init__(self, n, h, w, store="f32"):
        self.n, self.h, self.w = n, h, w
        self.npix = n * h * w
        self.store = store
        self.esize = 4 if store == "f32" else 2
        self.total = 0         # bytes of the full-resolution arena
        self.total_lo = 0      # bytes of the low-resolution (always fp32) arena.  The two never overlap across plans: a 16-bit plan
                               # laid over another shape's fp32 maps would read their bytes as bf16 / fp16 -- Inf / NaN patterns included
        self.buffers = []
        self.ops = []          # python dicts until finalize()

    def cpad(self, c):
        """channel pitch of a full-resolution buffer holding c channels: whole K chunks of the conv kernels
        (8 fp32 channels, 16 16-bit channels = 32 bytes either way)."""
        m = 8 if self.esize == 4 else 16
        return (c + m - 1) // m * m

    def buffer(self, name, pitch, h=None, w=None, blocked=False):
        """Full-resolution buffer by default; (h, w) gives a low-resolution fp32 one (ESA maps).  blocked: channel-blocked
        [n][pitch / 8][h][w][8] (fp32 plans; only as the dst1 of a split store and the input of the fused IMDB tail)."""
        lowres = h is not None
        assert not blocked or (self.esize == 4 and not lowres and pitch % 8 == 0)
        esize = 4 if lowres else self.esize
        assert (pitch * esize) % 16 == 0
        h = self.h if h is None else h
        w = self.w if w is None else w
        size = (self.n * h * w * pitch * esize + 255) // 256 * 256
        if lowres:
            b = Buffer(name, pitch, self.total_lo, h, w, esize, arena=1)
            self.total_lo += size
        else:
            b = Buffer(name, pitch, self.total, h, w, esize, blocked=blocked)
            self.total += size
        self.buffers.append(b)
        return b

    def planar(self, name, nseg, seg_pitch):
        """nseg equal full-resolution buffers back to back (see Planar); 16-bit plans only"""
        assert self.esize == 2 and seg_pitch % 8 == 0            # (whole 16-byte granules; a tight pitch -- 56 -- is not whole K chunks)
        return Planar([self.buffer(f"{name}.{j}", seg_pitch) for j in range(nseg)])

    def pair(self, name, pitch):
        """a hi + lo pair (esr_conv_desc.hilo): value = seg(0) + seg(1), two dense bf16 tensors `stride` bytes apart"""
        return self.planar(name, 2, pitch)

    def conv(self, wname, src, dst, cin, cout, k=3, act=L.ACT_NONE, slope=0.05,
             res=None, res_mode=L.RES_NONE, dst1=None, split=0, hw=None, counted=True, tail=None, post=None, cin_alg=None,
             border=None, bs_of=None, hilo=0):
        """src/dst/res: INPUT | OUTPUT | Buffer | (Buffer, coff, channels).  hw: spatial dims if not full-res.
        counted=False marks launches that are not an nn.Conv2d call of the reference (complexity counters).
        tail = dict(w=<1x1 weight name>, cat=<view of its other input channels>, cat_c, cout, mid_act): the 3x3 result
        (cout <= 16) feeds a fused 1x1 (esr_conv_desc.tail_*); dst/res/act/split then belong to the 1x1.
        post = dict(w=<1x1 weight name>, dst=<view>, cout, act): a 1x1 of this conv's activated output, stored to `dst`
        by the same launch (esr_conv_desc.post_*).  cin_alg: logical input channels when `cin` counts the pad slots of a
        padded concat buffer (algorithmic flops / bytes).  border: name of an esr_conv_desc.border_bias table.  bs_of: this
        dense 3x3 stands for a BSConvU (pointwise 1x1 + depthwise 3x3 with merged weights): its algorithmic flops and its
        complexity-counter terms are the BSConvU's.  hilo: L.HILO_IN | HILO_RES | HILO_OUT -- src / res / dst are hi + lo pairs
        (esr_conv_desc.hilo, bf16 plans): `Plan.pair(name, pitch)` objects, two dense tensors [high parts, low parts] one stride apart."""
        head = None
        if src is INPUT and self.esize == 2 and k == 3 and dst is not OUTPUT:
            # 16-bit plans: the NCHW fp32 input is first packed to 16-bit hi / lo slots (esr_pack_input_s16), the head convolution
            # then runs on conv_s16_kernel with hi / lo weights (`<name>#head#s16`, engine.pack_head_s16)
            x16 = self.buffer('in16', 16)
            self.ops.append(dict(kind="pack", src=INPUT, dst=x16, cin=cin, cout=16, hw=None, counted=False))
            src, head, wname = x16, cin, wname + '#head'
        self.ops.append(dict(kind="conv", w=wname, src=src, dst=dst, dst1=dst1, cin=cin, cout=cout, k=k, act=act,
                             slope=slope, res=res, res_mode=res_mode, split=split, hw=hw, counted=counted, tail=tail,
                             post=post, cin_alg=cin if cin_alg is None else cin_alg, border=border, bs_of=bs_of, head=head, hilo=hilo))

    def dwconv(self, wname, src, dst, c, act=L.ACT_NONE, slope=0.05, res=None, res_mode=L.RES_NONE, hw=None):
        """depthwise 3x3 + bias (+res) (+act): the dw half of BSConvU."""
        self.ops.append(dict(kind="dw", w=wname, src=src, dst=dst, dst1=None, cin=c, cout=c, k=3, act=act, slope=slope,
                             res=res, res_mode=res_mode, split=0, hw=hw, counted=True))

    def bsconv(self, pw, dw, src, dst, cin, c, act=L.ACT_NONE, slope=0.05, res=None, res_mode=L.RES_NONE, distill=None):
        """BSConvU in one launch (esr_bsconv_f32): dst = act(dw3x3(pw1x1(src)) [+ res]).  distill = dict(w=<1x1 weight
        name>, dst=<view>, cout, act): the enclosing block's distillation conv on the same input."""
        self.ops.append(dict(kind="bs", pw=pw, dw=dw, w=dw, src=src, dst=dst, cin=cin, cout=c, k=3, act=act, slope=slope,
                             res=res, res_mode=res_mode, distill=distill, hw=None, counted=True))

    def conv3x3s2(self, wname, src, dst, f):
        self.ops.append(dict(kind="s2", w=wname, src=src, dst=dst, f=f, cin=f, cout=f, k=3, act=L.ACT_NONE,
                             hw=(dst.h, dst.w), counted=True))

    def maxpool7s3(self, src, dst):
        self.ops.append(dict(kind="pool", src=src, dst=dst))

    def esa_lowres(self, mark, c1, pooled, dst, f, s2, layers):
        """ESA's low-resolution branch as ONE op (esr_esa_lowres_f32: two launches with halo recompute) in place of the ops appended
        since `mark = len(plan.ops)` -- conv3x3s2, maxpool7s3 and the 1..3 small 3x3 layers (RFDN / RLFN: dense convs, BSRN: pointwise +
        depthwise pairs); those op dicts stay attached as `replaces`: the complexity counters and the algorithmic costs are theirs.
        layers: [dict(kind=0|1, act, w=<path of the dense 3x3 / pointwise weights>, w_dw=<path of the depthwise weights>)]"""
        sub = self.ops[mark:]
        del self.ops[mark:]
        self.ops.append(dict(kind="lowres", src=c1, pooled=pooled, dst=dst, f=f, w=s2, layers=layers, replaces=sub,
                             cin=f, cout=f, k=3))

    def chain(self, mark):
        """The 3x3 convolutions appended since `mark = len(plan.ops)` -- a residual block's chain src -> t1 -> ... -> (+ src) -> post 1x1 ->
        post2 1x1, RLFB's c1_r -> c2_r -> c3_r -> c5 -> esa.conv1 (team04_rlfn.py:109-122) -- as ONE esr_conv_chain_s16 op (16-bit plans):
        the intermediate tensors stay in LDS (rlfb_chain_kernel).  The conv op dicts stay attached as `replaces`: weights, complexity
        counters and algorithmic costs are theirs; the result is bit-identical to running them one by one."""
        sub = self.ops[mark:]
        assert self.esize == 2 and len(sub) >= 2 and all(o["kind"] == "conv" and o["k"] == 3 for o in sub)
        for a, b in zip(sub[:-1], sub[1:]):
            assert b["src"] is a["dst"] and a["res"] is None and a.get("post") is None
        last = sub[-1]
        assert last["dst"] is None and last["res"] is sub[0]["src"] and last.get("post") is not None and last["post"].get("post2") is not None
        del self.ops[mark:]
        self.ops.append(dict(kind="chain", replaces=sub, w=sub[0]["w"], cin=sub[0]["cin"], cout=last["cout"], k=3))

    def esa_apply(self, wf, w4, x, c1, c3, dst, c, f, post=None, skip_y=False):
        """y = x * sigmoid(conv4(bilinear(c3) + conv_f(c1)));  two nn.Conv2d calls of the reference.
        post (16-bit plans): [dict(w=<1x1 path>, dst=<view>, cout, act, slope, res=<view>|None, linear=bool), ...] -- one or two 1x1
        convolutions evaluated in the same launch (esr_esa_desc.post[]): the first on y as stored, the second on the first's fp32
        result; their weights are ONE blob `<post[0].w>#apost` (engine.pack_apply_post, packed by the network).  skip_y: y itself
        is not stored (nothing but the chain reads it)."""
        self.ops.append(dict(kind="apply", wf=wf, w4=w4, x=x, c1=c1, c3=c3, dst=dst, c=c, f=f, post=post, skip_y=skip_y))

    @staticmethod
    def _addr(buf, base):
        """base = (workspace address, bytes reserved for the low-resolution arena in front of the full-resolution one)"""
        ws, lo_cap = base
        return ws + (buf.offset if buf.arena == 1 else lo_cap + buf.offset)

    @staticmethod
    def _view(v, base):
        if isinstance(v, Buffer):
            v = (v, 0, v.pitch)
        buf, coff, _ = v
        return L.View(ctypes.c_void_p(Plan._addr(buf, base)), buf.pitch, coff)

    def finalize(self, workspace, weights):
        """weights: name -> device blob tensor (16-bit-storage plans: `name#s16` for the NHWC convs).  Returns
        (Op array, input op indices, output op indices)."""
        st = L.STORE[self.store]
        arr = (L.Op * len(self.ops))()
        in_idx, out_idx = [], []
        base = workspace if isinstance(workspace, tuple) else ((workspace if isinstance(workspace, int) else (workspace.data_ptr() if workspace is not None else 0)), self.total_lo)
        for i, o in enumerate(self.ops):
            op = arr[i]
            if o["kind"] == "pack":
                op.kind = L.OP_PACK_INPUT
                d = op.conv
                d.n, d.h, d.w, d.cin, d.storage = self.n, self.h, self.w, o["cin"], st
                d.out0 = self._view(o["dst"], base)
                in_idx.append(i)
                continue
            if o["kind"] == "bs":
                op.kind = L.OP_BSCONV
                d = op.bs
                d.storage = st
                d.n, d.h, d.w, d.cin, d.c = self.n, self.h, self.w, o["cin"], o["cout"]
                d.act, d.slope, d.res_mode = o["act"], o["slope"], o["res_mode"]
                d.inp, d.out = self._view(o["src"], base), self._view(o["dst"], base)
                if o["res"] is not None:
                    d.res = self._view(o["res"], base)
                sfx = "#s16" if st else ""                        # 16-bit storage: hi + lo 1x1 blobs (esr_pack_conv_s16)
                d.pw_packed = ctypes.c_void_p(weights[o["pw"] + sfx].data_ptr())
                d.dw_packed = ctypes.c_void_p(weights[o["dw"]].data_ptr())
                t = o["distill"]
                if t is not None:
                    d.d_packed = ctypes.c_void_p(weights[t["w"] + sfx].data_ptr())
                    d.d_cout, d.d_act = t["cout"], t.get("act", L.ACT_NONE)
                    d.d_out = self._view(t["dst"], base)
                continue
            if o["kind"] == "lowres":
                op.kind = L.OP_ESA_LOWRES
                d = op.lo
                d.n, d.h, d.w, d.f, d.storage, d.n_layers = self.n, self.h, self.w, o["f"], st, len(o["layers"])
                d.x = self._view(o["src"], base)
                d.w_s2 = ctypes.c_void_p(weights[o["w"]].data_ptr())
                d.pooled = ctypes.c_void_p(self._addr(o["pooled"], base))
                d.y = ctypes.c_void_p(self._addr(o["dst"], base))
                for l, ly in enumerate(o["layers"]):
                    d.layer[l].kind, d.layer[l].act = ly["kind"], ly.get("act", L.ACT_NONE)
                    d.layer[l].w = ctypes.c_void_p(weights[ly["w"] + "#dense"].data_ptr())
                    if ly["kind"] == 1:
                        d.layer[l].w_dw = ctypes.c_void_p(weights[ly["w_dw"]].data_ptr())
                continue
            if o["kind"] == "chain":
                op.kind = L.OP_CONV_CHAIN
                d = op.chain
                sub = o["replaces"]
                first, last = sub[0], sub[-1]
                d.n, d.h, d.w, d.n_layers = self.n, self.h, self.w, len(sub)
                d.cin, d.cmid, d.cout = first["cin"], first["cout"], last["cout"]
                d.act, d.slope, d.res_mode = first["act"], first["slope"], last["res_mode"]
                d.storage = d.compute = st
                d.inp = self._view(first["src"], base)
                for l, so in enumerate(sub):
                    d.wpacked[l] = weights[so["w"] + "#s16"].data_ptr()
                t = last["post"]
                t2 = t["post2"]
                pc = min((t["cout"] + 15) // 16 * 16, t["dst"].pitch) if isinstance(t["dst"], Buffer) else t["cout"]    # (whole dense buffer: pad channels too)
                d.post_wpacked, d.post_out = ctypes.c_void_p(weights[t["w"] + "#post"].data_ptr()), self._view(t["dst"], base)
                d.post_cout, d.post_act = pc, t.get("act", L.ACT_NONE)
                d.post2_wpacked, d.post2_out = ctypes.c_void_p(weights[t2["w"] + "#post"].data_ptr()), self._view(t2["dst"], base)
                d.post2_cout = t2["cout"]
                if not L.lib().esr_conv_chain_supported(ctypes.byref(d)):
                    raise L.EsrError(f"{o['w']}: no chain kernel for this shape (the plan should have kept separate ops)")
                continue
            if o["kind"] not in ("conv", "dw"):
                e = op.esa
                e.n = self.n
                if o["kind"] == "apply":
                    op.kind = L.OP_ESA_APPLY
                    e.storage = st
                    e.h, e.w, e.c, e.f = self.h, self.w, o["c"], o["f"]
                    e.h_lo, e.w_lo = o["c3"].h, o["c3"].w
                    e.x, e.y = self._view(o["x"], base), self._view(o["dst"], base)
                    e.c1 = ctypes.c_void_p(self._addr(o["c1"], base))
                    e.c3 = ctypes.c_void_p(self._addr(o["c3"], base))
                    e.w0 = ctypes.c_void_p(weights[o["wf"]].data_ptr())
                    e.w1 = ctypes.c_void_p(weights[o["w4"]].data_ptr())
                    if o.get("post"):
                        e.post_w = ctypes.c_void_p(weights[o["post"][0]["w"] + "#apost"].data_ptr())
                        e.skip_y = 1 if o.get("skip_y") else 0
                        for k_, t in enumerate(o["post"]):
                            pp = e.post[k_]
                            pp.cout, pp.act, pp.slope = t["cout"], t.get("act", L.ACT_NONE), t.get("slope", 0.05)
                            pp.out = self._view(t["dst"], base)
                            if t.get("res") is not None:
                                pp.res_mode, pp.res = L.RES_PRE_ACT, self._view(t["res"], base)
                else:
                    op.kind = L.OP_CONV3X3S2 if o["kind"] == "s2" else L.OP_MAXPOOL7S3
                    e.h, e.w = o["src"].h, o["src"].w
                    e.h_lo, e.w_lo = o["dst"].h, o["dst"].w
                    e.x, e.y = self._view(o["src"], base), self._view(o["dst"], base)
                    if o["kind"] == "s2":
                        e.storage = st                           # reads the full-resolution conv1 map
                        e.f = o["f"]
                        e.w0 = ctypes.c_void_p(weights[o["w"]].data_ptr())
                continue
            op.kind = L.OP_CONV if o["kind"] == "conv" else L.OP_DWCONV
            d = op.conv
            d.n, d.h, d.w = self.n, self.h, self.w
            if o["hw"] is not None:
                d.h, d.w = o["hw"]
            if o.get("hilo", 0):                                  # hi + lo pairs (Plan.pair): the descriptor's views are the high-part tensors
                o = dict(o)
                for bit, key in ((L.HILO_IN, "src"), (L.HILO_RES, "res"), (L.HILO_OUT, "dst")):
                    if o["hilo"] & bit:
                        assert isinstance(o[key], Planar) and len(o[key].segs) == 2, "hilo: Plan.pair() buffers"
                        o[key + "_pair"], o[key] = o[key], o[key].seg(0)
            d.cin, d.cout, d.ksize = (3 * o["head"] if o.get("head") else o["cin"]), o["cout"], o["k"]
            d.act, d.slope, d.res_mode = o["act"], o["slope"], o["res_mode"]
            d.split = o["split"]
            if o["src"] is INPUT:
                d.in_layout = L.NCHW_IN
                in_idx.append(i)
            elif isinstance(o["src"], Planar):
                pl = o["src"]
                d.in_layout = L.NHWC
                d.inp = self._view(pl.seg(0), base)
                d.in_seg_stride, d.in_seg_chunks = pl.stride, (pl.pitch + 15) // 16      # (a tight pitch -- 56 for nf = 50 -- still reads whole K chunks)
            else:
                d.in_layout = L.NHWC
                d.inp = self._view(o["src"], base)
            if o["dst"] is OUTPUT:
                d.out_layout = L.NCHW_SHUFFLE4
                out_idx.append(i)
            elif o["dst"] is None:                                # consumed by the post chain only (out0.ptr == NULL)
                d.out_layout = L.NHWC
            else:
                d.out_layout = L.NHWC
                d.out0 = self._view(o["dst"], base)
            if o["dst1"] is not None:
                d.out1 = self._view(o["dst1"], base)
            def _blk(v):
                return v is not None and v is not INPUT and v is not OUTPUT and not isinstance(v, Planar) and (v if isinstance(v, Buffer) else v[0]).blocked
            assert o.get("tail") is not None or (not _blk(o["dst"]) and not _blk(o["res"])), \
                "blocked buffers: input of a Winograd conv / the fused tail, dst1 of a split store, dst / res of the fused tail"
            d.blocked8 = (L.BLOCKED_IN if _blk(o["src"]) else 0) | (L.BLOCKED_OUT1 if _blk(o["dst1"]) else 0) | \
                (L.BLOCKED_OUT0 if _blk(o["dst"]) else 0) | (L.BLOCKED_RES if _blk(o["res"]) else 0)
            if o["res"] is not None:
                d.res = self._view(o["res"], base)
            d.hilo = o.get("hilo", 0)
            if d.hilo:                                            # one stride for all pairs of the descriptor
                strides = {o[key + "_pair"].stride for bit, key in ((L.HILO_IN, "src"), (L.HILO_RES, "res"), (L.HILO_OUT, "dst")) if d.hilo & bit}
                assert len(strides) == 1, "hilo: pairs of one shape"
                d.hilo_stride = strides.pop()
            lowres = o["hw"] is not None
            d.storage = 0 if lowres else st                       # ESA low-resolution maps: fp32

            def full_width(dst, cout):
                """16-bit storage, destination = a WHOLE dense buffer: store the pad channels of its last K chunk too (they are
                zeros: zero weight rows, zero bias) -- a 24-of-32-channel store leaves every 64-byte run partial, and partial-line
                stores cost up to 2.3x a full one (tools/dbg/s16_1x1_probe.py).  Slices of shared buffers keep their width."""
                if st and not lowres and isinstance(dst, Buffer) and o["kind"] == "conv":
                    return min((cout + 15) // 16 * 16, dst.pitch)
                return cout
            if o["dst"] is not OUTPUT and o["dst"] is not None and not o["split"]:
                d.cout = full_width(o["dst"], o["cout"])
            if st and not lowres and o["kind"] == "conv" and o["src"] is not INPUT:
                d.wpacked = ctypes.c_void_p(weights[o["w"] + "#s16"].data_ptr())     # conv_s16_kernel
                d.compute = st
            else:
                d.wpacked = ctypes.c_void_p(weights[o["w"]].data_ptr())
                wn = weights.get(o["w"] + "#wino") if o["kind"] == "conv" else None
                if wn is not None and o.get("tail") is None and o.get("post") is None and L.lib().esr_wino_supported(ctypes.byref(d)):
                    d.wino_wpacked = ctypes.c_void_p(wn.data_ptr())      # Winograd F(2x2, 3x3): wino_f32_kernel
            if o.get("border") is not None:
                d.border_bias = ctypes.c_void_p(weights[o["border"]].data_ptr())
            t = o.get("tail")
            if t is not None:
                d.tail_wpacked = ctypes.c_void_p(weights[t["w"]].data_ptr())
                if isinstance(t["cat"], Planar):             # 16-bit tail (ABI v12): three dense tensors one stride apart; the 3x3's own width
                    d.tail_cat = self._view(t["cat"].seg(0), base)
                    d.tail_seg_stride16 = t["cat"].stride // 16
                    d.cout = o["cout"]
                    d.tail_cat_c, d.tail_cout, d.tail_mid_act = t["cat_c"], full_width(o["dst"], t["cout"]), t.get("mid_act", L.ACT_NONE)
                else:
                    d.tail_cat = self._view(t["cat"], base)
                    d.tail_cat_c, d.tail_cout, d.tail_mid_act = t["cat_c"], t["cout"], t.get("mid_act", L.ACT_NONE)
            t = o.get("post")
            if t is not None:
                psfx = "#post" if (st and not lowres) else ""     # 16-bit storage: esr_pack_post_s16 images
                d.post_wpacked = ctypes.c_void_p(weights[t["w"] + psfx].data_ptr())
                d.post_out = self._view(t["dst"], base)
                d.post_cout, d.post_act = full_width(t["dst"], t["cout"]) if psfx else t["cout"], t.get("act", L.ACT_NONE)
                t2 = t.get("post2")
                if t2 is not None:
                    d.post2_wpacked = ctypes.c_void_p(weights[t2["w"] + psfx].data_ptr())
                    d.post2_out = self._view(t2["dst"], base)
                    d.post2_cout = t2["cout"]
                if psfx and o.get("tail") is not None:
                    if not L.lib().esr_conv_tail_supported(ctypes.byref(d)):
                        raise L.EsrError(f"{o['w']}: no fused tail kernel for this shape (the plan should have built separate ops)")
                elif psfx and not L.lib().esr_conv_post_supported(ctypes.byref(d)):
                    raise L.EsrError(f"{o['w']}: no fused post-chain kernel for this shape (the plan should have built separate ops)")
        return arr, in_idx, out_idx


# ---- torch.library custom op -------------------------------------------------------------------------------------------
# `model(x)` goes through ONE registered operator, esr::sr_forward(x, handle) -> y, so that graph capture / FakeTensor
# tracing / torch.compile see an opaque op with a shape function instead of ctypes calls: the "thin PyTorch-ROCm custom-op
# layer" the drop-in modules sit on.  The real implementation replays the fused op list through the C ABI; the fake (meta)
# implementation only computes the output shape.  `handle` identifies the live module: a key from a process-wide counter handed out
# at construction and never reused (an `id()` can be recycled by another object once its owner is collected); the registry holds
# weak references, so an entry disappears with its module.
_LIVE = weakref.WeakValueDictionary()
_HANDLES = itertools.count(1)


@torch.library.custom_op("esr::sr_forward", mutates_args=())
def sr_forward(x: torch.Tensor, handle: int) -> torch.Tensor:
    m = _LIVE.get(handle)
    if m is None:
        raise L.EsrError(f"esr::sr_forward: no live model with handle {handle}")
    return m._forward_impl(x)


@sr_forward.register_fake
def _sr_forward_fake(x, handle):
    m = _LIVE.get(handle)
    if m is None:
        raise L.EsrError(f"esr::sr_forward: no live model with handle {handle}")
    if x.dim() != 4:
        raise L.EsrError("expected a 4-D float32 NCHW tensor (uint2tensor4 output)")
    return x.new_empty((x.shape[0], m.out_nc, x.shape[2] * m.upscale, x.shape[3] * m.upscale), dtype=torch.float32)


class _ModelLock:
    """threading.RLock that survives copy.deepcopy / pickling of the module that owns it (a copy gets a lock of its own)."""

    def __init__(self):
        self._l = threading.RLock()

    def __enter__(self):
        return self._l.__enter__()

    def __exit__(self, *a):
        return self._l.__exit__(*a)

    def __deepcopy__(self, memo):
        return _ModelLock()

    def __reduce__(self):
        return (_ModelLock, ())


class _Entry:
    """One cached shape: the Plan, its esr_op array finalized against workspace base `base`, and (after the second forward of the
    shape on this stream) the HIP graph of its launches (esr_graph_create)."""
    __slots__ = ("plan", "arr", "in_idx", "out_idx", "base", "graph", "uses")

    def __init__(self, plan):
        self.plan, self.arr, self.in_idx, self.out_idx, self.base = plan, None, (), (), None
        self.graph, self.uses = None, 0

    def drop_graph(self):
        if self.graph is not None:
            L.lib().esr_graph_destroy(self.graph)
            self.graph = None
        self.uses = 0

    def __del__(self):
        try:
            self.drop_graph()
        except Exception:          # interpreter shutdown: the library may be gone
            pass


class _StreamCtx:
    """What a forward needs that is private to ONE HIP stream: the workspace, the plans finalized against it and the
    per-kernel event sets.  Two forwards of the same model enqueued on two streams run concurrently on the GPU, so they must
    not share activation buffers; the packed weights (read-only) are shared."""
    __slots__ = ("plans", "ws", "ws_owner", "lo_cap", "profs")

    def __init__(self):
        self.plans = collections.OrderedDict()   # (n, c, h, w, device) -> _Entry, LRU
        self.ws = None             # ONE grow-only workspace (all cached plans lay their buffers out in it)
        self.ws_owner = None       # key of the plan that ran last in the workspace (None: content unknown -> zero before use)
        self.lo_cap = 0            # bytes reserved in front of the workspace for the plans' low-resolution fp32 maps (grow-only)
        self.profs = {}


class HipSRModel(nn.Module):
    """Base of the drop-in nn.Modules.  Subclasses register reference-compatible parameters
    with `_add_conv` and describe their forward with `_build_plan`."""

    def __init__(self):
        super().__init__()
        self._conv_specs = {}      # path -> (cin, cout, k, cin_map)
        self._dense_specs = {}     # path -> (cin_p, cout_p): small ESA weights in the plain dense layout
        self._dw_specs = []        # depthwise 3x3 parameter paths
        self._packed = None        # path -> device blob
        self._packed_sig = None
        self._dirty = True         # parameters may have changed since the last repack (load_state_dict / .to() / repack())
        self._ctxs = {}            # (device, HIP stream handle) -> _StreamCtx: workspace + plans of the forwards enqueued on that stream
        self.compute = "f32"       # "f32" | "bf16" | "f16": MFMA operand format of the full-resolution 3x3 convs
        self._fuse_esa_lowres = True  # ESA's low-resolution branch as one esr_esa_lowres_f32 op (two launches) instead of 3 .. 8 launches
        self._winograd = True      # fp32 plans: 3x3 convs whose shape qualifies run as Winograd F(2x2, 3x3) (esr_conv_desc.wino_wpacked)
        self._hilo_skip = True     # bf16 plans: the long skip's tensors (`fea`, `out_lr`) as hi + lo pairs (esr_conv_desc.hilo; Plan.hilo_skip)
        self._fuse_tail = True     # 16-bit RFDN plans: RFDB's c4 -> cat -> c5 -> esa.conv1 as one launch (rfdb_tail_kernel, ABI v12)
        self._tight_pitch = True   # 16-bit RFDN plans: nf-wide tensors at pitch round_up(nf, 8) instead of whole K chunks (56 for nf = 50; esr_conv2d_s16: tight pitch)
        self._fuse_chain = True    # 16-bit plans: a block's 3x3 chain as one esr_conv_chain_s16 launch where a kernel exists (Plan.chain)
        self.use_graphs = True     # forwards of at most GRAPH_MAX_PIXELS input pixels replay a captured HIP graph (esr_graph_launch)
        self._lock = _ModelLock()       # plan / workspace bookkeeping and the pointer patch + enqueue of one forward (see _forward_impl)
        self._prof_passes = 0      # >0: record HIP events around every op (bench roofline leg)
        self.handle = next(_HANDLES)       # the `handle` argument of esr::sr_forward
        _LIVE[self.handle] = self
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._mark_dirty())

    # -- parameter registration: same key names as the reference state_dict -------------------
    def _add_conv(self, path, cin, cout, k, cin_map=None, linear=False, dense=None, stride=1, padding=None, custom=False):
        """Create nested containers so that `path + '.weight'` / `path + '.bias'` are the
        state_dict keys (e.g. 'model.1.sub.0.conv1.0').  The leaf is an nn.Conv2d / nn.Linear
        used purely as a parameter holder with the reference's shapes and default init."""
        parts = path.split(".")
        mod = self
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        leaf = nn.Linear(cin, cout) if linear else nn.Conv2d(cin, cout, k, stride, (k - 1) // 2 if padding is None else padding)
        mod.add_module(parts[-1], leaf)
        if dense is not None:
            self._dense_specs[path] = dense
        elif not custom:
            self._conv_specs[path] = (cin, cout, k, cin_map)

    def _add_dw(self, path, c):
        """depthwise nn.Conv2d(c, c, 3, 1, 1, groups=c) parameter holder (`path.weight` is [c,1,3,3])."""
        parts = path.split(".")
        mod = self
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        mod.add_module(parts[-1], nn.Conv2d(c, c, 3, 1, 1, groups=c))
        self._dw_specs.append(path)

    def _leaf(self, path):
        mod = self
        for p in path.split("."):
            mod = mod._modules[p]
        return mod

    # Plan-shaping switches: which blobs repack() builds and which ops _build_plan emits depend on them, so a change re-packs and drops
    # the cached plans exactly like set_compute() (as plain attributes a late change was ignored or raised KeyError).
    def _set_flag(self, name, value):
        value = bool(value)
        with self._lock:               # (another thread may be inside forward(): plans / profiler handles must not vanish under it)
            if getattr(self, name) != value:
                setattr(self, name, value)
                self._dirty = True
                self._drop_plans()

    fuse_esa_lowres = property(lambda self: self._fuse_esa_lowres, lambda self, v: self._set_flag("_fuse_esa_lowres", v))
    winograd = property(lambda self: self._winograd, lambda self, v: self._set_flag("_winograd", v))
    hilo_skip = property(lambda self: self._hilo_skip, lambda self, v: self._set_flag("_hilo_skip", v))
    fuse_chain = property(lambda self: self._fuse_chain, lambda self, v: self._set_flag("_fuse_chain", v))
    fuse_tail = property(lambda self: self._fuse_tail, lambda self, v: self._set_flag("_fuse_tail", v))
    tight_pitch = property(lambda self: self._tight_pitch, lambda self, v: self._set_flag("_tight_pitch", v))

    def _skip_hilo(self, plan, c):
        """bf16 plans: keep the long skip `upsampler(LR_conv(body) + fea)` in hi + lo pairs?  (c = its channel count; the hi + lo kernels
        exist for 3 and 4 output tiles.)  Two bf16 roundings of the image itself cost 0.03-0.09 dB on near-detail-free content, the pairs
        bring that to <= 0.003 dB (LAB_NOTES 9.4; tools/emulate_skip.py)."""
        return bool(self._hilo_skip) and plan.store == "bf16" and (c + 15) // 16 in (3, 4)

    def set_compute(self, mode):
        """'f32': exact fp32 MFMA, fp32 activations.  'bf16' / 'f16' (BASELINE.json configs [2]-[4]): the full-resolution
        activations are STORED in that type and are the MFMA operands as they are (v_mfma_f32_16x16x32), accumulation,
        bias, residual and activation stay fp32, one rounding per stored value; the NCHW input / output and the ESA
        low-resolution branch stay fp32."""
        if mode not in L.COMPUTE:
            raise ValueError(f"compute must be one of {sorted(L.COMPUTE)}")
        with self._lock:
            if mode != self.compute:
                self.compute = mode
                self._dirty = True
        return self

    def _store(self):
        """storage type of the full-resolution activations for the current compute mode"""
        return self.compute

    def _s16_convs(self):
        """paths of the convolutions conv_s16_kernel runs in the 16-bit modes: every full-resolution NHWC conv"""
        plan = Plan(1, 32, 32, self._store())
        self._build_plan(plan, self.in_nc)
        paths = {o["w"] for o in _flat_ops(plan.ops) if o["kind"] == "conv" and o["hw"] is None and o["src"] is not INPUT and not o.get("head")}
        for o in plan.ops:
            if o["kind"] == "bs":                        # BSConvU: pointwise + distillation 1x1 weights as hi + lo blobs
                paths.add(o["pw"])
                if o["distill"] is not None:
                    paths.add(o["distill"]["w"])
        return paths

    # -- packing ------------------------------------------------------------------------------
    GRAPH_MAX_PIXELS = 4 * 512 * 512   # larger forwards are GPU-bound by a wide margin: esr_run_ops
    MAX_PLANS = 128                # DIV2K has ~100 distinct LR shapes; a plan is a few tens of KB of host memory
    # Plans of different shapes lay their buffers out in ONE workspace, so after a shape switch another shape's activations lie where
    # this plan keeps its pad channels.  That is harmless: every pad slot is only ever multiplied by a zero weight (packers), added
    # to an accumulator nobody stores, or copied into another pad slot -- stale FINITE values of the SAME element type cannot reach a
    # result (bit-identical outputs in every mode: tools/dbg/rezero_probe.py, test_hundred_shapes_one_workspace).  Same type is what
    # the two arenas of a Plan are for: the low-resolution fp32 maps of all plans live in front of the workspace (`_lo_cap` bytes),
    # the full-resolution buffers behind them, so a 16-bit plan never reads another shape's fp32 bytes (Inf / NaN patterns).
    # Re-zeroing 130 MB per forward was 26 us of a 0.7 ms image (every DIV2K image has its own shape): +3 % in DIV2K mode (A/B).
    # Set True to isolate forwards from a previous one whose activations overflowed to Inf / NaN.
    rezero_on_switch = False

    def _mark_dirty(self):
        self._dirty = True

    def invalidate_workspaces(self):
        """Forget what the per-stream workspaces hold: the next forward on each stream zero-fills before it runs.  For callers that
        saw a non-finite output (overflowing activations in a 16-bit mode): with rezero_on_switch = False another shape's pad
        channels may lie where Inf / NaN were stored, and 0 * Inf = NaN would reach its results."""
        with self._lock:
            for ctx in self._ctxs.values():
                ctx.ws_owner = None

    def _apply(self, fn, *args, **kwargs):          # .to() / .cuda() / .float(): parameters move or change
        self._dirty = True
        return super()._apply(fn, *args, **kwargs)

    def _signature(self):
        return tuple((p.data_ptr(), p._version, p.device) for p in self.parameters())

    def _drop_plans(self):
        with self._lock:
            for ctx in self._ctxs.values():
                for prof in ctx.profs.values():
                    L.lib().esr_prof_destroy(prof)
                ctx.profs = {}
                for ent in ctx.plans.values():
                    ent.drop_graph()
                ctx.plans.clear()
                ctx.ws_owner = None

    MAX_STREAMS = 16               # contexts kept (LRU): bench.py's DIV2K mode uses 8 streams + the default stream of the replay leg; the
                                   # least recently USED context beyond that is dropped (its workspace is freed, its plans are rebuilt on return)

    def _ctx(self, device):
        """The context of the CURRENT HIP stream of `device` (torch.cuda.stream(...) selects it, like every torch op)."""
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        ctx = self._ctxs.pop(key, None)
        if ctx is None:
            ctx = _StreamCtx()
        self._ctxs[key] = ctx          # (re)inserted at the end: dict order = least recently used first
        while len(self._ctxs) > self.MAX_STREAMS:
            old = next(iter(self._ctxs))
            for prof in self._ctxs.pop(old).profs.values():
                L.lib().esr_prof_destroy(prof)
        return ctx

    # the default-stream context under its historical names (tests, tools)
    def _ctx0(self):
        dev = next(self.parameters()).device
        return self._ctxs.get((dev, torch.cuda.default_stream(dev).cuda_stream)) or _StreamCtx()

    _plans = property(lambda self: self._ctx0().plans)
    _ws = property(lambda self: self._ctx0().ws)

    def _post_convs(self):
        """paths of the 1x1 convolutions evaluated in a 16-bit conv's epilogue (post / post2) in the current mode"""
        plan = Plan(1, 32, 32, self._store())
        self._build_plan(plan, self.in_nc)
        out = set()
        for o in _flat_ops(plan.ops):
            t = o.get("post") if o["kind"] == "conv" else None
            if t is not None:
                out.add(t["w"])
                if t.get("post2") is not None:
                    out.add(t["post2"]["w"])
        return out

    def _lowres_dense(self):
        """paths of the layers evaluated inside esr_esa_lowres_f32 (dense 16 x 16 blobs: esr_pack_dense_f32)"""
        plan = Plan(1, 32, 32, self._store())
        self._build_plan(plan, self.in_nc)
        return {ly["w"] for o in plan.ops if o["kind"] == "lowres" for ly in o["layers"]}

    def _head_convs(self):
        """paths of the convolutions that read the network input in the current 16-bit mode (lowered to pack + conv_s16)"""
        plan = Plan(1, 32, 32, self._store())
        self._build_plan(plan, self.in_nc)
        return {o["w"][:-len('#head')] for o in plan.ops if o["kind"] == "conv" and o.get("head")}

    def _cin_map(self, path, cin_map, store):
        """physical-slot -> logical-channel map of a conv reading a padded concat buffer; networks whose slice padding
        depends on the storage type override this"""
        return cin_map

    def repack(self, device):
        with self._lock:
            self._repack(device)

    def _repack(self, device):
        packed = {}
        s16 = self._s16_convs() if self._store() != "f32" else set()
        for path, (cin, cout, k, cin_map) in self._conv_specs.items():
            leaf = self._leaf(path)
            packed[path] = pack_conv(leaf.weight, leaf.bias, cin_map=self._cin_map(path, cin_map, "f32")).to(device)
            if k == 3 and self.winograd and self._store() == "f32" and cin >= 32 and cout <= 64:
                packed[path + "#wino"] = pack_wino(leaf.weight, leaf.bias, cin_map=self._cin_map(path, cin_map, "f32")).to(device)
            if path in s16:
                packed[path + "#s16"] = pack_conv_s16(leaf.weight, leaf.bias, self._store(),
                                                      cin_map=self._cin_map(path, cin_map, self._store())).to(device)
        for path in sorted(self._post_convs()) if self._store() != "f32" else ():
            leaf = self._leaf(path)
            packed[path + "#post"] = pack_post_s16(leaf.weight, leaf.bias, self._store()).to(device)
        for path in sorted(self._head_convs()) if self._store() != "f32" else ():
            if path in self._conv_specs and not self._conv_specs[path][3]:          # (custom-packed heads: _extra_pack)
                leaf = self._leaf(path)
                packed[path + "#head#s16"] = pack_head_s16(leaf.weight, leaf.bias, self._store()).to(device)
        for path, (cin_p, cout_p) in self._dense_specs.items():
            leaf = self._leaf(path)
            packed[path] = pack_dense(leaf.weight, leaf.bias, cin_p, cout_p).to(device)
        for path in sorted(self._lowres_dense()):        # 3x3 / pointwise layers of the fused ESA branch: plain dense layout
            leaf = self._leaf(path)
            packed[path + "#dense"] = pack_dense(leaf.weight, leaf.bias, L.ESA_FP, L.ESA_FP).to(device)
        for path in self._dw_specs:
            leaf = self._leaf(path)
            packed[path] = pack_dw(leaf.weight, leaf.bias).to(device)
        self._extra_pack(packed, device)
        if torch.device(device).type == "cuda":
            torch.cuda.synchronize(device)           # the blobs are read by forwards on ANY stream from here on
        self._packed = packed
        self._packed_sig = self._signature()
        self._dirty = False
        self._drop_plans()

    def _extra_pack(self, packed, device):
        pass

    # -- forward ------------------------------------------------------------------------------
    def _build_plan(self, plan):
        raise NotImplementedError

    def _ensure_packed(self, device):
        """Weights are re-packed after load_state_dict / .to() / set_compute (dirty flag) or when the blobs live on
        another device.  In-place edits of a parameter's storage are not tracked: call repack() after them."""
        if self._packed is None or self._dirty or next(iter(self._packed.values())).device != device:
            self.repack(device)

    def _entry(self, key, ctx=None):
        """Cached plan for (n, c, h, w, device) in the current stream's context: built on first use, finalized against that
        context's workspace."""
        n, c, h, w, device = key
        if ctx is None:
            ctx = self._ctx(device)
        ent = ctx.plans.get(key)
        if ent is None:
            plan = Plan(n, h, w, self._store())
            self._build_plan(plan, c)
            ent = _Entry(plan)
            ctx.plans[key] = ent
            while len(ctx.plans) > self.MAX_PLANS:
                old, _ = ctx.plans.popitem(last=False)
                prof = ctx.profs.pop(old, None)
                if prof is not None:
                    L.lib().esr_prof_destroy(prof)
        else:
            ctx.plans.move_to_end(key)
        if ent.plan.total_lo > ctx.lo_cap:
            ctx.lo_cap = ent.plan.total_lo             # the low-resolution arena grows: every plan's full-resolution buffers move
            ctx.ws_owner = None
        need = max(ctx.lo_cap + ent.plan.total, 256)
        if ctx.ws is None or ctx.ws.device != device or ctx.ws.numel() < need:
            ctx.ws = None                                               # release before the larger allocation
            ctx.ws = torch.zeros(need, dtype=torch.uint8, device=device)
            ctx.ws_owner = key                                          # fresh zeros: this plan's pad channels are 0
        base = (ctx.ws.data_ptr(), ctx.lo_cap)
        if ent.base != base:
            ent.drop_graph()                        # (the captured launches hold the old workspace addresses)
            ent.arr, ent.in_idx, ent.out_idx = ent.plan.finalize(base, self._packed)
            ent.base = base
        if ctx.ws_owner is None:
            ctx.ws.zero_()              # unknown content (plans dropped: another storage type's bytes; the arenas moved)
        elif ctx.ws_owner != key and self.rezero_on_switch:
            ctx.ws.zero_()              # isolation requested
        ctx.ws_owner = key
        return ent

    def prepare(self, shape, device=None):
        """Everything a forward of an [N, C, H, W] input needs besides the kernels themselves: weight packing (first
        call, or after load_state_dict / .to()), plan construction, workspace growth and zero fill.  Callers that time
        `model(x)` with an event pair (test_demo.py:429-432) call this first so that the bracket holds only the
        esr_run_ops launches; forward() does the same work itself when it was not called."""
        device = torch.device(device) if device is not None else next(self.parameters()).device
        if device.type != "cuda":
            raise L.EsrError(f"{type(self).__name__}.prepare: device {device} -- this engine only runs on an MI355X")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        n, c, h, w = (int(v) for v in shape)
        with self._lock:
            self._ensure_packed(device)
            return self._entry((n, c, h, w, device))

    def forward(self, x):
        """NCHW fp32 [N, in_nc, H, W] on the GPU -> NCHW fp32 [N, out_nc, 4H, 4W]: one esr::sr_forward call."""
        if (type(x) is torch.Tensor and x.is_cuda and not torch.compiler.is_compiling()
                and torch._C._get_tracing_state() is None and torch._C._len_torch_dispatch_stack() == 0):
            # plain eager call on a real tensor: straight to the C ABI (the registered operator costs ~8 us of dispatch per call -- a
            # fifth of a graph-launched forward's host time).  torch.jit.trace, make_fx, any active TorchDispatchMode (FlopCounterMode,
            # profiler modes), FakeTensor and torch.compile go through esr::sr_forward below: they must SEE the operator
            return self._forward_impl(x)
        if _LIVE.get(self.handle) is not self:          # a copy.deepcopy of a module carries its source's handle: take a fresh one
            self.handle = next(_HANDLES)
            _LIVE[self.handle] = self
        return torch.ops.esr.sr_forward(x, self.handle)

    def _forward_impl(self, x):
        if not x.is_cuda:
            raise L.EsrError(f"{type(self).__name__}: input is on {x.device}; this engine only runs on an "
                             "MI355X through libesr_hip.so and has no CPU fallback (use oracle/ for CPU checks)")
        if x.dtype != torch.float32 or x.dim() != 4:
            raise L.EsrError("expected a 4-D float32 NCHW tensor (uint2tensor4 output)")
        lib = L.lib()
        x = x.contiguous()
        n, c, h, w = x.shape
        key = (n, c, h, w, x.device)
        y = torch.empty((n, self.out_nc, h * self.upscale, w * self.upscale), dtype=torch.float32, device=x.device)
        if n == 0:
            return y                        # an empty batch: nn.Conv2d returns an empty tensor too (nothing to launch: a grid of 0 blocks is an error)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        # One lock per model around the host-side bookkeeping AND the enqueue: the cached esr_op array of a (stream, shape) carries
        # this call's x / y pointers from the patch below until esr_run_ops has read it (it only enqueues: ~0.1-0.25 ms of host
        # time, no device wait), and plan / workspace creation mutates dicts.  Two Python threads calling one model therefore
        # enqueue their forwards one after the other -- on their own streams the forwards still overlap on the GPU.
        with self._lock:
            self._ensure_packed(x.device)
            ctx = self._ctx(x.device)
            ent = self._entry(key, ctx)
            arr = ent.arr
            for i in ent.in_idx:
                arr[i].conv.inp.ptr = x.data_ptr()
            for i in ent.out_idx:
                arr[i].conv.out0.ptr = y.data_ptr()
            if self._prof_passes > 0:
                prof = ctx.profs.get(key)
                if prof is None:
                    prof = ctypes.c_void_p()
                    L.check(lib.esr_prof_create(len(arr), self._prof_passes, ctypes.byref(prof)), "esr_prof_create")
                    ctx.profs[key] = prof
                rc = lib.esr_run_ops_profiled(arr, len(arr), ctypes.c_void_p(stream), prof)
            elif self.use_graphs and n * h * w <= self.GRAPH_MAX_PIXELS:
                # small forwards (one image: test_demo.py:416-433) are host-bound -- 20-35 launches at 5-8 us each: from the second
                # forward of a (stream, shape) on, the launches are replayed as ONE hipGraphLaunch with x / y patched (esr_graph_launch)
                ent.uses += 1
                if ent.graph is None and ent.uses >= 2:
                    gh = ctypes.c_void_p()
                    rc = lib.esr_graph_create(arr, len(arr), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.byref(gh))
                    if rc == L.ESR_OK:
                        ent.graph = gh
                    else:
                        ent.uses = -(1 << 30)      # this plan's launches cannot be captured: stay on esr_run_ops
                if ent.graph is not None:
                    rc = lib.esr_graph_launch(ent.graph, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(stream))
                else:
                    rc = lib.esr_run_ops(arr, len(arr), ctypes.c_void_p(stream))
            else:
                rc = lib.esr_run_ops(arr, len(arr), ctypes.c_void_p(stream))
        L.check(rc, f"{type(self).__name__}.forward")
        return y

    # -- per-kernel timing (HIP events on the launch stream; see esr_run_ops_profiled) ---------
    def enable_profiling(self, max_passes):
        self.disable_profiling()
        self._prof_passes = int(max_passes)

    def disable_profiling(self):
        with self._lock:
            for ctx in self._ctxs.values():
                for prof in ctx.profs.values():
                    L.lib().esr_prof_destroy(prof)
                ctx.profs = {}
            self._prof_passes = 0

    def op_costs(self, plan, arr=None):
        """Per op of `plan`: device kernel symbol, ALGORITHMIC flops (2 x MACs of the matrix products the op stands
        for) and ALGORITHMIC HBM bytes (every input slice / residual / weight read once, every output written once,
        at the storage type of the buffer) -- the numerators of bench.py's roofline leg."""
        es = plan.esize
        out = []
        for i, o in enumerate(plan.ops):
            kind = o["kind"]
            hw = o.get("hw")
            npix = plan.npix if hw is None else plan.n * hw[0] * hw[1]
            e_act = es if hw is None else 4                       # low-resolution maps are fp32
            wino = False
            stored = None
            if kind == "pack":                      # the network input read once (fp32 NCHW), its 16 16-bit slots per pixel written once
                out.append(dict(name="pack_input", kernel="pack_input_kernel", cin=o["cin"], cout=16, k=0, flops=0.0, flops_exec=0.0,
                                read_bytes=float(npix * o["cin"] * 4), write_bytes=float(npix * 16 * 2),
                                stored_bytes=float(npix * (o["cin"] * 4 + 32))))
                continue
            if kind == "conv":
                nt = (o["cout"] + 15) // 16
                nw = L.lib().esr_conv_block_waves(ctypes.byref(arr[i].conv)) if arr is not None else 0
                kern = f"conv_f32_kernel<NT={nt},KS={o['k']},NCHW_IN={int(o['src'] is INPUT)},NW={nw}>"
                if plan.esize == 2 and hw is None and o["src"] is not INPUT:
                    kern = f"conv_s16_kernel<NT={nt},KS={o['k']},NW={nw or 8},{plan.store}>"
                if isinstance(o.get("dst1"), Buffer) and o["dst1"].blocked:
                    kern = kern[:-1] + ",BLK>"          # the instantiation with the channel-blocked split store
                wino = arr is not None and bool(arr[i].conv.wino_wpacked) and bool(L.lib().esr_wino_supported(ctypes.byref(arr[i].conv)))
                if wino:
                    # the device symbol as rocprofv3 prints it: wino_f32_kernel<ACT, RES, Y1BLK> (esr_wino.hip: esr_conv2d_wino)
                    rm = o.get("res_mode", L.RES_NONE) if o["res"] is not None else L.RES_NONE
                    at = o["act"] if (o["act"] == L.ACT_LRELU or (o["act"] == L.ACT_NONE and rm != L.RES_POST_ACT)) else -1
                    blk = isinstance(o.get("dst1"), Buffer) and o["dst1"].blocked
                    kern = f"wino_f32_kernel<{at}, {rm}, {2 if o['dst'] is OUTPUT else int(blk)}>"
                e_in = 4 if o["src"] is INPUT else e_act
                e_out = 4 if o["dst"] is OUTPUT else e_act
                ca = o["cin_alg"]
                # a residual that IS the conv's input (RFDB's c1_r..c3_r, RLFB's c3_r: the kernel adds the centre tap of the staged tile)
                # is not read again: VERDICT r05 weak #2
                res_read = o["res"] is not None and not _same_view(o["res"], o["src"])
                wb = 4.0 * ca * o["cout"] * o["k"] ** 2          # weight bytes (part of both the algorithmic and the stored figure)
                rd = npix * (ca * e_in + (o["cout"] * e_act if res_read else 0)) + wb
                if o.get("head"):
                    rd += npix * (16 * 2 - ca * e_in)  # the head reads the packed 16-slot copy (esr_pack_input_s16), not the fp32 input
                wr = float(npix * o["cout"] * e_out) if o["dst"] is not None else 0.0
                hl = o.get("hilo", 0)                # hi + lo tensors: twice the 16-bit bytes
                rd += npix * e_act * ((ca if hl & L.HILO_IN else 0) + (o["cout"] if (hl & L.HILO_RES and res_read) else 0))
                wr += npix * e_act * o["cout"] if hl & L.HILO_OUT else 0.0
                if hl:
                    kern = kern[:-1] + ",HILO>"
                flops = 2.0 * npix * ca * o["cout"] * o["k"] * o["k"]
                if o.get("bs_of") is not None:      # the BSConvU it stands for: pointwise GEMM + depthwise 3x3
                    flops = 2.0 * npix * (ca * o["cout"] + 9 * o["cout"])
                    wb = 4.0 * (ca * o["cout"] + 10 * o["cout"])
                    rd = npix * (ca * e_in + (o["cout"] * e_act if res_read else 0)) + wb
                    if o.get("head"):
                        rd += npix * (16 * 2 - ca * e_in)
                t = o.get("tail")
                if t is not None:                   # 3x3 -> 1x1 in one kernel: both GEMMs' flops, the 1x1's traffic
                    kern = f"conv_f32_kernel<NT={nt},KS=3,NCHW_IN=0,NW=4,TAIL={(t['cout'] + 15) // 16}>"
                    if plan.esize == 2:
                        kern = f"rfdb_tail_kernel<{plan.store}>"
                    if (o["cin"] + 7) // 8 == 6 and t["cat_c"] == 48 and t["cout"] == 64 and o.get("res_mode", L.RES_NONE) in (L.RES_NONE, L.RES_PRE_ACT):
                        kern = f"imdb_tail_kernel<FOLD={int(o['res'] is not None)}>"       # esr_hip.hip: imdb_tail_shape()
                    cat_alg = t.get("cat_c_alg", t["cat_c"])    # logical channels of the concat (16-bit tail: three dense 32-slot tensors of dc)
                    k1 = cat_alg + o["cout"]
                    flops += 2.0 * npix * k1 * t["cout"]
                    wb = 4.0 * (ca * o["cout"] * 9 + k1 * t["cout"])
                    rd = npix * e_act * (ca + cat_alg + (t["cout"] if res_read else 0)) + wb
                    wr = float(npix * e_act * t["cout"])
                t = o.get("post")
                if t is not None:                   # + the 1x1 of the activated output, stored by the same launch
                    if nw == 8:
                        kern = kern[:-1] + f",POST={(t['cout'] + 15) // 16}>"
                    post_in = o["cout"] if o.get("tail") is None else o["tail"]["cout"]      # (behind a tail: the 1x1 of the tail's result)
                    flops += 2.0 * npix * post_in * t["cout"]
                    wr += npix * e_act * t["cout"]
                    t2 = t.get("post2")
                    if t2 is not None:
                        kern = kern[:-1] + f"+{(t2['cout'] + 15) // 16}>"
                        flops += 2.0 * npix * t["cout"] * t2["cout"]
                        wr += npix * e_act * t2["cout"]
                # STORED bytes: what the launch moves when pad channels travel (pitch 64 for nf = 50, 32 for dc = 25): whole pitches of
                # Buffers, chunk-rounded slices -- next to the algorithmic bytes above, so the padding waste is a reported number
                chunk = 8 if es == 4 else 16
                hl_in, hl_res, hl_out = (2 if hl & L.HILO_IN else 1), (2 if hl & L.HILO_RES else 1), (2 if hl & L.HILO_OUT else 1)
                sc = lambda v, logical: _stored_channels(v.seg(0) if (isinstance(v, Planar) and hl) else v, logical, chunk)
                tl = o.get("tail")
                st = 16 * 2 * npix if o.get("head") else npix * e_in * hl_in * sc(o["src"], ca)
                if tl is not None:
                    st += npix * e_act * sc(tl["cat"], tl["cat_c"])
                if res_read:
                    st += npix * e_act * hl_res * sc(o["res"], o["cout"] if tl is None else tl["cout"])
                if o["dst"] is not None:
                    st += npix * e_out * hl_out * sc(o["dst"], o["split"] if o.get("dst1") is not None else (o["cout"] if tl is None else tl["cout"]))
                if o.get("dst1") is not None:
                    st += npix * e_act * sc(o["dst1"], o["cout"] - o["split"])
                pt = o.get("post")
                if pt is not None:
                    st += npix * e_act * sc(pt["dst"], pt["cout"])
                    if pt.get("post2") is not None:
                        st += npix * e_act * sc(pt["post2"]["dst"], pt["post2"]["cout"])
                if pt is not None:
                    wb += 4.0 * o["cout"] * pt["cout"] + (4.0 * pt["cout"] * pt["post2"]["cout"] if pt.get("post2") is not None else 0.0)
                stored = float(st) + wb
            elif kind == "chain":                   # the block's 3x3 chain + its two 1x1s in one launch: the input read once, only the 1x1 results written
                sub = o["replaces"]
                kern = f"rlfb_chain_kernel<{plan.store}>"
                flops = sum(2.0 * npix * so["cin_alg"] * so["cout"] * 9 for so in sub)
                t = sub[-1]["post"]
                t2 = t["post2"]
                flops += 2.0 * npix * (sub[-1]["cout"] * t["cout"] + t["cout"] * t2["cout"])
                rd = float(npix * sub[0]["cin_alg"] * es) + 4.0 * (sum(so["cin_alg"] * so["cout"] * 9 for so in sub) + sub[-1]["cout"] * t["cout"] + t["cout"] * t2["cout"])
                wr = float(npix * es * (t["cout"] + t2["cout"]))
            elif kind == "lowres":                  # conv2 (s2) + pooling + the 3x3 layers behind it: two launches, only the pooled map in between
                src, dst = o["src"], o["dst"]
                npl = plan.n * dst.h * dst.w
                h2, w2 = (src.h - 3) // 2 + 1, (src.w - 3) // 2 + 1
                f = o["f"]
                kern = f"esa_s2pool{'16' if plan.esize == 2 else ''}_kernel<{L.STORE[plan.store]}> + esa_chain_kernel"
                flops = 2.0 * 9 * f * f * plan.n * h2 * w2
                for ly in o["layers"]:
                    flops += 2.0 * npl * (9 * f * f if ly["kind"] == 0 else f * f + 9 * f)
                rd = float(plan.n * src.h * src.w * f * src.esize + npl * f * 4)
                wr = float(2 * npl * f * 4)
            elif kind == "bs":                      # pointwise (+ distillation) GEMM + depthwise, one launch
                t = o["distill"]
                dco = t["cout"] if t is not None else 0
                kern = f"bsconv_kernel<NTP={(o['cout'] + 15) // 16},NTD={(dco + 15) // 16}>"
                flops = 2.0 * npix * (o["cin"] * (o["cout"] + dco) + 9 * o["cout"])
                rd = npix * e_act * (o["cin"] + (o["cout"] if (o["res"] is not None and not _same_view(o["res"], o["src"])) else 0)) \
                    + 4.0 * (o["cin"] * (o["cout"] + dco) + 10 * o["cout"])
                wr = float(npix * e_act * (o["cout"] + dco))
            elif kind == "dw":
                kern = "dwconv3x3_kernel"
                flops = 2.0 * 9 * o["cout"] * npix
                rd = float(npix * e_act * (o["cin"] + (o["cout"] if (o["res"] is not None and not _same_view(o["res"], o["src"])) else 0)))
                wr = float(npix * e_act * o["cout"])
            elif kind == "s2":                      # reads the full-resolution conv1 map, writes the half-resolution one
                kern = "conv3x3s2_kernel"
                src = o["src"]
                flops = 2.0 * 9 * o["f"] * o["f"] * npix
                rd = float(plan.n * src.h * src.w * o["f"] * src.esize)
                wr = float(npix * o["f"] * 4)
            elif kind == "pool":
                kern = "maxpool7s3_kernel"
                flops = 0.0
                rd = float(plan.n * o["src"].h * o["src"].w * 16 * 4)
                wr = float(plan.n * o["dst"].h * o["dst"].w * 16 * 4)
            else:                                   # apply: conv_f (f x f) + conv4 (f x c) per pixel, x read, y written
                kern = "esa_apply_kernel"
                flops = 2.0 * plan.npix * (o["f"] * o["f"] + o["f"] * o["c"])
                rd = float(plan.npix * es * (o["c"] + o["f"]) + plan.n * o["c3"].h * o["c3"].w * o["f"] * 4)
                wr = 0.0 if o.get("skip_y") else float(plan.npix * es * o["c"])
                kin = o["c"]
                for t in o.get("post") or ():       # 1x1 convolutions in the same launch: their flops, residual reads and stores
                    flops += 2.0 * plan.npix * kin * t["cout"]
                    rd += float(plan.npix * es * t["cout"]) if t.get("res") is not None else 0.0
                    wr += float(plan.npix * es * t["cout"])
                    kin = t["cout"]
            # flops = ALGORITHMIC (direct-convolution) flops; flops_exec = what the matrix cores execute: Winograd F(2x2,3x3) does 16
            # multiplications per 2x2 outputs where the direct form does 36
            fexec = flops * (16.0 / 36.0) if (kind == "conv" and wino) else flops
            if kind == "conv" and o.get("hilo", 0) & L.HILO_IN:
                fexec = 2.0 * flops               # both halves of a hi + lo input meet the weights
            out.append(dict(name=o.get("w", kind), kernel=kern, cin=o.get("cin", 0), cout=o.get("cout", 0), k=o.get("k", 0),
                            flops=flops, flops_exec=fexec, read_bytes=rd, write_bytes=wr,
                            stored_bytes=(rd + wr) if stored is None else max(stored, 0.0)))
        return out

    def collect_profile(self):
        """After a device synchronise: list of dicts {name, kernel, flops, read_bytes, write_bytes, ms_sum, passes} per
        op, summed over the recorded passes of every cached shape."""
        out = []
        for ctx in self._ctxs.values():
            for key, prof in ctx.profs.items():
                ent = ctx.plans.get(key)
                if ent is None or ent.arr is None:
                    continue
                n = len(ent.arr)
                ms = (ctypes.c_double * n)()
                passes = ctypes.c_int(0)
                L.check(L.lib().esr_prof_collect(prof, ms, n, ctypes.byref(passes)), "esr_prof_collect")
                buf = ctypes.create_string_buffer(256)
                for i, c in enumerate(self.op_costs(ent.plan, ent.arr)):
                    c.update(ms_sum=ms[i], passes=passes.value, shape=key[:4], label=c["kernel"])
                    # the device symbol the library launched for this op (rocprofv3's spelling); `label` keeps the descriptive name
                    if L.lib().esr_prof_kernel_symbol(prof, i, buf, 256) == 0 and buf.value:
                        c["kernel"] = buf.value.decode()
                    out.append(c)
        return out

    def _complexity_terms(self, plan, o):
        """(flops, activations, n_conv) that the reference's model_summary hooks would count for op `o`."""
        flops = acts = nconv = 0
        if o["kind"] in ("lowres", "chain"):        # the fused ESA branch / 3x3 chain counts as the nn.Conv2d / nn.Linear calls it replaces
            for sub in o["replaces"]:
                f_, a_, n_ = self._complexity_terms(plan, sub)
                flops, acts, nconv = flops + f_, acts + a_, nconv + n_
            return flops, acts, nconv
        for (cin, cout, k, npix, act) in self._counted_convs(plan, o):
            flops += k * k * cin * cout * npix
            acts += cout * npix
            nconv += 1
            if act in (L.ACT_LRELU, L.ACT_RELU):
                flops += cout * npix
        return flops, acts, nconv

    def _counted_convs(self, plan, o):
        """nn.Conv2d calls of the REFERENCE graph that op `o` stands for: (cin, cout, k, pixels, act) tuples
        (logical channel counts as the reference's hooks see them)."""
        if o["kind"] == "conv":
            if not o.get("counted", True):
                return []
            npix = plan.npix if o["hw"] is None else plan.n * o["hw"][0] * o["hw"][1]
            t = o.get("tail")
            if t is not None:                       # two nn.Conv2d calls of the reference in one launch
                return [(o["cin"], o["cout"], o["k"], npix, t.get("mid_act", L.ACT_NONE)),
                        (t["cat_c"] + o["cout"], t["cout"], 1, npix, o["act"])]
            t = o.get("post")
            if t is not None:
                r = [(o["cin"], o["cout"], o["k"], npix, o["act"]), (o["cout"], t["cout"], 1, npix, t.get("act", L.ACT_NONE))]
                if t.get("post2") is not None:
                    r.append((t["cout"], t["post2"]["cout"], 1, npix, L.ACT_NONE))
                return r
            return [(o["cin"], o["cout"], o["k"], npix, o["act"])]
        if o["kind"] == "s2":
            return [(o["f"], o["f"], 3, plan.n * o["dst"].h * o["dst"].w, L.ACT_NONE)]
        if o["kind"] == "apply":   # conv_f (f->f) and conv4 (f->c), both full resolution (+ the 1x1s riding in the launch)
            r = [(o["f"], o["f"], 1, plan.npix, L.ACT_NONE), (o["f"], o["c"], 1, plan.npix, L.ACT_NONE)]
            kin = o["c"]
            for t in o.get("post") or ():
                r.append((t.get("cin_alg", kin), t.get("cout_alg", t["cout"]), 1, plan.npix, t.get("act", L.ACT_NONE)))
                kin = t["cout"]
            return r
        return []

    def workspace_bytes(self, n, h, w, c=3):
        plan = Plan(n, h, w, self._store())
        self._build_plan(plan, c)
        return plan.total + plan.total_lo
