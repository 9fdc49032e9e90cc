"""RLFN_cut (x4) on the HIP engine -- drop-in for `models.team04_rlfn.RLFN_cut` (team04_rlfn.py:124-155).

Same constructor keywords and the same 78 state_dict keys (`fea_conv`, `B{k}.{c1_r,c2_r,c3_r,c5}`,
`B{k}.esa.{conv1,conv_f,conv2,conv3,conv4}`, `LR_conv`, `upsampler.0`).  nf=46 lives in NHWC buffers of
pitch 48 with two zero pad channels.  Per RLFB (team04_rlfn.py:109-122): three fused 3x3+LeakyReLU (the third
adds the block input AFTER the activation), the 1x1 c5, then ESA as 1x1 conv1 -> 3x3/s2 -> maxpool 7/3 ->
3x3 at ~H/6 -> one fused full-resolution tail (bilinear + conv_f + conv4 + sigmoid * x).
"""
from . import _lib as L
from .engine import INPUT, OUTPUT, HipSRModel

FP = L.ESA_FP


def _pad8(c):
    return (c + 7) // 8 * 8


def _lowres(h, w):
    h2, w2 = (h - 3) // 2 + 1, (w - 3) // 2 + 1
    return h2, w2, (h2 - 7) // 3 + 1, (w2 - 7) // 3 + 1


class RLFN_cut(HipSRModel):
    def __init__(self, in_nc=3, out_nc=3, nf=46, mf=48, upscale=4):
        super().__init__()
        if upscale != 4 or nf > 64 or mf > 64 or in_nc > 4 or out_nc * 16 > 64:
            raise NotImplementedError('HIP RLFN_cut supports upscale=4, nf/mf <= 64, in_nc <= 4, out_nc <= 4')
        self.in_nc, self.out_nc, self.nf, self.mf, self.upscale = in_nc, out_nc, nf, mf, upscale
        self.esa_channels = 16
        self.scale_idx = 0
        f = self.esa_channels
        cp4 = (nf + 3) // 4 * 4
        self._add_conv('fea_conv', in_nc, nf, 3)
        for k in range(1, 5):
            b = f'B{k}.'
            self._add_conv(b + 'c1_r', nf, mf, 3)
            self._add_conv(b + 'c2_r', mf, mf, 3)
            self._add_conv(b + 'c3_r', mf, nf, 3)
            self._add_conv(b + 'c5', nf, nf, 1)
            self._add_conv(b + 'esa.conv1', nf, f, 1)
            self._add_conv(b + 'esa.conv_f', f, f, 1, dense=(FP, FP))
            self._add_conv(b + 'esa.conv2', f, f, 3, dense=(FP, FP), stride=2, padding=0)
            self._add_conv(b + 'esa.conv3', f, f, 3)
            self._add_conv(b + 'esa.conv4', f, nf, 1, dense=(FP, cp4))
        self._add_conv('LR_conv', nf, nf, 3)
        self._add_conv('upsampler.0', nf, out_nc * upscale * upscale, 3)

    def set_scale(self, scale_idx):
        self.scale_idx = scale_idx

    def _build_plan(self, plan, c):
        if c != self.in_nc:
            raise L.EsrError(f'RLFN_cut expects {self.in_nc} input channels, got {c}')
        if plan.h < 15 or plan.w < 15:
            raise L.EsrError('ESA needs H, W >= 15 (3x3/s2 then 7x7/s3 pooling)')
        nf, mf, f = self.nf, self.mf, self.esa_channels
        P, M = plan.cpad(nf), plan.cpad(mf)
        h2, w2, h3, w3 = _lowres(plan.h, plan.w)
        # bf16: `fea` and `out_lr` -- the long skip, team04_rlfn.py:149-150 -- are hi + lo pairs (Plan.pair: two dense tensors)
        hl = self._skip_hilo(plan, nf)
        fea2 = plan.pair('fea', P) if hl else None
        fea = fea2.seg(0) if hl else plan.buffer('fea', P)
        out_lr2 = plan.pair('out_lr', P) if hl else None
        xa, xb = plan.buffer('xa', P), plan.buffer('xb', P)
        t1, t2 = plan.buffer('t1', M), plan.buffer('t2', M)
        u, v = plan.buffer('u', P), plan.buffer('v', P)
        c1 = plan.buffer('esa_c1', FP)
        lo2 = plan.buffer('esa_s2', FP, h2, w2)
        lo3 = plan.buffer('esa_pool', FP, h3, w3)
        lo4 = plan.buffer('esa_c3', FP, h3, w3)
        act = dict(act=L.ACT_LRELU, slope=0.05)
        plan.conv('fea_conv', INPUT, fea2 if hl else fea, self.in_nc, nf, hilo=L.HILO_OUT if hl else 0)
        cur, nxt = fea, xa
        for k in range(1, 5):
            b = f'B{k}.'
            mark_c = len(plan.ops)
            plan.conv(b + 'c1_r', cur, t1, nf, mf, **act)
            plan.conv(b + 'c2_r', t1, t2, mf, mf, **act)
            if plan.esize == 2 and (nf + 15) // 16 == 3 and f <= 16:
                # 16-bit storage: c5 and esa.conv1 are evaluated in c3_r's epilogue on the fp32 tile (esr_conv_desc.post_* /
                # post2_*): u = lrelu(c3_r(..)) + x (team04_rlfn.py:117-119) is never stored, never rounded -- two launches
                # and three tensor passes less per block, and the rounding that cost RLFN bf16 most of its PSNR budget is gone
                plan.conv(b + 'c3_r', t2, None, mf, nf, res=cur, res_mode=L.RES_POST_ACT, **act,
                          post=dict(w=b + 'c5', dst=v, cout=nf, act=L.ACT_NONE,
                                    post2=dict(w=b + 'esa.conv1', dst=c1, cout=f)))
                if self.fuse_chain and (mf + 15) // 16 == 3:
                    # ... and the three 3x3s as ONE launch: a layer-per-SIMD pipeline with t1 / t2 / u in LDS (esr_conv_chain_s16, round 5)
                    plan.chain(mark_c)
            else:
                plan.conv(b + 'c3_r', t2, u, mf, nf, res=cur, res_mode=L.RES_POST_ACT, **act)
                plan.conv(b + 'c5', u, v, nf, nf, k=1)
                plan.conv(b + 'esa.conv1', v, c1, nf, f, k=1)
            mark = len(plan.ops)
            plan.conv3x3s2(b + 'esa.conv2', c1, lo2, f)
            plan.maxpool7s3(lo2, lo3)
            plan.conv(b + 'esa.conv3', lo3, lo4, f, f, hw=(h3, w3))
            if self.fuse_esa_lowres:
                plan.esa_lowres(mark, c1, lo3, lo4, f, b + 'esa.conv2', [dict(kind=0, act=L.ACT_NONE, w=b + 'esa.conv3')])
            plan.esa_apply(b + 'esa.conv_f', b + 'esa.conv4', v, c1, lo4, nxt, nf, f)
            cur = nxt
            nxt = xb if cur is xa else xa
        if hl:
            plan.conv('LR_conv', cur, out_lr2, nf, nf, res=fea2, res_mode=L.RES_PRE_ACT, hilo=L.HILO_RES | L.HILO_OUT)
            plan.conv('upsampler.0', out_lr2, OUTPUT, nf, self.out_nc * 16, hilo=L.HILO_IN)
        else:
            plan.conv('LR_conv', cur, u, nf, nf, res=fea, res_mode=L.RES_PRE_ACT)
            plan.conv('upsampler.0', u, OUTPUT, nf, self.out_nc * 16)
