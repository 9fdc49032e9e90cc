"""BSRN (x4) on the HIP engine -- drop-in for `models.team18_bsrn.BSRN` (team18_bsrn.py:182-236).

Same constructor keywords and the same 247 state_dict keys (`*.pw.{weight[out,in],bias}`, `*.dw.weight[C,1,3,3]`,
`B{k}.cw[1,C]`, `c1.weight[C,5C]`, `upsampler.upsampleOneStep.0.*`, ...).  Every nn.Linear on the permuted NHWC
tensor is a 1x1 convolution in this engine's native layout (the reference's 131 permutes vanish); BSConvU
(team18_bsrn.py:82-88) = 1x1 conv + depthwise 3x3 kernel with the residual / GELU fused into the depthwise
epilogue.  Two load-time folds, both exact up to one fp32 rounding:
  * `cat([x,x,x,x],1)` (team18_bsrn.py:218) -> the 4 column groups of fea_conv.pw.weight[C,12] are summed to [C,3];
  * `out * cw` (team18_bsrn.py:169) -> folded into the columns of conv_out.weight.
16-bit storage modes: a full-resolution BSConvU runs as ONE dense 3x3 on the matrix cores (conv_s16_kernel) with the merged
weights W[c,k,tap] = dw[c,tap] * pw[c,k] -- at 16x the fp32 matrix rate the 9x larger product is cheaper than the VALU
depthwise pass (0.12 ms against bsconv_kernel's 0.37 ms per launch at 32x256x256, C = 48).  The depthwise conv zero-pads the
pointwise OUTPUT (bias included), which a dense conv of the input cannot express at the image border: the bias term goes
through esr_conv_desc.border_bias (a 16-row table indexed by which sides of the pixel lie outside), see _merged_bsconv.
"""
import torch

from . import _lib as L
from .engine import INPUT, OUTPUT, HipSRModel, Planar, pack_apply_post, pack_conv, pack_conv_s16, pack_head_s16
from .rlfn import FP, _lowres, _pad8


class BSRN(HipSRModel):
    def __init__(self, num_in_ch=3, num_feat=50, num_block=4, num_out_ch=3, upscale=4,
                 conv='BSConvU', upsampler='pixelshuffledirect', p=0.25):
        super().__init__()
        if conv != 'BSConvU':
            raise NotImplementedError('HIP BSRN implements conv="BSConvU" (the registry configuration, test_demo.py:155-156)')
        if upsampler != 'pixelshuffledirect':
            raise NotImplementedError(("Check the Upsampeler. None or not support yet"))
        if upscale != 4 or num_feat > 64 or num_feat % 8 or num_in_ch > 4 or num_out_ch * 16 > 64:
            raise NotImplementedError('HIP BSRN supports upscale=4, num_feat in {8..64 step 8}, in<=4, out<=4')
        print(conv)                                               # team18_bsrn.py:189 prints the conv type
        self.in_nc, self.out_nc, self.C, self.nb, self.upscale = num_in_ch, num_out_ch, num_feat, num_block, upscale
        C = num_feat
        self.dc, self.f = C // 2, C // 4
        dc, f = self.dc, self.f
        cp4 = (C + 3) // 4 * 4
        # fea_conv = BSConvU(4*in, C): parameters as in the reference, packed through the folds in _extra_pack
        self._add_conv('fea_conv.pw', num_in_ch * 4, C, 1, linear=True, custom=True)
        self._add_dw('fea_conv.dw', C)
        for k in range(1, num_block + 1):
            b = f'B{k}.'
            for j in (1, 2, 3):
                self._add_conv(b + f'c{j}_d', C, dc, 1, linear=True)
                self._add_conv(b + f'c{j}_r.pw', C, C, 1, linear=True)
                self._add_dw(b + f'c{j}_r.dw', C)
            self._add_conv(b + 'c4.pw', C, dc, 1, linear=True)
            self._add_dw(b + 'c4.dw', dc)
            self._add_conv(b + 'c5', dc * 4, C, 1, linear=True)
            self._add_conv(b + 'esa.conv1', C, f, 1, linear=True)
            self._add_conv(b + 'esa.conv_f', f, f, 1, linear=True, dense=(FP, FP))
            for nm in ('conv_max', 'conv3', 'conv3_'):
                self._add_conv(b + f'esa.{nm}.pw', f, f, 1, linear=True)
                self._add_dw(b + f'esa.{nm}.dw', f)
            self._add_conv(b + 'esa.conv2', f, f, 3, dense=(FP, FP), stride=2, padding=0)
            self._add_conv(b + 'esa.conv4', f, C, 1, linear=True, dense=(FP, cp4))
            self._leaf(b.rstrip('.')).register_parameter('cw', torch.nn.Parameter(torch.normal(mean=1, std=0.2, size=(1, C))))
            self._add_conv(b + 'conv_out', C, C, 1, linear=True, custom=True)
        self._add_conv('c1', C * num_block, C, 1, linear=True)
        self._add_conv('c2.pw', C, C, 1, linear=True)
        self._add_dw('c2.dw', C)
        self._add_conv('upsampler.upsampleOneStep.0', C, num_out_ch * upscale * upscale, 3)

    @staticmethod
    def _merged_bsconv(pw, dw, wp=None):
        """BSConvU (team18_bsrn.py:82-88) as a dense 3x3: y[c] = sum_tap dw[c,tap] * pad0(pw[c,:] . x + bp[c]) + bd[c]
        -> weights W[c,k,tap] = dw[c,tap] * pw[c,k], bias bd[c] + bp[c] * sum_tap dw[c,tap] for interior pixels, and for a
        pixel whose taps with dx = -1 (mask bit 0), dx = +1 (bit 1), dy = -1 (bit 2), dy = +1 (bit 3) fall outside the image
        the table row [mask] = -bp[c] * sum over those taps of dw[c,tap]."""
        if wp is None:
            wp = pw.weight.detach().reshape(pw.weight.shape[0], -1)
        wp = wp.detach().double().cpu()                                               # [C, cin]   (host arithmetic, fp64)
        wd = dw.weight.detach().double().cpu().reshape(-1, 3, 3)                      # [C, ky, kx]
        bp = pw.bias.detach().double().cpu() if pw.bias is not None else torch.zeros(wp.shape[0], dtype=torch.float64)
        bd = dw.bias.detach().double().cpu() if dw.bias is not None else torch.zeros(wp.shape[0], dtype=torch.float64)
        w = wd[:, None, :, :] * wp[:, :, None, None]                                  # [C, cin, 3, 3]
        bias = bd + bp * wd.sum(dim=(1, 2))
        c = wp.shape[0]
        cp = (c + 15) // 16 * 16
        table = torch.zeros(16, cp, dtype=torch.float64)
        for m in range(1, 16):
            out = torch.zeros(3, 3, dtype=torch.bool)
            if m & 1: out[:, 0] = True
            if m & 2: out[:, 2] = True
            if m & 4: out[0, :] = True
            if m & 8: out[2, :] = True
            table[m, :c] = -bp * (wd * out.double()).sum(dim=(1, 2))
        return w.float(), bias.float(), table.float().contiguous()

    def _merged_paths(self):
        """BSConvU modules that run as dense 3x3 convolutions in the 16-bit modes (the full-resolution ones)"""
        return [f'B{k}.c{j}_r' for k in range(1, self.nb + 1) for j in (1, 2, 3)] + [f'B{k}.c4' for k in range(1, self.nb + 1)] + ['c2']

    def _cin_map(self, path, cin_map, store):
        if path.endswith('.c5') and store != "f32":       # the four distilled tensors are DP = round_up(dc, 16) channels wide (engine.Planar)
            dc = self.dc
            DP = (dc + 15) // 16 * 16
            return [(s // DP) * dc + s % DP if s % DP < dc else -1 for s in range(4 * DP)]
        return cin_map

    def _extra_pack(self, packed, device):
        C, ic = self.C, self.in_nc
        if self._store() != "f32":
            for path in self._merged_paths():
                w, bias, table = self._merged_bsconv(self._leaf(path + '.pw'), self._leaf(path + '.dw'))
                packed[path + '#bs3#s16'] = pack_conv_s16(w, bias, self._store()).to(device)
                packed[path + '#bs3#border'] = table.to(device)
            if 32 < C <= 48 and self.dc <= 32:               # c5 over cat(d1, d2, d3, r4) for the fused block tail
                from .engine import pack_tail_s16
                for k in range(1, self.nb + 1):
                    leaf = self._leaf(f'B{k}.c5')
                    packed[f'B{k}.c5#tail'] = pack_tail_s16(leaf.weight, leaf.bias, 3, self.dc, self.dc, self._store()).to(device)
        pw = self._leaf('fea_conv.pw')
        w = pw.weight.detach().float().reshape(C, 4, ic).sum(dim=1)           # [C,4*ic] -> [C,ic]: input replicated x4
        w3 = torch.zeros(C, ic, 3, 3)
        w3[:, :, 1, 1] = w                                                     # 1x1 as the centre tap of the NCHW-input 3x3 path
        packed['fea_conv.pw'] = pack_conv(w3, pw.bias).to(device)
        if self._store() != "f32":                                           # the 16-bit head (engine.Plan.conv lowers it to pack + conv_s16)
            packed['fea_conv.pw#head#s16'] = pack_head_s16(w3, pw.bias, self._store()).to(device)
            # ... and fea_conv as ONE dense 3x3 of the packed input (pointwise x depthwise merged like the blocks' BSConvUs): the
            # depthwise pass of the head was a launch of its own at 2.2x its algorithmic bytes
            wm, bm, table = self._merged_bsconv(pw, self._leaf('fea_conv.dw'), wp=w)
            packed['fea_conv#bs3#head#s16'] = pack_head_s16(wm, bm, self._store()).to(device)
            packed['fea_conv#bs3#border'] = table.to(device)
        for k in range(1, self.nb + 1):
            co = self._leaf(f'B{k}.conv_out')
            cw = self._leaf(f'B{k}').cw.detach().float().reshape(1, C)
            packed[f'B{k}.conv_out'] = pack_conv(co.weight.detach().float() * cw, co.bias).to(device)
            if self._store() != "f32":
                packed[f'B{k}.conv_out#s16'] = pack_conv_s16(co.weight.detach().float() * cw, co.bias, self._store()).to(device)
                if L.lib().esr_esa_apply_post_supported(C, C, self.dc):      # the chain of the ESA apply launch (engine.Plan.esa_apply)
                    nd = self._leaf(f'B{k + 1}.c1_d') if k < self.nb else None
                    packed[f'B{k}.conv_out#apost'] = pack_apply_post(co.weight.detach().float() * cw, co.bias, None if nd is None else nd.weight,
                                                                     None if nd is None else nd.bias, self._store()).to(device)

    def _build_plan(self, plan, c):
        if c != self.in_nc:
            raise L.EsrError(f'BSRN expects {self.in_nc} input channels, got {c}')
        if plan.h < 15 or plan.w < 15:
            raise L.EsrError('ESA needs H, W >= 15 (3x3/s2 then 7x7/s3 pooling)')
        C, dc, f, nb = self.C, self.dc, self.f, self.nb
        h2, w2, h3, w3 = _lowres(plan.h, plan.w)
        g = dict(act=L.ACT_GELU)
        merged = plan.esize == 2
        # the post-chain kernel variants that exist: 3 main tiles (C in 33..48) with a 2-tile post (dc in 17..32)
        fuse_d = merged and (C + 15) // 16 == 3 and (dc + 15) // 16 == 2

        def bs3(path, src, dst, cin, cout, **kw):
            plan.conv(path + '#bs3', src, dst, cin, cout, k=3, border=path + '#bs3#border', bs_of=path, **kw)
        # bf16: `fea` and `out_lr` -- the long skip, team18_bsrn.py:231-234 -- are hi + lo pairs (Plan.pair: two dense tensors)
        hl = merged and self._skip_hilo(plan, C)
        fea2 = plan.pair('fea', plan.cpad(C)) if hl else None
        out_lr2 = plan.pair('out_lr', plan.cpad(C)) if hl else None
        fea = fea2.seg(0) if hl else plan.buffer('fea', C)
        # block outputs, team18_bsrn.py:226.  16-bit storage: nb dense tensors (engine.Planar) -- as C-channel slices of one [.., nb C]
        # buffer every block wrote 96 of 384 bytes per pixel (partial lines) and the next block's first convolutions read them back
        # at 1.7x their algorithmic bytes (profiles/pmc_traffic.json, r03b)
        bplanar = merged and C % 16 == 0
        bcat = plan.planar('bcat', nb, C) if bplanar else plan.buffer('bcat', nb * C)
        # d1 d2 d3 r4, team18_bsrn.py:166.  16-bit storage: four dense tensors of DP = round_up(dc, 16) channels (engine.Planar):
        # a dc-channel slice of a [.., 4 dc] buffer is a partial-line store (48 of 192 bytes per pixel), 2.3x the cost of a dense one
        DP = (dc + 15) // 16 * 16
        cat = plan.planar('cat', 4, DP) if merged else plan.buffer('cat', 4 * dc)
        cs = (lambda j: cat.seg(j)) if merged else (lambda j: cat[j * dc:(j + 1) * dc])
        t = None if merged else plan.buffer('t', C)       # pointwise result feeding the head's depthwise
        r1, r2, v, u = plan.buffer('r1', C), plan.buffer('r2', C), plan.buffer('v', C), plan.buffer('u', C)
        c1 = plan.buffer('esa_c1', FP)
        lo2 = plan.buffer('esa_s2', FP, h2, w2)
        la, lb, lt = (plan.buffer(n, FP, h3, w3) for n in ('esa_a', 'esa_b', 'esa_t'))
        lo = (h3, w3)
        # a block's FIRST distillation Linear + GELU (c1_d reads the block input) rides in the epilogue of the launch that produces
        # that input: the head for block 1, the previous block's conv_out for the others (one launch and one read of the tensor less)
        apply_out = fuse_d and bplanar and bool(L.lib().esr_esa_apply_post_supported(C, C, dc))
        def first_d(k):
            return dict(w=f'B{k}.c1_d', dst=cs(0), cout=dc, act=L.ACT_GELU) if fuse_d else None
        if hl:
            plan.conv('fea_conv#bs3', INPUT, fea2, self.in_nc, C, k=3, border='fea_conv#bs3#border', bs_of='fea_conv', post=first_d(1), hilo=L.HILO_OUT)
        elif merged:
            plan.conv('fea_conv#bs3', INPUT, fea, self.in_nc, C, k=3, border='fea_conv#bs3#border', bs_of='fea_conv', post=first_d(1))
        else:
            plan.conv('fea_conv.pw', INPUT, t, self.in_nc, C, counted=False)
            plan.dwconv('fea_conv.dw', t, fea, C)
        # ESDB's tail -- c4 (BSConvU + GELU) -> cat(d1, d2, d3, r4) -> c5 -> esa.conv1, team18_bsrn.py:165-171, :109 -- as ONE launch
        # (rfdb_tail_kernel<.., 3, true>, ABI v12): r4 never reaches memory; from 256 tiles of 16 x 16
        fused_tail = (self.fuse_tail and merged and 32 < C <= 48 and DP == 32 and dc <= 32 and f <= 16 and FP == 16
                      and plan.n * ((plan.w + 15) // 16) * ((plan.h + 15) // 16) >= 256)
        cur = fea
        for k in range(1, nb + 1):
            b = f'B{k}.'
            src = cur
            for j, (rin, rout) in enumerate(((cur, r1), (r1, r2), (r2, r1)), start=1):
                if merged:
                    # 16-bit storage: BSConvU as one dense 3x3 on the matrix cores (+ input from the staged tile, GELU).  The
                    # NEXT distillation Linear + GELU (c{j+1}_d reads this launch's result r_j) rides in its epilogue on the
                    # fp32 tile; the block's first one (c1_d, of the block input) rides with the producer of the block input (first_d).
                    if j == 1 and not fuse_d:
                        plan.conv(b + 'c1_d', rin, cs(0), C, dc, k=1, counted=False, **g)
                    nxt = dict(w=b + f'c{j + 1}_d', dst=cs(j), cout=dc, act=L.ACT_GELU) if (j < 3 and fuse_d) else None
                    bs3(b + f'c{j}_r', rin, rout, C, C, res=rin, res_mode=L.RES_PRE_ACT, post=nxt, **g)
                    if j < 3 and not fuse_d:
                        plan.conv(b + f'c{j + 1}_d', rout, cs(j), C, dc, k=1, counted=False, **g)
                else:
                    # c{j}_d (Linear + GELU) and c{j}_r = BSConvU (+ input, GELU) read the same tensor: one launch, the
                    # pointwise result stays in LDS (team18_bsrn.py:150-163)
                    plan.bsconv(b + f'c{j}_r.pw', b + f'c{j}_r.dw', rin, rout, C, C, res=rin, res_mode=L.RES_PRE_ACT,
                                distill=dict(w=b + f'c{j}_d', dst=cs(j - 1), cout=dc, act=L.ACT_GELU), **g)
            if fused_tail:
                bs3(b + 'c4', r1, v, C, dc,
                    tail=dict(w=b + 'c5#tail', cat=Planar(cat.segs[:3]), cat_c=3 * DP, cat_c_alg=3 * dc, cout=C, mid_act=L.ACT_GELU),
                    post=dict(w=b + 'esa.conv1', dst=c1, cout=f, act=L.ACT_NONE))
            elif merged:
                bs3(b + 'c4', r1, cs(3), C, dc, **g)
            else:
                plan.bsconv(b + 'c4.pw', b + 'c4.dw', r1, cs(3), C, dc, **g)
            if fused_tail:
                pass
            elif plan.esize == 2 and (C + 15) // 16 in (3, 4) and f <= 16:
                # 16-bit storage: esa.conv1 rides in c5's epilogue on the fp32 tile (one launch less per block)
                plan.conv(b + 'c5', cat, v, 4 * DP if merged else 4 * dc, C, k=1, counted=False, cin_alg=4 * dc,
                          post=dict(w=b + 'esa.conv1', dst=c1, cout=f, act=L.ACT_NONE))
            else:
                plan.conv(b + 'c5', cat, v, 4 * DP if merged else 4 * dc, C, k=1, counted=False, cin_alg=4 * dc)
                plan.conv(b + 'esa.conv1', v, c1, C, f, k=1, counted=False)
            mark = len(plan.ops)
            plan.conv3x3s2(b + 'esa.conv2', c1, lo2, f)
            plan.maxpool7s3(lo2, la)
            plan.conv(b + 'esa.conv_max.pw', la, lt, f, f, k=1, hw=lo, counted=False)
            plan.dwconv(b + 'esa.conv_max.dw', lt, lb, f, hw=lo, **g)
            plan.conv(b + 'esa.conv3.pw', lb, lt, f, f, k=1, hw=lo, counted=False)
            plan.dwconv(b + 'esa.conv3.dw', lt, la, f, hw=lo, **g)
            plan.conv(b + 'esa.conv3_.pw', la, lt, f, f, k=1, hw=lo, counted=False)
            plan.dwconv(b + 'esa.conv3_.dw', lt, lb, f, hw=lo)
            if self.fuse_esa_lowres:
                # the eight launches above as one op of two (halo recompute; only the pooled map reaches memory)
                ga = g.get("act", L.ACT_NONE)
                plan.esa_lowres(mark, c1, la, lb, f, b + 'esa.conv2',
                                [dict(kind=1, act=ga, w=b + 'esa.conv_max.pw', w_dw=b + 'esa.conv_max.dw'),
                                 dict(kind=1, act=ga, w=b + 'esa.conv3.pw', w_dw=b + 'esa.conv3.dw'),
                                 dict(kind=1, act=L.ACT_NONE, w=b + 'esa.conv3_.pw', w_dw=b + 'esa.conv3_.dw')])
            out = bcat.seg(k - 1) if bplanar else bcat[(k - 1) * C:k * C]
            if apply_out:
                # conv_out (. cw, + block input) and the next block's c1_d in the ESA apply launch: the attention output never reaches memory
                chain = [dict(w=b + 'conv_out', dst=out, cout=C, act=L.ACT_NONE, res=src)]
                if k < nb:
                    chain.append(dict(w=f'B{k + 1}.c1_d', dst=cs(0), cout=dc, act=L.ACT_GELU))
                plan.esa_apply(b + 'esa.conv_f', b + 'esa.conv4', v, c1, lb, out, C, f, post=chain, skip_y=True)
            else:
                plan.esa_apply(b + 'esa.conv_f', b + 'esa.conv4', v, c1, lb, u, C, f)
                plan.conv(b + 'conv_out', u, out, C, C, k=1, res=src, res_mode=L.RES_PRE_ACT, counted=False,
                          post=first_d(k + 1) if (merged and k < nb) else None)
            cur = out
        plan.conv('c1', bcat, v, nb * C, C, k=1, counted=False, **g)
        if hl:
            bs3('c2', v, out_lr2, C, C, res=fea2, res_mode=L.RES_PRE_ACT, hilo=L.HILO_RES | L.HILO_OUT)
            plan.conv('upsampler.upsampleOneStep.0', out_lr2, OUTPUT, C, self.out_nc * 16, hilo=L.HILO_IN)
            return
        if merged:
            bs3('c2', v, u, C, C, res=fea, res_mode=L.RES_PRE_ACT)
        else:
            plan.bsconv('c2.pw', 'c2.dw', v, u, C, C, res=fea, res_mode=L.RES_PRE_ACT)
        plan.conv('upsampler.upsampleOneStep.0', u, OUTPUT, C, self.out_nc * 16)

    # -- complexity counters: what utils/model_summary.py reports for this graph -----------------------------
    def _complexity_terms(self, plan, o):
        """Conv2d hooks fire for the depthwise convs, esa.conv2 and the upsampler conv (43 calls); nn.Linear is
        counted by linear_flops_counter_hook (model_summary.py:305-312), which for a 4-D NHWC input adds
        input.shape[0]*input.shape[1]*output.shape[1] = 1*H*H -- the reference's own quirk, reproduced so the
        table matches (true MACs are 9.43 G); Linear outputs are not 'activations'."""
        hw = o.get("hw")
        h, w = (hw if hw else (plan.h, plan.w))
        if o["kind"] == "dw":
            return 9 * o["cin"] * plan.n * h * w, o["cout"] * plan.n * h * w, 1
        if o["kind"] == "bs":                                             # depthwise Conv2d + 1 or 2 Linear calls in one launch
            nlin = 1 + (o["distill"] is not None)
            return 9 * o["cout"] * plan.n * h * w + nlin * plan.n * h * h, o["cout"] * plan.n * h * w, 1
        if o["kind"] == "conv" and o.get("bs_of") is not None:           # a BSConvU run as a dense 3x3: one Linear + one depthwise Conv2d
            nlin = 1 + (o.get("post") is not None) + (o.get("tail") is not None)     # (+ the Linear(s) riding in its launch)
            return 9 * o["cout"] * plan.n * h * w + nlin * plan.n * h * h, o["cout"] * plan.n * h * w, 1
        if o["kind"] == "conv" and not o.get("counted", True):
            return (1 + (o.get("post") is not None)) * plan.n * h * h, 0, 0      # a Linear call (+ the one in its epilogue)
        if o["kind"] == "apply":
            return (2 + len(o.get("post") or ())) * plan.n * plan.h * plan.h, 0, 0       # conv_f and conv4 are Linear here (+ those riding in the launch)
        return super()._complexity_terms(plan, o)
