"""RFDN (x4) on the HIP engine -- drop-in for `models.rfdn_baseline.RFDN.RFDN` (RFDN.py:11-41; ids 0, 6, 22).

Same constructor keywords and the same 128 state_dict keys.  nf=50 lives in NHWC buffers of pitch 56 (six
zero pad channels); the 25-channel distilled maps are stored as 32-wide (one 128-byte line) slices of one 128-wide concat buffer
and the four block outputs as 56-wide slices of one 224-wide buffer, so neither torch.cat
(rfdn_baseline/block.py:163, RFDN.py:36) exists as a kernel: the following 1x1 convs are packed with a
`cin_map` that skips the pad slots.  Per RFDB (block.py:148-166): {1x1 distil + LeakyReLU, 3x3 + input
residual + LeakyReLU} x3, 3x3 50->25 + LeakyReLU, 1x1 c5 over the concat, ESA.
"""
from . import _lib as L
from .engine import INPUT, OUTPUT, HipSRModel, Planar
from .rlfn import FP, _lowres, _pad8


def _slice_map(n_slices, logical, padded):
    """physical slot -> logical channel for `n_slices` slices of `logical` channels padded to `padded`."""
    return [(s // padded) * logical + s % padded if s % padded < logical else -1 for s in range(n_slices * padded)]


class RFDN(HipSRModel):
    def __init__(self, in_nc=3, nf=50, num_modules=4, out_nc=3, upscale=4, block_residual=True, esa_f=None,
                 esa_conv_f=True):
        """The last three keywords are not in the reference's constructor; they select its two RFDN derivatives
        (SURVEY 8f N2): `models.team40_rfdn_pruned.RFDN` = RFDN(nf=40, block_residual=False, esa_f=12)
        (team40_rfdn_pruned.py:148-166 has no `+ input` in the refinement convs, :106 fixes the ESA width at 50 // 4) and
        `models.team08_sfdn.RFDN` = RFDN(block_residual=False, esa_conv_f=False) (team08_sfdn.py:137-147; its ESA adds
        conv1's output itself where the baseline adds conv_f of it, :118-129; the checkpoint is re-parameterised)."""
        super().__init__()
        if upscale != 4 or nf > 64 or in_nc > 4 or out_nc * 16 > 64 or num_modules != 4:
            raise NotImplementedError('HIP RFDN supports upscale=4, nf <= 64, 4 modules, in_nc <= 4, out_nc <= 4')
        self.in_nc, self.out_nc, self.nf, self.num_modules, self.upscale = in_nc, out_nc, nf, num_modules, upscale
        self.dc = nf // 2
        self.f = nf // 4 if esa_f is None else esa_f
        self.block_residual, self.esa_conv_f = block_residual, esa_conv_f
        self.scale_idx = 0
        nf, dc, f = self.nf, self.dc, self.f
        # distilled slices are padded to whole 128-byte lines: a 1x1 writing a 112-byte slice of every 448 bytes costs
        # 17 % more time than one writing 128 of every 512 (partial-line writes), the wider c5 read costs 5 %
        self.DP = (dc + 31) // 32 * 32
        cp4 = (nf + 3) // 4 * 4
        self._add_conv('fea_conv', in_nc, nf, 3)
        for k in range(1, 5):
            b = f'B{k}.'
            for j in (1, 2, 3):
                self._add_conv(b + f'c{j}_d', nf, dc, 1)
                self._add_conv(b + f'c{j}_r', nf, nf, 3)
            self._add_conv(b + 'c4', nf, dc, 3)
            self._add_conv(b + 'c5', dc * 4, nf, 1, cin_map=_slice_map(4, dc, self.DP))
            self._add_conv(b + 'esa.conv1', nf, f, 1)
            if esa_conv_f:
                self._add_conv(b + 'esa.conv_f', f, f, 1, dense=(FP, FP))
            self._add_conv(b + 'esa.conv_max', f, f, 3)
            self._add_conv(b + 'esa.conv2', f, f, 3, dense=(FP, FP), stride=2, padding=0)
            self._add_conv(b + 'esa.conv3', f, f, 3)
            self._add_conv(b + 'esa.conv3_', f, f, 3)
            self._add_conv(b + 'esa.conv4', f, nf, 1, dense=(FP, cp4))
        self._add_conv('c.0', nf * num_modules, nf, 1, cin_map=_slice_map(num_modules, nf, _pad8(nf)))
        self._add_conv('LR_conv', nf, nf, 3)
        self._add_conv('upsampler.0', nf, out_nc * upscale * upscale, 3)

    def set_scale(self, scale_idx):
        self.scale_idx = scale_idx

    def _build_plan(self, plan, c):
        if c != self.in_nc:
            raise L.EsrError(f'RFDN expects {self.in_nc} input channels, got {c}')
        if plan.h < 15 or plan.w < 15:
            raise L.EsrError('ESA needs H, W >= 15 (3x3/s2 then 7x7/s3 pooling)')
        nf, dc, f, DP = self.nf, self.dc, self.f, self.DP
        KP = plan.cpad(nf)                                # 56 fp32 channels / 64 16-bit channels: whole K chunks
        # 16-bit storage, round 6: the nf-wide tensors at a TIGHT pitch -- round_up(nf, 8) = 56 channels (112-byte pixels) instead of 64; a consumer's
        # last K chunk runs 8 channels into the next pixel against zero weight rows (esr_conv2d_s16).  12.5 % fewer bytes per launch of an
        # HBM-bound model
        P = _pad8(nf) if (plan.esize == 2 and self.tight_pitch) else KP
        h2, w2, h3, w3 = _lowres(plan.h, plan.w)
        # bf16: `fea` and `out_lr` -- the long skip, RFDN.py:44-47 -- are hi + lo pairs (Plan.pair: two dense tensors)
        hl = self._skip_hilo(plan, nf)
        fea2 = plan.pair('fea', P) if hl else None
        fea = fea2.seg(0) if hl else plan.buffer('fea', P)
        out_lr2 = plan.pair('out_lr', P) if hl else None
        # the four block outputs, RFDN.py:36: four dense tensors in the 16-bit modes (engine.Planar; same slot order as the padded
        # [.., 4 P] buffer, so c.0's weight blob does not change)
        bplanar = plan.esize == 2
        bcat = plan.planar('bcat', 4, P) if bplanar else plan.buffer('bcat', 4 * P)
        # d1 d2 d3 r4, block.py:163.  16-bit storage: four dense tensors (engine.Planar) -- a 32-channel slice of a 128-wide
        # buffer is a partial-line store there (64 of 256 bytes per pixel), 2.3x the cost of a dense one
        planar = plan.esize == 2
        cat = plan.planar('cat', 4, DP) if planar else plan.buffer('cat', _pad8(4 * DP))
        cs = (lambda j: cat.seg(j)) if planar else (lambda j: cat[j * DP:(j + 1) * DP])
        r1, r2 = plan.buffer('r1', P), plan.buffer('r2', P)
        v = plan.buffer('v', P)
        c1 = plan.buffer('esa_c1', FP)
        lo2 = plan.buffer('esa_s2', FP, h2, w2)
        la, lb = plan.buffer('esa_a', FP, h3, w3), plan.buffer('esa_b', FP, h3, w3)
        act = dict(act=L.ACT_LRELU, slope=0.05)
        lo = dict(hw=(h3, w3))
        res = (lambda v: dict(res=v, res_mode=L.RES_PRE_ACT)) if self.block_residual else (lambda v: {})
        fused_post = (48 < nf <= 64 and 16 < dc <= 32) if plan.esize == 4 else ((nf + 15) // 16 in (3, 4) and 16 < dc <= 32)
        # 16-bit modes: block 1's first distillation conv (c1_d of fea) rides in the head convolution's epilogue
        # the fused block tail needs rfdb_tail_kernel's shapes (esr_conv_tail_supported has the last word in Plan.finalize)
        fused_tail = (self.fuse_tail and plan.esize == 2 and fused_post and planar and 48 < nf <= 64 and DP == 32 and f <= 16 and FP == 16
                      and plan.n * ((plan.w + 15) // 16) * ((plan.h + 15) // 16) >= 256)
        head_d = plan.esize == 2 and fused_post
        plan.conv('fea_conv', INPUT, fea2 if hl else fea, self.in_nc, nf, post=dict(w='B1.c1_d', dst=cs(0), cout=dc, act=L.ACT_LRELU) if head_d else None,
                  hilo=L.HILO_OUT if hl else 0)
        # ... and the other blocks' in the ESA apply launch that produces their input (esr_esa_desc.post[])
        apply_d = plan.esize == 2 and fused_post and bool(L.lib().esr_esa_apply_post_supported(nf, dc, 0))
        cur = fea
        for k in range(1, 5):
            b = f'B{k}.'
            if not ((head_d and k == 1) or (apply_d and k > 1)):
                plan.conv(b + 'c1_d', cur, cs(0), nf, dc, k=1, **act)
            if fused_post:
                # the distillation conv of r_j rides in the epilogue of the conv that produces r_j (block.py:150-160)
                plan.conv(b + 'c1_r', cur, r1, nf, nf, **res(cur), **act,
                          post=dict(w=b + 'c2_d', dst=cs(1), cout=dc, act=L.ACT_LRELU))
                plan.conv(b + 'c2_r', r1, r2, nf, nf, **res(r1), **act,
                          post=dict(w=b + 'c3_d', dst=cs(2), cout=dc, act=L.ACT_LRELU))
            else:
                plan.conv(b + 'c1_r', cur, r1, nf, nf, **res(cur), **act)
                plan.conv(b + 'c2_d', r1, cs(1), nf, dc, k=1, **act)
                plan.conv(b + 'c2_r', r1, r2, nf, nf, **res(r1), **act)
                plan.conv(b + 'c3_d', r2, cs(2), nf, dc, k=1, **act)
            plan.conv(b + 'c3_r', r2, r1, nf, nf, **res(r2), **act)
            if fused_tail:
                # round 6 (ABI v12): c4 -> cat(d1, d2, d3, r4) -> c5 -> esa.conv1 in ONE launch (rfdb_tail_kernel; block.py:161-164, :117):
                # r4 stays in registers, d1 .. d3 are read once, v and esa.conv1's map are the only stores
                plan.conv(b + 'c4', r1, v, nf, dc, tail=dict(w=b + 'c5#tail', cat=Planar(cat.segs[:3]), cat_c=3 * DP, cat_c_alg=3 * dc, cout=nf,
                                                             mid_act=L.ACT_LRELU),
                          post=dict(w=b + 'esa.conv1', dst=c1, cout=f, act=L.ACT_NONE))
            else:
                plan.conv(b + 'c4', r1, cs(3), nf, dc, **act)
            if fused_tail:
                pass
            elif plan.esize == 2 and (nf + 15) // 16 in (3, 4) and f <= 16:
                # 16-bit storage: esa.conv1 rides in c5's epilogue on the fp32 tile (one launch less per block)
                plan.conv(b + 'c5', cat, v, 4 * DP, nf, k=1, cin_alg=4 * dc, post=dict(w=b + 'esa.conv1', dst=c1, cout=f, act=L.ACT_NONE))
            else:
                plan.conv(b + 'c5', cat, v, 4 * DP, nf, k=1, cin_alg=4 * dc)
                plan.conv(b + 'esa.conv1', v, c1, nf, f, k=1)
            mark = len(plan.ops)
            plan.conv3x3s2(b + 'esa.conv2', c1, lo2, f)
            plan.maxpool7s3(lo2, la)
            plan.conv(b + 'esa.conv_max', la, lb, f, f, act=L.ACT_RELU, **lo)
            plan.conv(b + 'esa.conv3', lb, la, f, f, act=L.ACT_RELU, **lo)
            plan.conv(b + 'esa.conv3_', la, lb, f, f, **lo)
            if self.fuse_esa_lowres:
                # the five launches above as one op of two (halo recompute; only the pooled map reaches memory)
                plan.esa_lowres(mark, c1, la, lb, f, b + 'esa.conv2',
                                [dict(kind=0, act=L.ACT_RELU, w=b + 'esa.conv_max'), dict(kind=0, act=L.ACT_RELU, w=b + 'esa.conv3'),
                                 dict(kind=0, act=L.ACT_NONE, w=b + 'esa.conv3_')])
            out = bcat.seg(k - 1) if bplanar else bcat[(k - 1) * P:k * P]
            nxt_d = [dict(w=f'B{k + 1}.c1_d', dst=cs(0), cout=dc, act=L.ACT_LRELU, slope=0.05)] if (apply_d and k < 4) else None
            plan.esa_apply(b + 'esa.conv_f', b + 'esa.conv4', v, c1, lb, out, nf, f, post=nxt_d)
            cur = out
        plan.conv('c.0', bcat, v, 4 * KP, nf, k=1, cin_alg=4 * nf, **act)
        if hl:
            plan.conv('LR_conv', v, out_lr2, nf, nf, res=fea2, res_mode=L.RES_PRE_ACT, hilo=L.HILO_RES | L.HILO_OUT)
            plan.conv('upsampler.0', out_lr2, OUTPUT, nf, self.out_nc * 16, hilo=L.HILO_IN)
        else:
            plan.conv('LR_conv', v, r1, nf, nf, res=fea, res_mode=L.RES_PRE_ACT)
            plan.conv('upsampler.0', r1, OUTPUT, nf, self.out_nc * 16)

    def _cin_map(self, path, cin_map, store):
        if path == 'c.0':                                 # the block-output slices are as wide as the storage type's K chunks
            return _slice_map(self.num_modules, self.nf, _pad8(self.nf) if store == "f32" else (self.nf + 15) // 16 * 16)
        return cin_map

    def _extra_pack(self, packed, device):
        if self._store() != "f32" and 48 < self.nf <= 64 and self.dc <= 32:
            from .engine import pack_tail_s16    # c5 over cat(d1, d2, d3, r4) for rfdb_tail_kernel
            for k in range(1, 5):
                leaf = self._leaf(f'B{k}.c5')
                packed[f'B{k}.c5#tail'] = pack_tail_s16(leaf.weight, leaf.bias, 3, self.dc, self.dc, self._store()).to(device)
        if self._store() != "f32":               # c1_d of blocks 2..4 as the post of the previous block's ESA apply launch
            from .engine import pack_apply_post
            if L.lib().esr_esa_apply_post_supported(self.nf, self.dc, 0):
                for k in range(2, 5):
                    leaf = self._leaf(f'B{k}.c1_d')
                    packed[f'B{k}.c1_d#apost'] = pack_apply_post(leaf.weight, leaf.bias, None, None, self._store()).to(device)
        if not self.esa_conv_f:                  # SFDN: c3 + c1_  ==  c3 + conv_f(c1_) with conv_f = identity
            import torch
            from .engine import pack_dense
            for k in range(1, 5):
                packed[f'B{k}.esa.conv_f'] = pack_dense(torch.eye(self.f)[:, :, None, None], torch.zeros(self.f), FP, FP).to(device)

    def _counted_convs(self, plan, o):
        """logical channel counts for the padded-concat 1x1 convs (the reference sees 100 / 200 inputs)."""
        r = super()._counted_convs(plan, o)
        if o["kind"] == "apply" and not self.esa_conv_f:
            return r[1:]                         # no conv_f call in the reference graph
        if o["kind"] == "conv" and o.get("tail") is not None and o["w"].endswith('.c4'):      # the fused block tail: c4, c5, esa.conv1
            return [(o["cin"], o["cout"], 3, plan.npix, L.ACT_LRELU), (self.dc * 4, o["tail"]["cout"], 1, plan.npix, L.ACT_NONE),
                    (o["tail"]["cout"], o["post"]["cout"], 1, plan.npix, L.ACT_NONE)]
        if o["kind"] == "conv" and o["w"].endswith('.c5'):
            return [(self.dc * 4, o["cout"], 1, plan.npix, o["act"])]
        if o["kind"] == "conv" and o["w"] == 'c.0':
            return [(self.nf * self.num_modules, o["cout"], 1, plan.npix, o["act"])]
        return r
