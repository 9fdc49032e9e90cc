"""IMDN (x4) on the HIP engine -- drop-in for `models.imdn_baseline.IMDN`.

Same constructor keywords (models/imdn_baseline.py:33) and the same 86 state_dict keys
(`model.0.*`, `model.1.sub.{i}.conv{1,2,3}.0.*`, `.conv4.*`, `.conv1x1.*`, `model.1.sub.{nb}.*`,
`model.2.*`) so `load_state_dict(torch.load('model_zoo/imdn_baseline.pth'), strict=True)`
(test_demo.py:22-23; id 26 = nb 7, test_demo.py:203-209) works unchanged.

Forward = 3 + 4*nb kernel launches (3 + 5*nb when nc != 64), no split/cat/add kernels:
  head     3->nc        NCHW in, NHWC out (FEA, kept for the global shortcut)
  block i  conv1 nc->nc  LReLU, split store: [0,d) -> CAT[0:d),   [d,nc) -> R1      basicblock.py:260
           conv2 r->nc   LReLU, split store:        CAT[d:2d),            R2        :261
           conv3 r->nc   LReLU, split store:        CAT[2d:3d),           R1        :262
           conv4 r->d    no act                     (accumulators only)             :263
           1x1  4d->nc   + block input (residual)   -> ping-pong X   same launch    :264-265
  tail     nc->nc 3x3 + FEA (ShortcutBlock, basicblock.py:197-199)
  up       nc->out_nc*16 3x3 fused with PixelShuffle(4) -> NCHW output             basicblock.py:446-449
"""
from . import _lib as L
from .engine import INPUT, OUTPUT, HipSRModel


class IMDN(HipSRModel):
    def __init__(self, in_nc=3, out_nc=3, nc=64, nb=8, upscale=4, act_mode='L',
                 upsample_mode='pixelshuffle', negative_slope=0.05):
        super().__init__()
        if 'R' not in act_mode and 'L' not in act_mode:
            raise AssertionError('Examples of activation function: R, L, BR, BL, IR, IL')
        if upsample_mode != 'pixelshuffle':
            raise NotImplementedError('upsample mode [{:s}] is not found'.format(upsample_mode))
        if upscale != 4 or nc not in (32, 64) or in_nc > 4 or (out_nc * 16) > 64:
            # r_nc = 3/4 nc must be a whole number of 8-channel K chunks (esr_conv2d_f32: in.coff + round_up(cin, 8) <= pitch)
            raise NotImplementedError('HIP IMDN supports upscale=4, nc in {32, 64}, in_nc<=4, out_nc<=4')
        self.in_nc, self.out_nc, self.nc, self.nb, self.upscale = in_nc, out_nc, nc, nb, upscale
        self.act = L.ACT_LRELU if 'L' in act_mode else L.ACT_RELU
        self.slope = negative_slope
        self.d_nc = int(nc * 0.25)
        self.r_nc = nc - self.d_nc
        self._add_conv('model.0', in_nc, nc, 3)
        for i in range(nb):
            p = f'model.1.sub.{i}.'
            self._add_conv(p + 'conv1.0', nc, nc, 3)
            self._add_conv(p + 'conv2.0', self.r_nc, nc, 3)
            self._add_conv(p + 'conv3.0', self.r_nc, nc, 3)
            self._add_conv(p + 'conv4', self.r_nc, self.d_nc, 3)
            self._add_conv(p + 'conv1x1', self.d_nc * 4, nc, 1)
        self._add_conv(f'model.1.sub.{nb}', nc, nc, 3)
        self._add_conv('model.2', nc, out_nc * upscale * upscale, 3)

    def _build_plan(self, plan, c):
        if c != self.in_nc:
            raise L.EsrError(f'IMDN expects {self.in_nc} input channels, got {c}')
        nc, d, r = self.nc, self.d_nc, self.r_nc
        # bf16: `fea` and the LR conv's output -- the long skip of the ShortcutBlock, models/imdn_baseline.py:61, basicblock.py:191-205 -- are hi + lo pairs (Plan.pair)
        hl = self._skip_hilo(plan, nc)
        fea2 = plan.pair('fea', plan.cpad(nc)) if hl else None
        out_lr2 = plan.pair('out_lr', plan.cpad(nc)) if hl else None
        fea = fea2.seg(0) if hl else plan.buffer('fea', nc)
        fused = d == 16 and 48 < nc <= 64 and self.compute == 'f32'    # conv4 + 1x1 in one launch (16-bit modes keep conv4 on the 16-bit kernel)
        # fused: conv4's slot never reaches memory.  16-bit storage (d a multiple of 16): four dense tensors (engine.Planar) instead
        # of 32-byte slices of a 128-byte pixel -- partial-line stores cost 2.3x a dense one
        planar = plan.esize == 2 and d % 16 == 0
        cat = plan.planar('cat', 4, d) if planar else plan.buffer('cat', plan.cpad(3 * d if fused else 4 * d))
        cs = (lambda j: cat.seg(j)) if planar else (lambda j: cat[j * d:(j + 1) * d])
        # pitches are whole K chunks of the plan's storage type (8 fp32 / 16 16-bit channels): r = 24 (nc = 32) needs 32 slots in bf16 / fp16
        # fused tail: its 3x3 input (conv3's remaining channels) is stored channel-blocked [n][r/8][h][w][8] -- imdb_tail_kernel
        # stages one 8-channel K chunk at a time, and in NHWC every 128-byte line of a 192-byte pixel would be fetched by four
        # stages microseconds apart (esr_conv_desc.blocked8).  Round 4: the same for r1 / r2, the inputs of conv2 / conv3 on
        # wino8_f32_kernel (its ablations put 15 % of a launch into the read traffic of 32-byte pieces of 192-byte pixels; the
        # direct kernel cannot read a blocked input, so only with the Winograd path on)
        blk = fused and r == 48 and nc == 64
        blk12 = blk and self.winograd
        # ... and for the block input / output x (xa / xb: written and read as a residual by the fused tail, read by conv1 and the LR conv
        # on wino_f32_kernel): a standalone 64 -> 64 layer reads a blocked input 7-9 % faster (tools/wino/blk_probe.py).  `fea` stays NHWC:
        # the NCHW head writes it, block 0 and the LR conv's residual read it
        xa, xb = plan.buffer('xa', nc, blocked=blk12), plan.buffer('xb', nc, blocked=blk12)
        lr = plan.buffer('lr', nc) if blk12 else None
        r1, r2 = plan.buffer('r1', plan.cpad(r), blocked=blk12), plan.buffer('r2', plan.cpad(r), blocked=blk12)
        r3 = plan.buffer('r3', r, blocked=True) if blk else r1
        act = dict(act=self.act, slope=self.slope)
        plan.conv('model.0', INPUT, fea2 if hl else fea, self.in_nc, nc, hilo=L.HILO_OUT if hl else 0)
        cur, nxt = fea, xa
        for i in range(self.nb):
            p = f'model.1.sub.{i}.'
            plan.conv(p + 'conv1.0', cur, cs(0), nc, nc, split=d, dst1=r1, **act)
            plan.conv(p + 'conv2.0', r1, cs(1), r, nc, split=d, dst1=r2, **act)
            plan.conv(p + 'conv3.0', r2, cs(2), r, nc, split=d, dst1=r3, **act)
            if fused:
                # conv4 -> cat -> conv1x1 -> + x in one kernel: the 16 conv4 channels go from the 3x3's accumulators
                # straight into the 1x1's K loop and never reach memory
                plan.conv(p + 'conv4', r3, nxt, r, d, res=cur, res_mode=L.RES_PRE_ACT,
                          tail=dict(w=p + 'conv1x1', cat=cat[0:3 * d], cat_c=3 * d, cout=nc))
            else:
                plan.conv(p + 'conv4', r1, cs(3), r, d)
                plan.conv(p + 'conv1x1', cat, nxt, 4 * d, nc, k=1, res=cur, res_mode=L.RES_PRE_ACT)
            cur = nxt
            nxt = xb if cur is xa else xa
        if lr is not None:
            nxt = lr                                   # the LR conv's epilogue (residual add) stores NHWC
        if hl:
            plan.conv(f'model.1.sub.{self.nb}', cur, out_lr2, nc, nc, res=fea2, res_mode=L.RES_PRE_ACT, hilo=L.HILO_RES | L.HILO_OUT)
            plan.conv('model.2', out_lr2, OUTPUT, nc, self.out_nc * 16, hilo=L.HILO_IN)
        else:
            plan.conv(f'model.1.sub.{self.nb}', cur, nxt, nc, nc, res=fea, res_mode=L.RES_PRE_ACT)
            plan.conv('model.2', nxt, OUTPUT, nc, self.out_nc * 16)
