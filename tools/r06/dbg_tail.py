"""debug rfdb_tail_kernel: which term of v is wrong"""
import ctypes, sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from ntire2022_esr_amd import _lib as L
from ntire2022_esr_amd.engine import pack_conv_s16, unpack_conv_s16, pack_tail_s16, pack_post_s16
DEV = "cuda:0"
compute = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dt = torch.bfloat16 if compute == "bf16" else torch.float16
n, hw, nf, dc, f = (int(sys.argv[2]), (int(sys.argv[3]), int(sys.argv[4])), int(sys.argv[5]), int(sys.argv[6]), 16) if len(sys.argv) > 2 else (1, (256, 256), 50, 25, 16)
g = torch.Generator().manual_seed(1)
r3 = F.pad(torch.randn(n, *hw, nf, generator=g), (0, 64 - nf)).to(dt).to(DEV)
ds = F.pad(torch.randn(3, n, *hw, dc, generator=g), (0, 32 - dc)).to(dt).to(DEV)
w4, b4 = torch.randn(dc, nf, 3, 3, generator=g) * 0.1, torch.randn(dc, generator=g)
wc, bc = torch.randn(f, nf, generator=g) * 0.2, torch.randn(f, generator=g)
def run(w5, b5):
    blob4 = pack_conv_s16(w4, b4, compute, cin_phys=64)
    w4e, _ = unpack_conv_s16(blob4, nf, dc, 3, compute, cin_phys=64)
    blob5 = pack_tail_s16(w5, b5, 3, dc, dc, compute).to(DEV); blobc = pack_post_s16(wc, bc, compute).to(DEV); blob4 = blob4.to(DEV)
    v = torch.full((n, *hw, 64), 7.0, dtype=dt, device=DEV); c1 = torch.full((n, *hw, 16), 7.0, dtype=dt, device=DEV)
    d = L.ConvDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.ksize = n, hw[0], hw[1], nf, dc, 3
    d.in_layout = d.out_layout = L.NHWC; d.storage = d.compute = L.STORE[compute]; d.act, d.slope = L.ACT_NONE, 0.05
    d.inp = L.View(ctypes.c_void_p(r3.data_ptr()), 64, 0); d.out0 = L.View(ctypes.c_void_p(v.data_ptr()), 64, 0)
    d.wpacked = ctypes.c_void_p(blob4.data_ptr()); d.tail_wpacked = ctypes.c_void_p(blob5.data_ptr())
    d.tail_cat = L.View(ctypes.c_void_p(ds.data_ptr()), 32, 0); d.tail_cat_c, d.tail_cout, d.tail_mid_act = 96, nf, L.ACT_LRELU
    d.tail_seg_stride16 = ds[0].numel() * 2 // 16
    d.post_wpacked = ctypes.c_void_p(blobc.data_ptr()); d.post_out = L.View(ctypes.c_void_p(c1.data_ptr()), 16, 0); d.post_cout, d.post_act = f, L.ACT_NONE
    assert L.lib().esr_conv_tail_supported(ctypes.byref(d)) == 1
    L.check(L.lib().esr_conv2d_f32(ctypes.byref(d), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "tail"); torch.cuda.synchronize()
    x = r3[..., :nf].permute(0, 3, 1, 2).double()
    r4 = F.leaky_relu(F.conv2d(x, w4e.double().to(DEV), b4.double().to(DEV), padding=1), 0.05).to(dt).double()
    cat = torch.cat([ds[j, ..., :dc].permute(0, 3, 1, 2).double() for j in range(3)] + [r4], 1)
    vref = torch.einsum("oc,nchw->nohw", w5.double().to(DEV), cat) + b5.double().to(DEV)[None, :, None, None]
    got = v.permute(0, 3, 1, 2)[:, :nf].double()
    e = (got - vref).abs()
    return float(e.max()), float(e.mean()), got, vref, c1
z = torch.zeros(nf, 4 * dc)
print("bias only:", run(z, torch.arange(nf).float() * 0.1)[:2])
for s in range(4):
    w = z.clone()
    for o in range(dc): w[o, s * dc + o] = 1.0
    em, ea, got, vref, _ = run(w, torch.zeros(nf))
    print(f"segment {s} -> v[0..{dc}): max {em:.4f} mean {ea:.5f}")
    if em > 0.05:
        bad = ((got - vref).abs() > 0.05)
        idx = bad.nonzero()[:6].tolist()
        print("   first bad (n, ch, y, x):", idx, [ (round(float(got[tuple(i)]), 3), round(float(vref[tuple(i)]), 3)) for i in idx])
        print("   bad fraction per channel:", [round(float(bad[0, c].float().mean()), 2) for c in range(0, nf, 5)])
        print("   bad fraction per row mod 4:", [round(float(bad[0, :, r::4].float().mean()), 2) for r in range(4)], " per col mod 16:", [round(float(bad[0, :, :, c::16].float().mean()), 2) for c in range(0, 16, 3)])
w = torch.randn(nf, 4 * dc, generator=g) * 0.15
print("random w5:", run(w, torch.randn(nf, generator=g))[:2])
