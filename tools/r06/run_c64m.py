"""run conv64m_kernel<.., POST> (RFDB c1_r shape) a few hundred times: a target for rocprofv3 PC sampling   usage: run_c64m.py [reps] [post 0|1]"""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from ntire2022_esr_amd import ops, _lib as L
from ntire2022_esr_amd.engine import pack_conv_s16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
post = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
g = torch.Generator().manual_seed(0)
c, pc = 50, 25
x = F.pad(torch.randn(32, 256, 256, c, generator=g), (0, 64 - c)).to(torch.bfloat16).cuda()
w, b = torch.randn(c, c, 3, 3, generator=g) * 0.1, torch.randn(c, generator=g)
wp, bp = torch.randn(pc, c, generator=g) * 0.2, torch.randn(pc, generator=g)
blob = pack_conv_s16(w, b, "bf16", cin_phys=64).cuda()
kw = dict(act=1, cin=c, packed=blob, res=x, res_mode=L.RES_PRE_ACT)
if post:
    kw.update(post_weight=wp, post_bias=bp, post_act=1)
for _ in range(reps):
    ops.conv2d(x, w, b, **kw)
torch.cuda.synchronize()
print("done", reps)
