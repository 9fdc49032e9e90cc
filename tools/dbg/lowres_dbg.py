import ctypes, os, sys, torch
import torch.nn.functional as F
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
from ntire2022_esr_amd import _lib as L
from ntire2022_esr_amd.engine import pack_dense
os.system("/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/dpp_probe.so " + os.path.join(R, "tools/dbg/dpp_probe.hip")); so = ctypes.CDLL("/tmp/dpp_probe.so")
o = torch.zeros(128, dtype=torch.int32, device="cuda")
so.dpp_probe(ctypes.c_void_p(o.data_ptr())); print("row_shl:1", o[:20].tolist()); print("row_shr:3", o[64:84].tolist())
lib = L.lib(); DEV = "cuda:0"
for storage in ("bf16", "f16"):
    f, n, h, w = 16, 1, 31, 30
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, f, h, w, generator=g)
    st = {"f32": (0, torch.float32), "bf16": (1, torch.bfloat16), "f16": (2, torch.float16)}[storage]
    xq = x.to(st[1]).float()
    w2, b2 = torch.randn(f, f, 3, 3, generator=g) * 0.2, torch.randn(f, generator=g) * 0.1
    wl, bl = torch.zeros(f, f, 3, 3), torch.zeros(f)
    pref = F.max_pool2d(F.conv2d(xq.double(), w2.double(), b2.double(), stride=2), 7, 3)
    h3, w3 = pref.shape[2:]
    xd = torch.zeros(n, h, w, 16); xd[..., :f] = xq.permute(0, 2, 3, 1); xd = xd.to(DEV).to(st[1]).contiguous()
    blobs = [pack_dense(w2, b2, 16, 16).to(DEV), pack_dense(wl, bl, 16, 16).to(DEV)]
    pooled = torch.full((n, h3, w3, 16), float("nan"), device=DEV); y = torch.full((n, h3, w3, 16), float("nan"), device=DEV)
    d = L.EsaLowresDesc(); d.n, d.h, d.w, d.f, d.storage, d.n_layers = n, h, w, f, st[0], 1
    d.x = L.View(ctypes.c_void_p(xd.data_ptr()), 16, 0); d.w_s2, d.pooled, d.y = blobs[0].data_ptr(), pooled.data_ptr(), y.data_ptr()
    d.layer[0].kind, d.layer[0].act, d.layer[0].w = 0, 0, blobs[1].data_ptr()
    L.check(lib.esr_esa_lowres_f32(ctypes.byref(d), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "lowres")
    torch.cuda.synchronize()
    got = pooled.cpu()[..., :f].permute(0, 3, 1, 2).double()
    print(storage, "max err", float((got - pref).abs().max()))
    print("got[0,:4]", got[0, :4].flatten().tolist()); print("ref[0,:4]", pref[0, :4].flatten().tolist())
    # conv2 alone, for comparing single values
    c2 = F.conv2d(xq.double(), w2.double(), b2.double(), stride=2)
    print("conv2[0,0,:3,:8]", c2[0, 0, :3, :8].tolist())
