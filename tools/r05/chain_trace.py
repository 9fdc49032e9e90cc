"""Where does a step of rlfb_chain_kernel go?  Builds csrc/esr_chain.hip with -DESR_CHAIN_TRACE as a library of its own (the kernel stamps
s_memtime before / after every step's barrier, per wave, for blocks 0..3), runs one launch and prints per wave: cycles of work per step
(after the previous barrier -> before this one) and cycles waited at the barrier.  Build: here (hipcc cross-compiles); run: on the GPU box.
usage: python tools/r05/chain_trace.py build | run [n h w]"""
import ctypes
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SO = os.path.join(REPO, "tools", "r05", "libchain_trace%s.so" % os.environ.get("CHAIN_VARIANT", ""))
STUB = r"""
#include <hip/hip_runtime.h>
#include <stdio.h>
void esr_set_err(const char* what, hipError_t e) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); }
int esr_check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { esr_set_err(what, e); return -3; } return 0; }
void esr_note_kernel(const char*, ...) {}
"""


def build(extra=()):
    csrc = os.path.join(REPO, "ntire2022_esr_amd", "csrc")
    stub = "/tmp/chain_trace_stub.hip"
    open(stub, "w").write(STUB)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DESR_CHAIN_TRACE", *extra,
                           "-I", os.path.join(REPO, "include"), "-I", csrc, os.path.join(csrc, "esr_chain.hip"), stub, "-o", SO])
    print("built", SO)


def run(n=1, h=339, w=510, dtype="bf16"):
    import numpy as np
    import torch
    sys.path.insert(0, REPO)
    from ntire2022_esr_amd import _lib as L
    from ntire2022_esr_amd.engine import pack_conv_s16, pack_post_s16
    lib = ctypes.CDLL(SO)
    lib.esr_conv_chain_s16.argtypes = [ctypes.POINTER(L.ChainDesc), ctypes.c_void_p]
    lib.esr_chain_set_trace.argtypes = [ctypes.c_void_p]
    dev = "cuda:0"
    dt = torch.bfloat16 if dtype == "bf16" else torch.float16
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, h, w, 48, generator=g).to(dt).to(dev)
    blobs = [pack_conv_s16(torch.randn(48, 48, 3, 3, generator=g) * 0.06, torch.randn(48, generator=g) * 0.1, dtype).to(dev) for _ in range(3)]
    p1 = pack_post_s16(torch.randn(48, 48, 1, 1, generator=g) * 0.1, torch.randn(48, generator=g) * 0.1, dtype).to(dev)
    p2 = pack_post_s16(torch.randn(16, 48, generator=g) * 0.1, torch.randn(16, generator=g) * 0.1, dtype).to(dev)
    v = torch.zeros(n, h, w, 48, dtype=dt, device=dev)
    c1 = torch.zeros(n, h, w, 16, dtype=dt, device=dev)
    d = L.ChainDesc()
    d.n, d.h, d.w, d.n_layers, d.cin, d.cmid, d.cout = n, h, w, 3, 48, 48, 48
    d.act, d.slope, d.res_mode, d.storage, d.compute = 1, 0.05, 2, L.STORE[dtype], L.STORE[dtype]
    d.inp = L.View(x.data_ptr(), 48, 0)
    for i in range(3):
        d.wpacked[i] = blobs[i].data_ptr()
    d.post_wpacked, d.post_out, d.post_cout = p1.data_ptr(), L.View(v.data_ptr(), 48, 0), 48
    d.post2_wpacked, d.post2_out, d.post2_cout = p2.data_ptr(), L.View(c1.data_ptr(), 16, 0), 16
    NS, TW = 40, 6
    NW = 4
    tr = torch.zeros(4 * NW * NS * TW, dtype=torch.int64, device=dev)
    lib.esr_chain_set_trace(ctypes.c_void_p(tr.data_ptr()))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert lib.esr_conv_chain_s16(ctypes.byref(d), st) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.esr_conv_chain_s16(ctypes.byref(d), st)
    e1.record()
    torch.cuda.synchronize()
    print(f"{n}x{h}x{w} {dtype}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch (traced build)")
    t = tr.cpu().numpy().reshape(4, NW, NS, TW)
    for blk in range(2):
        tb = t[blk]
        nst = int((tb[0, :, 0] != 0).sum())
        if nst < 3:
            continue
        nst = min(nst, 36)
        print(f"block {blk}: {nst} recorded steps; s_memtime ticks (100 MHz constant clock?) -- total {tb[0, nst - 1, 1] - tb[0, 0, 0]}")
        for wv in range(NW):
            work = tb[wv, 1:nst, 0] - tb[wv, 0:nst - 1, 1]       # after barrier k-1 -> before barrier k
            wait = tb[wv, 0:nst, 1] - tb[wv, 0:nst, 0]
            print(f"  wave {wv}: work/step median {np.median(work):.0f} mean {work.mean():.1f} max {work.max()}  | barrier wait median {np.median(wait):.0f} mean {wait.mean():.1f}", end="")
            if (wv & 3) != 3:
                b0 = tb[wv, 0:nst - 1, 1]                      # after the previous barrier
                seg = [np.median(tb[wv, 1:nst, 2] - b0), np.median(tb[wv, 1:nst, 3] - tb[wv, 1:nst, 2]), np.median(tb[wv, 1:nst, 4] - tb[wv, 1:nst, 3]),
                       np.median(tb[wv, 1:nst, 5] - tb[wv, 1:nst, 4]), np.median(tb[wv, 1:nst, 0] - tb[wv, 1:nst, 5])]
                print("  | top..m0 %d, groups 0-4 %d, 5-9 %d, 10-14 %d, tail %d" % tuple(seg))
            else:
                print()
        step = tb[0, 1:nst, 1] - tb[0, 0:nst - 1, 1]
        print(f"  step period (wave 0, barrier to barrier): median {np.median(step):.0f} mean {step.mean():.1f}; first 12: {step[:12].tolist()}")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        a = [int(v) for v in sys.argv[2:5]] if len(sys.argv) >= 5 else [1, 339, 510]
        run(*a)
        if len(sys.argv) < 5:
            run(32, 256, 256)
            os.environ["ESR_CHAIN_G"] = "3"
            run(32, 256, 256)
