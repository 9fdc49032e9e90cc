"""One-launch timings of the IMDN 3x3 shapes at batch 32: direct conv_f32_kernel vs wino_f32_kernel (same descriptor, wino_wpacked
set or not).  Prints ms, direct-equivalent TFLOP/s and the EXECUTED MFMA TFLOP/s (Winograd: 16/36 of the direct flops)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ntire2022_esr_amd import _lib as L
from ntire2022_esr_amd.engine import pack_conv, pack_wino
lib = L.lib(); dev = "cuda:0"
B = int(os.environ.get("BATCH", "32")); H = int(os.environ.get("HH", "256")); W = int(os.environ.get("WW", "256"))
for (cin, cout, split) in ((64, 64, 16), (48, 64, 16), (64, 64, 0)):
    x = torch.randn(B, H, W, cin, device=dev)
    y0 = torch.zeros(B, H, W, 64 if not split else 48, device=dev)
    y1 = torch.zeros(B, H, W, 48, device=dev)
    w = torch.randn(cout, cin, 3, 3) * 0.05; bias = torch.randn(cout)
    blob = pack_conv(w, bias).to(dev); wb = pack_wino(w, bias).to(dev)
    d = L.ConvDesc(); d.n, d.h, d.w, d.cin, d.cout, d.ksize = B, H, W, cin, cout, 3
    d.act, d.slope = 1, 0.05
    d.inp = L.View(x.data_ptr(), cin, 0)
    if split:
        d.split = split; d.out0 = L.View(y0.data_ptr(), 48, 16); d.out1 = L.View(y1.data_ptr(), 48, 0)
    else:
        d.out0 = L.View(y0.data_ptr(), 64, 0)
    d.wpacked = blob.data_ptr()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name, wp in (("direct", None), ("wino", wb.data_ptr())):
        d.wino_wpacked = wp
        for _ in range(int(os.environ.get('WARM', '30'))): assert lib.esr_conv2d_f32(ctypes.byref(d), st) == 0, lib.esr_last_hip_error()
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(20): lib.esr_conv2d_f32(ctypes.byref(d), st)
        e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 20
        fl = 2.0 * B * H * W * cin * cout * 9
        ex = fl * (16 / 36 if wp else 1.0)
        print(f"{cin:3d}->{cout:3d} split{split:2d} {name:8s}: {ms:.4f} ms  direct-equivalent {fl / ms / 1e9:6.1f} TFLOP/s  executed {ex / ms / 1e9:6.1f} TFLOP/s = {ex / ms / 1e9 / 157.3:.3f} of the fp32 MFMA peak", flush=True)
