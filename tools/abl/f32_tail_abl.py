#!/usr/bin/env python3
"""Ablation builds + timing of the fused IMDB tail (research tooling only): imdb_tail_kernel (csrc/imdb_tail.inc) and, as `old`,
conv_f32_kernel<1,3,false,4,TAIL=4>, which ran this shape until the end of round 2.

  python tools/abl/f32_tail_abl.py build [names]   (authoring container)
  python tools/abl/f32_tail_abl.py run             (GPU box: one tail launch per variant, B = 32, 256x256, IMDBlock shapes)

Variants are TEXT substitutions on a copy of csrc/esr_hip.hip + imdb_tail.inc (results of every variant but `prod` / `old` are wrong):
  prod     unchanged
  old      the generic TAIL variant (dispatch to imdb_tail_kernel disabled)
  nocat    no loads of the concat slices          nores   no loads of the residual
  no1x1    no MFMAs of the 1x1                    noconv  no MFMAs of the 3x3
  noepi    no epilogue (no stores)
  nodma    the DMAs of a stage are not issued
"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
SRC = os.path.join(REPO, "ntire2022_esr_amd", "csrc")

SUBS = {
    "prod": [],
    "old": [("    if (imdb_tail_shape(k)) {", "    if (false) {")],
    "nocat": [("                            bc[C][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, C * 64, 0));",
               "                            bc[C][r] = f32x4{(float)vo, 0.f, 0.f, 0.f};")],
    "nores": [("                            acc2[tt][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, tt * 64, 0));",
               "                            acc2[tt][r] = f32x4{(float)vo, 0.f, 0.f, 0.f};")],
    "no1x1": [("                            acc2[tt][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[tt][j], bf[r][j], acc2[tt][r], 0, 0, 0);\n                __builtin_amdgcn_sched_barrier(0);\n            };\n            __builtin_amdgcn_s_setprio(0);",
               "                            acc2[tt][r].x += a2[tt][j] * bf[r][j];\n                __builtin_amdgcn_sched_barrier(0);\n            };\n            __builtin_amdgcn_s_setprio(0);")],
    "noconv": [("                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cs][j], b[cs][r][j], acc[r], 0, 0, 0);",
                "                        acc[r].x += a[cs][j] * b[cs][r][j];")],
    "noepi": [("            epilogue_nhwc<TNT>(p, acc2, scr, cur.n, cur.x0, cur.y0, wv, lane, TILE);",
               "            if (acc2[0][0].x == 1.2345e-30f) epilogue_nhwc<TNT>(p, acc2, scr, cur.n, cur.x0, cur.y0, wv, lane, TILE);")],
    "nodma": [("            if (c + 2 < IT_NCH) issue((c + 2) % IT_R, cur, c + 2);\n            else issue((c + 2) % IT_R, nxt, c + 2 - IT_NCH);", "            ;")],
}


def build(only=None):
    base = open(os.path.join(SRC, "esr_hip.hip")).read()
    inc = '#include "imdb_tail.inc"'
    assert inc in base
    base = base.replace(inc, open(os.path.join(SRC, "imdb_tail.inc")).read())
    others = [os.path.join(SRC, f) for f in ("esr_s16.hip", "esr_esa.hip", "esr_bsconv.hip", "esr_ca.hip")]
    objs = []
    for f in others:
        o = os.path.join(HERE, "obj_" + os.path.basename(f).replace(".hip", "") + ".o")
        if not os.path.exists(o):
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I",
                                   os.path.join(REPO, "include"), "-I", SRC, f, "-o", o])
        objs.append(o)
    for name, subs in SUBS.items():
        if only and name not in only:
            continue
        s = base
        for a, b in subs:
            assert a in s, (name, a)
            s = s.replace(a, b)
        src = os.path.join(HERE, f"f32_{name}.hip")
        open(src, "w").write(s)
        vo = os.path.join(HERE, f"obj_f32_{name}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I",
                               os.path.join(REPO, "include"), "-I", SRC, src, "-o", vo])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", vo] + objs +
                              ["-o", os.path.join(HERE, f"libesr_t_{name}.so")])
        os.remove(src)
        os.remove(vo)


def run():
    import torch
    from ntire2022_esr_amd import _lib as L
    from ntire2022_esr_amd.engine import pack_conv
    dev = "cuda:0"
    n, h, w = 32, 256, 256
    x = torch.randn(n, h, w, 48, device=dev)                 # r3: the remaining 48 channels
    cat = torch.randn(n, h, w, 48, device=dev)               # d1 | d2 | d3
    res = torch.randn(n, h, w, 64, device=dev)               # block input
    y = torch.zeros(n, h, w, 64, device=dev)
    w3, b3 = torch.randn(16, 48, 3, 3) * 0.1, torch.randn(16)
    w1, b1 = torch.randn(64, 64, 1, 1) * 0.1, torch.randn(64)
    blob3, blob1 = pack_conv(w3, b3).to(dev), pack_conv(w1, b1).to(dev)
    import glob
    names = [n_ for n_ in SUBS if os.path.exists(os.path.join(HERE, f"libesr_t_{n_}.so"))]
    names += sorted(os.path.basename(f)[9:-3] for f in glob.glob(os.path.join(HERE, "libesr_t_*.so")) if os.path.basename(f)[9:-3] not in SUBS)
    for name in names:
        lib = ctypes.CDLL(os.path.join(HERE, f"libesr_t_{name}.so"))
        lib.esr_conv2d_f32.argtypes = [ctypes.POINTER(L.ConvDesc), ctypes.c_void_p]
        d = L.ConvDesc()
        d.n, d.h, d.w, d.cin, d.cout, d.ksize = n, h, w, 48, 16, 3
        d.act, d.slope, d.res_mode = 0, 0.05, 1
        d.inp = L.View(x.data_ptr(), 48, 0)
        d.out0 = L.View(y.data_ptr(), 64, 0)
        d.res = L.View(res.data_ptr(), 64, 0)
        d.wpacked = blob3.data_ptr()
        d.tail_wpacked = blob1.data_ptr()
        d.tail_cat = L.View(cat.data_ptr(), 48, 0)
        d.tail_cat_c, d.tail_cout, d.tail_mid_act = 48, 64, 0
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(3):
            rc = lib.esr_conv2d_f32(ctypes.byref(d), st)
            assert rc == 0, rc
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.esr_conv2d_f32(ctypes.byref(d), st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        gb = n * h * w * (48 + 48 + 64 + 64) * 4 / 1e9
        fl = 2.0 * n * h * w * (48 * 16 * 9 + 64 * 64)
        print(f"tail {name:10s} {ms:.4f} ms  {gb / ms:.0f} GB/s(alg)  {fl / ms / 1e9:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    build(sys.argv[2:]) if sys.argv[1] == "build" else run()
