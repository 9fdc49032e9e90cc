#!/usr/bin/env python3
"""Ablation builds + timing of conv64r_kernel (research tooling, never part of libesr_hip.so).

  python tools/abl/c64_abl.py build [names]   (authoring container: hipcc cross-compiles the variants into tools/abl/libesr_c64_*.so)
  python tools/abl/c64_abl.py run             (GPU box: one 3x3 launch per variant and shape, round-robin twice)

Variants are TEXT substitutions inside conv64r_kernel on a copy of csrc/esr_s16.hip (the product source carries no switches);
results of everything but `prod` are wrong on purpose."""
import ctypes, os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
SRC = os.path.join(REPO, "ntire2022_esr_amd", "csrc")
BEGIN, END = "// ---- conv64r_kernel: conv48r_kernel's plan", "// ---- conv48rp_kernel: RLFB's c3_r"

NOSTORE = ("            e_v[j] = (inx && ch < p.cout_store) ? base + (unsigned)ch * 2u : OOB;", "            e_v[j] = OOB;")
NODMA = ("            const bool ok = valid && part < (unsigned)GSL && sl < (unsigned)NSLOT && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;",
         "            const bool ok = false && valid && part < (unsigned)GSL && sl < (unsigned)NSLOT && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;")
NOB = ("                if (L + AHEAD < (RW / 2) * NG) read_b(L + AHEAD);", "                ;")
NOA = ("                if (L + 1 < (RW / 2) * NG) read_a(L + 1);", "                ;")
NOEPI = ("                            if (g >= 1 && g <= 2 * NT) epi_pack_step(par ^ 1, g - 1, m);\n                            if (g >= 2 * NT + 1 && g < 2 * NT + 1 + SPP) epi_store_step(g - 2 * NT - 1, r_prev, m);", "                            (void)m; (void)r_prev;")
NOBAR = ("""        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((RW / 2 - 1) * SPP) : "memory");
        __builtin_amdgcn_s_barrier();
        if (!more) break;""", "        if (!more) break;")
NODMAISSUE = ("                if (rp == 0 && g < PPW) dma_piece(g, more, nn, nx0, ny0, (k + 1) & 1);", "                ;")
SUBS = {
    "prod": [], "nostore": [NOSTORE], "nodma": [NODMA], "nomem": [NOSTORE, NODMA], "nomfma": ["MFMA"], "nob": [NOB], "nolds": [NOB, NOA], "noepi": [NOEPI],
    "nobar": [NOBAR], "mfmaonly": [NOSTORE, NODMAISSUE, NOB, NOA, NOEPI, NOBAR], "memonly": ["MFMA", NOB, NOA],
}


def variant(base, subs):
    i, j = base.index(BEGIN), base.index(END)
    body = base[i:j]
    for sub in subs:
        if sub == "MFMA":
            body, n = re.subn(r'asm\("v_mfma_f32_16x16x32_(bf16|f16) %0, %1, %2, %[03]"', 'asm("; no mfma %0 %1 %2"', body)
            assert n == 6, n
            continue
        a, b = sub
        assert body.count(a) == 1, (a, body.count(a))
        body = body.replace(a, b)
    return base[:i] + body + base[j:]


def build(only=None):
    base = open(os.path.join(SRC, "esr_s16.hip")).read()
    others = [os.path.join(SRC, f) for f in sorted(os.listdir(SRC)) if f.endswith(".hip") and f != "esr_s16.hip"]
    cc = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(REPO, "include"), "-I", SRC]
    objs = []
    for f in others:                                   # compiled once
        o = os.path.join(HERE, "obj_" + os.path.basename(f).replace(".hip", "") + ".o")
        subprocess.check_call(cc + ["-c", f, "-o", o], stderr=subprocess.DEVNULL)
        objs.append(o)
    for name, subs in SUBS.items():
        if only and name not in only:
            continue
        src = os.path.join(HERE, f"c64_{name}.hip")
        open(src, "w").write(variant(base, subs))
        vo = os.path.join(HERE, f"obj_c64_{name}.o")
        subprocess.check_call(cc + ["-c", src, "-o", vo], stderr=subprocess.DEVNULL)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", vo] + objs + ["-o", os.path.join(HERE, f"libesr_c64_{name}.so")])
        os.remove(src)
        os.remove(vo)
        print("built", name, flush=True)
    for o in objs:
        os.remove(o)


def run():
    import torch
    from ntire2022_esr_amd import _lib as L
    from ntire2022_esr_amd.engine import pack_conv_s16
    dev = "cuda:0"
    libs = {}
    for name in SUBS:
        so = os.path.join(HERE, f"libesr_c64_{name}.so")
        if os.path.exists(so):
            libs[name] = ctypes.CDLL(so)
            libs[name].esr_conv2d_f32.argtypes = [ctypes.POINTER(L.ConvDesc), ctypes.c_void_p]
    for (n, hw, cin, cout, res) in ((32, (256, 256), 64, 64, 1), (32, (256, 256), 64, 32, 0), (1, (339, 510), 64, 64, 1)):
        x = torch.randn(n, *hw, cin, device=dev).to(torch.bfloat16)
        y = torch.zeros(n, *hw, cout, device=dev, dtype=torch.bfloat16)
        w = torch.randn(cout, cin, 3, 3) * 0.1
        b = torch.randn(cout)
        blob = pack_conv_s16(w, b, "bf16").to(dev)
        d = L.ConvDesc()
        d.n, d.h, d.w, d.cin, d.cout, d.ksize = n, hw[0], hw[1], cin, cout, 3
        d.act, d.slope, d.storage, d.compute = 1, 0.05, 1, 1
        d.inp = L.View(x.data_ptr(), cin, 0)
        d.out0 = L.View(y.data_ptr(), cout, 0)
        if res:
            d.res_mode, d.res = 1, L.View(x.data_ptr(), cin, 0)
        d.wpacked = blob.data_ptr()
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        reps = 20 if n > 1 else 100
        for rnd in range(2):
            for name, lib in libs.items():
                for _ in range(3):
                    assert lib.esr_conv2d_f32(ctypes.byref(d), st) == 0
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    lib.esr_conv2d_f32(ctypes.byref(d), st)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / reps
                gb = n * hw[0] * hw[1] * (cin + cout) * 2 / 1e9
                print(f"{n:2d}x{hw[0]}x{hw[1]} {cin}->{cout} res{res} {name:9s} {ms * 1000:8.1f} us  {gb / ms:5.0f} GB/s(alg)  {2 * n * hw[0] * hw[1] * cin * cout * 9 / ms / 1e9:5.0f} TFLOP/s", flush=True)


def run1():
    """the product library: a few launches of the 64 -> 64 (+ x) and the 48 -> 48 layer at batch 32 (for rocprofv3 --pmc)"""
    import torch
    from ntire2022_esr_amd import _lib as L, ops
    from ntire2022_esr_amd.engine import pack_conv_s16
    for c in (64, 48):
        x = torch.randn(32, 256, 256, c, device="cuda:0").to(torch.bfloat16)
        w, b = torch.randn(c, c, 3, 3) * 0.1, torch.randn(c)
        blob = pack_conv_s16(w, b, "bf16").to("cuda:0")
        for _ in range(4):
            ops.conv2d(x, w, b, act=1, packed=blob, **(dict(res=x, res_mode=1) if c == 64 else {}))
        torch.cuda.synchronize()


if __name__ == "__main__":
    {"build": lambda: build(sys.argv[2:]), "run": run, "run1": run1}[sys.argv[1]]()
