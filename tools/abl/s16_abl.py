#!/usr/bin/env python3
"""Ablation builds + timing of conv_s16_kernel (research tooling, never part of libesr_hip.so).

  python tools/abl/s16_abl.py build     (authoring container: hipcc cross-compiles the variants into tools/abl/*.so)
  python tools/abl/s16_abl.py run       (GPU box: times one 3x3 conv launch per variant, B = 32, 256x256)

Variants are TEXT substitutions on a copy of csrc/esr_s16.hip (the product source carries no switches):
  prod      unchanged
  noepi     no epilogue (no stores)                          -> what the stores + transposes cost
  nomfma    no fragment reads / MFMAs                        -> the memory pipeline alone
  nowait    no vmcnt wait before the stage barrier (results wrong) -> compute + epilogue without memory stalls
  nostore   epilogue with its LDS transposes but (practically) no stores
  nolds     epilogue stores without the LDS transposes (stores garbage)
  nodma     no DMA issue at all (results wrong)              -> LDS + MFMA + epilogue only
"""
import ctypes, os, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
SRC = os.path.join(REPO, "ntire2022_esr_amd", "csrc")

MFMA_LINE = "acc[tt][r] = mfma32<BF16>(a[cs][tt], b[cs][r], acc[tt][r]);"
NEXT_LOADS = [("                        if (q + 1 < PAIRS && m < NFRAG) load_one(cs ^ 1, q + 1, m, false);", "                        ;", 2),
              ("                for (int i = NT * RW; i < NFRAG; ++i) load_one(cs ^ 1, q + 1, i, false);", "                ;")]
MF = [(MFMA_LINE, "acc[tt][r].x += 1.f;", 2), ("        for (int i = 0; i < NFRAG; ++i) load_one(0, 0, i, EPI);", "        ;")] + NEXT_LOADS
# the fragment reads of the stage's first pair only (both sets), every later MFMA reuses them: LDS read traffic / waits removed
NODS = [("        for (int i = 0; i < NFRAG; ++i) load_one(0, 0, i, EPI);", "        for (int i = 0; i < NFRAG; ++i) { load_one(0, 0, i, EPI); load_one(1, 0, i, EPI); }")] + NEXT_LOADS
WAIT = ("            wait_vm_dyn((R - 2) * n_my + epi_stores * __builtin_popcount(hist_st & hmask) + (GRES ? RES_LOADS : 0) * __builtin_popcount(hist_rs & hmask));", "            ;")
DMA = ("            else dma_buf16(dst, lvoff[i], lrsrc, lsoff);", "            else ;")
SUBS = {
    "prod": [],
    "nw16": [("constexpr int S16_NW = 8;", "constexpr int S16_NW = 16;")],      # 16 waves per tile, 2 rows each
    "nods": NODS,
    "nomfma": MF,                                   # the memory pipeline alone (DMA issue, waits, epilogue stores)
    "nowait": [WAIT],                               # no vmcnt wait before the stage barrier (results wrong)
    "nodma": [DMA, WAIT],                           # LDS reads + MFMA + epilogue only (results wrong)
    "nostore": [("        __builtin_amdgcn_raw_buffer_store_b128(o, __builtin_amdgcn_make_buffer_rsrc(e_y0, 0, e_y0n, 0x00020000), v0 + (unsigned)r * e_rowb0, 0, 0);",
                 "        if (o.x == 0x7fc12345) __builtin_amdgcn_raw_buffer_store_b128(o, __builtin_amdgcn_make_buffer_rsrc(e_y0, 0, e_y0n, 0x00020000), v0 + (unsigned)r * e_rowb0, 0, 0);")],
    "nobar": [("            if (!OWN_PIECES) __builtin_amdgcn_s_barrier();\n            slot = slot == R - 1 ? 0 : slot + 1;", "            slot = slot == R - 1 ? 0 : slot + 1;")],   # no stage barrier (results wrong)
}


def build(only=None):
    base = open(os.path.join(SRC, "esr_s16.hip")).read()
    others = [os.path.join(SRC, f) for f in sorted(os.listdir(SRC)) if f.endswith(".hip") and f != "esr_s16.hip"]
    objs = []
    for f in others:                                   # compiled once
        o = os.path.join(HERE, "obj_" + os.path.basename(f).replace(".hip", "") + ".o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I", os.path.join(REPO, "include"),
                               "-I", SRC, f, "-o", o])
        objs.append(o)
    for name, subs in SUBS.items():
        if only and name not in only:
            continue
        s = base
        for sub in subs:
            a, b = sub[0], sub[1]
            if a == "__END__":
                s += b
                continue
            assert a in s, (name, a)
            assert s.count(a) == (sub[2] if len(sub) > 2 else 1), (name, a, s.count(a))
            s = s.replace(a, b)
        src = os.path.join(HERE, f"s16_{name}.hip")
        open(src, "w").write(s)
        vo = os.path.join(HERE, f"obj_s16_{name}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I", os.path.join(REPO, "include"),
                               "-I", SRC, src, "-o", vo])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", vo] + objs +
                              ["-o", os.path.join(HERE, f"libesr_{name}.so")])
        os.remove(src)
        os.remove(vo)
    for o in objs:
        os.remove(o)


def run():
    import torch
    from ntire2022_esr_amd import _lib as L
    from ntire2022_esr_amd.engine import pack_conv_s16
    dev = "cuda:0"
    for (cin, cout, k, res) in ((48, 48, 3, 0), (64, 64, 3, 0), (64, 64, 3, 1), (48, 48, 3, 2), (48, 48, 1, 0), (128, 64, 1, 0), (32, 32, 3, 0)):
        x = torch.randn(32, 256, 256, cin, device=dev).to(torch.bfloat16)
        y = torch.zeros(32, 256, 256, cout, device=dev, dtype=torch.bfloat16)
        r = torch.randn(32, 256, 256, cout, device=dev).to(torch.bfloat16)
        w = torch.randn(cout, cin, k, k) * 0.1
        b = torch.randn(cout)
        blob = pack_conv_s16(w, b, "bf16").to(dev)
        for name in SUBS:
            if not os.path.exists(os.path.join(HERE, f"libesr_{name}.so")):
                continue
            lib = ctypes.CDLL(os.path.join(HERE, f"libesr_{name}.so"))
            lib.esr_conv2d_f32.argtypes = [ctypes.POINTER(L.ConvDesc), ctypes.c_void_p]
            d = L.ConvDesc()
            d.n, d.h, d.w, d.cin, d.cout, d.ksize = 32, 256, 256, cin, cout, k
            d.act, d.slope, d.storage, d.compute = 1, 0.05, 1, 1
            d.inp = L.View(x.data_ptr(), cin, 0)
            d.out0 = L.View(y.data_ptr(), cout, 0)
            if res:
                d.res_mode, d.res = (1, L.View(x.data_ptr(), cin, 0)) if res == 1 else (2, L.View(r.data_ptr(), cout, 0))
            d.wpacked = blob.data_ptr()
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            for _ in range(3):
                assert lib.esr_conv2d_f32(ctypes.byref(d), st) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                lib.esr_conv2d_f32(ctypes.byref(d), st)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            gb = 32 * 65536 * (cin + cout + (cout if res else 0)) * 2 / 1e9
            print(f"{cin:3d}->{cout:3d} k{k} res{res} {name:8s} {ms:.4f} ms  {gb / ms:.0f} GB/s(alg)  {2 * 32 * 65536 * cin * cout * k * k / ms / 1e9:.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    build(sys.argv[2:]) if sys.argv[1] == "build" else run()
