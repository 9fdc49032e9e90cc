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

SUBS = {
    "prod": [],
    "nw16": [],      # built with -DESR_S16_NW=16: 16 waves per tile, 2 rows each
    "noepi": [("                    epilogue(pn, px0, py0, dma_now);\n                    hist_st |= 1u;", "                    ;"),
              ("    if (pend) epilogue(pn, px0, py0, 0);", "    ;")],
    "nomfma": [("                    for (int r = 0; r < RW; ++r) acc[tt][r] = mfma32<BF16>(a[cs][tt], b[cs][r], acc[tt][r]);",
                "                    for (int r = 0; r < RW; ++r) acc[tt][r].x += 1.f;"),
               ("            if (NBUF == 2) load_frag(0, 0);", "            ;"),
               ("                    if (q + 1 < PAIRS) load_frag(cs ^ 1, q + 1);", "                    ;"),
               ("                    load_frag(0, q);", "                    ;")],
    # memory pipeline alone with perfectly coalesced DMA reads (same bytes, linear addresses; results wrong)
    "nomfma_lin": [("                    for (int r = 0; r < RW; ++r) acc[tt][r] = mfma32<BF16>(a[cs][tt], b[cs][r], acc[tt][r]);",
                "                    for (int r = 0; r < RW; ++r) acc[tt][r].x += 1.f;"),
               ("            if (NBUF == 2) load_frag(0, 0);", "            ;"),
               ("                    if (q + 1 < PAIRS) load_frag(cs ^ 1, q + 1);", "                    ;"),
               ("                    load_frag(0, q);", "                    ;"),
               ("        lrsrc = make_rsrc(q->x + (size_t)n * img_bytes, img_bytes);", "        lrsrc = make_rsrc(q->x, (size_t)q->N * img_bytes);"),
               ("            lvoff[r] = ok ? (unsigned)((gy * qW + gx) * qpitch + qcoff + 8 * plane) * 2u : OOB;",
                "            lvoff[r] = (unsigned)t * 49152u + (unsigned)pc * 1024u + (unsigned)lane * 16u + (ok ? 0u : 0u);"),
               ("        const unsigned soff = (unsigned)lc * 32u;", "        const unsigned soff = (unsigned)lc * 16384u;")],
    "nomfma_r64": [("                    for (int r = 0; r < RW; ++r) acc[tt][r] = mfma32<BF16>(a[cs][tt], b[cs][r], acc[tt][r]);",
                "                    for (int r = 0; r < RW; ++r) acc[tt][r].x += 1.f;"),
               ("            if (NBUF == 2) load_frag(0, 0);", "            ;"),
               ("                    if (q + 1 < PAIRS) load_frag(cs ^ 1, q + 1);", "                    ;"),
               ("                    load_frag(0, q);", "                    ;"),
               ("        lrsrc = make_rsrc(q->x + (size_t)n * img_bytes, img_bytes);", "        lrsrc = make_rsrc(q->x, (size_t)q->N * img_bytes);"),
               ("            lvoff[r] = ok ? (unsigned)((gy * qW + gx) * qpitch + qcoff + 8 * plane) * 2u : OOB;",
                "            lvoff[r] = (unsigned)t * 49152u + ((unsigned)pc * 16u + ((unsigned)lane >> 2)) * 96u + ((unsigned)lane & 3u) * 16u + (ok ? 0u : 0u);"),
               ("        const unsigned soff = (unsigned)lc * 32u;", "        const unsigned soff = (unsigned)lc * 16384u;")],
    "nomfma_r32": [("                    for (int r = 0; r < RW; ++r) acc[tt][r] = mfma32<BF16>(a[cs][tt], b[cs][r], acc[tt][r]);",
                "                    for (int r = 0; r < RW; ++r) acc[tt][r].x += 1.f;"),
               ("            if (NBUF == 2) load_frag(0, 0);", "            ;"),
               ("                    if (q + 1 < PAIRS) load_frag(cs ^ 1, q + 1);", "                    ;"),
               ("                    load_frag(0, q);", "                    ;"),
               ("        lrsrc = make_rsrc(q->x + (size_t)n * img_bytes, img_bytes);", "        lrsrc = make_rsrc(q->x, (size_t)q->N * img_bytes);"),
               ("            lvoff[r] = ok ? (unsigned)((gy * qW + gx) * qpitch + qcoff + 8 * plane) * 2u : OOB;",
                "            lvoff[r] = (unsigned)t * 49152u + ((unsigned)pc * 32u + ((unsigned)lane >> 1)) * 96u + ((unsigned)lane & 1u) * 16u + (ok ? 0u : 0u);"),
               ("        const unsigned soff = (unsigned)lc * 32u;", "        const unsigned soff = (unsigned)lc * 16384u;")],
    "nomfma_r16": [("                    for (int r = 0; r < RW; ++r) acc[tt][r] = mfma32<BF16>(a[cs][tt], b[cs][r], acc[tt][r]);",
                "                    for (int r = 0; r < RW; ++r) acc[tt][r].x += 1.f;"),
               ("            if (NBUF == 2) load_frag(0, 0);", "            ;"),
               ("                    if (q + 1 < PAIRS) load_frag(cs ^ 1, q + 1);", "                    ;"),
               ("                    load_frag(0, q);", "                    ;"),
               ("        lrsrc = make_rsrc(q->x + (size_t)n * img_bytes, img_bytes);", "        lrsrc = make_rsrc(q->x, (size_t)q->N * img_bytes);"),
               ("            lvoff[r] = ok ? (unsigned)((gy * qW + gx) * qpitch + qcoff + 8 * plane) * 2u : OOB;",
                "            lvoff[r] = (unsigned)t * 49152u + ((unsigned)pc * 64u + ((unsigned)lane)) * 96u + (ok ? 0u : 0u);"),
               ("        const unsigned soff = (unsigned)lc * 32u;", "        const unsigned soff = (unsigned)lc * 16384u;")],
    "nowait": [("                wait_vm_dyn(younger < 0 ? 0 : cnt);", "                ;")],
    "nostore": [("                __builtin_amdgcn_raw_buffer_store_b128(o, yr0, vo0, 0, 0);", "                if (o.x == 0x7fc12345) __builtin_amdgcn_raw_buffer_store_b128(o, yr0, vo0, 0, 0);")],
    "nolds": [("                const i32x4 o = *reinterpret_cast<const i32x4*>(scr + p8 * SCR_ROW + min(cb, NT * 16 - 8) * 2);",
               "                const i32x4 o = i32x4{(int)pk[0].x, (int)pk[0].y, (int)pk[NT - 1].x, (int)pk[NT - 1].y};"),
              ("                    for (int tt = 0; tt < NT; ++tt) *reinterpret_cast<uint2*>(scr + (px & 7) * SCR_ROW + (tt * 16 + kq * 4) * 2) = pk[tt];",
               "                    for (int tt = 0; tt < 1; ++tt) {}")],
    # per-phase shader-clock totals over all waves (esr_dbg_prof exports them): issue / epilogue / compute / vmcnt wait / barrier
    "ptime": [("namespace {", "__device__ unsigned long long g_ph[8];\nnamespace {", 1),
              ("            // ---- top of stage s ----", "            const unsigned long long T0 = clock64();\n            // ---- top of stage s ----", 1),
              ("            hist_st <<= 1;", "            const unsigned long long T1 = clock64(); ph[0] += T1 - T0;\n            hist_st <<= 1;", 1),
              ("            // ---- compute ----", "            const unsigned long long T2 = clock64(); ph[1] += T2 - T1;\n            // ---- compute ----", 1),
              ("            // ---- sync: stage s+1 has landed", "            const unsigned long long T3 = clock64(); ph[2] += T3 - T2;\n            // ---- sync: stage s+1 has landed", 1),
              ("                wait_vm_dyn(younger < 0 ? 0 : cnt);\n                __builtin_amdgcn_s_barrier();",
               "                wait_vm_dyn(younger < 0 ? 0 : cnt);\n                const unsigned long long T4 = clock64(); ph[3] += T4 - T3;\n                __builtin_amdgcn_s_barrier();\n                const unsigned long long T5 = clock64(); ph[4] += T5 - T4; ph[5] += 1;", 1),
              ("    int slot = 0;\n    int s = 0;", "    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};\n    const unsigned long long TK0 = clock64();\n    int slot = 0;\n    int s = 0;", 1),
              ("    if (pend) epilogue(pn, px0, py0, 0);\n}", "    const unsigned long long TE0 = clock64();\n    if (pend) epilogue(pn, px0, py0, 0);\n    const unsigned long long TE1 = clock64();\n    if (lane == 0) { for (int i = 0; i < 6; ++i) atomicAdd(&g_ph[i], ph[i]); atomicAdd(&g_ph[6], TE1 - TE0); atomicAdd(&g_ph[7], TE1 - TK0); }\n}", 1),
              ("__END__", "\nextern \"C\" void esr_dbg_prof(unsigned long long* out) { (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ph), 64); unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ph), z, 64); }\n", 1)],
    "nodma": [("                dma_buf16(dst0 + (unsigned)pc * 1024u, lvoff[r], lrsrc, soff);", "                ;"),
              ("                wait_vm_dyn(younger < 0 ? 0 : cnt);", "                ;")],
}


def build(only=None):
    base = open(os.path.join(SRC, "esr_s16.hip")).read()
    others = [os.path.join(SRC, f) for f in ("esr_hip.hip", "esr_esa.hip", "esr_bsconv.hip")]
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
            s = s.replace(a, b, *sub[2:])
        src = os.path.join(HERE, f"s16_{name}.hip")
        open(src, "w").write(s)
        vo = os.path.join(HERE, f"obj_s16_{name}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I", os.path.join(REPO, "include"),
                               "-I", SRC, src, "-o", vo] + (["-DESR_S16_NW=16"] if "nw16" in name else []))
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
            if name == "ptime":
                out = (ctypes.c_ulonglong * 8)()
                lib.esr_dbg_prof(out)          # reset
                lib.esr_conv2d_f32(ctypes.byref(d), st)
                lib.esr_dbg_prof(out)
                v = list(out); nw = 256 * 8
                print("ptime per wave [shader clocks]: issue %.0f  epilogue+res %.0f  compute %.0f  vmcnt-wait %.0f  barrier %.0f  stages %.0f  last-epi %.0f  total %.0f" % tuple(x / nw for x in v), flush=True)
            print(f"{cin:3d}->{cout:3d} k{k} res{res} {name:8s} {ms:.4f} ms  {gb / ms:.0f} GB/s(alg)  {2 * 32 * 65536 * cin * cout * k * k / ms / 1e9:.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    build(sys.argv[2:]) if sys.argv[1] == "build" else run()
