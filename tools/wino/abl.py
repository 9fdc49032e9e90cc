"""Ablations of wino_f32_kernel: textual variants of csrc/esr_wino.hip built into tools/wino/libesr_<name>.so (git-ignored), timed
on the 64->64 3x3 at batch 32.  Results of the ablated variants are WRONG by construction; only their time is read.
usage: abl.py build [names...] | abl.py run [names...]"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(REPO, "ntire2022_esr_amd", "csrc")

READ2 = [("    typedef const volatile __attribute__((address_space(3))) f32x2* lds_ptr;", "    typedef const __attribute__((address_space(3))) f32x2* lds_ptr;")]
EPI = [("                        if (pok[a][b] && cok) *reinterpret_cast<f32x4*>(dbase + dlane + (size_t)pix[a][b] * dps) = v;",
        "                        if (pok[a][b] && cok && v.x == 1.2345e30f) *reinterpret_cast<f32x4*>(dbase + dlane + (size_t)pix[a][b] * dps) = v;")]
BAR12 = lambda n: [("constexpr int WN_BARRIER_POS = 12;", f"constexpr int WN_BARRIER_POS = {n};")]
RAWPOS = lambda a, b, c: [("                if (pos == 2 || pos == 4 || pos == 6) {\n                    const int i = pos / 2 - 1;", f"                if (pos == {a} || pos == {b} || pos == {c}) {{\n                    const int i = pos == {a} ? 0 : pos == {b} ? 1 : 2;")]
SUBS = {
    "base": [], "read2": READ2, "noepi": EPI,
    "share8": [("constexpr int WN_FIRST_SHARE = 9; ", "constexpr int WN_FIRST_SHARE = 8; ")],
    "share10": [("constexpr int WN_FIRST_SHARE = 9; ", "constexpr int WN_FIRST_SHARE = 10;")],
    "raw012": RAWPOS(0, 1, 2), "raw91011": RAWPOS(9, 10, 11), "raw159": RAWPOS(1, 5, 9),
}


def build(names):
    base = open(os.path.join(SRC, "esr_wino.hip")).read()
    objdir = os.path.join(REPO, "build", "obj")
    others = [os.path.join(objdir, f) for f in sorted(os.listdir(objdir)) if f.endswith(".o") and f != "esr_wino.o"]
    for name in names or SUBS:
        s = base
        for a, b in SUBS[name]:
            assert a in s, (name, a)
            s = s.replace(a, b)
        src = os.path.join(HERE, f"wino_{name}.hip")
        open(src, "w").write(s)
        obj = os.path.join(HERE, f"wino_{name}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I", os.path.join(REPO, "include"),
                               "-I", SRC, src, "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj] + others + ["-o", os.path.join(HERE, f"libesr_{name}.so")])
        os.remove(src); os.remove(obj)
        print("built", name, flush=True)


def run(names):
    import torch
    sys.path.insert(0, REPO)
    from ntire2022_esr_amd import _lib as L
    from ntire2022_esr_amd.engine import pack_conv, pack_wino
    B, H, W, cin, cout = 32, 256, 256, int(os.environ.get("CIN", "64")), 64
    dev = "cuda:0"
    x = torch.randn(B, H, W, cin, device=dev); y = torch.zeros(B, H, W, cout, device=dev)
    w = torch.randn(cout, cin, 3, 3) * 0.05; bias = torch.randn(cout)
    blob = pack_conv(w, bias).to(dev); wb = pack_wino(w, bias).to(dev)
    for name in names or SUBS:
        so = os.path.join(HERE, f"libesr_{name}.so")
        if not os.path.exists(so):
            continue
        lib = ctypes.CDLL(so)
        lib.esr_conv2d_f32.argtypes = [ctypes.POINTER(L.ConvDesc), ctypes.c_void_p]
        d = L.ConvDesc(); d.n, d.h, d.w, d.cin, d.cout, d.ksize = B, H, W, cin, cout, 3
        d.act, d.slope = 1, 0.05
        d.inp = L.View(x.data_ptr(), cin, 0); d.out0 = L.View(y.data_ptr(), cout, 0)
        d.wpacked = blob.data_ptr(); d.wino_wpacked = wb.data_ptr()
        dbg = torch.zeros(512 * 4 * 8, dtype=torch.int64, device=dev)
        if name.startswith("probe"): d.res = L.View(dbg.data_ptr(), 64, 0)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(int(os.environ.get('WARM', '100'))): assert lib.esr_conv2d_f32(ctypes.byref(d), st) == 0
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(100): lib.esr_conv2d_f32(ctypes.byref(d), st)
        e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 100
        ex = 2.0 * B * H * W * cin * cout * 4
        print(f"{name:18s} {ms:.4f} ms   executed {ex / ms / 1e9:6.1f} TFLOP/s = {ex / ms / 1e9 / 157.3:.3f}", flush=True)
        if name == "probe2":
            raw = dbg.cpu().reshape(512 * 4, 8)
            for slot in (0, 1):
                sel = raw[(raw[:, 5] & 0xf) == slot].double()
                n = sel[:, 4].mean()
                print(f"    wave slot {slot}: {len(sel)} waves, {n:.0f} stages each; per stage: DMA issue {sel[:, 0].mean() / n:.0f}  pos loop {sel[:, 1].mean() / n:.0f}  vmcnt wait {sel[:, 2].mean() / n:.0f}  barrier {sel[:, 3].mean() / n:.0f} cycles")
            continue
        if name.startswith("probe"):
            raw = dbg.cpu().reshape(512 * 4, 8); t = raw.double()
            n = t[:, 4].mean()
            print(f"    per wave (s_memtime ticks, 100 MHz): items {n:.1f}  stages {t[:, 0].mean() / n:.1f} / item  epilogue {t[:, 1].mean() / n:.1f}  setup {t[:, 2].mean() / n:.1f}  total {t[:, 3].mean():.0f} (max {t[:, 3].max():.0f}, min {t[:, 3].min():.0f})  -> clock {t[:, 3].max() / ms / 1e6:.3f} GHz, MFMA pipe busy {n * 8 * 64 * 32 * 2 / t[:, 3].max():.3f}")
            if os.environ.get("DUMP"):
                import collections
                hw = raw[:, 5]; xcc = raw[:, 7] & 0xf
                cu = ((hw >> 8) & 0xf); sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 3; wid = hw & 0xf
                t0 = raw[:, 6] - raw[:, 6].min()
                per_cu = collections.defaultdict(list)
                for i in range(0, 2048, 4):
                    per_cu[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))].append((i // 4, int(raw[i, 3]), int(t0[i]), int(wid[i])))
                cnt = collections.Counter(len(v) for v in per_cu.values())
                print("    blocks per (xcc, se, sh, cu):", dict(cnt), " distinct CUs:", len(per_cu))
                for k in sorted(per_cu)[:12]:
                    print("     ", k, per_cu[k])
                dur = raw[::4, 3].double()
                for nb in sorted(cnt):
                    sel = [v[1] for vs in per_cu.values() if len(vs) == nb for v in vs]
                    print(f"    CUs with {nb} block(s): mean duration {sum(sel) / len(sel):.0f}")


if __name__ == "__main__":
    (build if sys.argv[1] == "build" else run)(sys.argv[2:])
