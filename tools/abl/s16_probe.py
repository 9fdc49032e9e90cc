#!/usr/bin/env python3
"""s_memtime stamps inside conv_s16_kernel (research tooling; the product source carries no probes).

  python tools/abl/s16_probe.py build [names]   authoring container: text-substituted copies of csrc/esr_s16.hip -> tools/abl/libesr_p_<name>.so
  python tools/abl/s16_probe.py run [names]     GPU box: per-wave sums of the phases of the stage loop, shader cycles (s_memtime)

Variants:
  probe     stamps at the stage boundaries: compute (MFMA groups + DMA issue + epilogue), cursor, vmcnt wait, barrier
  probe_dma probe + stamps around every DMA piece issue inside the MFMA groups (perturbs: each stamp drains lgkmcnt)
The debug buffer travels in esr_conv_desc.res (res_mode = NONE): [block][wave][8] dwords.
"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(REPO, "ntire2022_esr_amd", "csrc")

CLK = ("__device__ __forceinline__ void wait_vm_dyn(int cnt)",
       "__device__ __forceinline__ unsigned prof_clk() { unsigned long long t; asm volatile(\"s_memtime %0\\n s_waitcnt lgkmcnt(0)\" : \"=s\"(t) :: \"memory\"); return (unsigned)t; }\n"
       "__device__ __forceinline__ void wait_vm_dyn(int cnt)")
PROBE = [
    CLK,
    ("    int slot = 0;\n    bool pend = false;", "    unsigned pf0 = 0, pf1 = 0, pf2 = 0, pf3 = 0, pf4 = 0, pf5 = 0, pf6 = 0; const unsigned TS = prof_clk();\n    int slot = 0;\n    bool pend = false;"),
    ("            const bool last = c == nst - 1;\n            hist_rs", "            const bool last = c == nst - 1;\n            const unsigned T0 = prof_clk(); const bool epi_stage = c == 0 && pend;\n            hist_rs"),
    ("            cursor_advance();\n            // ---- sync", "            const unsigned T1 = prof_clk();\n            cursor_advance();\n            const unsigned T2 = prof_clk();\n            // ---- sync"),
    ("            if (!OWN_PIECES) __builtin_amdgcn_s_barrier();\n            slot = slot == R - 1 ? 0 : slot + 1;",
     "            const unsigned T3 = prof_clk();\n            if (!OWN_PIECES) __builtin_amdgcn_s_barrier();\n            const unsigned T4 = prof_clk();\n"
     "            if (epi_stage) pf1 += T1 - T0; else pf0 += T1 - T0;\n            pf2 += T2 - T1; pf3 += T3 - T2; pf4 += T4 - T3; pf5 += 1;\n"
     "            slot = slot == R - 1 ? 0 : slot + 1;"),
    ("    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");     // the trailing (zero-fill) DMA writes LDS: it must not outlive the block\n}",
     "    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n"
     "    if (lane == 0 && p.res && p.res_mode == 0) { unsigned* o = (unsigned*)p.res + (blockIdx.x * NW + wv) * 8;\n"
     "        o[0] = pf0; o[1] = pf1; o[2] = pf2; o[3] = pf3; o[4] = pf4; o[5] = pf5; o[6] = pf6; o[7] = prof_clk() - TS; }\n}"),
]
PROBE_DMA = PROBE + [
    ("                            if (m == DMA_AT && q < PPW) dma_piece(q);     // the DMA issue rides in the shadow of the matrix pipe",
     "                            if (m == DMA_AT && q < PPW) { const unsigned Ta = prof_clk(); dma_piece(q); pf6 += prof_clk() - Ta; }"),
]
SUBS = {"probe": PROBE, "probe_dma": PROBE_DMA}


def build(names):
    base = open(os.path.join(SRC, "esr_s16.hip")).read()
    objdir = os.path.join(REPO, "build", "obj")
    others = [os.path.join(objdir, f) for f in sorted(os.listdir(objdir)) if f.endswith(".o") and f != "esr_s16.o"]
    for name in names or SUBS:
        s = base
        for a, b in SUBS[name]:
            assert s.count(a) == 1, (name, a, s.count(a))
            s = s.replace(a, b)
        src = os.path.join(HERE, f"s16_{name}.hip")
        open(src, "w").write(s)
        obj = os.path.join(HERE, f"s16_{name}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I", os.path.join(REPO, "include"),
                               "-I", SRC, src, "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj] + others + ["-o", os.path.join(HERE, f"libesr_p_{name}.so")])
        os.remove(src); os.remove(obj)
        print("built", name, flush=True)


def run(names):
    import torch
    sys.path.insert(0, REPO)
    from ntire2022_esr_amd import _lib as L
    from ntire2022_esr_amd.engine import pack_conv_s16
    dev = "cuda:0"
    B, H, W = 32, 256, 256
    for (cin, cout, k) in ((48, 48, 3), (64, 64, 3), (48, 48, 1)):
        x = torch.randn(B, H, W, cin, device=dev).to(torch.bfloat16)
        y = torch.zeros(B, H, W, cout, device=dev, dtype=torch.bfloat16)
        blob = pack_conv_s16(torch.randn(cout, cin, k, k) * 0.1, torch.randn(cout), "bf16").to(dev)
        dbg = torch.zeros(512 * 8 * 8, dtype=torch.int32, device=dev)
        for name in ["old", "prod"] + list(names or SUBS):
            so = (os.path.join(REPO, "ntire2022_esr_amd", "libesr_hip.so") if name == "prod" else os.path.join(HERE, "libesr_prod.so") if name == "old"
                  else os.path.join(HERE, f"libesr_p_{name}.so"))
            if not os.path.exists(so):
                continue
            lib = ctypes.CDLL(so)
            lib.esr_conv2d_f32.argtypes = [ctypes.POINTER(L.ConvDesc), ctypes.c_void_p]
            lib.esr_conv_block_waves.argtypes = [ctypes.POINTER(L.ConvDesc)]
            d = L.ConvDesc(); d.n, d.h, d.w, d.cin, d.cout, d.ksize = B, H, W, cin, cout, k
            d.act, d.slope, d.storage, d.compute = 1, 0.05, 1, 1
            d.inp = L.View(x.data_ptr(), cin, 0); d.out0 = L.View(y.data_ptr(), cout, 0)
            d.wpacked = blob.data_ptr()
            if name not in ("prod", "old"):
                d.res = L.View(dbg.data_ptr(), 64, 0)
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            for _ in range(30):
                assert lib.esr_conv2d_f32(ctypes.byref(d), st) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                lib.esr_conv2d_f32(ctypes.byref(d), st)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            nw = lib.esr_conv_block_waves(ctypes.byref(d))
            print(f"{cin}->{cout} k{k} {name:10s} {ms:.4f} ms  ({nw} waves per block)", flush=True)
            if name not in ("prod", "old"):
                grid = 512 if nw == 4 else 256
                t = dbg.cpu().reshape(-1, 8)[:grid * nw].double()
                n = t[:, 5].mean()
                tot = t[:, 7]
                names8 = ["compute", "compute+epi", "cursor", "vmcnt wait", "barrier"]
                line = "  ".join(f"{names8[i]} {t[:, i].mean():.0f}" for i in range(5))
                print(f"    per wave, cycles: stages {n:.1f}  {line}  dma-issue {t[:, 6].mean():.0f}  total {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f});  clock {tot.max() / ms / 1e6:.2f} GHz")
                w0 = t.reshape(grid, nw, 8)
                print("    by wave index (compute, wait, barrier):", " ".join(f"[{w0[:, i, 0].mean() + w0[:, i, 1].mean():.0f} {w0[:, i, 3].mean():.0f} {w0[:, i, 4].mean():.0f}]" for i in range(nw)))


if __name__ == "__main__":
    (build if sys.argv[1] == "build" else run)(sys.argv[2:])
