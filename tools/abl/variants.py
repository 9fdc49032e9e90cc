#!/usr/bin/env python3
"""Whole-model A/B of conv_s16_kernel variants in ONE gpurun call (research tooling, never part of libesr_hip.so).

  python tools/abl/variants.py build            authoring container: tools/abl/libesr_v_<name>.so from patched copies of csrc/esr_s16.hip
  python tools/abl/variants.py run "team04_rlfn bf16" "rfdn_baseline bf16 --sizes div2k --streams 4" ...
                                                GPU box: bench.py per variant, alternating, two rounds (single-kernel loops are NOT
                                                representative: clocks and L2 state differ from the model's -- A/B at the model level)
A variant = (base revision of esr_s16.hip, [patch functions]).  `head` = the committed file, `cur` = the working tree.
"""
import json, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(REPO, "ntire2022_esr_amd", "csrc")


def head_src():
    return subprocess.check_output(["git", "-C", REPO, "show", "HEAD:ntire2022_esr_amd/csrc/esr_s16.hip"]).decode()


def cur_src():
    return open(os.path.join(SRC, "esr_s16.hip")).read()


def sub(s, a, b, cnt=1):
    assert s.count(a) == cnt, (a, s.count(a))
    return s.replace(a, b)


def p_cfg(s):
    """the per-launch loop constants in one opaque SGPR (no kernarg re-loads in the per-stage code)"""
    s = sub(s, "    const int R = p.ring;\n",
            "    int cfg = p.nchunks | ((p.nchunks + p.nres) << 6) | (p.seg_chunks << 12) | (p.ring << 18) | ((p.res_in ? 1 : 0) << 22) | ((p.border ? 1 : 0) << 23) |\n"
            "              ((p.act == ESR_ACT_GELU ? 1 : 0) << 24) | ((p.out_layout != ESR_NCHW_SHUFFLE4 ? 1 : 0) << 25);\n"
            "    asm volatile(\"\" : \"+s\"(cfg));\n#define S16_CFG(shift, bits) ((cfg >> (shift)) & ((1 << (bits)) - 1))\n#define S16_NCHUNKS S16_CFG(0, 6)\n"
            "    const int R = S16_CFG(18, 4);\n")
    s = sub(s, "    const int G = gridDim.x;\n", "    int G = gridDim.x;\n    asm volatile(\"\" : \"+s\"(G));\n")
    s = sub(s, "    const int ntiles = p.N * p.tiles_y * p.tiles_x;\n", "    int ntiles = p.N * p.tiles_y * p.tiles_x;\n    asm volatile(\"\" : \"+s\"(ntiles));\n")
    s = sub(s, "    const int nstages = p.nchunks + p.nres; ", "    const int nstages = S16_CFG(6, 6); ")
    s = sub(s, "            if (lc >= p.nchunks) dma_buf16(dst, LVR(i), lrsrcr, (unsigned)(lc - p.nchunks) * 32u);", "            if (lc >= S16_NCHUNKS) dma_buf16(dst, LVR(i), lrsrcr, (unsigned)(lc - S16_NCHUNKS) * 32u);")
    s = sub(s, "        if (++lcc == p.seg_chunks) { ", "        if (++lcc == S16_CFG(12, 6)) { ")
    s = sub(s, "(unsigned long long)p.seg_stride;", "(unsigned long long)seg_stride;")
    s = sub(s, "    auto cursor_advance = [&]() __attribute__((always_inline)) {", "    long long seg_stride = p.seg_stride;\n    asm volatile(\"\" : \"+s\"(seg_stride));\n    auto cursor_advance = [&]() __attribute__((always_inline)) {")
    s = sub(s, "    const bool swap_epi = p.out_layout != ESR_NCHW_SHUFFLE4;", "    const bool swap_epi = S16_CFG(25, 1) != 0;")
    s = sub(s, "    const bool act_gelu = p.act == ESR_ACT_GELU;", "    const bool act_gelu = S16_CFG(24, 1) != 0;")
    s = sub(s, "        wait_vm_dyn(p.nchunks * n_my);", "        wait_vm_dyn(S16_NCHUNKS * n_my);")
    s = sub(s, "        if (KS == 3 && p.res_in) {", "        if (KS == 3 && S16_CFG(22, 1)) {")
    s = sub(s, "        if (c == p.nchunks - 1 && p.border) border_fix(x0, y0);", "        if (S16_CFG(23, 1) && c == S16_NCHUNKS - 1) border_fix(x0, y0);")
    s = sub(s, "        if (post && c == p.nchunks) {", "        if (post && c == S16_NCHUNKS) {")
    s = sub(s, "            int cc = c - p.nchunks;", "            int cc = c - S16_NCHUNKS;")
    s = sub(s, "                if (c >= p.nchunks) residual_stage(c, last);", "                if (c >= S16_NCHUNKS) residual_stage(c, last);")
    s = sub(s, "    const int epi_stores = p.out_layout == ESR_NCHW_SHUFFLE4 ? RW * NT", "    int epi_stores = p.out_layout == ESR_NCHW_SHUFFLE4 ? RW * NT")
    s = sub(s, "+ P1_STORES + P2_STORES;   // stores per wave and tile\n", "+ P1_STORES + P2_STORES;   // stores per wave and tile\n    asm volatile(\"\" : \"+s\"(epi_stores));\n")
    return s


def p_clump(s):
    """the working tree's compute() with the next pair's fragment reads in ONE clump in front of the group (as the committed kernel has them)"""
    s = sub(s, "            } else {\n#pragma unroll\n                for (int tt = 0; tt < NT; ++tt)\n#pragma unroll\n                    for (int r = 0; r < RW; ++r) {\n                        acc[tt][r] = mfma32<BF16>(a[cs][tt], b[cs][r], acc[tt][r]);\n                        const int m = tt * RW + r;\n                        if (q + 1 < PAIRS && m < NFRAG) load_one(cs ^ 1, q + 1, m, false);\n",
            "            } else {\n                if (q + 1 < PAIRS) {\n#pragma unroll\n                    for (int i = 0; i < NFRAG; ++i) load_one(cs ^ 1, q + 1, i, false);\n                    __builtin_amdgcn_sched_barrier(0);\n                }\n#pragma unroll\n"
            "                for (int tt = 0; tt < NT; ++tt)\n#pragma unroll\n                    for (int r = 0; r < RW; ++r) {\n                        acc[tt][r] = mfma32<BF16>(a[cs][tt], b[cs][r], acc[tt][r]);\n                        const int m = tt * RW + r;\n")
    s = sub(s, "                for (int i = NT * RW; i < NFRAG; ++i) load_one(cs ^ 1, q + 1, i, false);", "                for (int i = NT * RW; i < NFRAG; ++i) if (q == 0 && EPI) load_one(cs ^ 1, q + 1, i, false);")
    return s


def p_nosb(s):
    """the working tree's compute() without the per-MFMA sched_barriers of the plain groups (hipcc orders reads / MFMAs itself)"""
    return sub(s, "                        if (m == DMA_AT && q < PPW) dma_piece(q);     // the DMA issue rides in the shadow of the matrix pipe\n                        __builtin_amdgcn_sched_barrier(0);\n",
               "                        if (m == DMA_AT && q < PPW) dma_piece(q);\n")


VARIANTS = {
    "head": (head_src, []),
    "cur": (cur_src, []),
    "head_cfg": (head_src, [p_cfg]),
    "cur_clump": (cur_src, [p_clump]),
    "cur_nosb": (cur_src, [p_nosb]),
}


def build(names):
    objdir = os.path.join(REPO, "build", "obj")
    others = [os.path.join(objdir, f) for f in sorted(os.listdir(objdir)) if f.endswith(".o") and f != "esr_s16.o"]
    procs = []
    for name in names or VARIANTS:
        base, patches = VARIANTS[name]
        s = base()
        for f in patches:
            s = f(s)
        src = os.path.join(HERE, f"v_{name}.hip")
        open(src, "w").write(s)
        obj = os.path.join(HERE, f"v_{name}.o")
        procs.append((name, src, obj, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I", os.path.join(REPO, "include"),
                                                        "-I", SRC, src, "-o", obj], stderr=subprocess.DEVNULL)))
    for name, src, obj, pr in procs:
        assert pr.wait() == 0, name
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj] + others + ["-o", os.path.join(HERE, f"libesr_v_{name}.so")])
        os.remove(src); os.remove(obj)
        print("built", name, flush=True)


def run(cases):
    names = [n for n in VARIANTS if os.path.exists(os.path.join(HERE, f"libesr_v_{n}.so"))]
    if os.environ.get("VARIANTS"):
        names = [n for n in os.environ["VARIANTS"].split(",") if n in names]
    res = {}
    for rnd in range(int(os.environ.get("ROUNDS", "2"))):
        for case in cases or ["team04_rlfn bf16"]:
            model, compute, *extra = case.split()
            for n in names:
                so = os.path.join(HERE, f"libesr_v_{n}.so")
                code = (f"import sys; sys.path.insert(0, {REPO!r}); import ntire2022_esr_amd._lib as L; L.SO_PATH = {so!r}; import runpy; "
                        f"sys.argv = ['bench.py', '--model', {model!r}, '--compute', {compute!r}, '--no-cpu-baseline', '--steps', '30'] + {extra!r}; "
                        f"runpy.run_path({os.path.join(REPO, 'bench.py')!r}, run_name='__main__')")
                out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=REPO)
                try:
                    d = json.loads(out.stdout.strip().splitlines()[-1])
                    res.setdefault((case, n), []).append(d["value"])
                    print(f"{case:45s} {n:10s} {d['value']:9.1f} img/s  {d['ms_per_step']:.3f} ms", flush=True)
                except Exception:
                    print(case, n, "FAILED", out.stderr[-300:], flush=True)
    print("---- best of rounds")
    for case in cases:
        ref = max(res.get((case, names[0]), [0]))
        for n in names:
            v = max(res.get((case, n), [0]))
            print(f"{case:45s} {n:10s} {v:9.1f}  {v / ref if ref else 0:.3f}x")


if __name__ == "__main__":
    (build if sys.argv[1] == "build" else run)(sys.argv[2:])
