#!/usr/bin/env python3
"""debug build of esa_apply_mfma_kernel that re-derives its LDS weight images when a block ends and counts corrupted elements
(a foreign write into the block's LDS shows up here): build -> tools/dbg/libesr_ldschk.so ; run on the GPU box"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(REPO, "ntire2022_esr_amd", "csrc")
if sys.argv[1] == "build":
    s = open(os.path.join(SRC, "esr_esa.hip")).read()
    a = "        if (!more) break;\n        cur = nxt;\n        grp = gn;\n    }\n}\n"
    assert s.count(a) == 1
    chk = '''        if (!more) break;
        cur = nxt;
        grp = gn;
    }
    if (NP0 == 0) {
        __syncthreads();
        unsigned bad = 0;
        for (int e = tid; e < (1 + 2 * NT) * 512; e += 256) {
            const int img = e >> 9, l = (e >> 3) & 63, j = e & 7;
            const int i = l & 15, kq2 = l >> 4;
            float v = 0.f;
            if (img == 0) {
                const float w = p.wf[(8 * (kq2 & 1) + j) * FP + i];
                const float hi = from16<ST>(to16<ST>(w));
                v = kq2 < 2 ? hi : w - hi;
            } else {
                const int t = (img - 1) % NT, lo = (img - 1) / NT;
                const int oc = 32 * (t >> 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3);
                const float w = oc < p.cp ? p.w4[(4 * kq2 + (j & 3)) * p.cp + oc] : 0.f;
                const float hi = from16<ST>(to16<ST>(w));
                v = lo ? (j < 4 ? w - hi : 0.f) : hi;
            }
            if (simg[e] != to16<ST>(v)) ++bad;
        }
        if (bad) atomicAdd(&g_lds_bad, bad);
    }
}
'''
    s = s.replace(a, chk)
    s = s.replace("template <int ST, int NP, int NP0 = 0, int NP1 = 0>     // NP = channel pairs", "__device__ unsigned g_lds_bad;\ntemplate <int ST, int NP, int NP0 = 0, int NP1 = 0>     // NP = channel pairs")
    s = s.replace("int esr_esa_apply_post_supported(int c, int cout0, int cout1)\n{", "unsigned esr_dbg_lds_bad(void) { unsigned v = 0; hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_lds_bad), 4); return v; }\nint esr_esa_apply_post_supported(int c, int cout0, int cout1)\n{")
    src = os.path.join(HERE, "esa_ldschk.hip"); open(src, "w").write(s)
    obj = os.path.join(HERE, "esa_ldschk.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I", os.path.join(REPO, "include"), "-I", SRC, src, "-o", obj])
    objdir = os.path.join(REPO, "build", "obj")
    others = [os.path.join(objdir, f) for f in sorted(os.listdir(objdir)) if f.endswith(".o") and f != "esr_esa.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj] + others + ["-o", os.path.join(HERE, "libesr_ldschk.so")])
    os.remove(src); os.remove(obj)
    print("built")
