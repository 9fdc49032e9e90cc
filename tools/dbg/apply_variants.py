#!/usr/bin/env python3
"""debug builds of esr_esa.hip for the multi-stream race hunt: tools/abl/libesr_av_<name>.so"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(REPO, "ntire2022_esr_amd", "csrc")
base = open(os.path.join(SRC, "esr_esa.hip")).read()
END = "        if (!more) break;\n        cur = nxt;\n        grp = gn;\n    }\n}\n"
assert base.count(END) == 1
KDEF = "__global__ __launch_bounds__(256) void esa_apply_mfma_kernel(const EsaK p)"
assert base.count(KDEF) == 1
V = {
    "endsync": base.replace(END, "        if (!more) break;\n        cur = nxt;\n        grp = gn;\n    }\n    __syncthreads();\n}\n"),
    "occ3": base.replace(KDEF, "__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 3))) void esa_apply_mfma_kernel(const EsaK p)"),
    "ldspad": base.replace(END, "        if (!more) break;\n        cur = nxt;\n        grp = gn;\n    }\n    { __shared__ char pad[44 * 1024]; if (p.N < 0) { pad[threadIdx.x] = 1; __syncthreads(); if (pad[threadIdx.x ^ 1] == 3) ys[0] = 1; } }\n}\n"),
}
objdir = os.path.join(REPO, "build", "obj")
others = [os.path.join(objdir, f) for f in sorted(os.listdir(objdir)) if f.endswith(".o") and f != "esr_esa.o"]
for name, s in V.items():
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    src = os.path.join(HERE, f"av_{name}.hip"); open(src, "w").write(s)
    obj = os.path.join(HERE, f"av_{name}.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I", os.path.join(REPO, "include"), "-I", SRC, src, "-o", obj], stderr=subprocess.DEVNULL)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj] + others + ["-o", os.path.join(REPO, "tools", "abl", f"libesr_av_{name}.so")])
    os.remove(src); os.remove(obj)
    print("built", name)
