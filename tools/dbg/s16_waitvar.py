#!/usr/bin/env python3
"""race hunt: conv_s16_kernel with CONSERVATIVE stage waits (the epilogue stores are waited for too) -> tools/abl/libesr_s16cons.so"""
import os, subprocess
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(REPO, "ntire2022_esr_amd", "csrc")
s = open(os.path.join(SRC, "esr_s16.hip")).read()
a = "            wait_vm_dyn((R - 2) * n_my + epi_stores * __builtin_popcount(hist_st & hmask) + (GRES ? RES_LOADS : 0) * __builtin_popcount(hist_rs & hmask));"
assert s.count(a) == 1
s = s.replace(a, "            wait_vm_dyn((R - 2) * n_my);")
src = os.path.join(HERE, "s16cons.hip"); open(src, "w").write(s)
obj = os.path.join(HERE, "s16cons.o")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I", os.path.join(REPO, "include"), "-I", SRC, src, "-o", obj], stderr=subprocess.DEVNULL)
objdir = os.path.join(REPO, "build", "obj")
others = [os.path.join(objdir, f) for f in sorted(os.listdir(objdir)) if f.endswith(".o") and f != "esr_s16.o"]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj] + others + ["-o", os.path.join(REPO, "tools", "abl", "libesr_s16cons.so")])
os.remove(src); os.remove(obj)
print("built")
