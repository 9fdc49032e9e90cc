#!/usr/bin/env python3
"""ISA lint of conv_s16_kernel (hipcc cross-compiles esr_s16.hip to assembly, no GPU needed).

The kernel's `s_waitcnt vmcnt(N)` arithmetic counts every vector-memory instruction the wave issues, and its residual
registers (post-chain variants) are written by asm loads hipcc believes complete at once.  Both break silently if the
compiler does one of these, so the build is checked for them:
  * scratch instructions / VGPR spills in any conv_s16_kernel variant (a scratch access is a VMEM instruction the counts do
    not know about, and hipcc guards it with its own vmcnt(0));
  * a copy (v_mov) or spill out of a register that a residual `buffer_load_dwordx2` (inline asm) writes: the copy would read
    the register before the data has arrived.
usage: lint_s16_isa.py [path of an existing .s]     exit status 0 = clean"""
import os, re, subprocess, sys, tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assemble():
    out = os.path.join(tempfile.mkdtemp(prefix="esr_lint_"), "esr_s16.s")
    src = os.path.join(REPO, "ntire2022_esr_amd", "csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(REPO, "include"), "-I", src,
                           "-S", "--cuda-device-only", os.path.join(src, "esr_s16.hip"), "-o", out], stderr=subprocess.DEVNULL)
    return out


def lint(path):
    txt = open(path).read()
    starts = [m.start() for m in re.finditer(r"^_ZN12_GLOBAL__N_115conv_s16_kernel\S*:", txt, re.M)]
    problems, n = [], 0
    for i, st in enumerate(starts):
        k = txt[st: starts[i + 1] if i + 1 < len(starts) else len(txt)]
        name = k[:k.find(":")]
        body = k[:k.find(".Lfunc_end")]
        n += 1
        if re.search(r"^\s+scratch_", body, re.M):
            problems.append(f"{name}: scratch instructions")
        regs = set()
        for mm in re.finditer(r"buffer_load_dwordx2 v\[(\d+):(\d+)\]", body):
            regs.update((int(mm.group(1)), int(mm.group(2))))
        for line in body.split("\n"):
            mm = re.match(r"\s+v_mov_b32_e32 v\d+, v(\d+)\s*$", line) or re.match(r"\s+v_mov_b64_e32 v\[\d+:\d+\], v\[(\d+):\d+\]", line)
            if mm and int(mm.group(1)) in regs:
                problems.append(f"{name}: copy out of a residual register: {line.strip()}")
                break
    for mm in re.finditer(r"\.name:\s+(_ZN12_GLOBAL__N_115conv_s16_kernel\S*)(.*?)\.vgpr_spill_count:\s+(\d+)", txt, re.S):
        if int(mm.group(3)):
            problems.append(f"{mm.group(1)}: {mm.group(3)} VGPR spills")
    return n, problems


if __name__ == "__main__":
    n, problems = lint(sys.argv[1] if len(sys.argv) > 1 else assemble())
    print(f"{n} conv_s16_kernel variants, {len(problems)} problems")
    for p in problems:
        print("  ", p)
    sys.exit(1 if problems or n == 0 else 0)
