#!/usr/bin/env python3
"""ISA lint of every kernel of libesr_hip.so (hipcc cross-compiles each translation unit to gfx950 assembly; no GPU needed).

Rule 1 -- the packed-fp32 op_sel erratum (round 4, LAB_NOTES.md; reproduced in isolation by tools/dbg/pk_opsel_probe.hip):
    a v_pk_{mul,add,fma}_f32 / v_pk_mov_b32 whose `op_sel:[...]` has a 1 -- a LOW result half reading the HIGH dword of a 64-bit source
    pair -- returns 0 in lanes 48..63 when a wave of another kernel issues MFMAs on the same SIMD (forwards on several HIP streams).
    hipcc emits that encoding whenever a scalar factor happens to sit in the odd register of a pair; the sources route such factors
    through esr_lone() (csrc/esr_internal.h).  No kernel of the library may contain the encoding.
Rule 3 -- prologues and wait states (round 5, LAB_NOTES.md 10.8): hipcc keeps ONE load in flight in a `for (e = tid; ..)` staging loop, sinks a
    load into the predicated store that uses it, and pads every dependent v_pk_fma_f32 with an s_nop.  Guards for the kernels that were fixed:
    esa_apply_mfma_kernel -- at most 3 waits for an empty load queue in front of the first barrier (24 before); esa_s2pool16_kernel -- every
    16-byte patch load in front of the first LDS store of the patch; conv48r / conv48rq_kernel with the GELU compiled in -- < 160 s_nop (510).
Rule 4 -- conv64m_kernel, rfdb_tail_kernel (esr_c64m.hip) and rlfb_chain_kernel (esr_chain.hip): no scratch access, no dynamic register
    indexing -- their s_waitcnt vmcnt(N) count every vector-memory instruction, and the DMA pieces leave m0 changed.
Rule 5 -- rlfb_chain_kernel: the post wave's counted wait (PIECES + STORES) (D - 2) + STORES is in the code, and between two barriers
    of the step loop there are exactly PIECES LDS-DMA loads and exactly STORES stores (ChGeo<G>): an extra vector-memory instruction would loosen it.
Rule 2 -- conv_s16_kernel's wait-count arithmetic (tools/lint_s16_isa.py, unchanged): no scratch / spills, no copies out of registers an
    in-flight asm load writes.

usage: lint_isa.py [--src FILE.hip ...]      exit status 0 = clean.   Assemblies are cached under build/isa/ by source hash."""
import hashlib, os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "ntire2022_esr_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BAD_OPSEL = re.compile(r"^\s+(v_pk_(?:mul|add|fma)_f32|v_pk_mov_b32)\b.*\bop_sel:\[([01,]+)\]", re.M)


def _deps_hash(src):
    h = hashlib.sha256(open(src, "rb").read())
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".inc", ".h")) and not f.endswith(".gen.h"):
            h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(REPO, "include", "esr_hip.h"), "rb").read())
    return h.hexdigest()[:16]


def assemble(src):
    """src (.hip) -> path of its gfx950 assembly (cached)."""
    d = os.path.join(REPO, "build", "isa")
    os.makedirs(d, exist_ok=True)
    out = os.path.join(d, f"{os.path.basename(src)[:-4]}.{_deps_hash(src)}.s")
    if not os.path.exists(out):
        tmp = out + f".tmp{os.getpid()}"
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(REPO, "include"), "-I", CSRC,
                               "-S", "--cuda-device-only", src, "-o", tmp], stderr=subprocess.DEVNULL)
        os.replace(tmp, out)
        for f in os.listdir(d):                                   # older assemblies of the same unit
            if f.startswith(os.path.basename(src)[:-4] + ".") and f.endswith(".s") and os.path.join(d, f) != out:
                os.remove(os.path.join(d, f))
    return out


def lint_opsel(path):
    """[(kernel symbol, instruction text)] for every packed-fp32 instruction with a 1 in op_sel."""
    txt = open(path).read()
    labels = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):", txt, re.M)]
    out = []
    for m in BAD_OPSEL.finditer(txt):
        if "1" not in m.group(2):
            continue
        name = "?"
        for pos, lab in labels:
            if pos > m.start():
                break
            name = lab
        out.append((name, " ".join(m.group(0).split())))
    return out


def _kernels(path):
    """[(symbol, [instruction lines])] of an assembly file (labels, directives and comments dropped)."""
    out, cur = [], None
    for l in open(path):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = (m.group(1), [])
            out.append(cur)
            continue
        if cur is None:
            continue
        if l.startswith(".Lfunc_end"):
            cur = None
            continue
        t = l.strip()
        if t and not t.startswith((";", ".")):
            cur[1].append(t)
    return out


def lint_prologues(path, unit):
    """rule 3 (see the module docstring): [problem strings]"""
    problems = []
    for name, ins in _kernels(path):
        if unit == "esr_esa.hip" and "esa_apply_mfma_kernel" in name:
            end = next((i for i, t in enumerate(ins) if t.startswith("s_barrier")), len(ins))
            drains = sum(1 for t in ins[:end] if t.startswith("s_waitcnt") and "vmcnt(0)" in t)
            if drains > 3:
                problems.append(f"{unit}: {name}: {drains} s_waitcnt vmcnt(0) in the prologue (weight-image loads serialised again)")
        if unit == "esr_esa_lowres.hip" and "esa_s2pool16_kernel" in name:
            end = next((i for i, t in enumerate(ins) if t.startswith("s_barrier")), len(ins))
            first_store = next((i for i, t in enumerate(ins[:end]) if t.startswith("ds_write_b128")), end)
            late = sum(1 for t in ins[first_store:end] if t.startswith("global_load_dwordx4") and "offset:1024" not in t)
            if late:
                problems.append(f"{unit}: {name}: {late} patch load(s) behind the first LDS store (sunk into the predicated stores again)")
        if unit == "esr_r16.hip" and re.search(r"conv48r_kernelILb[01]ELi[23]ELb1ELi[48]ELi[37]E|conv48rq_kernelILb[01]ELi15E", name):
            nops = sum(1 for t in ins if t.startswith("s_nop"))
            if nops >= 160:
                problems.append(f"{unit}: {name}: {nops} s_nop (the GELU's Horner chains are serial again)")
        # rule 4 (round 6): kernels that count their vector-memory instructions (exact s_waitcnt vmcnt) or leave m0 changed behind a DMA piece
        # must not spill (a scratch access is a VMEM instruction the count does not know) and must not index registers dynamically (the only
        # sequences in which hipcc keeps a value of its own in m0)
        if (unit == "esr_c64m.hip" and ("conv64m_kernel" in name or "rfdb_tail_kernel" in name)) or (unit == "esr_chain.hip" and "rlfb_chain_kernel" in name):
            bad = [t for t in ins if t.startswith(("scratch_", "s_set_gpr_idx", "s_movrel", "v_movrel"))]
            if bad:
                problems.append(f"{unit}: {name}: {len(bad)} scratch / dynamic register index instruction(s), first: {bad[0]}")
        # rule 5 (round 6): rlfb_chain_kernel's post wave waits with a COUNTED vmcnt -- (PIECES + STORES) (D - 2) + STORES -- which is
        # right only while a step issues exactly PIECES LDS-DMA loads and STORES buffer stores (esr_chain.hip: ChGeo<G>): between two barriers
        # of the step loop there are exactly that many, and the counted wait is in the code
        if unit == "esr_chain.hip" and "rlfb_chain_kernel" in name:
            m = re.search(r"rlfb_chain_kernelILb[01]ELi(\d)E", name)
            if m:
                G = int(m.group(1))
                pieces, stores = ((16 * G + 2) * 6 + 63) // 64, G + 2 * ((G + 1) // 2)
                D = 8 if (pieces + stores) * 6 + stores <= 63 else 6
                want = (pieces + stores) * (D - 2) + stores
                segs = [[]]
                for t in ins:
                    if t.startswith("s_barrier"):
                        segs.append([])
                    else:
                        segs[-1].append(t)
                if not any(t.startswith("s_waitcnt") and f"vmcnt({want})" in t for t in ins):
                    problems.append(f"{unit}: {name}: the post wave's counted wait s_waitcnt vmcnt({want}) is not in the code")
                for i, sg in enumerate(segs[1:], 1):          # (segment 0: the prologue -- weights and the first D input rows)
                    nl = sum(1 for t in sg if t.startswith("buffer_load") and " lds" in t)
                    ns = sum(1 for t in sg if t.startswith(("buffer_store", "global_store")))
                    if nl and nl != pieces:
                        problems.append(f"{unit}: {name}: {nl} LDS-DMA loads between two barriers of the step loop, the counted wait assumes {pieces}")
                    if ns and ns != stores:
                        problems.append(f"{unit}: {name}: {ns} stores between two barriers of the step loop, the counted wait assumes {stores}")
    return problems


def main(argv):
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    if "--src" in argv:
        srcs = argv[argv.index("--src") + 1:]
    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as ex:
        asms = list(ex.map(assemble, srcs))
    problems = []
    for src, a in zip(srcs, asms):
        hits = lint_opsel(a)
        by_kernel = {}
        for k, ins in hits:
            by_kernel.setdefault(k, []).append(ins)
        for k, v in by_kernel.items():
            problems.append(f"{os.path.basename(src)}: {k}: {len(v)} packed-fp32 instruction(s) with op_sel reading a high dword, e.g. `{v[0]}`")
        problems += lint_prologues(a, os.path.basename(src))
        if os.path.basename(src) == "esr_s16.hip":
            sys.path.insert(0, os.path.join(REPO, "tools"))
            import lint_s16_isa
            n, p2 = lint_s16_isa.lint(a)
            print(f"{n} conv_s16_kernel variants checked (tools/lint_s16_isa.py)")
            if n == 0:
                p2 = p2 + ["no conv_s16_kernel variant found in esr_s16.s"]
            problems += p2
    print(f"{len(srcs)} translation units, {len(problems)} problems")
    for p in problems:
        print("  ", p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
