#!/usr/bin/env python3
"""static census of a gfx950 .s file (hipcc -save-temps): per kernel registers / scratch / LDS and the opcode mix of its largest loop
usage: census.py file.s [kernel-name-substring]"""
import collections, re, sys
src = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
# split by kernel
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*\.end_amdhsa_kernel", src, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if want and want not in name:
        continue
    meta = {k: re.search(r"\.%s\s+(\S+)" % k, body) for k in ("amdhsa_next_free_vgpr", "amdhsa_accum_offset", "amdhsa_private_segment_fixed_size", "amdhsa_group_segment_fixed_size")}
    code = body.split(".section")[0]
    lines = [l.strip() for l in code.split("\n")]
    ins = [l for l in lines if l and not l.startswith((".", ";", "//")) and not l.endswith(":")]
    # largest basic-block-ish region between labels
    blocks, cur = [], []
    for l in lines:
        if l.startswith(".LBB") and ":" in l.split()[0]:
            blocks.append(cur); cur = []
        elif l and not l.startswith((".", ";", "//")):
            cur.append(l)
    blocks.append(cur)
    big = max(blocks, key=len)
    def kind(op):
        if op.startswith("v_mfma"): return "mfma"
        if op.startswith("s_waitcnt"): return "s_waitcnt"
        if op.startswith("s_nop"): return "s_nop"
        if op.startswith("s_"): return "salu"
        if op.startswith("ds_"): return "lds"
        if op.startswith(("buffer_", "global_", "flat_")): return "vmem"
        if op.startswith("v_pk_") and "f32" in op: return "valu_pk_f32"
        if op.startswith("v_accvgpr"): return "accvgpr_mov"
        return "valu"
    for label, seq in (("whole kernel", ins), ("largest block", big)):
        c = collections.Counter(kind(l.split()[0]) for l in seq)
        ops = collections.Counter(l.split()[0] for l in seq)
        nm = c.get("mfma", 0)
        print(f"{name[:70]} [{label}]: {len(seq)} instr, {nm} MFMA, {(len(seq) - nm) / max(nm, 1):.2f} others per MFMA")
        print("   ", dict(c))
        print("    top:", ", ".join(f"{n} {o}" for o, n in ops.most_common(14)))
    print("    meta:", {k: (v.group(1) if v else None) for k, v in meta.items()})
    nops = sum(int(l.split()[1]) + 1 for l in big if l.startswith("s_nop"))
    print(f"    s_nop wait states in largest block: {nops}")
