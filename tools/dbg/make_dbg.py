"""Build an ablated / instrumented variant of csrc/esr_hip.hip -> tools/dbg/libesr_dbg_<variant>.so
usage: make_dbg.py [plain|noepi|waits]"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(R, 'ntire2022_esr_amd/csrc/esr_hip.hip')).read()
def rep(a, b, count=1):
    global s
    assert a in s, a[:70]
    s = s.replace(a, b, count)
VAR = sys.argv[1] if len(sys.argv) > 1 else 'plain'
SB = "__builtin_amdgcn_sched_barrier(0);"
EPI = '''        if (p.out_layout == ESR_NCHW_SHUFFLE4) epilogue_shuffle<NT>(p, acc, cur.n, cur.x0, cur.y0, wv, lane);
        else epilogue_nhwc<NT>(p, acc, scr, cur.n, cur.x0, cur.y0, wv, lane);'''
if VAR == 'noepi':
    rep(EPI, '''        for (int tt = 0; tt < NT; ++tt) for (int r = 0; r < 4; ++r) asm volatile("" :: "v"(acc[tt][r]));''')
dbg_api = 'void esr_set_dbg(void* p) { (void)p; }'
if VAR == 'waits':
    # per-wave cycle sums: [0] wait for in_reg, [1] ds_write issue, [2] request ahead, [3] wait DMA (s_waitcnt), [4] barrier, [5] chunk top (glds issue + first frag), [6] MFMA phase
    rep("    int tiles_x, tiles_y;\n", "    int tiles_x, tiles_y;\n    unsigned long long* dbg;\n")
    rep("    k.tiles_y = (d->h + TILE - 1) / TILE;\n", "    k.tiles_y = (d->h + TILE - 1) / TILE;\n    k.dbg = g_dbg;\n")
    rep('thread_local char g_err[256] = "";', 'thread_local char g_err[256] = "";\nunsigned long long* g_dbg = nullptr;')
    dbg_api = 'void esr_set_dbg(void* p) { g_dbg = (unsigned long long*)p; }'
    rep("    int sbuf = 0;\n\n    // bias", "    int sbuf = 0;\n    unsigned long long W0 = 0, W1 = 0, W2 = 0, W3 = 0, W4 = 0, W5 = 0, W6 = 0, W7 = 0, W8 = 0, Ta, Tb, Tc, Td; const unsigned long long Tk0 = clock64();\n\n    // bias") if "    int sbuf = 0;\n\n    // bias" in s else None
    if "unsigned long long W0" not in s:
        rep("    stage_barrier(inflight);\n    int sbuf = 0;\n", "    stage_barrier(inflight);\n    int sbuf = 0;\n    unsigned long long W0 = 0, W1 = 0, W2 = 0, W3 = 0, W4 = 0, W5 = 0, W6 = 0, W7 = 0, W8 = 0, Ta, Tb, Tc, Td; const unsigned long long Tk0 = clock64();\n")
    rep("            const bool more = c + 1 < p.nchunks;\n            if (more) load_weights(c + 1, sbuf ^ 1);", f"            const bool more = c + 1 < p.nchunks;\n            {SB} Ta = clock64(); {SB}\n            if (more) load_weights(c + 1, sbuf ^ 1);")
    rep("            const char* s = smem + sbuf * STAGE_BYTES;\n            // fragment reads run one tap ahead", f"            {SB} Tc = clock64(); W7 += Tc - Ta; {SB}\n            const char* s = smem + sbuf * STAGE_BYTES;\n            // fragment reads run one tap ahead")

    rep("            load_frag(0, 0);\n            __builtin_amdgcn_s_setprio(0);", f"            load_frag(0, 0);\n            {SB} Td = clock64(); W8 += Td - Tc; {SB}\n            asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n            {SB} Tb = clock64(); W5 += Tb - Ta; {SB}\n            __builtin_amdgcn_s_setprio(0);")
    rep("            __builtin_amdgcn_s_setprio(3);\n            if (more || has_next) store_stage(sbuf ^ 1);\n            inflight = false;",
        f"            {SB} Ta = clock64(); W6 += Ta - Tb; {SB}\n            __builtin_amdgcn_s_setprio(3);\n            asm volatile(\"s_waitcnt vmcnt(%0)\" :: \"n\"(W_ROUNDS) : \"memory\");\n            {SB} Tb = clock64(); W0 += Tb - Ta; {SB}\n            if (more || has_next) store_stage(sbuf ^ 1);\n            {SB} Ta = clock64(); W1 += Ta - Tb; {SB}\n            inflight = false;")
    rep("            stage_barrier(inflight);\n            sbuf ^= 1;\n        }", f"            {SB} Tb = clock64(); W2 += Tb - Ta; {SB}\n            if (inflight) asm volatile(\"s_waitcnt vmcnt(%0) lgkmcnt(0)\" ::\"n\"(IN_ROUNDS) : \"memory\"); else asm volatile(\"s_waitcnt vmcnt(0) lgkmcnt(0)\" ::: \"memory\");\n            {SB} Ta = clock64(); W3 += Ta - Tb; {SB}\n            __builtin_amdgcn_s_barrier();\n            {SB} Tb = clock64(); W4 += Tb - Ta; {SB}\n            sbuf ^= 1;\n        }}")
    rep("        tn = tile_index(k + 1);\n        if (tn >= 0) setup_tile(tn, nxt);\n    }\n}", '''        tn = tile_index(k + 1);
        if (tn >= 0) setup_tile(tn, nxt);
    }
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 4 + wv) * 8;
        d[0] = W0; d[1] = W1; d[2] = W2; d[3] = W3; d[4] = W4; d[5] = W5; d[6] = W6; d[7] = W7 | (W8 << 32);
    }
}''')
if VAR == 'tall':
    # 8-wave kernel, per-wave cycle sums: [0] staging block, [1] stage-end wait (vmcnt/lgkm), [2] s_barrier, [3] whole chunk, [4] epilogue, [5] first-frag wait
    rep("    int tiles_x, tiles_y;\n", "    int tiles_x, tiles_y;\n    unsigned long long* dbg;\n")
    rep("    k.tiles_y = (d->h + TILE - 1) / TILE;\n", "    k.tiles_y = (d->h + TILE - 1) / TILE;\n    k.dbg = g_dbg;\n")
    rep('thread_local char g_err[256] = "";', 'thread_local char g_err[256] = "";\nunsigned long long* g_dbg = nullptr;')
    dbg_api = 'void esr_set_dbg(void* p) { g_dbg = (unsigned long long*)p; }'
    rep("    stage_barrier(inflight);\n    int sbuf = 0;\n", "    stage_barrier(inflight);\n    int sbuf = 0;\n    unsigned long long W0 = 0, W1 = 0, W2 = 0, W3 = 0, W4 = 0, W5 = 0, Ta, Tb, Tc;\n")
    rep("            if (STAGGER) {\n                if (!late) staging_block();\n", f"            {SB} Tc = clock64(); {SB}\n            if (STAGGER) {{\n                if (!late) {{ staging_block(); {SB} Ta = clock64(); W0 += Ta - Tc; {SB} }}\n")
    rep("                    staging_block();\n                    __builtin_amdgcn_s_setprio(0);", f"                    {SB} Ta = clock64(); {SB}\n                    staging_block();\n                    {SB} Tb = clock64(); W0 += Tb - Ta; {SB}\n                    __builtin_amdgcn_s_setprio(0);")
    rep("            load_frag(0, 0);\n            __builtin_amdgcn_s_setprio(0);", f"            {SB} Ta = clock64(); {SB}\n            load_frag(0, 0);\n            asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n            {SB} Tb = clock64(); W5 += Tb - Ta; {SB}\n            __builtin_amdgcn_s_setprio(0);")
    rep("            stage_barrier(inflight);\n            sbuf ^= 1;\n        }", f"            {SB} Ta = clock64(); {SB}\n            if (inflight) asm volatile(\"s_waitcnt vmcnt(%0) lgkmcnt(0)\" ::\"n\"(IN_ROUNDS) : \"memory\"); else asm volatile(\"s_waitcnt vmcnt(0) lgkmcnt(0)\" ::: \"memory\");\n            {SB} Tb = clock64(); W1 += Tb - Ta; {SB}\n            __builtin_amdgcn_s_barrier();\n            {SB} Ta = clock64(); W2 += Ta - Tb; W3 += Ta - Tc; {SB}\n            sbuf ^= 1;\n        }}")
    rep("        else epilogue_nhwc<NT>(p, acc, scr, cur.n, cur.x0, cur.y0, wv, lane, TILE_H);\n", f"        else epilogue_nhwc<NT>(p, acc, scr, cur.n, cur.x0, cur.y0, wv, lane, TILE_H);\n        {SB} Tb = clock64(); W4 += Tb - Ta; {SB}\n")
    rep("        tn = tile_index(k + 1);\n        if (tn >= 0) setup_tile(tn, nxt);\n    }\n}", """        tn = tile_index(k + 1);
        if (tn >= 0) setup_tile(tn, nxt);
    }
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wv) * 8;
        d[0] = W0; d[1] = W1; d[2] = W2; d[3] = W3; d[4] = W4; d[5] = W5;
    }
}""")
rep('int esr_abi_version(void) { return ESR_ABI_VERSION; }', 'int esr_abi_version(void) { return ESR_ABI_VERSION; }\n' + dbg_api)
src = '/tmp/esr_dbg.hip'
open(src, 'w').write(s)
csrc = os.path.join(R, 'ntire2022_esr_amd/csrc')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
                       '-I', os.path.join(R, 'include'), '-I', csrc, '-o', os.path.join(R, f'tools/dbg/libesr_dbg_{VAR}.so'),
                       src, os.path.join(csrc, 'esr_esa.hip'), os.path.join(csrc, 'esr_bsconv.hip')])
print('built', VAR)
