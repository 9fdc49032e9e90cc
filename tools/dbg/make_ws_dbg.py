"""Instrumented build of conv_f32_ws_kernel -> tools/dbg/libesr_dbg_ws.so  (s_memtime sums per wave)
consumer wave slots: [0] barrier wait, [1] whole chunk loop (per tile), [2] epilogue, [3] first-frag wait at tile start
loader wave slots:   [0] barrier wait, [1] write (vmcnt wait + ds_write), [2] request issue
usage: make_ws_dbg.py [variant]   variants: plain | probe | noepi | solo | solo_noepi | nowrite_noepi | noreq_noepi | noboth_noepi"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(R, 'ntire2022_esr_amd/csrc/esr_hip.hip')).read()
# the product source carries no research hooks: put them back into this private copy (tools/dbg/ws_hooks.py)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ws_hooks
WS_INC = open(os.path.join(R, 'tools/dbg/conv_ws.inc')).read()
for anchor, text in ws_hooks.HOOKS:
    assert anchor in s, anchor[:60]
    if text is None:                       # the launcher goes in FRONT of launch_conv_nt
        s = s.replace(anchor, ws_hooks.LAUNCHER + "\n" + anchor, 1)
    else:
        s = s.replace(anchor, anchor + text.replace('@@CONV_WS_INC@@', WS_INC), 1)
def rep(a, b, count=1):
    global s
    assert a in s, a[:70]
    s = s.replace(a, b, count)
VAR = sys.argv[1] if len(sys.argv) > 1 else 'probe'
SB = "__builtin_amdgcn_sched_barrier(0);"
rep("    int tiles_x, tiles_y;\n", "    int tiles_x, tiles_y;\n    unsigned long long* dbg;\n")
rep("    k.tiles_y = (d->h + TILE - 1) / TILE;\n", "    k.tiles_y = (d->h + TILE - 1) / TILE;\n    k.dbg = g_dbg;\n")
rep('thread_local char g_err[256] = "";', 'thread_local char g_err[256] = "";\nunsigned long long* g_dbg = nullptr;')
rep('int esr_abi_version(void) { return ESR_ABI_VERSION; }', 'int esr_abi_version(void) { return ESR_ABI_VERSION; }\nvoid esr_set_dbg(void* p) { g_dbg = (unsigned long long*)p; }')
if VAR == 'probe':
    # loader
    rep("        int wbuf = 0;\n        for (int s = 0; s < nstages; ++s) {", "        int wbuf = 0;\n        unsigned long long W0 = 0, W1 = 0, W2 = 0, Ta, Tb;\n        for (int s = 0; s < nstages; ++s) {\n            " + SB + " Ta = clock64(); " + SB)
    rep("            write(wbuf);\n            bump(cw);\n            if (s + 1 < nstages) request();\n",
        f"            {SB} Tb = clock64(); W0 += Tb - Ta; {SB}\n            write(wbuf);\n            bump(cw);\n            asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n            {SB} Ta = clock64(); W1 += Ta - Tb; {SB}\n            if (s + 1 < nstages) request();\n            {SB} Tb = clock64(); W2 += Tb - Ta; {SB}\n")
    # consumer
    rep("    int rbuf = 0;\n    int s = 0; ", "    unsigned long long W0 = 0, W1 = 0, W2 = 0, W3 = 0, Ta, Tb, Tc;\n    int rbuf = 0;\n    int s = 0; ")
    rep("            if (s < nstages) wait_ready(seen, s + 2 <= nstages ? s + 2 : nstages);", f"            {SB} Ta = clock64(); {SB}\n            if (s < nstages) wait_ready(seen, s + 2 <= nstages ? s + 2 : nstages);\n            {SB} Tb = clock64(); W0 += Tb - Ta; {SB}")
    rep("        for (int c = 0; c < p.nchunks; ++c) {\n            if (s & 1) chunk(", f"        {SB} Tc = clock64(); {SB}\n        for (int c = 0; c < p.nchunks; ++c) {{\n            if (s & 1) chunk(")
    rep("        // the scratch rows may still hold the previous handed-over tile\n", f"        {SB} Ta = clock64(); W1 += Ta - Tc; {SB}\n")
    rep("        while (counter(12 + wv) < handed) __builtin_amdgcn_s_sleep(1);\n        asm volatile(\"\" ::: \"memory\");\n", f"        while (counter(12 + wv) < handed) __builtin_amdgcn_s_sleep(1);\n        asm volatile(\"\" ::: \"memory\");\n        {SB} Tb = clock64(); W3 += Tb - Ta; {SB}\n")
    rep("            epilogue_nhwc_checked<NT>(p, acc, scr, n, tx * TILE, ty * TILE, wv, lane);\n        }\n    }\n}\n", f"            epilogue_nhwc_checked<NT>(p, acc, scr, n, tx * TILE, ty * TILE, wv, lane);\n        }}\n        asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n        {SB} Tb = clock64(); W2 += Tb - Ta; {SB}\n    }}\n    if (p.dbg && lane == 0) {{ unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wv) * 4; d[0] = W0; d[1] = W1; d[2] = W2; d[3] = W3; }}\n}}\n")
    rep("            drain(false);\n        }\n        while (ek < my_tiles) drain(true);\n        return;", f"            {SB} Ta = clock64(); {SB}\n            drain(false);\n            {SB} Tb = clock64(); W3 += Tb - Ta; {SB}\n        }}\n        while (ek < my_tiles) drain(true);\n        if (p.dbg && lane == 0) {{ unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wv) * 4; d[0] = W0; d[1] = W1; d[2] = W2; d[3] = W3; }}\n        return;")
    rep("unsigned long long W0 = 0, W1 = 0, W2 = 0, Ta, Tb;\n        for (int s = 0; s < nstages; ++s) {", "unsigned long long W0 = 0, W1 = 0, W2 = 0, W3 = 0, Ta, Tb;\n        for (int s = 0; s < nstages; ++s) {")
elif VAR in ('solo', 'solo_noepi'):
    # loaders leave at once, consumers never wait: the bare fragment + MFMA loop with one wave per SIMD
    rep("    if (wv >= 4) {\n", "    if (wv >= 4) return;\n    if (false) {\n")
    rep("        while (seen < need) {\n", "        while (false) {\n")
    if VAR == 'solo_noepi':
        rep("        if (epilogue_offloadable<NT>(p, tx * TILE, ty * TILE)) {\n#pragma unroll\n            for (int r = 0; r < 4; ++r)\n#pragma unroll\n                for (int tt = 0; tt < NT; ++tt)\n                    *reinterpret_cast<f32x4*>(scr",
            "        for (int tt = 0; tt < NT; ++tt) for (int r = 0; r < 4; ++r) asm volatile(\"\" :: \"v\"(acc[tt][r]));\n        if (false) {\n#pragma unroll\n            for (int r = 0; r < 4; ++r)\n#pragma unroll\n                for (int tt = 0; tt < NT; ++tt)\n                    *reinterpret_cast<f32x4*>(scr")
        rep("        } else if (p.out_layout == ESR_NCHW_SHUFFLE4) {\n            epilogue_shuffle<NT>(p, acc, n, tx * TILE, ty * TILE, wv, lane);\n        } else {\n            epilogue_nhwc_checked<NT>(p, acc, scr, n, tx * TILE, ty * TILE, wv, lane);\n        }\n    }\n}\n", "        }\n    }\n}\n")
        rep("        while (ek < my_tiles) drain(true);\n", "")
elif VAR in ('nowrite', 'noreq', 'noboth', 'noboth_noepi', 'nowrite_noepi', 'noreq_noepi', 'noepi'):
    if VAR.startswith('nowrite') or VAR.startswith('noboth'):
        rep("            write(wbuf);\n            bump(cw);", "            for (int r = 0; r < IN_ROUNDS; ++r) asm volatile(\"\" :: \"v\"(in_reg[r]));\n            for (int r = 0; r < W_ROUNDS; ++r) asm volatile(\"\" :: \"v\"(w_reg[r]));\n            bump(cw);")
    if VAR.startswith('noreq') or VAR.startswith('noboth'):
        rep("        if (nstages > 0) request();\n", "        for (int r = 0; r < IN_ROUNDS; ++r) in_reg[r] = f32x4{0.f, 0.f, 0.f, 0.f};\n        for (int r = 0; r < W_ROUNDS; ++r) w_reg[r] = f32x4{0.f, 0.f, 0.f, 0.f};\n")
        rep("            if (s + 1 < nstages) request();\n", "")
    if VAR.endswith('noepi'):
        rep("        if (epilogue_offloadable<NT>(p, tx * TILE, ty * TILE)) {\n#pragma unroll\n            for (int r = 0; r < 4; ++r)\n#pragma unroll\n                for (int tt = 0; tt < NT; ++tt)\n                    *reinterpret_cast<f32x4*>(scr",
            "        for (int tt = 0; tt < NT; ++tt) for (int r = 0; r < 4; ++r) asm volatile(\"\" :: \"v\"(acc[tt][r]));\n        if (false) {\n#pragma unroll\n            for (int r = 0; r < 4; ++r)\n#pragma unroll\n                for (int tt = 0; tt < NT; ++tt)\n                    *reinterpret_cast<f32x4*>(scr")
        rep("        } else if (p.out_layout == ESR_NCHW_SHUFFLE4) {\n            epilogue_shuffle<NT>(p, acc, n, tx * TILE, ty * TILE, wv, lane);\n        } else {\n            epilogue_nhwc_checked<NT>(p, acc, scr, n, tx * TILE, ty * TILE, wv, lane);\n        }\n    }\n}\n", "        }\n    }\n}\n")
        rep("        while (ek < my_tiles) drain(true);\n", "")
src = '/tmp/esr_ws_dbg.hip'
open(src, 'w').write(s)
csrc = os.path.join(R, 'ntire2022_esr_amd/csrc')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-DESR_EXPERIMENTAL_WS',
                       '-I', os.path.join(R, 'include'), '-I', csrc, '-o', os.path.join(R, f'tools/dbg/libesr_dbg_ws{"" if VAR == "probe" else "_" + VAR}.so'),
                       src, os.path.join(csrc, 'esr_esa.hip'), os.path.join(csrc, 'esr_bsconv.hip')])
print('built', VAR)
