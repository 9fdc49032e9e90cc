"""Instrumented build of conv_f32_ws_kernel -> tools/dbg/libesr_dbg_ws.so  (s_memtime sums per wave)
consumer wave slots: [0] barrier wait, [1] whole chunk loop (per tile), [2] epilogue, [3] first-frag wait at tile start
loader wave slots:   [0] barrier wait, [1] write (vmcnt wait + ds_write), [2] request issue
usage: make_ws_dbg.py [variant]   variants: probe | noepi | nobar"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(R, 'ntire2022_esr_amd/csrc/esr_hip.hip')).read()
def rep(a, b, count=1):
    global s
    assert a in s, a[:70]
    s = s.replace(a, b, count)
VAR = sys.argv[1] if len(sys.argv) > 1 else 'probe'
SB = "__builtin_amdgcn_sched_barrier(0);"
rep("    int tiles_x, tiles_y;\n};", "    int tiles_x, tiles_y;\n    unsigned long long* dbg;\n};")
rep("    k.tiles_y = (d->h + TILE - 1) / TILE;\n", "    k.tiles_y = (d->h + TILE - 1) / TILE;\n    k.dbg = g_dbg;\n")
rep('thread_local char g_err[256] = "";', 'thread_local char g_err[256] = "";\nunsigned long long* g_dbg = nullptr;')
rep('int esr_abi_version(void) { return ESR_ABI_VERSION; }', 'int esr_abi_version(void) { return ESR_ABI_VERSION; }\nvoid esr_set_dbg(void* p) { g_dbg = (unsigned long long*)p; }')
if VAR == 'probe':
    # loader
    rep("        int issued = 0, wbuf = 0;\n", "        int issued = 0, wbuf = 0;\n        unsigned long long W0 = 0, W1 = 0, W2 = 0, Ta, Tb;\n")
    rep("""            if (pending) {
                write(wbuf);
                pending = false;
                if (issued < nstages) { request(); ++issued; pending = true; }
            }
            wbuf = wbuf + 1 == WS_STAGES ? 0 : wbuf + 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;""", f"""            {SB} Ta = clock64(); {SB}
            if (pending) {{
                write(wbuf);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                {SB} Tb = clock64(); W1 += Tb - Ta; {SB}
                pending = false;
                if (issued < nstages) {{ request(); ++issued; pending = true; }}
                {SB} Ta = clock64(); W2 += Ta - Tb; {SB}
            }}
            wbuf = wbuf + 1 == WS_STAGES ? 0 : wbuf + 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {SB} Ta = clock64(); {SB}
            __builtin_amdgcn_s_barrier();
            {SB} Tb = clock64(); W0 += Tb - Ta; {SB}
        }}
        if (p.dbg && lane == 0) {{ unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wv) * 4; d[0] = W0; d[1] = W1; d[2] = W2; }}
        return;""")
    # consumer
    rep("    __syncthreads();\n    int rbuf = 0;\n", "    __syncthreads();\n    int rbuf = 0;\n    unsigned long long W0 = 0, W1 = 0, W2 = 0, W3 = 0, Ta, Tb, Tc;\n")
    rep("            __builtin_amdgcn_s_barrier();\n            rbuf = nbuf;", f"            {SB} Ta = clock64(); {SB}\n            __builtin_amdgcn_s_barrier();\n            {SB} Tb = clock64(); W0 += Tb - Ta; {SB}\n            rbuf = nbuf;")
    rep("        if (WS_STAGES >= 3) {\n            load_frag(smem + rbuf * STAGE_BYTES, 0, 0);", f"        {SB} Tc = clock64(); {SB}\n        if (WS_STAGES >= 3) {{\n            load_frag(smem + rbuf * STAGE_BYTES, 0, 0);")
    rep("""        if (p.out_layout == ESR_NCHW_SHUFFLE4) epilogue_shuffle<NT>(p, acc, n, tx * TILE, ty * TILE, wv, lane);
        else epilogue_nhwc<NT>(p, acc, scr, n, tx * TILE, ty * TILE, wv, lane);
    }
}""", f"""        {SB} Ta = clock64(); W1 += Ta - Tc; {SB}
        if (p.out_layout == ESR_NCHW_SHUFFLE4) epilogue_shuffle<NT>(p, acc, n, tx * TILE, ty * TILE, wv, lane);
        else epilogue_nhwc<NT>(p, acc, scr, n, tx * TILE, ty * TILE, wv, lane);
        {SB} Tb = clock64(); W2 += Tb - Ta; {SB}
    }}
    if (p.dbg && lane == 0) {{ unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wv) * 4; d[0] = W0; d[1] = W1; d[2] = W2; d[3] = W3; }}
}}""")
elif VAR in ('solo', 'solo_noepi'):
    # loaders leave at once, consumers never wait at a barrier: the bare fragment + MFMA loop with one wave per SIMD
    rep("    if (wv >= 4) {\n", "    if (wv >= 4) return;\n    if (false) {\n")
    rep("    __syncthreads();\n    int rbuf = 0;\n", "    int rbuf = 0;\n")
    rep("            __builtin_amdgcn_s_barrier();\n            rbuf = nbuf;", "            rbuf = nbuf;")
    if VAR == 'solo_noepi':
        rep("""        if (p.out_layout == ESR_NCHW_SHUFFLE4) epilogue_shuffle<NT>(p, acc, n, tx * TILE, ty * TILE, wv, lane);
        else epilogue_nhwc<NT>(p, acc, scr, n, tx * TILE, ty * TILE, wv, lane);
    }
}""", """        for (int tt = 0; tt < NT; ++tt) for (int r = 0; r < 4; ++r) asm volatile("" :: "v"(acc[tt][r]));
    }
}""")
elif VAR == 'noepi':
    rep("""        if (p.out_layout == ESR_NCHW_SHUFFLE4) epilogue_shuffle<NT>(p, acc, n, tx * TILE, ty * TILE, wv, lane);
        else epilogue_nhwc<NT>(p, acc, scr, n, tx * TILE, ty * TILE, wv, lane);
    }
}""", """        for (int tt = 0; tt < NT; ++tt) for (int r = 0; r < 4; ++r) asm volatile("" :: "v"(acc[tt][r]));
    }
}""")
src = '/tmp/esr_ws_dbg.hip'
open(src, 'w').write(s)
csrc = os.path.join(R, 'ntire2022_esr_amd/csrc')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
                       '-I', os.path.join(R, 'include'), '-I', csrc, '-o', os.path.join(R, f'tools/dbg/libesr_dbg_ws{"" if VAR == "probe" else "_" + VAR}.so'),
                       src, os.path.join(csrc, 'esr_esa.hip')])
print('built', VAR)
