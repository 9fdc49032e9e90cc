"""Generate + build a LIGHTLY instrumented variant of csrc/esr_hip.hip -> tools/dbg/libesr_dbg.so.
Per wave: kernel start/end, cycles summed over its tiles for {K loop, epilogue}, first-tile prologue, HW_ID."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(R, 'ntire2022_esr_amd/csrc/esr_hip.hip')).read()
def rep(a, b):
    global s
    assert a in s, a[:70]
    s = s.replace(a, b)
SB = "__builtin_amdgcn_sched_barrier(0);"
rep("    int tiles_x, tiles_y;\n};", "    int tiles_x, tiles_y;\n    unsigned long long* dbg;\n};")
rep("    const int tid = threadIdx.x;\n    const int lane = tid & 63;\n    const int wv = tid >> 6;\n    const int px = lane & 15;\n    const int kq = lane >> 4;\n    float* const scr",
    f"    const unsigned long long T0 = clock64(); {SB}\n    unsigned long long Tloop = 0, Tepi = 0, Tpro = 0; int ntile = 0;\n    const int tid = threadIdx.x;\n    const int lane = tid & 63;\n    const int wv = tid >> 6;\n    const int px = lane & 15;\n    const int kq = lane >> 4;\n    float* const scr")
rep("    int sbuf = 0;\n\n    for (;;) {", f"    int sbuf = 0;\n    {SB} Tpro = clock64() - T0; {SB}\n\n    for (;;) {{")
rep("        for (int c = 0; c < p.nchunks; ++c) {\n            const bool more", f"        {SB} const unsigned long long Ta = clock64(); {SB}\n        for (int c = 0; c < p.nchunks; ++c) {{\n            const bool more")
rep("        if (p.out_layout == ESR_NCHW_SHUFFLE4) epilogue_shuffle<NT>(p, acc, cur.n", f"        {SB} const unsigned long long Tb = clock64(); {SB}\n        if (p.out_layout == ESR_NCHW_SHUFFLE4) epilogue_shuffle<NT>(p, acc, cur.n")
rep("        if (!has_next) break;\n        cur = nxt;", f"        {SB} const unsigned long long Tc = clock64(); {SB}\n        Tloop += Tb - Ta; Tepi += Tc - Tb; ++ntile;\n        if (!has_next) break;\n        cur = nxt;")
rep("        cur = nxt;\n        ++k;\n    }\n}", '''        cur = nxt;
        ++k;
    }
    if (p.dbg && lane == 0) {
        unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        ulonglong4 a = {T0, clock64(), Tloop, Tepi};
        ulonglong4 b = {xcc, hwid, Tpro, (unsigned long long)ntile};
        unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 4 + wv) * 8;
        *reinterpret_cast<ulonglong4*>(d) = a;
        *reinterpret_cast<ulonglong4*>(d + 4) = b;
    }
}''')
rep("    k.tiles_y = (d->h + TILE - 1) / TILE;\n", "    k.tiles_y = (d->h + TILE - 1) / TILE;\n    k.dbg = g_dbg;\n")
rep('thread_local char g_err[256] = "";', 'thread_local char g_err[256] = "";\nunsigned long long* g_dbg = nullptr;')
rep('int esr_abi_version(void) { return ESR_ABI_VERSION; }', 'int esr_abi_version(void) { return ESR_ABI_VERSION; }\nvoid esr_set_dbg(void* p) { g_dbg = (unsigned long long*)p; }')
VAR = sys.argv[2] if len(sys.argv) > 2 else ''
if VAR == 'nostore':
    rep('''            if (!has_split) {
                if (NT == 4 || to0) *reinterpret_cast<f32x4*>(p.y0 + pu * p.y0_pitch + off0) = o;
            } else {
                if (to0) *reinterpret_cast<f32x4*>(p.y0 + pu * p.y0_pitch + off0) = o;
                if (to1) *reinterpret_cast<f32x4*>(p.y1 + pu * p.y1_pitch + off1) = o;
            }''','''            asm volatile("" :: "v"(o), "s"(pu), "v"(off0), "v"(off1));
            {''')
if VAR == 'nolds':
    rep('''        for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(scr + px * EPI_PITCH + t * 16 + kq * 4) = acc[t][r];
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(scr + (4 * i + prow) * EPI_PITCH + rd);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 o;
            if (RES == ESR_RES_PRE_ACT)''','''        for (int t = 0; t < 1; ++t) {}
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[i % NT][r];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 o;
            if (RES == ESR_RES_PRE_ACT)''')
if len(sys.argv) > 1 and sys.argv[1] == 'epi':
    rep('''    const int rd = (NT == 4) ? cb : min(cb, NT * 16 - 4);

    f32x4 rv[4][4];''', '''    const int rd = (NT == 4) ? cb : min(cb, NT * 16 - 4);
    unsigned long long E[6];
    __builtin_amdgcn_sched_barrier(0); E[0] = clock64(); __builtin_amdgcn_sched_barrier(0);
    f32x4 rv[4][4];''')
    rep('''            }
        }
    }
}

template <int ACT, int NT>
__device__ __forceinline__ void epilogue_nhwc_fast_res(''', '''            }
        }
        __builtin_amdgcn_sched_barrier(0); E[r + 1] = clock64(); __builtin_amdgcn_sched_barrier(0);
    }
    if (p.dbg && lane == 0 && p.dbg[((size_t)blockIdx.x * 4 + wv) * 8 + 512 * 4 * 8] == 0)
        p.dbg[((size_t)blockIdx.x * 4 + wv) * 8 + 512 * 4 * 8] = ((E[1]-E[0]) & 0xffff) | (((E[2]-E[1]) & 0xffff) << 16) | (((E[3]-E[2]) & 0xffff) << 32) | (((E[4]-E[3]) & 0xffff) << 48);
}

template <int ACT, int NT>
__device__ __forceinline__ void epilogue_nhwc_fast_res(''')
src = '/tmp/esr_dbg.hip'
open(src, 'w').write(s)
csrc = os.path.join(R, 'ntire2022_esr_amd/csrc')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
                       '-I', os.path.join(R, 'include'), '-I', csrc, '-o', os.path.join(R, 'tools/dbg/libesr_dbg.so'),
                       src, os.path.join(csrc, 'esr_esa.hip')])
print('built')
