"""Generate + build a LIGHTLY instrumented variant of csrc/esr_hip.hip -> tools/dbg/libesr_dbg.so
(4 s_memtime stamps per wave: kernel start, K-loop start, K-loop end, kernel end + HW_ID)."""
import os, subprocess
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(R, 'ntire2022_esr_amd/csrc/esr_hip.hip')).read()
def rep(a, b):
    global s
    assert a in s, a[:60]
    s = s.replace(a, b)
rep("    int tiles_x, tiles_y;\n};", "    int tiles_x, tiles_y;\n    unsigned long long* dbg;\n};")
rep('''    for (int c = 0; c < p.nchunks; ++c) {
        const bool more = c + 1 < p.nchunks;''', '''    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long Tstart = clock64();
    __builtin_amdgcn_sched_barrier(0);
    for (int c = 0; c < p.nchunks; ++c) {
        const bool more = c + 1 < p.nchunks;''')
rep('''    switch (p.act) {
        case ESR_ACT_LRELU: epilogue<ESR_ACT_LRELU, NT>''', '''    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long Tend = clock64();
    __builtin_amdgcn_sched_barrier(0);
    switch (p.act) {
        case ESR_ACT_LRELU: epilogue<ESR_ACT_LRELU, NT>''')
rep('''        default: epilogue<ESR_ACT_NONE, NT>(p, acc, n, x0, y0, wv, px, kq); break;
    }
}''', '''        default: epilogue<ESR_ACT_NONE, NT>(p, acc, n, x0, y0, wv, px, kq); break;
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long Tx = clock64();
    __builtin_amdgcn_sched_barrier(0);
    if (p.dbg && lane == 0) {
        unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        ulonglong4 a = {Tstart0, Tstart, Tend, Tx};
        ulonglong2 b = {xcc, hwid};
        unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 4 + wv) * 8;
        *reinterpret_cast<ulonglong4*>(d) = a;
        *reinterpret_cast<ulonglong2*>(d + 4) = b;
    }
}''')
rep('''    __builtin_amdgcn_s_setprio(3);

    const int tid = threadIdx.x;''', '''    __builtin_amdgcn_s_setprio(3);
    const unsigned long long Tstart0 = clock64();
    __builtin_amdgcn_sched_barrier(0);
    const int tid = threadIdx.x;''')
rep("    k.tiles_y = (d->h + TILE - 1) / TILE;\n", "    k.tiles_y = (d->h + TILE - 1) / TILE;\n    k.dbg = g_dbg;\n")
rep('thread_local char g_err[256] = "";', 'thread_local char g_err[256] = "";\nunsigned long long* g_dbg = nullptr;')
rep('int esr_abi_version(void) { return ESR_ABI_VERSION; }', 'int esr_abi_version(void) { return ESR_ABI_VERSION; }\nvoid esr_set_dbg(void* p) { g_dbg = (unsigned long long*)p; }')
import sys
if len(sys.argv)>1 and sys.argv[1]=='small':
    rep('''        case ESR_ACT_RELU: epilogue<ESR_ACT_RELU, NT>(p, acc, n, x0, y0, wv, px, kq); break;
        case ESR_ACT_GELU: epilogue<ESR_ACT_GELU, NT>(p, acc, n, x0, y0, wv, px, kq); break;
''','')
if len(sys.argv)>1 and sys.argv[1]=='noepi':
    rep('''    const int gx = x0 + px;
    if (gx >= p.W) return;
    const bool shuffle''','''    const int gx = x0 + px;
    for (int t = 0; t < NT; ++t) for (int r = 0; r < 4; ++r) asm volatile("" :: "v"(acc[t][r]));
    if (gx >= 0) return;
    const bool shuffle''')
if len(sys.argv)>1 and sys.argv[1]=='epistamps':
    rep('''template <int ACT, int NT>
__device__ __forceinline__ void epilogue(const ConvK& p, f32x4 (&acc)[NT][4], int n, int x0, int y0, int wv,
                                         int px, int kq)
{''','''template <int ACT, int NT>
__device__ __forceinline__ void epilogue(const ConvK& p, f32x4 (&acc)[NT][4], int n, int x0, int y0, int wv,
                                         int px, int kq)
{
    unsigned long long E[5];
    __builtin_amdgcn_sched_barrier(0); E[0] = clock64(); __builtin_amdgcn_sched_barrier(0);''')
    rep('''                *reinterpret_cast<f32x4*>(p.y1 + (size_t)(pix * p.y1_pitch + p.y1_coff + (cb - p.split))) = v;
            }
        }
    }
}''','''                *reinterpret_cast<f32x4*>(p.y1 + (size_t)(pix * p.y1_pitch + p.y1_coff + (cb - p.split))) = v;
            }
        }
        __builtin_amdgcn_sched_barrier(0); E[r + 1] = clock64(); __builtin_amdgcn_sched_barrier(0);
    }
    if (p.dbg && (threadIdx.x & 63) == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 4 + wv) * 8 + 6;
        d[0] = ((E[1]-E[0]) & 0xffff) | (((E[2]-E[1]) & 0xffff) << 16) | (((E[3]-E[2]) & 0xffff) << 32) | (((E[4]-E[3]) & 0xffff) << 48);
    }
}''')
if len(sys.argv)>1 and sys.argv[1]=='nostore':
    rep('''                *reinterpret_cast<f32x4*>(dst) = v;
            } else if (cb < p.split) {
                *reinterpret_cast<f32x4*>(p.y0 + (size_t)(pix * p.y0_pitch + p.y0_coff + cb)) = v;
            } else {
                *reinterpret_cast<f32x4*>(p.y1 + (size_t)(pix * p.y1_pitch + p.y1_coff + (cb - p.split))) = v;
            }''','''                asm volatile("" :: "v"(dst), "v"(v));
            } else {
                asm volatile("" :: "v"(v), "v"(pix));
            }''')
src = '/tmp/esr_dbg.hip'
open(src, 'w').write(s)
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
                       '-I', os.path.join(R, 'include'), '-o', os.path.join(R, 'tools/dbg/libesr_dbg.so'), src])
print('built')
