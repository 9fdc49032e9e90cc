"""Generate + build the s_memtime-instrumented debug variant of csrc/esr_hip.hip -> tools/dbg/libesr_dbg.so"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(R, 'ntire2022_esr_amd/csrc/esr_hip.hip')).read()
def rep(a, b):
    global s
    assert a in s, a[:60]
    s = s.replace(a, b)
rep("    int tiles_x, tiles_y;\n};", "    int tiles_x, tiles_y;\n    unsigned long long* dbg;\n};")
rep('''    for (int c = 0; c < p.nchunks; ++c) {
        const bool more = c + 1 < p.nchunks;
        if (more) load_stage(c + 1);''', '''    unsigned long long T[8][6];
    const unsigned long long Tstart = clock64();
    for (int c = 0; c < p.nchunks; ++c) {
        const bool more = c + 1 < p.nchunks;
        T[c&7][0] = clock64();
        if (more) load_stage(c + 1);
        T[c&7][1] = clock64();''')
rep('''        load_frag(0, 0);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {''', '''        load_frag(0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        T[c&7][2] = clock64();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {''')
rep('''        __builtin_amdgcn_s_setprio(3);
        if (more) store_stage((c + 1) & 1);
        __syncthreads();
    }
''', '''        __builtin_amdgcn_sched_barrier(0);
        T[c&7][3] = clock64();
        __builtin_amdgcn_s_setprio(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        T[c&7][4] = clock64();
        if (more) store_stage((c + 1) & 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        T[c&7][5] = clock64();
        __syncthreads();
    }
    const unsigned long long Tend = clock64();
''')
rep('''    // ---- epilogue: bias -> (+res) -> act -> (+res) -> store''', '''    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 4 + wv) * 64;
        d[0] = Tstart0; d[1] = Tstart; d[2] = Tend;
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        d[4] = xcc; d[5] = hwid;
        for (int c = 0; c < 8; ++c) for (int k = 0; k < 6; ++k) d[8 + c * 6 + k] = T[c][k];
    }
    // ---- epilogue: bias -> (+res) -> act -> (+res) -> store''')
# stamp the very end of the kernel (after the epilogue stores are issued)
idx = s.index('thread_local char g_err[256]')
kend = s.rfind('}\n\n', 0, idx)   # end of anonymous-namespace kernel template region is before; find kernel closing
rep('''                *reinterpret_cast<f32x4*>(p.y1 + pix * p.y1_pitch + p.y1_coff + (cb - p.split)) = v;
            }
        }
    }
}''', '''                *reinterpret_cast<f32x4*>(p.y1 + pix * p.y1_pitch + p.y1_coff + (cb - p.split)) = v;
            }
        }
    }
    if (p.dbg && lane == 0) p.dbg[((size_t)blockIdx.x * 4 + wv) * 64 + 3] = clock64();
}''')
rep('''    const int tid = threadIdx.x;
    const int lane = tid & 63;''', '''    const unsigned long long Tstart0 = clock64();
    const int tid = threadIdx.x;
    const int lane = tid & 63;''')
rep("    k.tiles_y = (d->h + TILE - 1) / TILE;\n", "    k.tiles_y = (d->h + TILE - 1) / TILE;\n    k.dbg = g_dbg;\n")
rep('thread_local char g_err[256] = "";', 'thread_local char g_err[256] = "";\nunsigned long long* g_dbg = nullptr;')
rep('int esr_abi_version(void) { return ESR_ABI_VERSION; }', 'int esr_abi_version(void) { return ESR_ABI_VERSION; }\nvoid esr_set_dbg(void* p) { g_dbg = (unsigned long long*)p; }')
src = '/tmp/esr_dbg.hip'
open(src, 'w').write(s)
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
                       '-I', os.path.join(R, 'include'), '-o', os.path.join(R, 'tools/dbg/libesr_dbg.so'), src])
print('built')
