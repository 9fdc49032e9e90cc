"""Build an ablated / instrumented variant of csrc/esr_hip.hip -> tools/dbg/libesr_dbg.so
usage: make_dbg.py [plain|noepi|nostore|nolds]"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(R, 'ntire2022_esr_amd/csrc/esr_hip.hip')).read()
def rep(a, b):
    global s
    assert a in s, a[:70]
    s = s.replace(a, b)
VAR = sys.argv[1] if len(sys.argv) > 1 else 'plain'
NOEPI = '''        for (int tt = 0; tt < NT; ++tt) for (int r = 0; r < 4; ++r) asm volatile("" :: "v"(acc[tt][r]));'''
EPI = '''        if (p.out_layout == ESR_NCHW_SHUFFLE4) epilogue_shuffle<NT>(p, acc, cur.n, cur.x0, cur.y0, wv, lane);
        else epilogue_nhwc<NT>(p, acc, scr, cur.n, cur.x0, cur.y0, wv, lane);'''
if VAR in ('nostage', 'nobarrier'):
    rep('''            if (more) {
                load_stage(cur, c + 1);
            } else if (has_next) {          // cross-tile prefetch: next tile's first stage under this tile's last chunk
                setup_tile(tn, nxt);
                load_stage(nxt, 0);
            }''', '''            if (!more && has_next) setup_tile(tn, nxt);''')
    rep('''            if (more || has_next) store_stage(sbuf ^ 1);
            __syncthreads();
            sbuf ^= 1;
        }

        if (p.out_layout''', '''            %s
            sbuf ^= 1;
        }

        if (p.out_layout''' % ('__syncthreads();' if VAR == 'nostage' else ''))
if VAR == 'nowrite':
    rep('''                *reinterpret_cast<f32x4*>(s + (idx & 1) * (NPX * 16) + (idx >> 1) * 16) = in_reg[r];''', '''                asm volatile("" :: "v"(in_reg[r]), "v"(s));''')
    rep('''                *reinterpret_cast<f32x4*>(s + IN_BYTES + idx * 16) = w_reg[r];''', '''                asm volatile("" :: "v"(w_reg[r]), "v"(s));''')
    rep(EPI, NOEPI)
if VAR == 'noload':
    rep('''            if (more || has_next) store_stage(sbuf ^ 1);
            __syncthreads();
            sbuf ^= 1;
        }

        if (p.out_layout''', '''            __syncthreads();
            sbuf ^= 1;
        }

        if (p.out_layout''')
if VAR == 'noepi':
    rep('''        if (p.out_layout == ESR_NCHW_SHUFFLE4) epilogue_shuffle<NT>(p, acc, cur.n, cur.x0, cur.y0, wv, lane);
        else epilogue_nhwc<NT>(p, acc, scr, cur.n, cur.x0, cur.y0, wv, lane);''',
        '''        for (int tt = 0; tt < NT; ++tt) for (int r = 0; r < 4; ++r) asm volatile("" :: "v"(acc[tt][r]));''')
if VAR == 'nostore':
    rep('''            if (!has_split) {
                if (NT == 4 || to0) *reinterpret_cast<f32x4*>(p.y0 + pu * p.y0_pitch + off0) = o;
            } else {
                if (to0) *reinterpret_cast<f32x4*>(p.y0 + pu * p.y0_pitch + off0) = o;
                if (to1) *reinterpret_cast<f32x4*>(p.y1 + pu * p.y1_pitch + off1) = o;
            }''', '''            asm volatile("" :: "v"(o), "s"(pu), "v"(off0), "v"(off1), "s"(has_split), "v"(to0 ? 1 : 0), "v"(to1 ? 1 : 0));''')
if VAR == 'nolds':
    rep('''        for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(scr + px * EPI_PITCH + t * 16 + kq * 4) = acc[t][r];
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(scr + (4 * i + prow) * EPI_PITCH + rd);''',
        '''        for (int t = 0; t < 1; ++t) {}
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[i % NT][r];''')
if VAR in ('nostage', 'nobarrier', 'noload'):
    rep(EPI, NOEPI)
rep('int esr_abi_version(void) { return ESR_ABI_VERSION; }', 'int esr_abi_version(void) { return ESR_ABI_VERSION; }\nvoid esr_set_dbg(void* p) { (void)p; }')
src = '/tmp/esr_dbg.hip'
open(src, 'w').write(s)
csrc = os.path.join(R, 'ntire2022_esr_amd/csrc')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
                       '-I', os.path.join(R, 'include'), '-I', csrc, '-o', os.path.join(R, f'tools/dbg/libesr_dbg_{VAR}.so'),
                       src, os.path.join(csrc, 'esr_esa.hip')])
print('built', VAR)
