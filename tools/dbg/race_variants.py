#!/usr/bin/env python3
"""Variants of libesr_hip.so for the multi-stream race hunt (research tooling):  python tools/dbg/race_variants.py [name ...]
Each variant = patched copies of some translation units, linked with the product's other objects (build/obj) into
tools/abl/libesr_r_<name>.so; run with tools/dbg/streams_race.py <model> <compute> <rounds> <so>."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(REPO, "ntire2022_esr_amd", "csrc"); OBJ = os.path.join(REPO, "build", "obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(REPO, "include"), "-I", SRC]


def sub(s, a, b, cnt=1):
    assert s.count(a) == cnt, (a, s.count(a))
    return s.replace(a, b)


def v_pool3(src):      # s2pool16 at three blocks per CU (LDS padded)
    s = src["esr_esa_lowres.hip"]
    s = sub(s, "    __shared__ __attribute__((aligned(16))) float hb[CT * PT * FP];           // horizontal",
            "    __shared__ __attribute__((aligned(16))) float hb[CT * PT * FP + 3600];           // horizontal")
    s = sub(s, "    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);\n    const int lane = threadIdx.x & 63;\n    const int j = lane & 15, kq = lane >> 4;\n    int t = blockIdx.x;",
            "    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);\n    const int lane = threadIdx.x & 63;\n    const int j = lane & 15, kq = lane >> 4;\n    if (H < 0) hb[CT * PT * FP + threadIdx.x * 14] = 1.f;\n    int t = blockIdx.x;")
    return {"esr_esa_lowres.hip": s}


def v_nw8(src):        # no two-blocks-per-CU conv_s16 shape
    s = src["esr_s16.hip"]
    s = sub(s, "    const int nt = esr_round_up(d->cout, 16) / 16, nchunks = esr_round_up(d->cin, 16) / 16;\n    const bool res_hbm",
            "    return 8;\n    const int nt = esr_round_up(d->cout, 16) / 16, nchunks = esr_round_up(d->cin, 16) / 16;\n    const bool res_hbm")
    return {"esr_s16.hip": s}


def v_tail(src):       # conv_s16: barrier + a pause behind the final vmcnt(0)
    s = src["esr_s16.hip"]
    s = sub(s, '    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the trailing',
            '    asm volatile("s_waitcnt vmcnt(0)\\n\\ts_barrier\\n\\ts_sleep 64" ::: "memory");     // the trailing')
    return {"esr_s16.hip": s}


def v_wait0(src):      # conv_s16: every stage waits for everything (vmcnt(0))
    s = src["esr_s16.hip"]
    s = sub(s, "            wait_vm_dyn((R - 2) * n_my + epi_stores", "            wait_vm_dyn(0 * (R - 2) * n_my + 0 * epi_stores")
    return {"esr_s16.hip": s}


def v_oldpool(src):    # 16-bit plans on the fp32-MFMA s2pool kernel again
    s = src["esr_esa_lowres.hip"]
    s = sub(s, 'hipLaunchKernelGGL(esa_s2pool16_kernel<ESR_STORE_BF16>, grid, dim3(256), 0, st, d->x.ptr, w0, pooled, d->h, d->w, H3, W3, tx, ty)',
            'hipLaunchKernelGGL(esa_s2pool_kernel<ESR_STORE_BF16>, grid, dim3(256), s2_lds<ESR_STORE_BF16>(), st, d->x.ptr, w0, pooled, d->h, d->w, H2, W2, H3, W3, tx, ty)')
    return {"esr_esa_lowres.hip": s}


def v_zero(src):       # s2pool16: out-of-image patch pixels are zeros (as the fp32 kernel) instead of clamped reads
    s = src["esr_esa_lowres.hip"]
    s = sub(s, "        v[i] = *reinterpret_cast<const uint4*>(xb + ((size_t)gy * W + gx) * (FP * 2) + (rem & 1) * 16);",
            "        const bool inb = iy0 + row < H && ix0 + (rem >> 1) < W;\n        v[i] = *reinterpret_cast<const uint4*>(xb + ((size_t)gy * W + gx) * (FP * 2) + (rem & 1) * 16);\n        if (!inb) v[i] = uint4{0u, 0u, 0u, 0u};")
    return {"esr_esa_lowres.hip": s}


def v_check(src):      # every 16-bit s2pool launch is shadowed by the fp32-MFMA kernel on the same input; a compare kernel counts differences
    s = src["esr_esa_lowres.hip"]
    s = sub(s, "}  // namespace\n\nextern \"C\" int esr_esa_lowres_f32", """__device__ unsigned g_dbg_bad[4];
__global__ void dbg_cmp_kernel(const float* a, const float* b, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float d = fabsf(a[i] - b[i]);
    if (!(d <= 1e-3f * fmaxf(1.f, fabsf(a[i])))) { atomicAdd(&g_dbg_bad[0], 1u); atomicMax(&g_dbg_bad[1], (unsigned)i); }
    if (i == 0) atomicAdd(&g_dbg_bad[2], 1u);
}
}  // namespace
extern "C" unsigned esr_dbg_pool_bad(int which) { unsigned h[4] = {0, 0, 0, 0}; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dbg_bad), sizeof(h)); return h[which]; }

extern "C" int esr_esa_lowres_f32""")
    s = sub(s, "    int rc = esr_check_launch(\"esa_s2pool_kernel launch\");", """    if (d->storage == ESR_STORE_BF16) {
        static float* ring[16]; static std::atomic<unsigned> nxt{0};
        const unsigned slot = nxt.fetch_add(1u) & 15u;
        if (!ring[slot]) hipMalloc(&ring[slot], 4 << 20);
        const int nel = d->n * H3 * W3 * FP;
        if ((size_t)nel * 4 <= (4u << 20)) {
            hipLaunchKernelGGL(esa_s2pool_kernel<ESR_STORE_BF16>, grid, dim3(256), s2_lds<ESR_STORE_BF16>(), st, d->x.ptr, w0, ring[slot], d->h, d->w, H2, W2, H3, W3, tx, ty);
            hipLaunchKernelGGL(dbg_cmp_kernel, dim3((nel + 255) / 256), dim3(256), 0, st, (const float*)pooled, (const float*)ring[slot], nel);
        }
    }
    int rc = esr_check_launch("esa_s2pool_kernel launch");""")
    return {"esr_esa_lowres.hip": s}


def v_log(src):        # esr_run_ops: a position-weighted checksum of every op's outputs into a device log [call][op][slot]
    s = src["esr_hip.hip"]
    s = sub(s, "int esr_run_ops(const esr_op* ops, int n_ops, void* hip_stream)\n{\n    if (!ops || n_ops < 0) return ESR_ERR_BAD_ARG;\n    for (int i = 0; i < n_ops; ++i) {\n        const int rc = run_one(ops[i], hip_stream);\n        if (rc != ESR_OK) return rc;\n    }\n    return ESR_OK;\n}",
"""int esr_run_ops(const esr_op* ops, int n_ops, void* hip_stream)
{
    if (!ops || n_ops < 0) return ESR_ERR_BAD_ARG;
    if (!g_dbg_log) { hipMalloc(&g_dbg_log, DBG_CALLS * 64 * 4 * 8); hipMemset(g_dbg_log, 0, DBG_CALLS * 64 * 4 * 8); hipDeviceSynchronize(); }
    const unsigned call = g_dbg_call.fetch_add(1u);
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    for (int i = 0; i < n_ops; ++i) {
        const int rc = run_one(ops[i], hip_stream);
        if (rc != ESR_OK) return rc;
        if (call >= DBG_CALLS || i >= 64) continue;
        const esr_op& o = ops[i];
        const void* ptr[4] = {nullptr, nullptr, nullptr, nullptr}; size_t bytes[4] = {0, 0, 0, 0};
        if (o.kind == ESR_OP_CONV || o.kind == ESR_OP_PACK_INPUT) {
            const esr_conv_desc& c = o.conv; const size_t es = c.storage ? 2 : 4; const size_t px = (size_t)c.n * c.h * c.w;
            if (c.out_layout == ESR_NCHW_SHUFFLE4) { ptr[0] = c.out0.ptr; bytes[0] = px * c.cout * 4; }
            else {
                if (c.out0.ptr) { ptr[0] = c.out0.ptr; bytes[0] = px * c.out0.pitch * es; }
                if (c.split > 0 && c.split < c.cout && c.out1.ptr) { ptr[1] = c.out1.ptr; bytes[1] = px * c.out1.pitch * es; }
                if (c.post_wpacked && c.post_out.ptr) { ptr[2] = c.post_out.ptr; bytes[2] = px * c.post_out.pitch * es; }
                if (c.post2_wpacked && c.post2_out.ptr) { ptr[3] = c.post2_out.ptr; bytes[3] = px * c.post2_out.pitch * es; }
            }
        } else if (o.kind == ESR_OP_ESA_APPLY) {
            const esr_esa_desc& e = o.esa; const size_t es = e.storage ? 2 : 4; const size_t px = (size_t)e.n * e.h * e.w;
            if (!e.skip_y) { ptr[0] = e.y.ptr; bytes[0] = px * e.y.pitch * es; }
            for (int k = 0; k < 2; ++k) if (e.post_w && e.post[k].cout > 0) { ptr[1 + k] = e.post[k].out.ptr; bytes[1 + k] = px * e.post[k].out.pitch * es; }
        } else if (o.kind == ESR_OP_ESA_LOWRES) {
            const esr_esa_lowres_desc& l = o.lo; const int H2 = (l.h - 3) / 2 + 1, W2 = (l.w - 3) / 2 + 1, H3 = (H2 - 7) / 3 + 1, W3 = (W2 - 7) / 3 + 1;
            ptr[0] = l.pooled; bytes[0] = (size_t)l.n * H3 * W3 * 64; ptr[1] = l.y; bytes[1] = bytes[0];
            ptr[2] = l.x.ptr; bytes[2] = (size_t)l.n * l.h * l.w * 16 * (l.storage ? 2 : 4);     // its INPUT as the launch could see it afterwards
        }
        if (o.kind == ESR_OP_ESA_APPLY && g_dbg_cap && !o.esa.skip_y) {
            const esr_esa_desc& e = o.esa; const size_t nb = (size_t)e.n * e.h * e.w * e.y.pitch * (e.storage ? 2 : 4), nc = (size_t)e.n * e.h_lo * e.w_lo * 64;
            char* dst = g_dbg_cap + ((size_t)(call % 64) * 4 + (size_t)(g_dbg_napply++ % 4)) * g_dbg_cap_bytes;
            const size_t nx = (size_t)e.n * e.h * e.w * e.x.pitch * (e.storage ? 2 : 4), n1 = (size_t)e.n * e.h * e.w * 16 * (e.storage ? 2 : 4);
            if (nb + nc + nx + n1 <= g_dbg_cap_bytes) { hipMemcpyAsync(dst, e.y.ptr, nb, hipMemcpyDeviceToDevice, st); hipMemcpyAsync(dst + nb, e.c3, nc, hipMemcpyDeviceToDevice, st);
                hipMemcpyAsync(dst + nb + nc, e.x.ptr, nx, hipMemcpyDeviceToDevice, st); hipMemcpyAsync(dst + nb + nc + nx, e.c1, n1, hipMemcpyDeviceToDevice, st); }
        }
        for (int k = 0; k < 4; ++k)
            if (ptr[k]) hipLaunchKernelGGL(dbg_sum_kernel, dim3(64), dim3(256), 0, st, static_cast<const unsigned*>(ptr[k]), bytes[k] / 4, g_dbg_log + ((size_t)call * 64 + i) * 4 + k);
    }
    return ESR_OK;
}
void esr_dbg_capture(void* base, unsigned long long bytes_per_apply) { g_dbg_cap = static_cast<char*>(base); g_dbg_cap_bytes = bytes_per_apply; }
unsigned esr_dbg_log_read(unsigned long long* dst, unsigned max_calls)
{
    const unsigned n = g_dbg_call.load() < max_calls ? g_dbg_call.load() : max_calls;
    hipDeviceSynchronize();
    if (g_dbg_log && n) hipMemcpy(dst, g_dbg_log, (size_t)n * 64 * 4 * 8, hipMemcpyDeviceToHost);
    return n;
}""")
    s = sub(s, "static int run_one(const esr_op& op, void* hip_stream)\n{", """constexpr unsigned DBG_CALLS = 4096;
static unsigned long long* g_dbg_log = nullptr;
static std::atomic<unsigned> g_dbg_call{0};
static char* g_dbg_cap = nullptr; static size_t g_dbg_cap_bytes = 0; static unsigned g_dbg_napply = 0;
__global__ void dbg_sum_kernel(const unsigned* p, size_t nwords, unsigned long long* out)
{
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256) s += (unsigned long long)p[i] * (1ull + (i & 4095));
    atomicAdd(out, s);
}
static int run_one(const esr_op& op, void* hip_stream)
{""")
    if "#include <atomic>" not in s:
        s = s.replace("#include <hip/hip_runtime.h>", "#include <hip/hip_runtime.h>\n#include <atomic>", 1)
    return {"esr_hip.hip": s}


def _sysload(expr):
    return ("[&]() { const float* q_ = " + expr + "; f32x4 r_; for (int e_ = 0; e_ < 4; ++e_) r_[e_] = __hip_atomic_load(q_ + e_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return r_; }()")


def v_c3sc(src):       # apply: the four bilinear gathers of c3 as system-scope loads (past the caches)
    s = src["esr_esa.hip"]
    for nm, ix in (("ta", "((size_t)y0 * p.w3 + x0)"), ("tb", "((size_t)y0 * p.w3 + x1)"), ("tc", "((size_t)y1 * p.w3 + x0)"), ("td", "((size_t)y1 * p.w3 + x1)")):
        s = sub(s, f"        g.{nm} = *reinterpret_cast<const f32x4*>(cb + {ix} * FP);", f"        g.{nm} = " + _sysload(f"cb + {ix} * FP") + ";", 1)
    return {"esr_esa.hip": s}


def v_c3ag(src):       # ... as agent-scope loads
    s = v_c3sc(src)["esr_esa.hip"]
    return {"esr_esa.hip": s.replace("__HIP_MEMORY_SCOPE_SYSTEM", "__HIP_MEMORY_SCOPE_AGENT")}


def v_c3wg(src):       # ... as workgroup-scope loads
    s = v_c3sc(src)["esr_esa.hip"]
    return {"esr_esa.hip": s.replace("__HIP_MEMORY_SCOPE_SYSTEM", "__HIP_MEMORY_SCOPE_WORKGROUP")}


def v_inv(src):        # apply: agent-scope acquire (buffer_inv sc1) at kernel entry
    s = src["esr_esa.hip"]
    s = sub(s, "__global__ __launch_bounds__(256) void esa_apply_mfma_kernel(const EsaK p)\n{\n", "__global__ __launch_bounds__(256) void esa_apply_mfma_kernel(const EsaK p)\n{\n    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"agent\");\n")
    return {"esr_esa.hip": s}


def _asmload(expr, bits):
    return ("[&]() { const float* q_ = " + expr + "; f32x4 r_; asm volatile(\"global_load_dwordx4 %0, %1, off " + bits + "\\n\\ts_waitcnt vmcnt(0)\" : \"=v\"(r_) : \"v\"(q_) : \"memory\"); return r_; }()")


def _c3asm(src, bits):
    s = src["esr_esa.hip"]
    for nm, ix in (("ta", "((size_t)y0 * p.w3 + x0)"), ("tb", "((size_t)y0 * p.w3 + x1)"), ("tc", "((size_t)y1 * p.w3 + x0)"), ("td", "((size_t)y1 * p.w3 + x1)")):
        s = sub(s, f"        g.{nm} = *reinterpret_cast<const f32x4*>(cb + {ix} * FP);", f"        g.{nm} = " + _asmload(f"cb + {ix} * FP", bits) + ";", 1)
    return {"esr_esa.hip": s}


def _dwload(expr, bits):
    return ("[&]() { const float* q_ = " + expr + "; f32x4 r_; float t0_, t1_, t2_, t3_; asm volatile(\"global_load_dword %0, %4, off " + bits + "\\n\\tglobal_load_dword %1, %4, off offset:4 " + bits + "\\n\\tglobal_load_dword %2, %4, off offset:8 " + bits + "\\n\\tglobal_load_dword %3, %4, off offset:12 " + bits + "\\n\\ts_waitcnt vmcnt(0)\" : \"=&v\"(t0_), \"=&v\"(t1_), \"=&v\"(t2_), \"=&v\"(t3_) : \"v\"(q_) : \"memory\"); r_ = f32x4{t0_, t1_, t2_, t3_}; return r_; }()")


def _c3dw(src, bits):
    s = src["esr_esa.hip"]
    for nm, ix in (("ta", "((size_t)y0 * p.w3 + x0)"), ("tb", "((size_t)y0 * p.w3 + x1)"), ("tc", "((size_t)y1 * p.w3 + x0)"), ("td", "((size_t)y1 * p.w3 + x1)")):
        s = sub(s, f"        g.{nm} = *reinterpret_cast<const f32x4*>(cb + {ix} * FP);", f"        g.{nm} = " + _dwload(f"cb + {ix} * FP", bits) + ";", 1)
    return {"esr_esa.hip": s}


def v_dwplain(src):    # the gathers as four plain dword loads each (asm, vmcnt(0))
    return _c3dw(src, "")


def v_dwsc0(src):
    return _c3dw(src, "sc0")


def v_nopref(src):     # apply: no prefetch of the next group (fetch -> finish in the same iteration)
    s = src["esr_esa.hip"]
    s = sub(s, """    Grp cur, nxt;
    fetch(grp, cur);
    for (;;) {
        const long long gn = grp + gstep;
        const bool more = gn < ngroups;                          // wave-uniform
        if (more) fetch(gn, nxt);
        finish(cur);
        if (!more) break;
        cur = nxt;
        grp = gn;
    }""", """    for (; grp < ngroups; grp += gstep) {
        Grp cur;
        fetch(grp, cur);
        finish(cur);
    }""")
    return {"esr_esa.hip": s}


LOOP = """    Grp cur, nxt;
    fetch(grp, cur);
    for (;;) {
        const long long gn = grp + gstep;
        const bool more = gn < ngroups;                          // wave-uniform
        if (more) fetch(gn, nxt);
        finish(cur);
        if (!more) break;
        cur = nxt;
        grp = gn;
    }"""


def v_prefwait(src):   # prefetch kept, but everything has landed before finish() starts
    s = src["esr_esa.hip"]
    return {"esr_esa.hip": sub(s, LOOP, LOOP.replace("        if (more) fetch(gn, nxt);\n", "        if (more) fetch(gn, nxt);\n        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n        __builtin_amdgcn_sched_barrier(0);\n"))}


def v_prefafter(src):  # the next group is fetched BEHIND the current group's stores (the copy at the latch stays)
    s = src["esr_esa.hip"]
    return {"esr_esa.hip": sub(s, LOOP, LOOP.replace("        if (more) fetch(gn, nxt);\n        finish(cur);\n", "        finish(cur);\n        __builtin_amdgcn_sched_barrier(0);\n        if (more) fetch(gn, nxt);\n"))}


def v_pingpong(src):   # prefetch kept, no register copies: the loop body twice with the roles of the two groups swapped
    s = src["esr_esa.hip"]
    return {"esr_esa.hip": sub(s, LOOP, """    Grp ga, gb;
    fetch(grp, ga);
    for (;;) {
        long long gn = grp + gstep;
        bool more = gn < ngroups;
        if (more) fetch(gn, gb);
        finish(ga);
        if (!more) break;
        grp = gn;
        gn = grp + gstep;
        more = gn < ngroups;
        if (more) fetch(gn, ga);
        finish(gb);
        if (!more) break;
        grp = gn;
    }""")}


def v_nops(src):       # 16 extra wait states behind every MFMA of esr_esa.hip before its result can be read
    s = src["esr_esa.hip"]
    s = sub(s, """    if (ST == ESR_STORE_BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// Channel <-> MFMA row map""", """    f32x4 r_;
    if (ST == ESR_STORE_BF16) r_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    else r_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    asm volatile("s_nop 15" : "+v"(r_));
    return r_;
}

// Channel <-> MFMA row map""")
    return {"esr_esa.hip": s}


def v_abl_nosig(src):  # ABLATION (wrong results): no v_exp / v_rcp in the apply -- y = x * m
    s = src["esr_esa.hip"]
    return {"esr_esa.hip": sub(s, "        return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * m));", "        return m;")}


def v_abl_nomath(src): # ABLATION (wrong results): finish() stores x unchanged -- no MFMA, no bilinear, no sigmoid
    s = src["esr_esa.hip"]
    s = sub(s, "        return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * m));", "        return m;")
    s = sub(s, "                ow[d] = (unsigned)to16<ST>(xa * sigmoid(mm[2 * d])) | ((unsigned)to16<ST>(xb * sigmoid(mm[2 * d + 1])) << 16);", "                ow[d] = xw[d] + (unsigned)g.bc1.x + __builtin_bit_cast(unsigned, g.ta.x + g.tb.x + g.tc.x + g.td.x);")
    return {"esr_esa.hip": s}


def v_x4plain(src):    # the gathers as asm dwordx4 loads + vmcnt(0), no scope bits (timing control for x4sc0 / x4sc1)
    return _c3asm(src, "")


def v_x4sc0(src):
    return _c3asm(src, "sc0")


def v_x4sc1(src):
    return _c3asm(src, "sc1")


def v_chainfence(src): # chain kernel: system-scope fence behind its stores
    s = src["esr_esa_lowres.hip"]
    s = sub(s, "            S = So;\n            ++oy; ++ox;\n        }\n    }\n}\n", "            S = So;\n            ++oy; ++ox;\n        }\n    }\n    __threadfence_system();\n}\n")
    return {"esr_esa_lowres.hip": s}


def v_nop(src):        # an empty launch in front of the MFMA apply kernel
    s = src["esr_esa.hip"]
    s = sub(s, "template <int ST>\nint launch_esa_mfma(const EsaK& k, int np0, int np1, hipStream_t st)\n{", "__global__ void dbg_nop_kernel() {}\ntemplate <int ST>\nint launch_esa_mfma(const EsaK& k, int np0, int np1, hipStream_t st)\n{\n    hipLaunchKernelGGL(dbg_nop_kernel, dim3(1), dim3(64), 0, st);")
    return {"esr_esa.hip": s}


VARIANTS = {"abl_nosig": v_abl_nosig, "abl_nomath": v_abl_nomath, "nops": v_nops, "prefwait": v_prefwait, "prefafter": v_prefafter, "pingpong": v_pingpong, "nopref": v_nopref, "dwplain": v_dwplain, "dwsc0": v_dwsc0, "x4plain": v_x4plain, "x4sc0": v_x4sc0, "x4sc1": v_x4sc1, "inv": v_inv, "c3sc": v_c3sc, "c3ag": v_c3ag, "c3wg": v_c3wg, "chainfence": v_chainfence, "nop": v_nop, "log": v_log, "check": v_check, "oldpool": v_oldpool, "zero": v_zero, "pool3": v_pool3, "nw8": v_nw8, "tail": v_tail, "wait0": v_wait0}


def build(name):
    src = {f: open(os.path.join(SRC, f)).read() for f in os.listdir(SRC) if f.endswith(".hip")}
    changed = VARIANTS[name](src)
    d = os.path.join("/tmp", "race_" + name); os.makedirs(d, exist_ok=True)
    objs = []
    for f in src:
        o = os.path.join(OBJ, f[:-4] + ".o")
        if f in changed:
            p = os.path.join(d, f); open(p, "w").write(changed[f])
            o = os.path.join(d, f[:-4] + ".o")
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", p, "-o", o], stderr=subprocess.DEVNULL)
        objs.append(o)
    out = os.path.join(REPO, "tools", "abl", f"libesr_r_{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    names = sys.argv[1:] or list(VARIANTS)
    with ThreadPoolExecutor(len(names)) as ex:
        for o in ex.map(build, names):
            print(o)
