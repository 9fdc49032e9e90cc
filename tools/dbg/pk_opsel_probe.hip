// pk_opsel_probe.hip -- does a packed-fp32 VALU instruction whose LOW half reads the HIGH dword of a 64-bit source (op_sel:[0,1])
// return wrong results in the last 16 lanes of a wave when waves of OTHER kernels share the SIMD?  (round 4: what the dumps of
// tools/dbg/race_dump.py showed for esa_apply_mfma_kernel's bilinear term: lx * tb == 0 in lanes 48..63, low half only.)
//
//   hipcc --offload-arch=gfx950 -O2 -o pk_opsel_probe tools/dbg/pk_opsel_probe.hip && ./pk_opsel_probe [seconds per pairing]
//
// victim kernels (one wave per SIMD, 256 blocks x 256 threads) loop one packed multiply form on lane-dependent data and compare
// every result with the scalar products; a partner kernel on ANOTHER stream loops one instruction class (or launches / retires waves
// at a high rate).  Prints wrong results per pairing with the lanes and halves they hit.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Stat { unsigned long long bad_lo, bad_hi, iters; unsigned lane_hist[4]; unsigned first_got, first_want, first_lane; };

// FORM 0: op_sel:[0,1]  (lo = a.lo * b.hi, hi = a.hi * b.hi)   -- the failing form of the apply kernel
// FORM 1: op_sel_hi:[1,0] (lo = a.lo * b.lo, hi = a.hi * b.lo) -- the form the passing builds use
// FORM 2: default         (lo = a.lo * b.lo, hi = a.hi * b.hi)
// FORM 3: v_pk_fma op_sel:[1,0,0] op_sel_hi:[1,1,1] (lo = a.hi * b.lo + c.lo, hi = a.hi * b.hi + c.hi) -- esa_apply_kernel (fp32 plans)
template <int FORM>
__global__ __launch_bounds__(256) void victim(Stat* st, int iters, float seed)
{
    const int lane = threadIdx.x & 63;
    f32x2 a = {seed + 0.001f * threadIdx.x, 1.5f + 0.002f * threadIdx.x};
    f32x2 b = {0.75f + 0.003f * lane, 1.25f + 0.004f * blockIdx.x};
    unsigned long long bad_lo = 0, bad_hi = 0;
    unsigned fg = 0, fw = 0;
    for (int i = 0; i < iters; ++i) {
        f32x2 r, w;
        if (FORM == 0) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b)); w.x = a.x * b.y; w.y = a.y * b.y; }
        else if (FORM == 1) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b)); w.x = a.x * b.x; w.y = a.y * b.x; }
        else if (FORM == 2) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); w.x = a.x * b.x; w.y = a.y * b.y; }
        else if (FORM == 3) { f32x2 c = {0.5f, 0.25f}; asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); w.x = __builtin_fmaf(a.y, b.x, c.x); w.y = __builtin_fmaf(a.y, b.y, c.y); }
        else if (FORM == 4) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b)); w.x = a.x + b.y; w.y = a.y + b.y; }
        else if (FORM == 5) { f32x2 c = {0.5f, 0.25f}; asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); w.x = __builtin_fmaf(a.x, b.y, c.x); w.y = __builtin_fmaf(a.y, b.y, c.y); }
        else { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b)); w.x = a.y * b.x; w.y = a.y * b.y; }
        asm volatile("" : "+v"(w));
        if (__builtin_bit_cast(unsigned, r.x) != __builtin_bit_cast(unsigned, w.x)) { if (!bad_lo && !bad_hi) { fg = __builtin_bit_cast(unsigned, r.x); fw = __builtin_bit_cast(unsigned, w.x); } ++bad_lo; }
        if (__builtin_bit_cast(unsigned, r.y) != __builtin_bit_cast(unsigned, w.y)) { if (!bad_lo && !bad_hi) { fg = __builtin_bit_cast(unsigned, r.y); fw = __builtin_bit_cast(unsigned, w.y); } ++bad_hi; }
        a.x += 0.000001f * (i & 7);           // keep the operands moving
        b.y += 0.000002f;
    }
    if (bad_lo | bad_hi) {
        atomicAdd(&st->bad_lo, bad_lo); atomicAdd(&st->bad_hi, bad_hi);
        atomicAdd(&st->lane_hist[lane >> 4], 1u);
        if (atomicCAS(&st->first_lane, 0xffffffffu, (unsigned)lane) == 0xffffffffu) { st->first_got = fg; st->first_want = fw; }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) st->iters = (unsigned long long)iters;
}

// ---- partners --------------------------------------------------------------------------------------------------------------------
enum { P_NONE, P_MFMA, P_MFMA32, P_VALU, P_PKHI, P_DPP, P_PERMSWAP, P_SDWA, P_TRANS, P_LDS, P_VMEM, P_CVTPK, P_ACC, P_LAUNCH, P_COUNT };
const char* PNAME[P_COUNT] = {"none", "mfma 16x16x32 bf16", "mfma 16x16x4 f32", "v_fma_f32", "v_pk_fma op_sel_hi", "dpp row_shr", "v_permlane16_swap", "sdwa", "v_exp/v_rcp",
                              "ds_read/ds_write", "global_load/store", "v_cvt_pk_bf16_f32", "v_accvgpr_write/read", "wave launches (tiny blocks)"};

template <int P>
__global__ __launch_bounds__(256) void partner(float* out, int iters)
{
    __shared__ float sm[1024];
    float x = 1.0f + 0.001f * threadIdx.x, y = 0.5f;
    f32x2 p2 = {x, y}, q2 = {y, x};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    sm[threadIdx.x] = x; sm[threadIdx.x + 256] = y;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
        if (P == P_VALU) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y)); }
        else if (P == P_PKHI) { asm volatile("v_pk_fma_f32 %0, %1, %0, %1 op_sel_hi:[0,1,1]" : "+v"(p2) : "v"(q2)); }
        else if (P == P_DPP) { asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y)); }
        else if (P == P_PERMSWAP) { asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y)); }
        else if (P == P_SDWA) { asm volatile("v_add_f16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "+v"(x) : "v"(y)); }
        else if (P == P_TRANS) { asm volatile("v_exp_f32 %0, %0\n\tv_rcp_f32 %0, %0" : "+v"(x)); }
        else if (P == P_MFMA) { bf16x8_t a8, b8; for (int e = 0; e < 8; ++e) { a8[e] = (__bf16)x; b8[e] = (__bf16)y; } acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc, 0, 0, 0); }
        else if (P == P_MFMA32) { acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0); }
        else if (P == P_LDS) { sm[(threadIdx.x + i) & 1023] = x; x += sm[(threadIdx.x * 3 + i) & 1023]; }
        else if (P == P_VMEM) { out[4096 + ((blockIdx.x * 256 + threadIdx.x + i * 64) & 0xfffff)] = x; x += out[4096 + ((blockIdx.x * 256 + threadIdx.x * 5 + i) & 0xfffff)]; }
        else if (P == P_CVTPK) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); x += (float)(r & 1); }
        else if (P == P_ACC) { asm volatile("v_accvgpr_write_b32 a0, %0\n\ts_nop 1\n\tv_accvgpr_read_b32 %0, a0" : "+v"(x) :: "a0"); }
    }
    if (out && x + y + p2.x + acc.x == 12345.678f) out[threadIdx.x] = x;
}

__global__ void tiny(float* out) { if (out && threadIdx.x == 1000) out[0] = 1.f; }

template <int P> void run_partner(hipStream_t s, float* buf, int iters) { hipLaunchKernelGGL(partner<P>, dim3(512), dim3(256), 0, s, buf, iters); }

int main(int argc, char** argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 0.5;
    const bool quick = argc > 2 && !strcmp(argv[2], "quick");          // partners none / MFMA only
    hipStream_t sv, sp;
    CK(hipStreamCreate(&sv)); CK(hipStreamCreate(&sp));
    Stat* st; CK(hipMalloc(&st, sizeof(Stat)));
    float* buf; CK(hipMalloc(&buf, (1 << 20) * 4 + 65536)); CK(hipMemset(buf, 0, (1 << 20) * 4 + 65536));
    const char* FNAME[7] = {"v_pk_mul op_sel:[0,1] (lo reads src1.hi)", "v_pk_mul op_sel_hi:[1,0] (hi reads lo)", "v_pk_mul default", "v_pk_fma op_sel:[1,0,0] (src0.hi)",
                            "v_pk_add op_sel:[0,1] (src1.hi)", "v_pk_fma op_sel:[0,1,0] (src1.hi)", "v_pk_mul op_sel:[1,0] (src0.hi)"};
    for (int form = 0; form < 7; ++form) {
        for (int p = 0; p < (quick ? 3 : P_COUNT); ++p) {
            Stat h; memset(&h, 0, sizeof(h)); h.first_lane = 0xffffffffu;
            CK(hipMemcpy(st, &h, sizeof(h), hipMemcpyHostToDevice));
            unsigned long long launches = 0;
            auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
                const int it = 20000;
                switch (form) {
                    case 0: hipLaunchKernelGGL(victim<0>, dim3(256), dim3(256), 0, sv, st, it, 1.0f); break;
                    case 1: hipLaunchKernelGGL(victim<1>, dim3(256), dim3(256), 0, sv, st, it, 1.0f); break;
                    case 2: hipLaunchKernelGGL(victim<2>, dim3(256), dim3(256), 0, sv, st, it, 1.0f); break;
                    case 3: hipLaunchKernelGGL(victim<3>, dim3(256), dim3(256), 0, sv, st, it, 1.0f); break;
                    case 4: hipLaunchKernelGGL(victim<4>, dim3(256), dim3(256), 0, sv, st, it, 1.0f); break;
                    case 5: hipLaunchKernelGGL(victim<5>, dim3(256), dim3(256), 0, sv, st, it, 1.0f); break;
                    default: hipLaunchKernelGGL(victim<6>, dim3(256), dim3(256), 0, sv, st, it, 1.0f); break;
                }
                ++launches;
                const int pit = 4000;
                switch (p) {
                    case P_NONE: break;
                    case P_VALU: run_partner<P_VALU>(sp, buf, pit); break;
                    case P_PKHI: run_partner<P_PKHI>(sp, buf, pit); break;
                    case P_DPP: run_partner<P_DPP>(sp, buf, pit); break;
                    case P_PERMSWAP: run_partner<P_PERMSWAP>(sp, buf, pit); break;
                    case P_SDWA: run_partner<P_SDWA>(sp, buf, pit); break;
                    case P_TRANS: run_partner<P_TRANS>(sp, buf, pit); break;
                    case P_MFMA: run_partner<P_MFMA>(sp, buf, pit); break;
                    case P_MFMA32: run_partner<P_MFMA32>(sp, buf, pit); break;
                    case P_LDS: run_partner<P_LDS>(sp, buf, pit); break;
                    case P_VMEM: run_partner<P_VMEM>(sp, buf, pit / 8); break;
                    case P_CVTPK: run_partner<P_CVTPK>(sp, buf, pit); break;
                    case P_ACC: run_partner<P_ACC>(sp, buf, pit); break;
                    case P_LAUNCH: for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(tiny, dim3(65536), dim3(64), 0, sp, (float*)nullptr); break;
                }
                if ((launches & 7) == 0) { CK(hipStreamSynchronize(sv)); }
            }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(&h, st, sizeof(h), hipMemcpyDeviceToHost));
            const double execs = (double)launches * 20000.0 * 256 * 4;      // wave-instructions
            printf("FORM %d %-42s | partner %-28s: %llu wave-launches, bad lo %llu hi %llu of %.3g wave-instr; lanes 0-15/16-31/32-47/48-63: %u %u %u %u",
                   form, FNAME[form], PNAME[p], launches, h.bad_lo, h.bad_hi, execs, h.lane_hist[0], h.lane_hist[1], h.lane_hist[2], h.lane_hist[3]);
            if (h.first_lane != 0xffffffffu) printf("; first: lane %u got 0x%08x want 0x%08x", h.first_lane, h.first_got, h.first_want);
            printf("\n"); fflush(stdout);
        }
    }
    return 0;
}
