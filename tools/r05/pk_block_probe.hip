// Do packed fp32 VALU instructions (v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32) of ONE wave hold up the MFMAs of ANOTHER wave on the same SIMD?
// (round 5: one wave pays ~16 cycles for a v_pk_* beside its own MFMAs, 2.5 for a plain one -- valu_cost_probe; the fp32 Winograd kernels run two waves
// per SIMD and issue 240 v_pk_add_f32 per 384 MFMAs.)  Block = 8 waves: waves 0-3 (one per SIMD) stream independent MFMAs, waves 4-7 (their partners)
// stream VALU instructions of one kind, or idle.  Reports the MFMA wave's cycles per MFMA and the partner's cycles per instruction.
// hipcc --offload-arch=gfx950 -O3 -w -o pk_block_probe pk_block_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MF, int KIND>
__global__ __launch_bounds__(512) void probe(unsigned long long* out, int iters, float seed)
{
    const int wv = threadIdx.x >> 6;
    unsigned long long t0, t1;
    if (wv < 4) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        i32x4 a = {0x3f803f80, 0x3f803f80 + (int)threadIdx.x, 0x3f803f80, 0x3f803f80}, b = {0x3f803f80, 0x3f803f80, 0x3f803f81, 0x3f803f80};
        float fa = seed + threadIdx.x, fb = seed * 0.5f;
        asm volatile("" : "+v"(a), "+v"(b), "+v"(fa), "+v"(fb));
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (MF == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(fa), "v"(fb));
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i].x;
        if (s == 12345.678f) out[7] = 1;
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    } else {
        f32x2 pv[8], pw[8];
        float v[8], w[8];
        for (int i = 0; i < 8; ++i) { v[i] = seed + i; w[i] = seed - i; pv[i] = f32x2{v[i], w[i]}; pw[i] = f32x2{w[i], v[i]}; asm volatile("" : "+v"(v[i]), "+v"(w[i]), "+v"(pv[i]), "+v"(pw[i])); }
        const int n2 = KIND == 0 ? 0 : iters * 4;        // the partner runs about as long as the MFMA wave (8 instructions per pass)
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
        for (int it = 0; it < n2; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(w[j]));
                else if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pv[j]) : "v"(pw[j]));
                else if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pv[j]) : "v"(pw[j]));
                else if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j]) : "v"(w[j]));
                else if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pv[j]) : "v"(pw[j]));
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += v[i] + pv[i].x + pv[i].y;
        if (s == 12345.678f) out[7] = 1;
        if (threadIdx.x == 256 && blockIdx.x == 0) out[1] = t1 - t0;
    }
}

template <int MF, int KIND>
void run(const char* mf, const char* kind, unsigned long long* d)
{
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<MF, KIND>), dim3(256), dim3(512), 0, 0, d, iters, 1.0f);
    hipDeviceSynchronize();
    unsigned long long t[2];
    hipMemcpy(t, d, 16, hipMemcpyDeviceToHost);
    printf("%-26s partner: %-14s  MFMA wave %6.1f cycles per MFMA", mf, kind, (double)t[0] / (iters * 8));
    if (KIND) printf("   partner %6.1f cycles per instruction", (double)t[1] / (iters * 4 * 8));
    printf("\n");
}

int main()
{
    unsigned long long* d;
    hipMalloc(&d, 64);
    run<0, 0>("v_mfma_f32_16x16x32_bf16", "idle", d); run<0, 1>("v_mfma_f32_16x16x32_bf16", "v_add_f32", d); run<0, 4>("v_mfma_f32_16x16x32_bf16", "v_fma_f32", d);
    run<0, 2>("v_mfma_f32_16x16x32_bf16", "v_pk_add_f32", d); run<0, 3>("v_mfma_f32_16x16x32_bf16", "v_pk_fma_f32", d); run<0, 5>("v_mfma_f32_16x16x32_bf16", "v_pk_mul_f32", d);
    run<1, 0>("v_mfma_f32_16x16x4_f32", "idle", d); run<1, 1>("v_mfma_f32_16x16x4_f32", "v_add_f32", d); run<1, 4>("v_mfma_f32_16x16x4_f32", "v_fma_f32", d);
    run<1, 2>("v_mfma_f32_16x16x4_f32", "v_pk_add_f32", d); run<1, 3>("v_mfma_f32_16x16x4_f32", "v_pk_fma_f32", d); run<1, 5>("v_mfma_f32_16x16x4_f32", "v_pk_mul_f32", d);
    return 0;
}
