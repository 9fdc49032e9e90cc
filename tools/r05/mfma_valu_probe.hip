// Can ONE wave overlap its own VALU / LDS instructions with its own MFMAs on gfx950?  (round 5: the chain kernel's ablations say a wave alone
// on its SIMD pays MFMA + VALU, not max.)  Kernel<K, KIND>: per loop pass 8 independent v_mfma_f32_16x16x32_bf16, each followed by K
// independent instructions of KIND (0: v_fma_f32, 1: v_pk_fma_f32, 2: v_cvt_pk_bf16_f32, 3: ds_read_b128 (conflict-free), 4: v_mul + dependent v_max).
// Reports cycles per MFMA with 1 and 2 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o mfma_valu_probe mfma_valu_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int K, int KIND, bool SAME = false, int NACC = 8>
__global__ __launch_bounds__(1024) void probe(unsigned long long* out, int iters, float seed)
{
    __shared__ i32x4 lds[512];
    lds[threadIdx.x] = i32x4{(int)threadIdx.x, 1, 2, 3};
    __syncthreads();
    f32x4 acc[8];
    i32x4 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = i32x4{0x3f803f80 + i, 0x3f803f80, 0x3f803f80 + (int)threadIdx.x, 0x3f803f80};
    for (int i = 0; i < 4; ++i) b[i] = i32x4{0x3f803f80, 0x3f803f80 + i, 0x3f803f80, 0x3f803f80 + (int)threadIdx.x};
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]));
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(b[i]));
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8];
    f32x2 pv[8];
    i32x4 lv[4];
    for (int i = 0; i < 8; ++i) { v[i] = seed + i + threadIdx.x; pv[i] = f32x2{seed + i, seed - i}; }
    for (int i = 0; i < 4; ++i) lv[i] = i32x4{0, 0, 0, 0};
    const i32x4* lp = lds + (threadIdx.x & 63);
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (SAME) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a[0]), "v"(a[0]));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m % NACC]) : "v"(a[m]), "v"(b[m & 3]));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int j = (m * K + k) & 7;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j]) : "v"(seed));
                else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pv[j]) : "v"(pv[(j + 1) & 7]));
                else if (KIND == 2) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(v[j]), "v"(v[(j + 1) & 7])); asm volatile("" :: "v"(r)); }
                else if (KIND == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(lv[j & 3]) : "v"((unsigned)(size_t)lp * 0u + (unsigned)((threadIdx.x & 63) * 16)));
                else { float tt; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(tt) : "v"(seed), "v"(v[j])); asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[j]) : "v"(tt)); }
            }
        }
        if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i].x + v[i] + pv[i].x;
    for (int i = 0; i < 4; ++i) s += (float)lv[i].x;
    if (s == 12345.678f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int K, int KIND, bool SAME = false, int NACC = 8>
void run(const char* name, unsigned long long* d)
{
    for (int waves = 4; waves <= 16; waves *= 2) {
        const int iters = 2000;
        hipLaunchKernelGGL((probe<K, KIND, SAME, NACC>), dim3(256), dim3(64 * waves), 0, 0, d, iters, 1.0f);
        hipLaunchKernelGGL((probe<K, KIND, SAME, NACC>), dim3(256), dim3(64 * waves), 0, 0, d, iters, 1.0f);
        hipDeviceSynchronize();
        unsigned long long t;
        hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
        printf("%-22s K=%d  %d wave(s)/SIMD: %6.1f cycles per MFMA (per wave)   %6.1f per MFMA per SIMD\n", name, K, waves / 4, (double)t / (iters * 8), (double)t / (iters * 8) / (waves / 4));
    }
}

int main()
{
    unsigned long long* d;
    hipMalloc(&d, 64);
    run<0, 0, true>("mfma only, A == B", d);
    run<0, 0>("mfma only", d);
    run<0, 0, false, 3>("mfma only, 3 accs", d);
    run<0, 0, false, 4>("mfma only, 4 accs", d);
    run<2, 0, false, 3>("v_fma, 3 accs", d);
    run<0, 0, false, 5>("mfma only, 5 accs", d);
    run<0, 0, false, 6>("mfma only, 6 accs", d);
    run<0, 0, false, 7>("mfma only, 7 accs", d);
    run<2, 0, false, 6>("2 v_fma, 6 accs", d);
    run<3, 0, false, 6>("3 v_fma, 6 accs", d);
    run<4, 0, false, 6>("4 v_fma, 6 accs", d);
    run<1, 3, false, 6>("1 ds_read_b128, 6 accs", d);
    run<2, 4, false, 6>("2 (v_mul+v_max), 6 accs", d);
    return 0;
}
