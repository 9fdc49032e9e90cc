// Why do the two waves of a SIMD not interleave their MFMAs in rlfb_chain_kernel?  Mimics its inner loop: per step 45 v_mfma_f32_16x16x32_bf16 over
// NA distinct A-operand register quads (45 = the layer's weight fragments), 3 accumulators (reuse distance 3), B from a ring of 4 quads,
// optionally an s_barrier per step.  8 waves per block (2 per SIMD), one block per CU.  Prints cycles per step per wave (first / second wave of SIMD 0).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int NA, bool BARRIER, int WAVES, int NACC, int PAT = 0>
__global__ __launch_bounds__(64 * WAVES) void probe(unsigned long long* out, int steps)
{
    i32x4 a[NA], b[4];
    f32x4 acc[NACC];
    for (int i = 0; i < NA; ++i) { a[i] = i32x4{0x3f803f80 + i, 0x3f803f80, 0x3f803f80 + (int)threadIdx.x, 0x3f803f80}; asm volatile("" : "+v"(a[i])); }
    for (int i = 0; i < 4; ++i) { b[i] = i32x4{0x3f803f80, 0x3f803f80 + i, 0x3f803f80, 0x3f803f80 + (int)threadIdx.x}; asm volatile("" : "+v"(b[i])); }
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int m = 0; m < 45; ++m)
        {
            // PAT 0: B shared by 3 consecutive MFMAs, A distinct (the chain kernel's order); 1: B changes with every MFMA, A distinct;
            // 2: A shared by 2 consecutive MFMAs, B alternates; 3: A and B both change, accumulators in the same order
            const int ai = PAT == 2 ? (m / 2) % NA : m % NA;
            const int bi = PAT == 0 ? (m / 3) & 3 : (PAT == 2 ? m & 1 : m & 3);
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m % NACC]) : "v"(a[ai]), "v"(b[bi]));
        }
        if (BARRIER) asm volatile("s_barrier" ::: "memory");
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float sum = 0.f;
    for (int i = 0; i < NACC; ++i) sum += acc[i].x;
    if (sum == 12345.678f) out[63] = 1;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
}

template <int NA, bool BARRIER, int WAVES, int NACC, int PAT = 0>
void run(unsigned long long* d)
{
    const int steps = 400;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((probe<NA, BARRIER, WAVES, NACC, PAT>), dim3(256), dim3(64 * WAVES), 0, 0, d, steps);
    hipDeviceSynchronize();
    unsigned long long t[16];
    hipMemcpy(t, d, sizeof(t), hipMemcpyDeviceToHost);
    printf("pat %d  A quads %2d  accs %d  barrier %d  waves %d: cycles per 45-MFMA step, waves 0..%d:", PAT, NA, NACC, (int)BARRIER, WAVES, WAVES - 1);
    for (int w = 0; w < WAVES; ++w) printf(" %5.0f", (double)t[w] / steps);
    printf("   -> %.1f cycles per MFMA per SIMD\n", (double)t[WAVES - 1] / steps / 45.0 / (WAVES / 4));
}

int main()
{
    unsigned long long* d;
    hipMalloc(&d, 1024);
    run<45, false, 8, 3, 0>(d); run<45, false, 8, 3, 1>(d); run<45, false, 8, 6, 1>(d); run<45, false, 8, 6, 2>(d); run<45, false, 8, 8, 1>(d);
    run<8, false, 8, 8, 1>(d); run<45, false, 4, 6, 1>(d); run<45, false, 4, 6, 2>(d); run<45, true, 8, 6, 1>(d); run<45, true, 8, 6, 2>(d);
    return 0;
}
