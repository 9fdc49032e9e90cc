// What does ONE more instruction of a given kind cost a wave that is alone on its SIMD and issues MFMAs back to back?  (round 5, after dot2_probe: v_add_f32 is
// nearly free twice per MFMA, v_lshlrev_b32 never.)  Per loop pass 8 independent v_mfma_f32_16x16x32_bf16 (6 accumulators), each followed by K independent
// instructions of one kind.  hipcc --offload-arch=gfx950 -O3 -w -o valu_cost_probe valu_cost_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define KINDS(X) \
    X(0, "v_add_f32 v, v, sgpr   (in place)", "v_add_f32 %1, %3, %1") \
    X(1, "v_add_f32 v, v, v2     (in place)", "v_add_f32 %1, %1, %2") \
    X(2, "v_add_f32 w, v, v2     (3-address)", "v_add_f32 %0, %1, %2") \
    X(3, "v_add_f32 w, v, sgpr   (3-address)", "v_add_f32 %0, %3, %1") \
    X(4, "v_mul_f32 w, 0.05, v   (literal)", "v_mul_f32 %0, 0x3d4ccccd, %1") \
    X(5, "v_mov_b32 w, v", "v_mov_b32 %0, %1") \
    X(6, "v_add_f32 v, v, v      (in place)", "v_add_f32 %1, %1, %1") \
    X(7, "v_max_f32 v, v, w      (in place, w just written)", "v_max_f32 %1, %1, %0") \
    X(8, "v_mul_f32 w, sgpr, v", "v_mul_f32 %0, %3, %1") \
    X(9, "s_nop 0", "s_nop 0")

template <int K, int KIND>
__global__ __launch_bounds__(256) void cost(unsigned long long* out, int iters, float seed)
{
    f32x4 acc[6];
    i32x4 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = i32x4{0x3f803f80 + i, 0x3f803f80, 0x3f803f80 + (int)threadIdx.x, 0x3f803f80};
    for (int i = 0; i < 4; ++i) b[i] = i32x4{0x3f803f80, 0x3f803f80 + i, 0x3f803f80, 0x3f803f80 + (int)threadIdx.x};
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]));
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(b[i]));
    for (int i = 0; i < 6; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float v[8], w[8];
    f32x2 pv[8], pw[8];
    unsigned sreg = 0;
    float fs = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, seed)));
    for (int i = 0; i < 8; ++i) { v[i] = seed + i + threadIdx.x; w[i] = seed * 0.5f + i; pv[i] = f32x2{v[i], w[i]}; pw[i] = f32x2{w[i], v[i]}; asm volatile("" : "+v"(v[i]), "+v"(w[i]), "+v"(pv[i]), "+v"(pw[i])); }
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(v[0]), "v"(w[0]) : "vcc");
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m % 6]) : "v"(a[m]), "v"(b[m & 3]));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int j = (m * K + k) & 7, j2 = (j + 3) & 7;
#define X(id, name, text) \
                if (KIND == id) asm volatile(text : "+v"(w[j]), "+v"(v[j]) : "v"(v[j2]), "s"(fs));
                KINDS(X)
#undef X
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = (float)sreg;
    for (int i = 0; i < 6; ++i) s += acc[i].x;
    for (int i = 0; i < 8; ++i) s += v[i] + w[i] + pw[i].x + pw[i].y;
    if (s == 12345.678f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int K, int KIND>
double run1(unsigned long long* d)
{
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((cost<K, KIND>), dim3(256), dim3(256), 0, 0, d, iters, 1.0f);
    hipDeviceSynchronize();
    unsigned long long t;
    hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
    return (double)t / (iters * 8);
}

template <int KIND>
void run(const char* name, unsigned long long* d)
{
    printf("%-36s K=1 %5.1f  K=2 %5.1f  K=3 %5.1f  K=4 %5.1f  K=6 %5.1f   cycles per MFMA slot\n", name, run1<1, KIND>(d), run1<2, KIND>(d), run1<3, KIND>(d), run1<4, KIND>(d), run1<6, KIND>(d));
}

int main()
{
    unsigned long long* d;
    hipMalloc(&d, 64);
    printf("MFMA only: %5.1f cycles per slot (8 independent v_mfma_f32_16x16x32_bf16 per pass, 6 accumulators, one wave per SIMD)\n", run1<0, 0>(d));
#define X(id, name, text) run<id>(name, d);
    KINDS(X)
#undef X
    return 0;
}
