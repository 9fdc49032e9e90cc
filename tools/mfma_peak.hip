// Practical fp32-MFMA ceiling on this box: a pure v_mfma_f32_16x16x4_f32 stream (16 independent
// accumulators per wave, like conv_f32_kernel<4,3>) on random vs zero operands, with the effective
// shader clock (clock64 = s_memtime shader cycles, wall_clock64 = 100 MHz constant).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak tools/mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef float f32x16 __attribute__((ext_vector_type(16)));
// same FLOPs per iteration with v_mfma_f32_32x32x2_f32 (4 accumulators x 16 regs, 64-cycle issue)
__global__ __launch_bounds__(256) void mfma_stream32(const float* in, float* out, int iters, unsigned long long* clk)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float a[2], b[2];
    for (int i = 0; i < 2; ++i) { a[i] = in[(tid * 8 + i) & 0xFFFFF]; b[i] = in[(tid * 8 + 4 + i) & 0xFFFFF]; }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    acc[t * 2 + r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[r], acc[t * 2 + r], 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[tid] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

template <int WAVES_PER_BLOCK>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void mfma_stream(const float* in, float* out, int iters,
                                                                     unsigned long long* clk)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(tid * 8 + i) & 0xFFFFF]; b[i] = in[(tid * 8 + 4 + i) & 0xFFFFF]; }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[t * 4 + r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[r], acc[t * 4 + r], 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    f32x4 s = acc[0];
    for (int i = 1; i < 16; ++i) s += acc[i];
    out[tid] = s.x + s.y + s.z + s.w;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

template <int WPB>
void run(const char* label, const float* d_in, float* d_out, unsigned long long* d_clk, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_stream<WPB><<<blocks, WPB * 64>>>(d_in, d_out, iters, d_clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int k = 0; k < 5; ++k) mfma_stream<WPB><<<blocks, WPB * 64>>>(d_in, d_out, iters, d_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    std::vector<unsigned long long> clk(blocks * 2);
    hipMemcpy(clk.data(), d_clk, sizeof(unsigned long long) * blocks * 2, hipMemcpyDeviceToHost);
    double sc = 0, sw = 0;
    for (int i = 0; i < blocks; ++i) { sc += clk[2 * i]; sw += clk[2 * i + 1]; }
    const double flops = 2.0 * 16 * 16 * 4 * 16.0 * iters * (double)blocks * WPB;
    printf("%-34s blocks=%5d waves/blk=%d  %8.3f ms  %7.1f TFLOP/s  shader clk %.3f GHz  cyc/MFMA/wave %.1f\n", label, blocks,
           WPB, ms, flops / ms / 1e9, sc / sw * 0.1, sc / blocks / (16.0 * iters));
}

int main()
{
    const int N = 1 << 20;
    std::vector<float> h(N);
    srand(1);
    for (int i = 0; i < N; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *d_rand, *d_zero, *d_out; unsigned long long* d_clk;
    hipMalloc(&d_rand, N * 4); hipMalloc(&d_zero, N * 4); hipMalloc(&d_out, 4096 * 512 * 4); hipMalloc(&d_clk, 4096 * 16);
    hipMemcpy(d_rand, h.data(), N * 4, hipMemcpyHostToDevice);
    hipMemset(d_zero, 0, N * 4);
    const int iters = 4000;
    run<4>("random, 1 wave/SIMD", d_rand, d_out, d_clk, 256, iters);
    run<4>("random, 2 waves/SIMD (2 blk/CU)", d_rand, d_out, d_clk, 512, iters);
    run<4>("random, 4 waves/SIMD", d_rand, d_out, d_clk, 1024, iters);
    run<4>("zeros,  2 waves/SIMD", d_zero, d_out, d_clk, 512, iters);
    run<4>("random, 2 waves/SIMD again", d_rand, d_out, d_clk, 512, iters);
    // sustained: ~150 ms per launch, 5 launches back to back
    run<4>("SUSTAINED 16x16x4 random 2 waves/SIMD", d_rand, d_out, d_clk, 512, 300000);
    run<4>("SUSTAINED 16x16x4 zeros  2 waves/SIMD", d_zero, d_out, d_clk, 512, 300000);
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int it32 = 300000;
        mfma_stream32<<<512, 256>>>(d_rand, d_out, 1000, d_clk); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int k = 0; k < 5; ++k) mfma_stream32<<<512, 256>>>(d_rand, d_out, it32, d_clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        std::vector<unsigned long long> clk(1024);
        hipMemcpy(clk.data(), d_clk, 1024 * 8, hipMemcpyDeviceToHost);
        double sc = 0, sw = 0; for (int i = 0; i < 512; ++i) { sc += clk[2 * i]; sw += clk[2 * i + 1]; }
        const double flops = 2.0 * 32 * 32 * 2 * 8.0 * it32 * 512.0 * 4;
        printf("%-38s %8.3f ms  %7.1f TFLOP/s  shader clk %.3f GHz\n", "SUSTAINED 32x32x2 random 2 waves/SIMD", ms, flops / ms / 1e9, sc / sw * 0.1);
    }
    return 0;
}
