// Feasibility probe for a Winograd F(2x2, 3x3) fp32 convolution on v_mfma_f32_16x16x4_f32: how busy does the
// matrix pipe stay in the TRANSFORMED-DOMAIN GEMM stage, where operand reuse is low?  No global traffic, no real transforms: the 16
// position GEMMs [64 cout x 8 cin] x [8 cin x 64 tiles] of one K chunk run from static LDS images (U: weights, V: input), which is
// the part a direct convolution does not have.  Two blockings of the 16 x 4 x 4 (position, cout tile, tile group) accumulators:
//   A  8 waves per CU (2 per SIMD), 128 accumulator VGPRs: a wave owns all 16 positions of (1 cout tile, 2 tile groups):
//      per position 1 A + 2 B ds_read_b64 for 4 MFMAs
//   B  4 waves per CU (1 per SIMD), 256 accumulator VGPRs: all 16 positions of (2 cout tiles, 2 tile groups):
//      per position 2 A + 2 B ds_read_b64 for 8 MFMAs
// XFORM = 1 adds, per stage, the LDS traffic and VALU work of the input transform (raw reads, 32 adds per (tile, channel), writes of
// the 32 KB V image) spread over the MFMA stream.
//   hipcc --offload-arch=gfx950 -O3 -o wino_probe tools/wino_probe.hip && ./wino_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int POS = 16;
constexpr int IMG = POS * 4 * 64 * 8;       // bytes of U (or V): [pos][tile of 16][lane][2 floats]

template <int WAVES, int XFORM>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void wino_gemm(const float* in, float* out, int stages, unsigned long long* clk)
{
    constexpr int NCT = WAVES == 8 ? 1 : 2;      // cout tiles per wave
    constexpr int NTG = 2;                        // tile groups per wave
    __shared__ __attribute__((aligned(16))) char smem[2 * IMG + 16384];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < (2 * IMG + 16384) / 4; i += WAVES * 64) reinterpret_cast<float*>(smem)[i] = in[i & 0xFFFF];
    __syncthreads();
    const int ct0 = WAVES == 8 ? (wv & 3) : (wv & 1) * 2;
    const int tg0 = WAVES == 8 ? (wv >> 2) * 2 : (wv >> 1) * 2;
    const char* U = smem + lane * 8;
    const char* V = smem + IMG + lane * 8;
    float* raw = reinterpret_cast<float*>(smem + 2 * IMG);

    f32x4 acc[POS][NCT][NTG];
#pragma unroll
    for (int p = 0; p < POS; ++p)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int t = 0; t < NTG; ++t) acc[p][c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned long long c0 = clock64();
    for (int s = 0; s < stages; ++s) {
        f32x2 a[2][NCT], b[2][NTG];
        auto load = [&](int slot, int p) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) a[slot][c] = *reinterpret_cast<const f32x2*>(U + (p * 4 + ct0 + c) * 512);
#pragma unroll
            for (int t = 0; t < NTG; ++t) b[slot][t] = *reinterpret_cast<const f32x2*>(V + (p * 4 + tg0 + t) * 512);
        };
        load(0, 0);
        f32x4 d[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int p = 0; p < POS; ++p) {
            const int cs = p & 1;
            if (p + 1 < POS) load(cs ^ 1, p + 1);
            if (XFORM) {
                // one sixteenth of this wave's share of the input transform per position: 2 (tile, channel) patches per lane and stage
                // = 32 raw reads, 64 adds, 32 writes per lane and stage (8 waves; twice that with 4) -> per position 2 b64 reads,
                // 8 adds, 2 b64 writes (x2 for 4 waves)
#pragma unroll
                for (int rep = 0; rep < (WAVES == 8 ? 1 : 2); ++rep) {
                    const int it = (p * 2 + rep) & 15;
                    const f32x2 r0 = *reinterpret_cast<const f32x2*>(raw + (it * 128 + lane * 2));
                    const f32x2 r1 = *reinterpret_cast<const f32x2*>(raw + (it * 128 + 2048 + lane * 2));
                    d[0].x = r0.x - r1.x + d[0].x * 0.5f; d[0].y = r0.y + r1.y - d[0].y;
                    d[0].z = r1.x - r0.y + d[0].z;        d[0].w = r0.x - r1.y + d[0].w * 0.25f;
                    *reinterpret_cast<f32x2*>(smem + IMG + (p * 4 + (wv & 3)) * 512 + lane * 8) = f32x2{d[0].x, d[0].y};
                    *reinterpret_cast<f32x2*>(smem + IMG + (p * 4 + ((wv + 1) & 3)) * 512 + lane * 8) = f32x2{d[0].z, d[0].w};
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int c = 0; c < NCT; ++c)
#pragma unroll
                    for (int t = 0; t < NTG; ++t)
                        acc[p][c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cs][c][j], b[cs][t][j], acc[p][c][t], 0, 0, 0);
        }
        if (XFORM) __syncthreads();              // the stage hand-over a real kernel needs (V is rewritten every stage)
    }
    const unsigned long long c1 = clock64();
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < POS; ++p)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int t = 0; t < NTG; ++t) sum += acc[p][c][t];
    out[blockIdx.x * WAVES * 64 + tid] = sum.x + sum.y + sum.z + sum.w;
    if (tid == 0) clk[blockIdx.x] = c1 - c0;
}

template <int WAVES, int XFORM>
void run(const char* name, const float* in, float* out, unsigned long long* clk, int blocks)
{
    const int stages = 4096;
    hipLaunchKernelGGL((wino_gemm<WAVES, XFORM>), dim3(blocks), dim3(WAVES * 64), 0, 0, in, out, 64, clk);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((wino_gemm<WAVES, XFORM>), dim3(blocks), dim3(WAVES * 64), 0, 0, in, out, stages, clk);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), clk, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double cyc = 0;
    for (auto v : h) cyc += (double)v;
    cyc /= blocks;
    // MFMAs per SIMD and stage: 512 per CU / 4 SIMDs = 128, 32 cycles each (8 passes x 4)
    const double ideal = 128.0 * 32.0 * stages;
    const double flops = 2.0 * 16 * 64 * 64 * 8 * (double)stages * blocks;       // 16 positions x (64 x 64 x 8) MACs
    const hipError_t e = hipGetLastError();
    printf("%-34s %8.3f ms  %7.1f TFLOP/s (transformed domain; x2.25 = direct-conv equivalent %7.1f)  MFMA pipe %.3f busy%s\n", name, ms,
           flops / ms / 1e9, 2.25 * flops / ms / 1e9, ideal / cyc, e == hipSuccess ? "" : "  (HIP ERROR)");
}

int main()
{
    float *in, *out;
    unsigned long long* clk;
    hipMalloc(&in, 65536 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8);
    std::vector<float> h(65536);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    run<8, 0>("A: 8 waves, 128 acc VGPRs", in, out, clk, 256);
    run<4, 0>("B: 4 waves, 256 acc VGPRs", in, out, clk, 256);
    run<8, 1>("A + input-transform traffic", in, out, clk, 256);
    run<4, 1>("B + input-transform traffic", in, out, clk, 256);
    return 0;
}
