// v_dot2c_f32_bf16 / v_dot2_f32_bf16 as "unpack a bf16 half and add / subtract it" (round 5): is it EXACTLY v_lshlrev / v_and + v_add_f32 (v_sub_f32) for
// every input class the epilogues see, and what does it cost next to MFMAs?     hipcc --offload-arch=gfx950 -O3 -o dot2_probe dot2_probe.hip
//   add_lo(c, r)  = c + bf16_lo(r)      <- v_dot2c_f32_bf16 c, 0x00003f80, r        add_hi: literal 0x3f800000
//   sub_lo(c, h)  = c - bf16_lo(h)      <- v_dot2_f32_bf16 d, h, 0x0000bf80, c      sub_hi: 0xbf800000
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void check(const float* c, const unsigned* r, float* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float cv = c[i];
    const unsigned rv = r[i];
    float a0 = cv, a1 = cv, s0, s1;
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a0) : "v"(0x00003f80u), "v"(rv));
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a1) : "v"(0x3f800000u), "v"(rv));
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(s0) : "v"(rv), "v"(0x0000bf80u), "v"(cv));
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(s1) : "v"(rv), "v"(0xbf800000u), "v"(cv));
    out[4 * i + 0] = a0; out[4 * i + 1] = a1; out[4 * i + 2] = s0; out[4 * i + 3] = s1;
}

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
    float v[8];
    unsigned pk[8];
    for (int i = 0; i < 8; ++i) { v[i] = seed + i + threadIdx.x; pk[i] = 0x3f803f80u + i; asm volatile("" : "+v"(pk[i])); }
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m % 6]) : "v"(a[m]), "v"(b[m & 3]));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int j = (m * K + k) & 7;
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(seed));
                else if (KIND == 1) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v[j]) : "v"(0x00003f80u), "v"(pk[j]));
                else if (KIND == 2) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(v[j]) : "v"(pk[j]), "v"(0x0000bf80u));
                else asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(v[j]) : "v"(pk[j]));
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.f;
    for (int i = 0; i < 6; ++i) s += acc[i].x;
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int K, int KIND>
void run(const char* name, unsigned long long* d)
{
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((cost<K, KIND>), dim3(256), dim3(256), 0, 0, d, iters, 1.0f);
    hipDeviceSynchronize();
    unsigned long long t;
    hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
    printf("%-28s K=%d: %6.1f cycles per MFMA slot (one wave per SIMD)\n", name, K, (double)t / (iters * 8));
}

static float bf(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main()
{
    const int n = 1 << 22;
    float* hc = (float*)malloc(n * 4);
    unsigned* hr = (unsigned*)malloc(n * 4);
    float* ho = (float*)malloc((size_t)n * 16);
    srand(7);
    auto rnd32 = []() { return ((unsigned)rand() << 17) ^ ((unsigned)rand() << 2) ^ (unsigned)rand(); };
    const unsigned special[] = {0u, 0x80000000u, 0x00000001u, 0x80000001u, 0x007fffffu, 0x00800000u, 0x7f7fffffu, 0xff7fffffu, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0x3f800000u, 0xbf800000u, 0x00400000u};
    const int ns = sizeof(special) / 4;
    for (int i = 0; i < n; ++i) {
        unsigned cb = rnd32(), rb = rnd32();
        if (i < ns * ns * 4) { cb = special[i % ns]; const unsigned s2 = special[(i / ns) % ns]; rb = ((i / (ns * ns)) & 1) ? (s2 >> 16) | (rnd32() & 0xffff0000u) : (s2 & 0xffff0000u) | (rnd32() >> 16); if ((i / (ns * ns)) & 2) rb = (s2 >> 16) | (s2 & 0xffff0000u); }
        else if (i & 1) {            // realistic: activation-sized values, the packed pair = two roundings of nearby values
            const float x = ((float)rand() / RAND_MAX - 0.5f) * 8.f, y = ((float)rand() / RAND_MAX - 0.5f) * 8.f;
            memcpy(&cb, &x, 4);
            unsigned xb, yb; memcpy(&xb, &x, 4); memcpy(&yb, &y, 4);
            rb = ((xb + 0x7fffu + ((xb >> 16) & 1u)) >> 16) | ((yb + 0x7fffu + ((yb >> 16) & 1u)) & 0xffff0000u);
        }
        memcpy(&hc[i], &cb, 4); hr[i] = rb;
    }
    float *dc, *dout; unsigned* dr; unsigned long long* dt;
    hipMalloc(&dc, n * 4); hipMalloc(&dr, n * 4); hipMalloc(&dout, (size_t)n * 16); hipMalloc(&dt, 64);
    hipMemcpy(dc, hc, n * 4, hipMemcpyHostToDevice); hipMemcpy(dr, hr, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(n / 256), dim3(256), 0, 0, dc, dr, dout, n);
    hipMemcpy(ho, dout, (size_t)n * 16, hipMemcpyDeviceToHost);
    long bad[4] = {0, 0, 0, 0}, badfinite[4] = {0, 0, 0, 0}, shown = 0;
    for (int i = 0; i < n; ++i) {
        const float lo = bf((unsigned short)(hr[i] & 0xffffu)), hi = bf((unsigned short)(hr[i] >> 16));
        const float ref[4] = {hc[i] + lo, hc[i] + hi, hc[i] - lo, hc[i] - hi};
        for (int k = 0; k < 4; ++k) {
            unsigned a, b; memcpy(&a, &ho[4 * i + k], 4); memcpy(&b, &ref[k], 4);
            const bool bothnan = (ho[4 * i + k] != ho[4 * i + k]) && (ref[k] != ref[k]);
            if (a != b && !bothnan) {
                ++bad[k];
                const bool fin = isfinite(lo) && isfinite(hi) && isfinite(hc[i]);
                if (fin) ++badfinite[k];
                if (fin && shown < 12) { unsigned cb; memcpy(&cb, &hc[i], 4); printf("  mismatch k=%d c=%08x r=%08x got=%08x want=%08x\n", k, cb, hr[i], a, b); ++shown; }
            }
        }
    }
    printf("exactness over %d cases (incl. %d special): mismatches add_lo %ld add_hi %ld sub_lo %ld sub_hi %ld; with all inputs finite: %ld %ld %ld %ld\n", n, ns * ns * 4, bad[0], bad[1], bad[2], bad[3],
           badfinite[0], badfinite[1], badfinite[2], badfinite[3]);
    run<0, 0>("mfma only", dt);
    run<1, 0>("v_add_f32", dt); run<2, 0>("v_add_f32", dt); run<3, 0>("v_add_f32", dt); run<4, 0>("v_add_f32", dt);
    run<1, 1>("v_dot2c_f32_bf16 (literal)", dt); run<2, 1>("v_dot2c_f32_bf16 (literal)", dt); run<3, 1>("v_dot2c_f32_bf16 (literal)", dt); run<4, 1>("v_dot2c_f32_bf16 (literal)", dt);
    run<1, 2>("v_dot2_f32_bf16", dt); run<2, 2>("v_dot2_f32_bf16", dt); run<3, 2>("v_dot2_f32_bf16", dt); run<4, 2>("v_dot2_f32_bf16", dt);
    run<2, 3>("v_lshlrev_b32", dt); run<4, 3>("v_lshlrev_b32", dt);
    return 0;
}
