// esr_esa.hip -- the ESA (enhanced spatial attention) kernels of libesr_hip.so.  Interface: include/esr_hip.h.
//
// ESA.forward (models/rfdn_baseline/block.py:117-129, models/team04_rlfn.py:76-89):
//     c1_ = conv1(x)                      1x1 C->f           -> conv_f32_kernel (esr_hip.hip)
//     c1  = conv2(c1_)                    3x3 s2 p0 f->f     -> conv3x3s2_kernel
//     v   = max_pool2d(c1, 7, 3)                             -> maxpool7s3_kernel
//     c3  = conv stack on v (41x41)       3x3 p1 f->f        -> conv_f32_kernel
//     y   = x * sigmoid(conv4(bilinear(c3) + conv_f(c1_)))   -> esa_apply_kernel (one full-res pass)
// All three are memory/latency-bound VALU kernels (<= 1 % of a network's MACs); the f-wide maps are
// NHWC with pitch 16 and zero pad channels, so every access is a float4.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include "esr_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int FP = ESR_ESA_FP;   // 16

// Element access for the full-resolution views: ST = esr_storage (0 fp32, 1 bf16, 2 fp16); idx counts elements and is a
// multiple of 4.  Arithmetic is fp32 in every case; a store rounds once (RNE).
template <int ST>
__device__ __forceinline__ f32x4 ld4(const void* base, size_t idx)
{
    if (ST == ESR_STORE_F32) return *reinterpret_cast<const f32x4*>(static_cast<const float*>(base) + idx);
    const uint2 u = *reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(base) + idx);
    f32x4 v;
    if (ST == ESR_STORE_BF16) {
        v.x = __builtin_bit_cast(float, u.x << 16); v.y = __builtin_bit_cast(float, u.x & 0xffff0000u);
        v.z = __builtin_bit_cast(float, u.y << 16); v.w = __builtin_bit_cast(float, u.y & 0xffff0000u);
    } else {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 a = __builtin_bit_cast(h2, u.x), b = __builtin_bit_cast(h2, u.y);
        v.x = (float)a[0]; v.y = (float)a[1]; v.z = (float)b[0]; v.w = (float)b[1];
    }
    return v;
}

template <int ST>
__device__ __forceinline__ void st4(void* base, size_t idx, f32x4 v)
{
    if (ST == ESR_STORE_F32) {
        *reinterpret_cast<f32x4*>(static_cast<float*>(base) + idx) = v;
        return;
    }
    uint2 u;
    if (ST == ESR_STORE_BF16) {
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        b2 a, b;
        a[0] = (__bf16)v.x; a[1] = (__bf16)v.y; b[0] = (__bf16)v.z; b[1] = (__bf16)v.w;
        u.x = __builtin_bit_cast(unsigned, a); u.y = __builtin_bit_cast(unsigned, b);
    } else {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 a, b;
        a[0] = (_Float16)v.x; a[1] = (_Float16)v.y; b[0] = (_Float16)v.z; b[1] = (_Float16)v.w;
        u.x = __builtin_bit_cast(unsigned, a); u.y = __builtin_bit_cast(unsigned, b);
    }
    *reinterpret_cast<uint2*>(static_cast<unsigned short*>(base) + idx) = u;
}

// ---- 3x3 stride 2, no padding, FP -> FP --------------------------------------------------------
// thread = (output pixel, quad of 4 output channels); weights [tap][cin][cout] in LDS.
template <int ST>
__global__ __launch_bounds__(256) void conv3x3s2_kernel(const void* __restrict__ x, const float* __restrict__ wp,
                                                        float* __restrict__ y, int N, int H, int W, int Ho, int Wo)
{
    __shared__ __attribute__((aligned(16))) float sw[9 * FP * FP + FP];
    for (int i = threadIdx.x; i < 9 * FP * FP + FP; i += 256) sw[i] = wp[i];
    __syncthreads();
    const int q = threadIdx.x & 3;
    const long long pix = (long long)blockIdx.x * 64 + (threadIdx.x >> 2);
    if (pix >= (long long)N * Ho * Wo) return;
    const int ox = (int)(pix % Wo);
    const int oy = (int)((pix / Wo) % Ho);
    const int n = (int)(pix / ((long long)Wo * Ho));
    f32x4 acc = *reinterpret_cast<const f32x4*>(sw + 9 * FP * FP + q * 4);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const size_t xi = (((size_t)n * H + (oy * 2 + ky)) * W + (ox * 2 + kx)) * FP;
            const float* wt = sw + (ky * 3 + kx) * FP * FP + q * 4;
#pragma unroll
            for (int cq = 0; cq < FP / 4; ++cq) {
                const f32x4 xv = ld4<ST>(x, xi + cq * 4);
                acc += esr_lone(xv.x) * *reinterpret_cast<const f32x4*>(wt + (cq * 4 + 0) * FP);     // esr_lone: see esr_internal.h
                acc += esr_lone(xv.y) * *reinterpret_cast<const f32x4*>(wt + (cq * 4 + 1) * FP);
                acc += esr_lone(xv.z) * *reinterpret_cast<const f32x4*>(wt + (cq * 4 + 2) * FP);
                acc += esr_lone(xv.w) * *reinterpret_cast<const f32x4*>(wt + (cq * 4 + 3) * FP);
            }
        }
    *reinterpret_cast<f32x4*>(y + (size_t)pix * FP + q * 4) = acc;
}

// ---- max pool 7x7 stride 3, floor mode, no padding ---------------------------------------------
__global__ __launch_bounds__(256) void maxpool7s3_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H,
                                                         int W, int Ho, int Wo)
{
    const int q = threadIdx.x & 3;
    const long long pix = (long long)blockIdx.x * 64 + (threadIdx.x >> 2);
    if (pix >= (long long)N * Ho * Wo) return;
    const int ox = (int)(pix % Wo);
    const int oy = (int)((pix / Wo) % Ho);
    const int n = (int)(pix / ((long long)Wo * Ho));
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int ky = 0; ky < 7; ++ky) {
        const float* row = x + (((size_t)n * H + (oy * 3 + ky)) * W + ox * 3) * FP + q * 4;
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + kx * FP);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    *reinterpret_cast<f32x4*>(y + (size_t)pix * FP + q * 4) = m;
}

// ---- fused ESA tail ------------------------------------------------------------------------------
// 16 lanes per pixel (4 pixels per wave, 16 per block).  Lane g first builds ONE element of the FP-wide vector
// s = bilinear(c3) + conv_f(c1_) of its pixel (16 FMAs + one 4-tap lerp), the 16 lanes exchange s through 1 KB
// of LDS, then lane g produces output channels 4g..4g+3 (16 x 4 FMAs, sigmoid, multiply).  x is read once and
// y written once, both as contiguous float4 per lane.
struct EsaK {
    const void* x; const void* c1; const float* c3; const float* wf; const float* w4; void* y;
    int x_pitch, x_coff, y_pitch, y_coff;
    int N, H, W, Cp4, cp, h3, w3;
    float sh, sw;
    // post chain (esr_esa_desc.post_w / post[]: 16-bit storage, esa_apply_mfma_kernel<.., NP0 > 0, ..>)
    const unsigned short* pimg;
    const void* pres; void* pout0; void* pout1;
    int pres_pitch, pres_coff, pout0_pitch, pout0_coff, pout1_pitch, pout1_coff;
    int p0_c8, p1_c8;           // channels stored (multiples of 8)
    int p0_act, p1_act, p0_res;
    float p0_slope, p1_slope;
    int skip_y;
};

template <int ST>
__global__ __launch_bounds__(256) void esa_apply_kernel(const EsaK p)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];      // wf: FP*FP + FP ; w4: FP*cp + cp ; s: 16*FP
    const int nwf = FP * FP + FP, nw4 = FP * p.cp + p.cp;
    for (int i = threadIdx.x; i < nwf; i += 256) sm[i] = p.wf[i];
    for (int i = threadIdx.x; i < nw4; i += 256) sm[nwf + i] = p.w4[i];
    float* sx = sm + nwf + ((nw4 + 3) & ~3);
    __syncthreads();
    const int g = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const long long npix = (long long)p.N * p.H * p.W;
    const long long ngroups = (npix + 15) / 16;
    // grid-stride over groups of 16 pixels: the 4.3 KB of weights above are fetched once per block, not once per 16
    // pixels.  The 16 lanes of a pixel sit in one wave and sx[pl] is private to them, so the exchange needs no block
    // barrier: LDS runs a wave's accesses in order, the fences only pin the compiler.
    for (long long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const long long pix = min(grp * 16 + pl, npix - 1);                         // clamp: every lane takes part in the exchange
    const bool live = grp * 16 + pl < npix && g * 4 < p.Cp4;
    // block-uniform 32-bit division of the group's first pixel (scalar unit), a carry per lane (see esa_apply_mfma_kernel)
    const unsigned pix0 = (unsigned)grp * 16u;
    const unsigned row0 = pix0 / (unsigned)p.W;
    const unsigned n0 = row0 / (unsigned)p.H;
    int ox = (int)(pix0 - row0 * (unsigned)p.W) + pl;
    int oy = (int)(row0 - n0 * (unsigned)p.H);
    int n = (int)n0;
    for (int wrap = 0; wrap < 2; ++wrap)
        if (ox >= p.W) {
            ox -= p.W;
            if (++oy == p.H) { oy = 0; ++n; }
        }
    if (grp * 16 + pl >= npix) { ox = p.W - 1; oy = p.H - 1; n = p.N - 1; }

    // bilinear source coordinates: ATen area_pixel_compute_source_index, fused multiply-add (see oracle)
    float fy = fmaf((float)oy + 0.5f, p.sh, -0.5f);
    fy = fy < 0.f ? 0.f : fy;
    const int y0 = (int)fy, y1 = y0 + (y0 < p.h3 - 1 ? 1 : 0);
    const float ly = fy - (float)y0, hy = 1.f - ly;
    float fx = fmaf((float)ox + 0.5f, p.sw, -0.5f);
    fx = fx < 0.f ? 0.f : fx;
    const int x0 = (int)fx, x1 = x0 + (x0 < p.w3 - 1 ? 1 : 0);
    const float lx = fx - (float)x0, hx = 1.f - lx;
    const float* cb = p.c3 + (size_t)n * p.h3 * p.w3 * FP + g;
    const float a = cb[((size_t)y0 * p.w3 + x0) * FP], b = cb[((size_t)y0 * p.w3 + x1) * FP];
    const float c = cb[((size_t)y1 * p.w3 + x0) * FP], d = cb[((size_t)y1 * p.w3 + x1) * FP];
    float sg = hy * (hx * a + lx * b) + ly * (hx * c + lx * d) + sm[FP * FP + g];
    // + conv_f(c1_)[g] = sum_i c1[i] * Wf[i][g]
#pragma unroll
    for (int iq = 0; iq < FP / 4; ++iq) {
        const f32x4 cv = ld4<ST>(p.c1, (size_t)pix * FP + iq * 4);
        sg = fmaf(cv.x, sm[(iq * 4 + 0) * FP + g], sg);
        sg = fmaf(cv.y, sm[(iq * 4 + 1) * FP + g], sg);
        sg = fmaf(cv.z, sm[(iq * 4 + 2) * FP + g], sg);
        sg = fmaf(cv.w, sm[(iq * 4 + 3) * FP + g], sg);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    sx[pl * FP + g] = sg;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!live) continue;
    // conv4 for this lane's 4 channels, sigmoid, multiply
    const float* w4 = sm + nwf;
    f32x4 m = *reinterpret_cast<const f32x4*>(w4 + FP * p.cp + g * 4);
#pragma unroll
    for (int iq = 0; iq < FP / 4; ++iq) {
        const f32x4 sv = *reinterpret_cast<const f32x4*>(sx + pl * FP + iq * 4);
        m += esr_lone(sv.x) * *reinterpret_cast<const f32x4*>(w4 + (iq * 4 + 0) * p.cp + g * 4);     // esr_lone: see esr_internal.h
        m += esr_lone(sv.y) * *reinterpret_cast<const f32x4*>(w4 + (iq * 4 + 1) * p.cp + g * 4);
        m += esr_lone(sv.z) * *reinterpret_cast<const f32x4*>(w4 + (iq * 4 + 2) * p.cp + g * 4);
        m += esr_lone(sv.w) * *reinterpret_cast<const f32x4*>(w4 + (iq * 4 + 3) * p.cp + g * 4);
    }
    const f32x4 xv = ld4<ST>(p.x, (size_t)pix * p.x_pitch + p.x_coff + g * 4);
    f32x4 o;
    o.x = xv.x * (1.f / (1.f + expf(-m.x)));
    o.y = xv.y * (1.f / (1.f + expf(-m.y)));
    o.z = xv.z * (1.f / (1.f + expf(-m.z)));
    o.w = xv.w * (1.f / (1.f + expf(-m.w)));
    st4<ST>(p.y, (size_t)pix * p.y_pitch + p.y_coff + g * 4, o);
    }
}

// ---- fused ESA tail on 16-bit storage: the two 1x1 convolutions on the matrix cores ------------------------------------
// One wave = 16 pixels per step.  S = Wf^T . c1 + bf + bilinear(c3) is ONE v_mfma_f32_16x16x32 (B operand = the 16 bytes of
// c1 as stored; the upper 16 k slots carry the low parts of Wf: w = hi + lo), its D fragment -- lane (px, kq): s[4kq .. 4kq+3]
// of pixel px, fp32 -- becomes the B operand of conv4 WITHOUT leaving the lane: k slots (kq, 0..3) = the 16-bit high parts
// of those four values, (kq, 4..7) = their low parts (s = hi + lo keeps fp32 accuracy); per 16-channel output tile two
// MFMAs (W4 high parts x s, W4 low parts x s_hi).  x is read and y written in the D layout (8 bytes per lane).
// The VALU version above spends its time on LDS weight reads (0.31 ms per launch at batch 32 against 0.12 ms of memory time).
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

template <int ST>
__device__ __forceinline__ unsigned short to16(float v)
{
    if (ST == ESR_STORE_BF16) return __builtin_bit_cast(unsigned short, (__bf16)v);
    return __builtin_bit_cast(unsigned short, (_Float16)v);
}
template <int ST>
__device__ __forceinline__ float from16(unsigned short h)
{
    if (ST == ESR_STORE_BF16) return __builtin_bit_cast(float, (unsigned)h << 16);
    return (float)__builtin_bit_cast(_Float16, h);
}
template <int ST>
__device__ __forceinline__ f32x4 mfma_k32(i32x4_t a, i32x4_t b, f32x4 c)
{
    if (ST == ESR_STORE_BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// Channel <-> MFMA row map of conv4: tile t, row i (= 4 kq + r) computes channel 32 (t >> 1) + 8 kq + 4 (t & 1) + r, so a lane's
// results of a tile PAIR are 8 CONSECUTIVE channels: x is read and y written as 16 bytes per lane, 64 contiguous bytes per
// pixel and instruction (the natural 16 t + 4 kq + r map gives 8-byte pieces, 32 bytes per pixel: twice the requests).
//
// Post chain (NP0 / NP1 = channel pairs of tiles of post 0 / post 1, esr_esa_desc.post[]): the packed 16-bit result of a tile pair
// -- 8 consecutive channels per lane -- IS the B operand of the next 1x1's K block of 32 channels (k slot (kq, j) = channel
// 32 q + 8 kq + j), so post 0 runs on the values a separate launch would have read back from memory; its D fragments (same row
// map, so again 8 consecutive channels per lane and tile pair) feed post 1 as fp32 high + low parts.  RFDN: y + the next block's
// c1_d; BSRN: conv_out (+ block input) and the next block's c1_d, the attention output itself never reaches memory.
template <int ST, int NP, int NP0 = 0, int NP1 = 0>     // NP = channel pairs of tiles = ceil(C / 32)
// Occupancy target: four waves per SIMD (<= 128 registers).  Without it hipcc takes 132 for the plain kernel once both groups' loads are
// unconditional -- three waves -- and the launch is 5-10 % slower (round 5 A/B, tools/r05/l_esa_fetch.sh: targets none / 3 / 4).  The variant
// with two post stages on bf16 needs 160 whatever the target (hipcc says so); fp16's meets it with two spilled registers.
#ifndef ESA_APPLY_WAVES
#define ESA_APPLY_WAVES 4
#endif
__global__ __launch_bounds__(256, ESA_APPLY_WAVES) void esa_apply_mfma_kernel(const EsaK p)
{
#pragma clang fp contract(off)
    constexpr int NT = 2 * NP;
    constexpr int NT0 = 2 * NP0, NT1 = 2 * NP1;
    constexpr int LO1 = ST == ESR_STORE_BF16 ? 2 : 1;                        // post 1: high (+ low) weight images; post 0: always both
    constexpr int P0_IMGS = 2 * NT0 * NP, P1_IMGS = LO1 * NT1 * NT0;
    __shared__ __attribute__((aligned(16))) unsigned short spimg[(P0_IMGS + P1_IMGS > 0 ? P0_IMGS + P1_IMGS : 1) * 512];
    __shared__ __attribute__((aligned(16))) float spbias[(NT0 + NT1 > 0 ? NT0 + NT1 : 1) * 16];
    // Prologue (round 5): every global load of the block's weight images is issued BEFORE the first one is waited for.  As plain loops
    // (for e = tid; e < n; e += 256) hipcc kept one load in flight -- 18 + 6 dependent L2 round trips, 3-7 us in front of a block whose
    // waves then work for 8-16 us: a fifth of a single image's launch.  The trip counts are compile-time constants, the image index of
    // a pass too (512 elements per image, 256 threads: pass i builds half of image i / 2).
    constexpr int PCOPY = ((P0_IMGS + P1_IMGS) * 64 + 255) / 256;
    i32x4_t pcp[PCOPY > 0 ? PCOPY : 1];
    float pbias = 0.f;
    if (NP0 > 0) {
        // esr_pack_apply_post blob: the images in this order, then the biases
        const i32x4_t* src = reinterpret_cast<const i32x4_t*>(p.pimg);
#pragma unroll
        for (int i = 0; i < PCOPY; ++i) {
            const int e = (int)threadIdx.x + 256 * i;
            pcp[i] = src[e < (P0_IMGS + P1_IMGS) * 64 ? e : 0];
        }
        const float* bsrc = reinterpret_cast<const float*>(p.pimg + (size_t)(P0_IMGS + P1_IMGS) * 512);
        static_assert((NT0 + NT1) * 16 <= 256, "one bias per thread");
        pbias = bsrc[(int)threadIdx.x < (NT0 + NT1) * 16 ? threadIdx.x : 0];
    }
    // LDS: A images, lane-linear 16 bytes per lane: [Wf][W4 hi x NT][W4 lo x NT], then bf[16] and b4[NT*16] as floats
    __shared__ __attribute__((aligned(16))) unsigned short simg[(1 + 2 * NT) * 64 * 8];
    __shared__ __attribute__((aligned(16))) float sbias[16 + NT * 16];
    const int tid = threadIdx.x;
    constexpr int NPASS = (1 + 2 * NT) * 2;
    float wraw[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int e = tid + 256 * ps;
        const int img = ps >> 1, l = (e >> 3) & 63, j = e & 7;
        const int i = l & 15, kq = l >> 4;
        if (img == 0) {
            // conv_f: k slot (kq, j) = input channel 8 (kq & 1) + j, high part for kq < 2, low part for kq >= 2
            wraw[ps] = p.wf[(8 * (kq & 1) + j) * FP + i];
        } else {
            const int t = (img - 1) % NT;
            const int oc = 32 * (t >> 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3);
            // (clamped address + select: a load under a lane mask is a branch, and hipcc drains the load queue in front of each)
            const float w = p.w4[(4 * kq + (j & 3)) * p.cp + (oc < p.cp ? oc : p.cp - 1)];
            wraw[ps] = oc < p.cp ? w : 0.f;
        }
    }
    static_assert(NT * 16 <= 240, "one bias per thread");
    float braw = 0.f;
    {
        const int e = tid < 16 ? 0 : (tid < 16 + NT * 16 ? tid - 16 : 0), t = e >> 4, i = e & 15;
        const int oc = 32 * (t >> 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3);
        const float* const src = tid < 16 ? p.wf + FP * FP + tid : p.w4 + FP * p.cp + (oc < p.cp ? oc : p.cp - 1);
        braw = *src;
        braw = (tid < 16 || oc < p.cp) ? braw : 0.f;
    }
    if (NP0 > 0) {
#pragma unroll
        for (int i = 0; i < PCOPY; ++i) {
            const int e = tid + 256 * i;
            if (e < (P0_IMGS + P1_IMGS) * 64) reinterpret_cast<i32x4_t*>(spimg)[e] = pcp[i];
        }
        if (tid < (NT0 + NT1) * 16) spbias[tid] = pbias;
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int e = tid + 256 * ps;
        const int img = ps >> 1, l = (e >> 3) & 63, j = e & 7;
        const int kq = l >> 4;
        const float w = wraw[ps];
        const float hi = from16<ST>(to16<ST>(w));
        float v;
        if (img == 0) v = kq < 2 ? hi : w - hi;
        else {
            const int lo = (img - 1) / NT;
            v = lo ? (j < 4 ? w - hi : 0.f) : hi;           // hi image: against s_hi (j < 4) and s_lo (j >= 4); lo image: against s_hi only
        }
        simg[e] = to16<ST>(v);
    }
    if (tid < 16 + NT * 16) sbias[tid] = braw;
    __syncthreads();

    const int lane = tid & 63, wv = tid >> 6;
    const int px = lane & 15, kq = lane >> 4;
    const i32x4_t a_f = *reinterpret_cast<const i32x4_t*>(simg + lane * 8);
    const f32x4 bf4 = *reinterpret_cast<const f32x4*>(sbias + kq * 4);
    const long long npix = (long long)p.N * p.H * p.W;
    const long long ngroups = (npix + 15) / 16;
    const unsigned short* c1 = static_cast<const unsigned short*>(p.c1);
    const unsigned short* xs = static_cast<const unsigned short*>(p.x);
    unsigned short* ys = static_cast<unsigned short*>(p.y);
    const long long gstep = (long long)gridDim.x * 4;
    bool chan_ok[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) chan_ok[q] = 32 * q + 8 * kq < p.Cp4;
    const unsigned short* rs = static_cast<const unsigned short*>(p.pres);
    unsigned short* o0 = static_cast<unsigned short*>(p.pout0);
    unsigned short* o1 = static_cast<unsigned short*>(p.pout1);

    // everything a group reads, requested one group ahead of its arithmetic
    struct Grp {
        long long pix;
        bool live;
        i32x4_t bc1;
        i32x4_t xv[NP];
        i32x4_t rv[NP0 > 0 ? NP0 : 1];          // residual of post 0 (8 channels per pair)
        f32x4 ta, tb, tc, td;
        float ly, lx;
    };
    auto fetch = [&](long long grp, Grp& g) __attribute__((always_inline)) {
        // the 16 pixels of a group are consecutive: ONE wave-uniform 32-bit division chain per group (scalar unit), a carry per lane
        const unsigned pix0 = (unsigned)grp * 16u;                       // npix < 2^31 (host)
        const unsigned row0 = pix0 / (unsigned)p.W;
        const unsigned n0 = row0 / (unsigned)p.H;
        int ox = (int)(pix0 - row0 * (unsigned)p.W) + px;
        int oy = (int)(row0 - n0 * (unsigned)p.H);
        int n = (int)n0;
        if (ox >= p.W) {                                                 // W >= 15 here: at most two wraps
            ox -= p.W;
            if (++oy == p.H) { oy = 0; ++n; }
        }
        if (ox >= p.W) {
            ox -= p.W;
            if (++oy == p.H) { oy = 0; ++n; }
        }
        const long long pixr = (long long)pix0 + px;
        g.live = pixr < npix;
        g.pix = pixr;
        if (!g.live) { g.pix = npix - 1; ox = p.W - 1; oy = p.H - 1; n = p.N - 1; }
        // B operand of conv_f straight from memory: 8 channels (16 bytes) of c1, lanes kq >= 2 read the same bytes again
        g.bc1 = *reinterpret_cast<const i32x4_t*>(c1 + (size_t)g.pix * FP + 8 * (kq & 1));
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            // UNCONDITIONAL (round 5): a lane past the last channel reads the pixel's first 16 bytes instead and finish() zeroes what it got.
            // As `zero; if (chan_ok) load` each load sat in a branch of its own, and hipcc waited for the group's FIRST loads before it issued
            // the second group's -- the two groups of an iteration were fetched one after the other, not together.
            g.xv[q] = *reinterpret_cast<const i32x4_t*>(xs + (size_t)g.pix * p.x_pitch + p.x_coff + (chan_ok[q] ? 32 * q + 8 * kq : 0));
        }
        if (NP0 > 0) {
#pragma unroll
            for (int q = 0; q < (NP0 > 0 ? NP0 : 1); ++q) {
                // (the residual likewise: without one, rs = the block input x -- any readable address, the value is dropped in finish())
                const bool rok = 32 * q + 8 * kq < p.p0_c8;
                g.rv[q] = *reinterpret_cast<const i32x4_t*>((p.p0_res ? rs + (size_t)g.pix * p.pres_pitch + p.pres_coff : xs + (size_t)g.pix * p.x_pitch + p.x_coff) + (rok ? 32 * q + 8 * kq : 0));
            }
        }
        // bilinear source coordinates: ATen area_pixel_compute_source_index, fused multiply-add (see oracle)
        float fy = fmaf((float)oy + 0.5f, p.sh, -0.5f);
        fy = fy < 0.f ? 0.f : fy;
        const int y0 = (int)fy, y1 = y0 + (y0 < p.h3 - 1 ? 1 : 0);
        g.ly = fy - (float)y0;
        float fx = fmaf((float)ox + 0.5f, p.sw, -0.5f);
        fx = fx < 0.f ? 0.f : fx;
        const int x0 = (int)fx, x1 = x0 + (x0 < p.w3 - 1 ? 1 : 0);
        g.lx = fx - (float)x0;
        const float* cb = p.c3 + (size_t)n * p.h3 * p.w3 * FP + kq * 4;
        g.ta = *reinterpret_cast<const f32x4*>(cb + ((size_t)y0 * p.w3 + x0) * FP);
        g.tb = *reinterpret_cast<const f32x4*>(cb + ((size_t)y0 * p.w3 + x1) * FP);
        g.tc = *reinterpret_cast<const f32x4*>(cb + ((size_t)y1 * p.w3 + x0) * FP);
        g.td = *reinterpret_cast<const f32x4*>(cb + ((size_t)y1 * p.w3 + x1) * FP);
    };
    auto sigmoid = [](float m) __attribute__((always_inline)) {
        // 1 / (1 + 2^(-m log2 e)) on v_exp_f32 / v_rcp_f32: ~1 ulp, far below the 16-bit rounding of the stored product
        return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * m));
    };
    auto finish = [&](const Grp& g) __attribute__((always_inline)) {
        // The four interpolation weights in registers of their own (esr_lone, esr_internal.h): when lx and ly travel as a register PAIR
        // (every loop shape that carried a fetched group into the next iteration did that) hipcc encodes `lx * tb` as v_pk_mul_f32 ...
        // op_sel:[0,1], the encoding that returns 0 in lanes 48..63 beside another kernel's MFMAs -- the overlapped-forward defect of
        // round 3 (LAB_NOTES.md).
        const float ly_ = esr_lone(g.ly), lx_ = esr_lone(g.lx);
        const float hy = esr_lone(1.f - ly_), hx = esr_lone(1.f - lx_);
        f32x4 sacc = hy * (hx * g.ta + lx_ * g.tb) + ly_ * (hx * g.tc + lx_ * g.td) + bf4;        // same evaluation order as the VALU kernel
        sacc = mfma_k32<ST>(a_f, g.bc1, sacc);
        // s -> B operand of conv4: high parts in k slots 0..3, low parts in 4..7
        unsigned short h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = to16<ST>(sacc[j]);
            l[j] = to16<ST>(sacc[j] - from16<ST>(h[j]));
        }
        i32x4_t bs;
        bs.x = (int)((unsigned)h[0] | ((unsigned)h[1] << 16)); bs.y = (int)((unsigned)h[2] | ((unsigned)h[3] << 16));
        bs.z = (int)((unsigned)l[0] | ((unsigned)l[1] << 16)); bs.w = (int)((unsigned)l[2] | ((unsigned)l[3] << 16));
        i32x4_t yv[NP];                  // the result as stored: 8 consecutive channels per lane and pair
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            f32x4 m0 = *reinterpret_cast<const f32x4*>(sbias + 16 + (2 * q) * 16 + kq * 4);
            f32x4 m1 = *reinterpret_cast<const f32x4*>(sbias + 16 + (2 * q + 1) * 16 + kq * 4);
            {
                // conv4's images are re-read from LDS for every group (made loop-variant on purpose): as loop-invariant registers they are
                // 32 VGPRs that decide between three and four waves per SIMD (plain kernel: 154 -> ~120 registers).  The build of round 3's
                // first half that did this for the plain kernel was withdrawn because forwards overlapping on several HIP streams differed
                // from serial ones -- that turned out to be the loop that carried a fetched group over its back edge (below, DESIGN.md
                // section 8), not the occupancy.
                int sio = lane * 8;
                asm volatile("" : "+v"(sio));
                const unsigned short* const sim = simg + sio;
                m0 = mfma_k32<ST>(*reinterpret_cast<const i32x4_t*>(sim + (1 + 2 * q) * 512), bs, m0);
                m1 = mfma_k32<ST>(*reinterpret_cast<const i32x4_t*>(sim + (1 + 2 * q + 1) * 512), bs, m1);
                m0 = mfma_k32<ST>(*reinterpret_cast<const i32x4_t*>(sim + (1 + NT + 2 * q) * 512), bs, m0);
                m1 = mfma_k32<ST>(*reinterpret_cast<const i32x4_t*>(sim + (1 + NT + 2 * q + 1) * 512), bs, m1);
            }
            const i32x4_t xq = chan_ok[q] ? g.xv[q] : i32x4_t{0, 0, 0, 0};
            const unsigned xw[4] = {(unsigned)xq.x, (unsigned)xq.y, (unsigned)xq.z, (unsigned)xq.w};
            const float mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
            unsigned ow[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float xa = from16<ST>((unsigned short)(xw[d] & 0xffffu)), xb = from16<ST>((unsigned short)(xw[d] >> 16));
                ow[d] = (unsigned)to16<ST>(xa * sigmoid(mm[2 * d])) | ((unsigned)to16<ST>(xb * sigmoid(mm[2 * d + 1])) << 16);
            }
            // only the store is predicated: the MFMAs above must run with every lane active (a lane supplies A / B operands)
            yv[q] = i32x4_t{(int)ow[0], (int)ow[1], (int)ow[2], (int)ow[3]};
            if (g.live && chan_ok[q] && !(NP0 > 0 && p.skip_y))
                *reinterpret_cast<i32x4_t*>(ys + (size_t)g.pix * p.y_pitch + p.y_coff + 32 * q + 8 * kq) = yv[q];
        }
        if constexpr (NP0 > 0) {
            auto pact = [&](f32x4 v, int act, float slope) __attribute__((always_inline)) -> f32x4 {
                if (act == ESR_ACT_GELU) return f32x4{esr_gelu16(v.x), esr_gelu16(v.y), esr_gelu16(v.z), esr_gelu16(v.w)};
                return f32x4{fmaxf(v.x, slope * v.x), fmaxf(v.y, slope * v.y), fmaxf(v.z, slope * v.z), fmaxf(v.w, slope * v.w)};     // slope 1: none, 0: ReLU
            };
            auto pack8 = [&](f32x4 a, f32x4 b) __attribute__((always_inline)) -> i32x4_t {
                return i32x4_t{(int)((unsigned)to16<ST>(a.x) | ((unsigned)to16<ST>(a.y) << 16)), (int)((unsigned)to16<ST>(a.z) | ((unsigned)to16<ST>(a.w) << 16)),
                               (int)((unsigned)to16<ST>(b.x) | ((unsigned)to16<ST>(b.y) << 16)), (int)((unsigned)to16<ST>(b.z) | ((unsigned)to16<ST>(b.w) << 16))};
            };
            // the weight images are re-read from LDS for every group: made loop-variant on purpose -- hoisted out of the group loop they
            // are 100+ registers (hipcc did exactly that: 364 VGPRs, one wave per SIMD)
            int pio = lane * 8;
            asm volatile("" : "+v"(pio));
            const unsigned short* const pim = spimg + pio;
            f32x4 d0[NT0];
#pragma unroll
            for (int t = 0; t < NT0; ++t) {
                d0[t] = *reinterpret_cast<const f32x4*>(spbias + t * 16 + kq * 4);
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    d0[t] = mfma_k32<ST>(*reinterpret_cast<const i32x4_t*>(pim + ((0 * NT0 + t) * NP + q) * 512), yv[q], d0[t]);
                    d0[t] = mfma_k32<ST>(*reinterpret_cast<const i32x4_t*>(pim + ((1 * NT0 + t) * NP + q) * 512), yv[q], d0[t]);
                }
            }
#pragma unroll
            for (int q = 0; q < NP0; ++q) {
                const i32x4_t rq = (p.p0_res && 32 * q + 8 * kq < p.p0_c8) ? g.rv[q] : i32x4_t{0, 0, 0, 0};
                const unsigned rw[4] = {(unsigned)rq.x, (unsigned)rq.y, (unsigned)rq.z, (unsigned)rq.w};
                f32x4 ra, rb;          // (zeros when there is no residual)
                ra.x = from16<ST>((unsigned short)(rw[0] & 0xffffu)); ra.y = from16<ST>((unsigned short)(rw[0] >> 16));
                ra.z = from16<ST>((unsigned short)(rw[1] & 0xffffu)); ra.w = from16<ST>((unsigned short)(rw[1] >> 16));
                rb.x = from16<ST>((unsigned short)(rw[2] & 0xffffu)); rb.y = from16<ST>((unsigned short)(rw[2] >> 16));
                rb.z = from16<ST>((unsigned short)(rw[3] & 0xffffu)); rb.w = from16<ST>((unsigned short)(rw[3] >> 16));
                d0[2 * q] = pact(d0[2 * q] + ra, p.p0_act, p.p0_slope);
                d0[2 * q + 1] = pact(d0[2 * q + 1] + rb, p.p0_act, p.p0_slope);
                if (g.live && 32 * q + 8 * kq < p.p0_c8)
                    *reinterpret_cast<i32x4_t*>(o0 + (size_t)g.pix * p.pout0_pitch + p.pout0_coff + 32 * q + 8 * kq) = pack8(d0[2 * q], d0[2 * q + 1]);
            }
            if constexpr (NP1 > 0) {
                f32x4 d1[NT1];
#pragma unroll
                for (int t = 0; t < NT1; ++t) d1[t] = *reinterpret_cast<const f32x4*>(spbias + (NT0 + t) * 16 + kq * 4);
#pragma unroll
                for (int T = 0; T < NT0; ++T) {
                    // fp32 fragment -> B operand: k slots 0..3 = the 16-bit high parts, 4..7 = the low parts (bf16; fp16's 11 bits are the storage precision)
                    unsigned short h[4], l[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        h[j] = to16<ST>(d0[T][j]);
                        l[j] = LO1 == 2 ? to16<ST>(d0[T][j] - from16<ST>(h[j])) : (unsigned short)0;
                    }
                    const i32x4_t bs = {(int)((unsigned)h[0] | ((unsigned)h[1] << 16)), (int)((unsigned)h[2] | ((unsigned)h[3] << 16)),
                                        (int)((unsigned)l[0] | ((unsigned)l[1] << 16)), (int)((unsigned)l[2] | ((unsigned)l[3] << 16))};
#pragma unroll
                    for (int t = 0; t < NT1; ++t) {
                        d1[t] = mfma_k32<ST>(*reinterpret_cast<const i32x4_t*>(pim + (P0_IMGS + (0 * NT1 + t) * NT0 + T) * 512), bs, d1[t]);
                        if (LO1 == 2)
                            d1[t] = mfma_k32<ST>(*reinterpret_cast<const i32x4_t*>(pim + (P0_IMGS + (1 * NT1 + t) * NT0 + T) * 512), bs, d1[t]);
                    }
                }
#pragma unroll
                for (int q = 0; q < NP1; ++q) {
                    d1[2 * q] = pact(d1[2 * q], p.p1_act, p.p1_slope);
                    d1[2 * q + 1] = pact(d1[2 * q + 1], p.p1_act, p.p1_slope);
                    if (g.live && 32 * q + 8 * kq < p.p1_c8)
                        *reinterpret_cast<i32x4_t*>(o1 + (size_t)g.pix * p.pout1_pitch + p.pout1_coff + 32 * q + 8 * kq) = pack8(d1[2 * q], d1[2 * q + 1]);
                }
            }
        }
    };
    // TWO groups per iteration, both fetched, then both finished: the second group's loads are in flight while the first is computed
    // and stored (the kernel is latency x concurrency bound: ~3 us of load latency against ~0.5 us of arithmetic per group).  The second
    // fetch is UNCONDITIONAL (behind the last group it re-reads the first one's addresses): with a branch around it hipcc's wait-count
    // pass merges "7 younger loads" with "none" and puts vmcnt(0) in front of finish(ga) -- no overlap at all (which is also what it did
    // to the loop this replaces: the next group fetched in front of finish(cur) and carried over the back edge; see DESIGN.md section 8
    // for why nothing fetched is carried across iterations any more).  Both inlined copies of finish() must round alike, whichever
    // copy a group meets (batch == per-image, test_16bit_batch_equals_per_image): floating-point contraction is off in this kernel.
    for (long long grp = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wv); grp < ngroups; grp += 2 * gstep) {
        Grp ga, gb;
        const long long g2 = grp + gstep;
        const bool two = g2 < ngroups;                           // wave-uniform
        fetch(grp, ga);
        fetch(two ? g2 : grp, gb);
        finish(ga);
        if (two) finish(gb);
    }
}

// the post-chain shapes that exist: (pairs of the apply, pairs of post 0, pairs of post 1)
static bool esa_post_variant(int np, int np0, int np1) { return np == 2 && ((np0 == 1 && np1 == 0) || (np0 == 2 && np1 <= 1)); }

template <int ST>
int launch_esa_mfma(const EsaK& k, int np0, int np1, hipStream_t st)
{
    const long long npix = (long long)k.N * k.H * k.W;
    // every block first builds the MFMA weight images (~2 us; with a post chain it also copies 16-32 KB of images): give each wave
    // at least ~4 (8) pixel groups of work
    const long long ngroups = (npix + 15) / 16;
    long long nwg = np0 > 0 ? (ngroups + 31) / 32 : (ngroups + 15) / 16;
    nwg = nwg < 1 ? 1 : (nwg > 4096 ? 4096 : nwg);      // (round 5 A/B of the cap, 512 .. 4096 blocks: 2048 / 4096 best, within 1.5 % of each other)
    const unsigned grid = (unsigned)nwg;
    const int np = (k.Cp4 + 31) / 32;
    if (np0 > 0) {
        if (!esa_post_variant(np, np0, np1)) return ESR_ERR_UNSUPPORTED;
        esr_note_kernel("esa_apply_mfma_kernel<%d, 2, %d, %d>", ST, np0, np1);
        if (np0 == 1) hipLaunchKernelGGL((esa_apply_mfma_kernel<ST, 2, 1, 0>), dim3(grid), dim3(256), 0, st, k);
        else if (np1 == 0) hipLaunchKernelGGL((esa_apply_mfma_kernel<ST, 2, 2, 0>), dim3(grid), dim3(256), 0, st, k);
        else hipLaunchKernelGGL((esa_apply_mfma_kernel<ST, 2, 2, 1>), dim3(grid), dim3(256), 0, st, k);
        return esr_check_launch("esa_apply_mfma_kernel launch");
    }
    switch (np) {
        case 1: esr_note_kernel("esa_apply_mfma_kernel<%d, 1, 0, 0>", ST); hipLaunchKernelGGL((esa_apply_mfma_kernel<ST, 1>), dim3(grid), dim3(256), 0, st, k); break;
        case 2: esr_note_kernel("esa_apply_mfma_kernel<%d, 2, 0, 0>", ST); hipLaunchKernelGGL((esa_apply_mfma_kernel<ST, 2>), dim3(grid), dim3(256), 0, st, k); break;
        default: return ESR_ERR_UNSUPPORTED;
    }
    return esr_check_launch("esa_apply_mfma_kernel launch");
}

// 16-bit rounding on the host (RNE), as the kernels' conversions
static unsigned short host_to16(float f, int storage)
{
    if (storage == ESR_STORE_BF16) {
        unsigned u;
        memcpy(&u, &f, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (unsigned short)(u >> 16);
    }
    const _Float16 h = (_Float16)f;
    unsigned short r;
    memcpy(&r, &h, 2);
    return r;
}
static float host_from16(unsigned short h, int storage)
{
    if (storage == ESR_STORE_BF16) {
        const unsigned u = (unsigned)h << 16;
        float f;
        memcpy(&f, &u, 4);
        return f;
    }
    _Float16 v;
    memcpy(&v, &h, 2);
    return (float)v;
}

// ---- depthwise 3x3, zero padding, fused residual / activation -------------------------------------
// thread = (pixel, quad of 4 channels); weights [tap][cp] + bias[cp] in LDS.  Memory-bound: the 9 taps of a
// pixel are served from L1/L2 (each input float4 is read by 9 neighbouring threads).
struct DwK {
    const void* x; const float* wp; const void* res; void* y;
    int N, H, W, cp, nq;
    int x_pitch, x_coff, r_pitch, r_coff, y_pitch, y_coff;
    int act, res_mode;
    float slope;
};

__device__ __forceinline__ float dw_act(float v, int act, float slope)
{
    if (act == ESR_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    if (act == ESR_ACT_LRELU) return fmaxf(v, v * slope);
    if (act == ESR_ACT_RELU) return fmaxf(v, 0.f);
    return v;
}

template <int ST>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const DwK p)
{
    extern __shared__ __attribute__((aligned(16))) float sdw[];          // 10 * cp floats
    for (int i = threadIdx.x; i < 10 * p.cp; i += 256) sdw[i] = p.wp[i];
    __syncthreads();
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long pix = gid / p.nq;
    const int q = (int)(gid - pix * p.nq);
    if (pix >= (long long)p.N * p.H * p.W) return;
    const int ox = (int)(pix % p.W);
    const int oy = (int)((pix / p.W) % p.H);
    f32x4 acc = *reinterpret_cast<const f32x4*>(sdw + 9 * p.cp + q * 4);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy + ky - 1;
        if (iy < 0 || iy >= p.H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox + kx - 1;
            if (ix < 0 || ix >= p.W) continue;
            const f32x4 xv = ld4<ST>(p.x, (size_t)(pix + (long long)(ky - 1) * p.W + (kx - 1)) * p.x_pitch + p.x_coff + q * 4);
            acc += xv * *reinterpret_cast<const f32x4*>(sdw + (ky * 3 + kx) * p.cp + q * 4);
        }
    }
    f32x4 rv = {0.f, 0.f, 0.f, 0.f};
    if (p.res_mode != ESR_RES_NONE) rv = ld4<ST>(p.res, (size_t)pix * p.r_pitch + p.r_coff + q * 4);
    if (p.res_mode == ESR_RES_PRE_ACT) acc += rv;
    acc.x = dw_act(acc.x, p.act, p.slope); acc.y = dw_act(acc.y, p.act, p.slope);
    acc.z = dw_act(acc.z, p.act, p.slope); acc.w = dw_act(acc.w, p.act, p.slope);
    if (p.res_mode == ESR_RES_POST_ACT) acc += rv;
    st4<ST>(p.y, (size_t)pix * p.y_pitch + p.y_coff + q * 4, acc);
}

// ---- post-processing: tensor2uint and squared error on uint8 ----------------------------------------
__global__ __launch_bounds__(256) void tensor2uint_kernel(const float* __restrict__ x, uint8_t* __restrict__ y, int C, int H,
                                                          int W, float dr, float scale)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;        // output element index (HWC)
    const long long n = (long long)H * W * C;
    if (i >= n) return;
    const int c = (int)(i % C);
    const long long hw = i / C;
    float v = x[(size_t)c * H * W + hw];
    v = v < 0.f ? 0.f : (v > dr ? dr : v);                                // clamp_(0, data_range); NaN -> propagates like torch? clamp keeps NaN
    y[i] = (uint8_t)__float2int_rn(__fmul_rn(v, scale));                  // numpy .round(): half to even
}

__global__ __launch_bounds__(256) void sqerr_u8_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int H, int W,
                                                       int C, int border, unsigned long long* out)
{
    const int hh = H - 2 * border, ww = W - 2 * border;
    const long long n = (long long)hh * ww * C;
    unsigned long long acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long p = i / C;
        const int x = (int)(p % ww) + border, yy = (int)(p / ww) + border;
        const size_t idx = ((size_t)yy * W + x) * C + c;
        const int d = (int)a[idx] - (int)b[idx];
        acc += (unsigned long long)(d * d);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

}  // namespace

extern "C" {

int esr_tensor2uint_u8(const float* x, uint8_t* y, int c, int h, int w, float data_range, void* hip_stream)
{
    if (!x || !y || c <= 0 || h <= 0 || w <= 0 || !(data_range > 0.f)) return ESR_ERR_BAD_ARG;
    const long long n = (long long)c * h * w;
    hipLaunchKernelGGL(tensor2uint_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                       x, y, c, h, w, data_range, 255.0f / data_range);
    return esr_check_launch("tensor2uint_kernel launch");
}

int esr_sqerr_u8(const uint8_t* a, const uint8_t* b, int h, int w, int c, int border, unsigned long long* sum_out,
                 void* hip_stream)
{
    if (!a || !b || !sum_out || h <= 0 || w <= 0 || c <= 0 || border < 0 || 2 * border >= h || 2 * border >= w)
        return ESR_ERR_BAD_ARG;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if (hipMemsetAsync(sum_out, 0, sizeof(unsigned long long), st) != hipSuccess) return ESR_ERR_LAUNCH;
    const long long n = (long long)(h - 2 * border) * (w - 2 * border) * c;
    const unsigned grid = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(sqerr_u8_kernel, dim3(grid), dim3(256), 0, st, a, b, h, w, c, border, sum_out);
    return esr_check_launch("sqerr_u8_kernel launch");
}

size_t esr_packed_dw_bytes(int c) { return c <= 0 ? 0 : (size_t)10 * esr_round_up(c, 4) * sizeof(float); }

int esr_pack_dw_f32(const float* w, const float* bias, int c, void* out, size_t out_bytes)
{
    if (!w || !out || c <= 0 || out_bytes < esr_packed_dw_bytes(c)) return ESR_ERR_BAD_ARG;
    const int cp = esr_round_up(c, 4);
    float* o = static_cast<float*>(out);
    memset(o, 0, esr_packed_dw_bytes(c));
    for (int ch = 0; ch < c; ++ch) {
        for (int t = 0; t < 9; ++t) o[t * cp + ch] = w[ch * 9 + t];
        if (bias) o[9 * cp + ch] = bias[ch];
    }
    return ESR_OK;
}

int esr_dwconv3x3_f32(const esr_conv_desc* d, void* hip_stream)
{
    if (!d || !d->in.ptr || !d->out0.ptr || !d->wpacked) return ESR_ERR_BAD_ARG;
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cin <= 0 || d->cin != d->cout || d->ksize != 3) return ESR_ERR_BAD_ARG;
    if (d->in_layout != ESR_NHWC || d->out_layout != ESR_NHWC) return ESR_ERR_UNSUPPORTED;
    const int cp = esr_round_up(d->cin, 4);
    if ((d->in.pitch & 3) || (d->in.coff & 3) || d->in.coff + cp > d->in.pitch) return ESR_ERR_BAD_ARG;
    if ((d->out0.pitch & 3) || (d->out0.coff & 3) || d->out0.coff + cp > d->out0.pitch) return ESR_ERR_BAD_ARG;
    if (d->res_mode != ESR_RES_NONE &&
        (!d->res.ptr || (d->res.pitch & 3) || (d->res.coff & 3) || d->res.coff + cp > d->res.pitch))
        return ESR_ERR_BAD_ARG;
    DwK k;
    k.x = d->in.ptr; k.wp = static_cast<const float*>(d->wpacked);
    k.res = d->res.ptr; k.y = d->out0.ptr;
    k.N = d->n; k.H = d->h; k.W = d->w; k.cp = cp; k.nq = cp / 4;
    k.x_pitch = d->in.pitch; k.x_coff = d->in.coff; k.r_pitch = d->res.pitch; k.r_coff = d->res.coff;
    k.y_pitch = d->out0.pitch; k.y_coff = d->out0.coff;
    k.act = d->act; k.res_mode = d->res_mode; k.slope = d->slope;
    const long long nthreads = (long long)d->n * d->h * d->w * k.nq;
    const dim3 grid((unsigned)((nthreads + 255) / 256));
    const size_t lds = (size_t)10 * cp * sizeof(float);
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    switch (d->storage) {
        case ESR_STORE_F32: esr_note_kernel("dwconv3x3_kernel<0>"); hipLaunchKernelGGL(dwconv3x3_kernel<ESR_STORE_F32>, grid, dim3(256), lds, st, k); break;
        case ESR_STORE_BF16: esr_note_kernel("dwconv3x3_kernel<1>"); hipLaunchKernelGGL(dwconv3x3_kernel<ESR_STORE_BF16>, grid, dim3(256), lds, st, k); break;
        case ESR_STORE_F16: esr_note_kernel("dwconv3x3_kernel<2>"); hipLaunchKernelGGL(dwconv3x3_kernel<ESR_STORE_F16>, grid, dim3(256), lds, st, k); break;
        default: return ESR_ERR_BAD_ARG;
    }
    return esr_check_launch("dwconv3x3_kernel launch");
}

size_t esr_packed_dense_bytes(int cin_p, int cout_p, int ksize)
{
    if (cin_p <= 0 || cout_p <= 0 || (ksize != 1 && ksize != 3)) return 0;
    return ((size_t)ksize * ksize * cin_p * cout_p + cout_p) * sizeof(float);
}

int esr_pack_dense_f32(const float* w, const float* bias, int cin, int cout, int ksize, int cin_p, int cout_p,
                       void* out, size_t out_bytes)
{
    if (!w || !out || cin <= 0 || cout <= 0 || cin > cin_p || cout > cout_p) return ESR_ERR_BAD_ARG;
    const size_t need = esr_packed_dense_bytes(cin_p, cout_p, ksize);
    if (need == 0 || out_bytes < need) return ESR_ERR_BAD_ARG;
    float* o = static_cast<float*>(out);
    memset(o, 0, need);
    const int taps = ksize * ksize;
    for (int oc = 0; oc < cout; ++oc)
        for (int c = 0; c < cin; ++c)
            for (int t = 0; t < taps; ++t)
                o[((size_t)t * cin_p + c) * cout_p + oc] = w[((size_t)oc * cin + c) * taps + t];
    if (bias)
        for (int oc = 0; oc < cout; ++oc) o[(size_t)taps * cin_p * cout_p + oc] = bias[oc];
    return ESR_OK;
}

static int lowres_args_ok(const esr_esa_desc* d)
{
    if (!d || !d->x.ptr || !d->y.ptr) return ESR_ERR_BAD_ARG;
    if (d->n <= 0 || d->h <= 0 || d->w <= 0) return ESR_ERR_BAD_ARG;
    if (d->x.pitch != FP || d->y.pitch != FP || d->x.coff || d->y.coff) return ESR_ERR_BAD_ARG;
    return ESR_OK;
}

int esr_conv3x3s2_f32(const esr_esa_desc* d, void* hip_stream)
{
    int rc = lowres_args_ok(d);
    if (rc != ESR_OK) return rc;
    if (!d->w0) return ESR_ERR_BAD_ARG;
    if (d->h < 3 || d->w < 3) return ESR_ERR_TOO_SMALL;
    const int Ho = (d->h - 3) / 2 + 1, Wo = (d->w - 3) / 2 + 1;
    if (d->h_lo != Ho || d->w_lo != Wo) return ESR_ERR_BAD_ARG;
    const long long npix = (long long)d->n * Ho * Wo;
    const dim3 grid((unsigned)((npix + 63) / 64));
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const float* w0 = static_cast<const float*>(d->w0);
    float* y = static_cast<float*>(d->y.ptr);                 // the half-resolution map is fp32 in every storage mode
    switch (d->storage) {
        case ESR_STORE_F32: esr_note_kernel("conv3x3s2_kernel<0>"); hipLaunchKernelGGL(conv3x3s2_kernel<ESR_STORE_F32>, grid, dim3(256), 0, st, d->x.ptr, w0, y, d->n, d->h, d->w, Ho, Wo); break;
        case ESR_STORE_BF16: esr_note_kernel("conv3x3s2_kernel<1>"); hipLaunchKernelGGL(conv3x3s2_kernel<ESR_STORE_BF16>, grid, dim3(256), 0, st, d->x.ptr, w0, y, d->n, d->h, d->w, Ho, Wo); break;
        case ESR_STORE_F16: esr_note_kernel("conv3x3s2_kernel<2>"); hipLaunchKernelGGL(conv3x3s2_kernel<ESR_STORE_F16>, grid, dim3(256), 0, st, d->x.ptr, w0, y, d->n, d->h, d->w, Ho, Wo); break;
        default: return ESR_ERR_BAD_ARG;
    }
    return esr_check_launch("conv3x3s2_kernel launch");
}

int esr_maxpool7s3_f32(const esr_esa_desc* d, void* hip_stream)
{
    int rc = lowres_args_ok(d);
    if (rc != ESR_OK) return rc;
    if (d->h < 7 || d->w < 7) return ESR_ERR_TOO_SMALL;
    if (d->storage != ESR_STORE_F32) return ESR_ERR_BAD_ARG;          // low-resolution maps are always fp32
    const int Ho = (d->h - 7) / 3 + 1, Wo = (d->w - 7) / 3 + 1;
    if (d->h_lo != Ho || d->w_lo != Wo) return ESR_ERR_BAD_ARG;
    const long long npix = (long long)d->n * Ho * Wo;
    esr_note_kernel("maxpool7s3_kernel");
    hipLaunchKernelGGL(maxpool7s3_kernel, dim3((unsigned)((npix + 63) / 64)), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                       static_cast<const float*>(d->x.ptr), static_cast<float*>(d->y.ptr), d->n, d->h, d->w, Ho, Wo);
    return esr_check_launch("maxpool7s3_kernel launch");
}

int esr_esa_apply_post_supported(int c, int cout0, int cout1)
{
    if (c <= 0 || cout0 <= 0 || cout1 < 0) return 0;
    return esa_post_variant((esr_round_up(c, 4) + 31) / 32, (cout0 + 31) / 32, (cout1 + 31) / 32) ? 1 : 0;
}

size_t esr_packed_apply_post_bytes(int cin, int cout0, int cout1, int storage)
{
    if (cin <= 0 || cout0 <= 0 || cout1 < 0 || (storage != ESR_STORE_BF16 && storage != ESR_STORE_F16)) return 0;
    const int np = (esr_round_up(cin, 4) + 31) / 32, nt0 = 2 * ((cout0 + 31) / 32), nt1 = 2 * ((cout1 + 31) / 32);
    const int lo1 = storage == ESR_STORE_BF16 ? 2 : 1;
    return (size_t)(2 * nt0 * np + lo1 * nt1 * nt0) * 1024 + (size_t)(nt0 + nt1) * 64;
}

int esr_pack_apply_post(const float* w0, const float* b0, const float* w1, const float* b1, int cin, int cout0, int cout1,
                        int storage, void* out, size_t out_bytes)
{
    const size_t need = esr_packed_apply_post_bytes(cin, cout0, cout1, storage);
    if (!need || !w0 || !out || out_bytes < need || (cout1 > 0 && !w1)) return ESR_ERR_BAD_ARG;
    const int np = (esr_round_up(cin, 4) + 31) / 32, nt0 = 2 * ((cout0 + 31) / 32), nt1 = 2 * ((cout1 + 31) / 32);
    const int lo1 = storage == ESR_STORE_BF16 ? 2 : 1;
    unsigned short* img = static_cast<unsigned short*>(out);
    // row i of tile t <-> channel 32 (t >> 1) + 8 (i >> 2) + 4 (t & 1) + (i & 3): a lane's results of a tile pair are 8 consecutive channels
    auto chan = [](int t, int i) { return 32 * (t >> 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3); };
    for (int lo = 0; lo < 2; ++lo)
        for (int t = 0; t < nt0; ++t)
            for (int q = 0; q < np; ++q)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int oc = chan(t, l & 15), ic = 32 * q + 8 * (l >> 4) + j;
                        const float w = (oc < cout0 && ic < cin) ? w0[(size_t)oc * cin + ic] : 0.f;
                        const unsigned short hi = host_to16(w, storage);
                        img[((size_t)((lo * nt0 + t) * np + q) * 64 + l) * 8 + j] = lo ? host_to16(w - host_from16(hi, storage), storage) : hi;
                    }
    unsigned short* img1 = img + (size_t)2 * nt0 * np * 512;
    for (int lo = 0; lo < lo1; ++lo)
        for (int t = 0; t < nt1; ++t)
            for (int T = 0; T < nt0; ++T)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        // k slot (kq, j): the high (j < 4) / low (j >= 4) part of the lane's value j & 3 of post 0's tile T
                        const int oc = chan(t, l & 15), ic = chan(T, 4 * (l >> 4) + (j & 3));
                        const float w = (oc < cout1 && ic < cout0) ? w1[(size_t)oc * cout0 + ic] : 0.f;
                        const unsigned short hi = host_to16(w, storage);
                        const unsigned short v = lo == 0 ? hi : (j < 4 ? host_to16(w - host_from16(hi, storage), storage) : (unsigned short)0);
                        img1[((size_t)((lo * nt1 + t) * nt0 + T) * 64 + l) * 8 + j] = (storage == ESR_STORE_F16 && j >= 4) ? (unsigned short)0 : v;
                    }
    float* bias = reinterpret_cast<float*>(img1 + (size_t)lo1 * nt1 * nt0 * 512);
    for (int t = 0; t < nt0; ++t)
        for (int i = 0; i < 16; ++i) bias[t * 16 + i] = (b0 && chan(t, i) < cout0) ? b0[chan(t, i)] : 0.f;
    for (int t = 0; t < nt1; ++t)
        for (int i = 0; i < 16; ++i) bias[(nt0 + t) * 16 + i] = (b1 && chan(t, i) < cout1) ? b1[chan(t, i)] : 0.f;
    return ESR_OK;
}

int esr_esa_apply_f32(const esr_esa_desc* d, void* hip_stream)
{
    if (!d || !d->x.ptr || !d->y.ptr || !d->c1 || !d->c3 || !d->w0 || !d->w1) return ESR_ERR_BAD_ARG;
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->c <= 0 || d->c > 64 || d->f <= 0 || d->f > FP) return ESR_ERR_BAD_ARG;
    if (d->h_lo <= 0 || d->w_lo <= 0) return ESR_ERR_TOO_SMALL;
    const int cp4 = esr_round_up(d->c, 4);
    if ((d->x.pitch & 3) || (d->x.coff & 3) || d->x.coff + cp4 > d->x.pitch) return ESR_ERR_BAD_ARG;
    if ((d->y.pitch & 3) || (d->y.coff & 3) || d->y.coff + cp4 > d->y.pitch) return ESR_ERR_BAD_ARG;
    if (d->storage != ESR_STORE_F32) {
        // 16-bit storage moves 8 channels (16 bytes) per lane: granules of 8, padded channels are read and written as zeros
        const int cp8 = esr_round_up(d->c, 8);
        if ((d->x.pitch & 7) || (d->x.coff & 7) || d->x.coff + cp8 > d->x.pitch) return ESR_ERR_BAD_ARG;
        if ((d->y.pitch & 7) || (d->y.coff & 7) || d->y.coff + cp8 > d->y.pitch) return ESR_ERR_BAD_ARG;
    }
    EsaK k;
    k.x = d->x.ptr; k.c1 = d->c1;
    k.c3 = static_cast<const float*>(d->c3); k.wf = static_cast<const float*>(d->w0);
    k.w4 = static_cast<const float*>(d->w1); k.y = d->y.ptr;
    k.x_pitch = d->x.pitch; k.x_coff = d->x.coff; k.y_pitch = d->y.pitch; k.y_coff = d->y.coff;
    k.N = d->n; k.H = d->h; k.W = d->w; k.Cp4 = cp4; k.cp = cp4; k.h3 = d->h_lo; k.w3 = d->w_lo;
    k.sh = (float)d->h_lo / (float)d->h;      // ATen area_pixel_compute_scale: float(in) / out
    k.sw = (float)d->w_lo / (float)d->w;
    const long long npix = (long long)d->n * d->h * d->w;
    if (npix >= 2147483647LL) return ESR_ERR_UNSUPPORTED;                 // 32-bit pixel indices inside the kernels
    // post chain (ABI v8)
    int np0 = 0, np1 = 0;
    k.pimg = nullptr; k.pres = nullptr; k.pout0 = nullptr; k.pout1 = nullptr;
    k.pres_pitch = k.pres_coff = k.pout0_pitch = k.pout0_coff = k.pout1_pitch = k.pout1_coff = 0;
    k.p0_c8 = k.p1_c8 = 0; k.p0_act = k.p1_act = ESR_ACT_NONE; k.p0_res = 0; k.p0_slope = k.p1_slope = 1.f; k.skip_y = 0;
    if (d->post_w) {
        if (d->storage == ESR_STORE_F32) return ESR_ERR_UNSUPPORTED;
        const esr_esa_post& p0 = d->post[0];
        const esr_esa_post& p1 = d->post[1];
        if (p0.cout <= 0 || p0.cout > 64 || p1.cout < 0 || p1.cout > 32) return ESR_ERR_UNSUPPORTED;
        auto view_ok = [](const esr_view& v, int c8) { return v.ptr && !(v.pitch & 7) && !(v.coff & 7) && v.coff + c8 <= v.pitch; };
        auto act_ok = [](int a) { return a == ESR_ACT_NONE || a == ESR_ACT_LRELU || a == ESR_ACT_RELU || a == ESR_ACT_GELU; };
        k.p0_c8 = esr_round_up(p0.cout, 8);
        if (!view_ok(p0.out, k.p0_c8) || !act_ok(p0.act)) return ESR_ERR_BAD_ARG;
        if (p0.res_mode != ESR_RES_NONE && p0.res_mode != ESR_RES_PRE_ACT) return ESR_ERR_UNSUPPORTED;
        if (p0.res_mode == ESR_RES_PRE_ACT && !view_ok(p0.res, k.p0_c8)) return ESR_ERR_BAD_ARG;
        np0 = (p0.cout + 31) / 32;
        k.pimg = static_cast<const unsigned short*>(d->post_w);
        k.pout0 = p0.out.ptr; k.pout0_pitch = p0.out.pitch; k.pout0_coff = p0.out.coff;
        k.p0_res = p0.res_mode == ESR_RES_PRE_ACT;
        if (k.p0_res) { k.pres = p0.res.ptr; k.pres_pitch = p0.res.pitch; k.pres_coff = p0.res.coff; }
        k.p0_act = p0.act;
        k.p0_slope = p0.act == ESR_ACT_LRELU ? p0.slope : (p0.act == ESR_ACT_RELU ? 0.f : 1.f);
        if (p1.cout > 0) {
            k.p1_c8 = esr_round_up(p1.cout, 8);
            if (!view_ok(p1.out, k.p1_c8) || !act_ok(p1.act) || p1.res_mode != ESR_RES_NONE) return ESR_ERR_BAD_ARG;
            np1 = (p1.cout + 31) / 32;
            k.pout1 = p1.out.ptr; k.pout1_pitch = p1.out.pitch; k.pout1_coff = p1.out.coff;
            k.p1_act = p1.act;
            k.p1_slope = p1.act == ESR_ACT_LRELU ? p1.slope : (p1.act == ESR_ACT_RELU ? 0.f : 1.f);
        }
        k.skip_y = d->skip_y ? 1 : 0;
        if (!esa_post_variant((cp4 + 31) / 32, np0, np1)) return ESR_ERR_UNSUPPORTED;
    } else if (d->skip_y) {
        return ESR_ERR_BAD_ARG;
    }
    const size_t lds = ((size_t)FP * FP + FP + (((size_t)FP * cp4 + cp4 + 3) & ~(size_t)3) + 16 * FP) * sizeof(float);
    const long long ngroups = (npix + 15) / 16;
    const unsigned grid = (unsigned)(ngroups < 8192 ? ngroups : 8192);        // 256 CUs x 8 blocks x 4 rounds
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    switch (d->storage) {
        case ESR_STORE_F32: esr_note_kernel("esa_apply_kernel<0>"); hipLaunchKernelGGL(esa_apply_kernel<ESR_STORE_F32>, dim3(grid), dim3(256), lds, st, k); break;
        case ESR_STORE_BF16: return launch_esa_mfma<ESR_STORE_BF16>(k, np0, np1, st);
        case ESR_STORE_F16: return launch_esa_mfma<ESR_STORE_F16>(k, np0, np1, st);
        default: return ESR_ERR_BAD_ARG;
    }
    return esr_check_launch("esa_apply_kernel launch");
}

}  // extern "C"
