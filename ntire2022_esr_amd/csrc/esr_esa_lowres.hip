// esr_esa_lowres.hip -- ESA's low-resolution branch in TWO launches instead of 3 .. 8.  Interface: esr_esa_lowres_f32 (include/esr_hip.h).
//
//     c1 = conv2(c1_)            3x3 stride 2, no padding   (models/rfdn_baseline/block.py:110,119; team04_rlfn.py:69,78; team18_bsrn.py:102,112)
//     v  = max_pool2d(c1, 7, 3)                             (block.py:120)
//     c3 = 1 .. 3 convolutions on v, 3x3 pad 1 (+ ReLU)     (block.py:121-123: conv_max, conv3, conv3_; RLFN: conv3 alone)
//          or BSConvU = pointwise + depthwise 3x3 (+ GELU)  (team18_bsrn.py:113-116)
//
// On one DIV2K image (339x510 -> 169x254 -> 55x83 x f <= 16 channels) every one of those is a launch of a few microseconds of work
// and ~10-15 us of latency: a third of a forward's launches for < 1 % of its arithmetic.  The grid-barrier fusion of round 2 lost
// (a co-resident grid runs every latency-bound layer at a quarter of its occupancy); this one needs no barrier -- HALO RECOMPUTE:
//   s2pool_kernel   one block = a 4x4 tile of the POOLED map: the 16x16 conv2 outputs under it go to LDS (1.8x recompute at the tile
//                   seams), the 7x7/3 maxima are taken from there.  conv2's output never reaches memory.
//   chain_kernel    one block = an 8x8 tile of the LAST layer: the (8 + 2L)^2 pooled patch goes to LDS and the L layers run on
//                   shrinking patches in two LDS buffers; positions outside the map are written as zeros, which is each layer's
//                   zero padding.  Weights of all layers sit in LDS.
// VALU fp32 kernels (thread = output pixel x quad of channels, float4 everywhere): the branch is 0.3 % of a network's MACs.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include "esr_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int FP = ESR_ESA_FP;                 // 16: channel pitch of every map here
constexpr int PT = 4;                          // pooled tile edge of s2pool_kernel
constexpr int CT = 3 * PT + 4;                 // conv2 outputs under it per edge: 16
constexpr int OT = 8;                          // output tile edge of chain_kernel
constexpr int ML = ESR_ESA_MAX_LAYERS;
constexpr int PMAX = OT + 2 * ML;              // 14

template <int ST>
__device__ __forceinline__ f32x4 lo_ld4(const void* base, size_t idx)
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

// ---- conv2 (3x3 / 2) + max pool (7 / 3) ---------------------------------------------------------------------------------------
template <int ST>
__global__ __launch_bounds__(256) void esa_s2pool_kernel(const void* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y,
                                                         int H, int W, int H2, int W2, int H3, int W3, int tiles_x, int tiles_y)
{
    __shared__ __attribute__((aligned(16))) float sw[9 * FP * FP + FP];
    __shared__ __attribute__((aligned(16))) float sc[CT * CT * FP];          // conv2 outputs of this tile, [y][x][16]
    for (int i = threadIdx.x; i < 9 * FP * FP + FP; i += 256) sw[i] = wp[i];
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int n = t / tiles_y;
    const int cy0 = 3 * PT * ty, cx0 = 3 * PT * tx;                          // first conv2 output of the tile
    __syncthreads();
    // conv2: 256 outputs x 4 channel quads over 256 threads
    for (int it = threadIdx.x; it < CT * CT * 4; it += 256) {
        const int q = it & 3, pl = it >> 2;
        const int ly = pl / CT, lx = pl - ly * CT;
        const int oy = cy0 + ly, ox = cx0 + lx;
        f32x4 acc = *reinterpret_cast<const f32x4*>(sw + 9 * FP * FP + q * 4);
        if (oy < H2 && ox < W2) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const size_t xi = (((size_t)n * H + (oy * 2 + ky)) * W + (ox * 2 + kx)) * FP;
                    const float* wt = sw + (ky * 3 + kx) * FP * FP + q * 4;
#pragma unroll
                    for (int cq = 0; cq < FP / 4; ++cq) {
                        const f32x4 xv = lo_ld4<ST>(x, xi + cq * 4);
                        acc += xv.x * *reinterpret_cast<const f32x4*>(wt + (cq * 4 + 0) * FP);
                        acc += xv.y * *reinterpret_cast<const f32x4*>(wt + (cq * 4 + 1) * FP);
                        acc += xv.z * *reinterpret_cast<const f32x4*>(wt + (cq * 4 + 2) * FP);
                        acc += xv.w * *reinterpret_cast<const f32x4*>(wt + (cq * 4 + 3) * FP);
                    }
                }
        }
        *reinterpret_cast<f32x4*>(sc + pl * FP + q * 4) = acc;               // (outside the map: never read by a valid window)
    }
    __syncthreads();
    if (threadIdx.x < PT * PT * 4) {
        const int q = threadIdx.x & 3, pl = threadIdx.x >> 2;
        const int py = pl / PT, pxx = pl - py * PT;
        const int gy = PT * ty + py, gx = PT * tx + pxx;
        if (gy < H3 && gx < W3) {
            f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int ky = 0; ky < 7; ++ky)
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(sc + ((3 * py + ky) * CT + 3 * pxx + kx) * FP + q * 4);
                    m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                }
            *reinterpret_cast<f32x4*>(y + (((size_t)n * H3 + gy) * W3 + gx) * FP + q * 4) = m;
        }
    }
}

// ---- the convolutions behind the pooling ---------------------------------------------------------------------------------------
struct ChainK {
    const float* x;            // pooled map [n][H3][W3][16]
    float* y;
    int H3, W3, tiles_x, tiles_y, n_layers;
    int kind[ML], act[ML], cp[ML];
    const float* w[ML];        // kind 0: dense 3x3 [tap][16][16] + bias[16]; kind 1: pointwise [16][16] + bias[16]
    const float* wdw[ML];      // kind 1: depthwise [tap][cp] + bias[cp]
};

__device__ __forceinline__ float lo_act(float v, int act)
{
    switch (act) {
        case ESR_ACT_RELU: return fmaxf(v, 0.f);
        case ESR_ACT_LRELU: return fmaxf(v, 0.05f * v);
        case ESR_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        default: return v;
    }
}

__global__ __launch_bounds__(256) void esa_chain_kernel(const ChainK p)
{
    constexpr int WL = 9 * FP * FP + FP;                       // floats reserved per layer's weights (dense 3x3 is the largest)
    __shared__ __attribute__((aligned(16))) float sw[ML * WL];
    __shared__ __attribute__((aligned(16))) float sdw[ML * (10 * FP)];
    __shared__ __attribute__((aligned(16))) float buf[2][PMAX * PMAX * FP];
    const int L = p.n_layers;
    for (int l = 0; l < L; ++l) {
        const int nw = p.kind[l] == 0 ? 9 * FP * FP + FP : FP * FP + FP;
        for (int i = threadIdx.x; i < nw; i += 256) sw[l * WL + i] = p.w[l][i];
        if (p.kind[l] == 1)
            for (int i = threadIdx.x; i < 10 * p.cp[l]; i += 256) sdw[l * 10 * FP + i] = p.wdw[l][i];
    }
    int t = blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int n = t / p.tiles_y;
    // patch of the pooled map: rows y0 - L .. y0 + OT + L - 1; outside the map = 0 (the first layer's zero padding)
    int S = OT + 2 * L;
    int oy = OT * ty - L, ox = OT * tx - L;                    // map coordinates of the current patch's (0, 0)
    for (int it = threadIdx.x; it < S * S * 4; it += 256) {
        const int q = it & 3, pl = it >> 2;
        const int ly = pl / S, lx = pl - ly * S;
        const int gy = oy + ly, gx = ox + lx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gy >= 0 && gy < p.H3 && gx >= 0 && gx < p.W3) v = *reinterpret_cast<const f32x4*>(p.x + (((size_t)n * p.H3 + gy) * p.W3 + gx) * FP + q * 4);
        *reinterpret_cast<f32x4*>(buf[0] + pl * FP + q * 4) = v;
    }
    __syncthreads();
    int cur = 0;
    for (int l = 0; l < L; ++l) {
        const float* wl = sw + l * WL;
        const int act = p.act[l];
        if (p.kind[l] == 1) {
            // BSConvU, first half: pointwise 1x1 in place geometry (S x S -> S x S); outside the map the depthwise conv must see
            // zeros, not the pointwise bias (team18_bsrn.py:82-88 pads the pointwise OUTPUT)
            for (int it = threadIdx.x; it < S * S * 4; it += 256) {
                const int q = it & 3, pl = it >> 2;
                const int ly = pl / S, lx = pl - ly * S;
                const int gy = oy + ly, gx = ox + lx;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (gy >= 0 && gy < p.H3 && gx >= 0 && gx < p.W3) {
                    acc = *reinterpret_cast<const f32x4*>(wl + FP * FP + q * 4);
                    const float* in = buf[cur] + pl * FP;
#pragma unroll
                    for (int cq = 0; cq < FP / 4; ++cq) {
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(in + cq * 4);
                        acc += xv.x * *reinterpret_cast<const f32x4*>(wl + (cq * 4 + 0) * FP + q * 4);
                        acc += xv.y * *reinterpret_cast<const f32x4*>(wl + (cq * 4 + 1) * FP + q * 4);
                        acc += xv.z * *reinterpret_cast<const f32x4*>(wl + (cq * 4 + 2) * FP + q * 4);
                        acc += xv.w * *reinterpret_cast<const f32x4*>(wl + (cq * 4 + 3) * FP + q * 4);
                    }
                }
                *reinterpret_cast<f32x4*>(buf[cur ^ 1] + pl * FP + q * 4) = acc;
            }
            __syncthreads();
            cur ^= 1;
        }
        // 3x3 (dense, or depthwise for BSConvU): S x S -> (S - 2) x (S - 2); outside the map -> 0 (the next layer's padding)
        const int So = S - 2;
        const bool last = l == L - 1;
        const float* dwl = sdw + l * 10 * FP;
        const int cp = p.cp[l];
        for (int it = threadIdx.x; it < So * So * 4; it += 256) {
            const int q = it & 3, pl = it >> 2;
            const int ly = pl / So, lx = pl - ly * So;
            const int gy = oy + 1 + ly, gx = ox + 1 + lx;
            const bool inside = gy >= 0 && gy < p.H3 && gx >= 0 && gx < p.W3;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (inside) {
                if (p.kind[l] == 0) {
                    acc = *reinterpret_cast<const f32x4*>(wl + 9 * FP * FP + q * 4);
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const float* in = buf[cur] + ((ly + ky) * S + lx + kx) * FP;
                            const float* wt = wl + (ky * 3 + kx) * FP * FP + q * 4;
#pragma unroll
                            for (int cq = 0; cq < FP / 4; ++cq) {
                                const f32x4 xv = *reinterpret_cast<const f32x4*>(in + cq * 4);
                                acc += xv.x * *reinterpret_cast<const f32x4*>(wt + (cq * 4 + 0) * FP);
                                acc += xv.y * *reinterpret_cast<const f32x4*>(wt + (cq * 4 + 1) * FP);
                                acc += xv.z * *reinterpret_cast<const f32x4*>(wt + (cq * 4 + 2) * FP);
                                acc += xv.w * *reinterpret_cast<const f32x4*>(wt + (cq * 4 + 3) * FP);
                            }
                        }
                } else if (q * 4 < cp) {
                    acc = *reinterpret_cast<const f32x4*>(dwl + 9 * cp + q * 4);
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
                            acc += *reinterpret_cast<const f32x4*>(buf[cur] + ((ly + ky) * S + lx + kx) * FP + q * 4) *
                                   *reinterpret_cast<const f32x4*>(dwl + (ky * 3 + kx) * cp + q * 4);
                }
                acc.x = lo_act(acc.x, act); acc.y = lo_act(acc.y, act); acc.z = lo_act(acc.z, act); acc.w = lo_act(acc.w, act);
            }
            if (last) {
                if (inside) *reinterpret_cast<f32x4*>(p.y + (((size_t)n * p.H3 + gy) * p.W3 + gx) * FP + q * 4) = acc;
            } else {
                *reinterpret_cast<f32x4*>(buf[cur ^ 1] + pl * FP + q * 4) = acc;
            }
        }
        __syncthreads();
        cur ^= 1;
        S = So;
        ++oy; ++ox;
    }
}

}  // namespace

extern "C" int esr_esa_lowres_f32(const esr_esa_lowres_desc* d, void* hip_stream)
{
    if (!d || !d->x.ptr || !d->w_s2 || !d->pooled || !d->y) return ESR_ERR_BAD_ARG;
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->f <= 0 || d->f > FP) return ESR_ERR_BAD_ARG;
    if (d->x.pitch != FP || d->x.coff) return ESR_ERR_BAD_ARG;
    if (d->n_layers < 1 || d->n_layers > ML) return ESR_ERR_BAD_ARG;
    if (d->h < 15 || d->w < 15) return ESR_ERR_TOO_SMALL;                         // (15 - 3) / 2 + 1 = 7: one pooling window
    const int H2 = (d->h - 3) / 2 + 1, W2 = (d->w - 3) / 2 + 1;
    const int H3 = (H2 - 7) / 3 + 1, W3 = (W2 - 7) / 3 + 1;
    ChainK k;
    memset(&k, 0, sizeof(k));
    for (int l = 0; l < d->n_layers; ++l) {
        const int kind = d->layer[l].kind;
        if ((kind != 0 && kind != 1) || !d->layer[l].w || (kind == 1 && !d->layer[l].w_dw)) return ESR_ERR_BAD_ARG;
        k.kind[l] = kind; k.act[l] = d->layer[l].act; k.cp[l] = esr_round_up(d->f, 4);
        k.w[l] = static_cast<const float*>(d->layer[l].w);
        k.wdw[l] = static_cast<const float*>(d->layer[l].w_dw);
    }
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const int tx = (W3 + PT - 1) / PT, ty = (H3 + PT - 1) / PT;
    if ((long long)d->n * tx * ty >= 2147483647LL) return ESR_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)(d->n * tx * ty));
    const float* w0 = static_cast<const float*>(d->w_s2);
    float* pooled = static_cast<float*>(d->pooled);
    switch (d->storage) {
        case ESR_STORE_F32: esr_note_kernel("esa_s2pool_kernel<0>"); hipLaunchKernelGGL(esa_s2pool_kernel<ESR_STORE_F32>, grid, dim3(256), 0, st, d->x.ptr, w0, pooled, d->h, d->w, H2, W2, H3, W3, tx, ty); break;
        case ESR_STORE_BF16: esr_note_kernel("esa_s2pool_kernel<1>"); hipLaunchKernelGGL(esa_s2pool_kernel<ESR_STORE_BF16>, grid, dim3(256), 0, st, d->x.ptr, w0, pooled, d->h, d->w, H2, W2, H3, W3, tx, ty); break;
        case ESR_STORE_F16: esr_note_kernel("esa_s2pool_kernel<2>"); hipLaunchKernelGGL(esa_s2pool_kernel<ESR_STORE_F16>, grid, dim3(256), 0, st, d->x.ptr, w0, pooled, d->h, d->w, H2, W2, H3, W3, tx, ty); break;
        default: return ESR_ERR_BAD_ARG;
    }
    int rc = esr_check_launch("esa_s2pool_kernel launch");
    if (rc != ESR_OK) return rc;
    k.x = pooled; k.y = static_cast<float*>(d->y);
    k.H3 = H3; k.W3 = W3; k.n_layers = d->n_layers;
    k.tiles_x = (W3 + OT - 1) / OT; k.tiles_y = (H3 + OT - 1) / OT;
    esr_note_kernel("esa_chain_kernel");
    hipLaunchKernelGGL(esa_chain_kernel, dim3((unsigned)(d->n * k.tiles_x * k.tiles_y)), dim3(256), 0, st, k);
    return esr_check_launch("esa_chain_kernel launch");
}
