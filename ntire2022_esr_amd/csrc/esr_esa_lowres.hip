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
//   s2pool_kernel   one block = a 4x4 tile of the POOLED map: its 33x33 input patch is staged in LDS, the 16x16 conv2 outputs under it go to LDS (1.8x recompute at the tile
//                   seams), the 7x7/3 maxima are taken from there.  conv2's output never reaches memory.
//   chain_kernel    one block = a 6x6 tile of the LAST layer: the (6 + 2L)^2 pooled patch goes to LDS and the L layers run on
//                   shrinking patches in two LDS buffers; positions outside the map are written as zeros, which is each layer's
//                   zero padding.
// The convolutions run on the matrix cores in exact fp32 with their weights in registers; pooling and the depthwise 3x3 on the
// vector units.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <atomic>

#include "esr_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 lo_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 lo_f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int FP = ESR_ESA_FP;                 // 16: channel pitch of every map here
constexpr int PT = 4;                          // pooled tile edge of s2pool_kernel
constexpr int CT = 3 * PT + 4;                 // conv2 outputs under it per edge: 16 (= one MFMA pixel group per row)
constexpr int OT = 6;                          // output tile edge of chain_kernel (339x510: 10 x 14 = 140 blocks)
constexpr int BP = 20;                         // floats per pixel of the LDS patches (16 + 4: adjacent pixels on different banks)
constexpr int ML = ESR_ESA_MAX_LAYERS;
constexpr int PMAX = OT + 2 * ML;              // 12

// The small convolutions run on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32): D[cout][pixel] += W[cout][cin] X[cin][pixel]
// per tap, 16 pixels per group.  Lane (j = l & 15, g = l >> 4): A = W[cout j][cin 4g + s], B = X[pixel j][cin 4g + s] for the four
// K steps s of a tap -- one 16-byte read of the pixel's channels 4g .. 4g+3 per tap; D: lane (j, g) holds output channels
// 4g .. 4g+3 of pixel j (one float4 of an NHWC pixel).  The weights of a layer sit in REGISTERS (9 taps x float4 per lane) for the
// life of the block.  (As VALU kernels with per-lane or scalar weight loads these were LDS-return- or latency-bound: 50 - 90 us per
// ESA block on one image, no better than the five launches they replace.)
// tile of this block.  Blocks are dealt round-robin to the 8 XCDs (XCD = blockIdx & 7), each with its own L2: XCD k takes the k-th eighth of
// the tiles in raster order, so that the halo a tile shares with its right and lower neighbours (s2pool: 33 x 33 input pixels per 24 x 24
// of the tile, 1.9x) is fetched by the SAME L2 shortly after (round 5; before: blockIdx = tile, every neighbour on another XCD)
__device__ __forceinline__ int lo_xcd_tile()
{
    const int G8 = (int)gridDim.x & ~7, b = (int)blockIdx.x;
    return b < G8 ? (b & 7) * (G8 >> 3) + (b >> 3) : b;
}

template <int TAPS>
__device__ __forceinline__ void lo_load_weights(const float* __restrict__ wp, f32x4 (&wr)[TAPS], f32x4& bias, int lane)
{
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        f32x4 v;
        v.x = wp[(t * FP + 4 * g + 0) * FP + j]; v.y = wp[(t * FP + 4 * g + 1) * FP + j];
        v.z = wp[(t * FP + 4 * g + 2) * FP + j]; v.w = wp[(t * FP + 4 * g + 3) * FP + j];
        wr[t] = v;
    }
    bias = *reinterpret_cast<const f32x4*>(wp + TAPS * FP * FP + 4 * g);
}

__device__ __forceinline__ f32x4 lo_mfma4(f32x4 a, f32x4 b, f32x4 acc)
{
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    return acc;
}

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
// The 33x33-pixel input patch of the tile is staged in LDS first (every load of the block in flight at once), in the storage
// type, pixel pitch padded so that the sixteen stride-2 pixels a wave reads per tap fall on different banks.
constexpr int IT = 2 * CT + 1;                                               // 33 input pixels per edge
template <int ST> struct S2In {
    static constexpr int PITCH = ST == ESR_STORE_F32 ? 80 : 40;              // bytes per staged pixel (16 channels + padding)
    static constexpr int BYTES = IT * IT * PITCH;
};

template <int ST>
__global__ __launch_bounds__(256) void esa_s2pool_kernel(const void* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y,
                                                         int H, int W, int H2, int W2, int H3, int W3, int tiles_x, int tiles_y)
{
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    float* const sc = reinterpret_cast<float*>(dyn);                         // conv2 outputs of this tile, [y][x][16]
    char* const sin = reinterpret_cast<char*>(sc + CT * CT * FP);            // the input patch
    constexpr int PITCH = S2In<ST>::PITCH;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    int t = lo_xcd_tile();
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int n = t / tiles_y;
    const int cy0 = 3 * PT * ty, cx0 = 3 * PT * tx;                          // first conv2 output of the tile
    const int iy0 = 2 * cy0, ix0 = 2 * cx0;
    f32x4 wr[9], bias;
    lo_load_weights<9>(wp, wr, bias, lane);
    for (int it = threadIdx.x; it < IT * IT * 4; it += 256) {
        const int qq = it & 3, pl = it >> 2;
        const int ly = pl / IT, lx = pl - ly * IT;
        const int gy = iy0 + ly, gx = ix0 + lx;
        if (ST == ESR_STORE_F32) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy < H && gx < W) v = *reinterpret_cast<const f32x4*>(static_cast<const float*>(x) + (((size_t)n * H + gy) * W + gx) * FP + qq * 4);
            *reinterpret_cast<f32x4*>(sin + pl * PITCH + qq * 16) = v;
        } else {
            uint2 v = {0u, 0u};
            if (gy < H && gx < W) v = *reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(x) + (((size_t)n * H + gy) * W + gx) * FP + qq * 4);
            *reinterpret_cast<uint2*>(sin + pl * PITCH + qq * 8) = v;
        }
    }
    __syncthreads();
    // conv2 on the matrix cores: one pixel group = one row of 16 conv2 outputs; wave w takes rows w, w + 4, ...
    for (int ly = wv; ly < CT; ly += 4) {
        f32x4 acc = bias;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                acc = lo_mfma4(wr[ky * 3 + kx], lo_ld4<ST>(sin + ((2 * ly + ky) * IT + 2 * j + kx) * PITCH, 4 * g), acc);
        *reinterpret_cast<f32x4*>(sc + (ly * CT + j) * FP + 4 * g) = acc;    // (outputs outside the map are never read by a valid window)
    }
    __syncthreads();
    if (threadIdx.x < PT * PT * 4) {
        const int qq = threadIdx.x & 3, pl = threadIdx.x >> 2;
        const int py = pl / PT, pxx = pl - py * PT;
        const int gy = PT * ty + py, gx = PT * tx + pxx;
        if (gy < H3 && gx < W3) {
            f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int ky = 0; ky < 7; ++ky)
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(sc + ((3 * py + ky) * CT + 3 * pxx + kx) * FP + qq * 4);
                    m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                }
            *reinterpret_cast<f32x4*>(y + (((size_t)n * H3 + gy) * W3 + gx) * FP + qq * 4) = m;
        }
    }
}

// ---- 16-bit storage: conv2 on the 16-bit matrix cores, pooling out of the accumulators -------------------------------------------
// Round 3 (second version).  At batch 32 the kernel above took 115 us per launch (RLFN: 13 % of a step) for 67 MB of input: its fp32
// MFMAs alone are 32 us of matrix-pipe time (16x16x4: 36 per 16 outputs, one dependent chain per wave), the pooling phase ran on 64
// of 256 threads with 49 LDS reads each, and 60 KB of LDS allowed two blocks per CU.  Here the patch stays as stored (bf16 / fp16):
//   * B operand of v_mfma_f32_16x16x32 = a TAP PAIR x 16 channels, read as stored: lane (j, kq) takes the 16 bytes of channel half
//     kq & 1 of pixel 2j + kx of tap 2q + (kq >> 1) -- ONE ds_read_b128 per pair, 5 pairs per 16 outputs (the tenth tap slot has
//     zero weights).  The patch is split by column parity ([row][x & 1][x >> 1][32 B]) so that the stride-2 pixels of a tap are
//     consecutive 32-byte slots: conflict-free without padding (the 16 lanes of an LDS group cover 16 different 16-byte slots).
//   * A operand = the fp32 weights split into 16-bit parts at block start (bf16: hi + mid + lo = 24 mantissa bits, fp16: hi + lo =
//     22), so the products are those of the fp32 weights: 15 / 10 MFMAs of 16 cycles per 16 outputs instead of 36 of 32 cycles.
//   * four independent accumulators per wave (rows w, w+4, w+8, w+12);
//   * the horizontal 7-window maximum is taken in the D fragments (lane j = conv2 column j: three DPP row shifts), only the 4 pooled
//     columns of each row go to LDS (4 KB), the vertical maximum reads 7 rows from there.  max() is exact: same result as above.
// 40 KB of LDS: four blocks per CU.
constexpr int S16_HALF = 17 * 32;                        // one column-parity half of a patch row: 17 pixels x 16 channels x 2 B
constexpr int S16_ROW = 2 * S16_HALF;
constexpr int S16_PATCH = IT * S16_ROW;                  // 35 904 B
constexpr int S16_ITEMS = IT * IT * 2;                   // 16-byte items of the patch
constexpr int S16_PER = (S16_ITEMS + 255) / 256;         // 9 per thread

template <int ST>
__device__ __forceinline__ f32x4 lo_mfma32(i32x4 a, i32x4 b, f32x4 c)
{
    if (ST == ESR_STORE_BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lo_bf16x8, a), __builtin_bit_cast(lo_bf16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(lo_f16x8, a), __builtin_bit_cast(lo_f16x8, b), c, 0, 0, 0);
}

template <int ST>
__device__ __forceinline__ unsigned lo_to16(float v, float& back)
{
    if (ST == ESR_STORE_BF16) {
        const unsigned short h = __builtin_bit_cast(unsigned short, (__bf16)v);
        back = __builtin_bit_cast(float, (unsigned)h << 16);
        return h;
    }
    const _Float16 h = (_Float16)v;
    back = (float)h;
    return __builtin_bit_cast(unsigned short, h);
}

template <int CTRL>
__device__ __forceinline__ f32x4 lo_max_shl(f32x4 m)
{
    // lane i takes max(own, lane i + n of its row of 16); lanes whose source lies outside the row keep their value
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float f = m[e];                                                  // (a copy: __builtin_bit_cast of the element lvalue m[e] reads element 0)
        const int own = __builtin_bit_cast(int, f);
        r[e] = fmaxf(f, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(own, own, CTRL, 0xf, 0xf, false)));
    }
    return r;
}

template <int ST>
__global__ __launch_bounds__(256) void esa_s2pool16_kernel(const void* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y,
                                                           int H, int W, int H3, int W3, int tiles_x, int tiles_y)
{
    constexpr int NPART = ST == ESR_STORE_BF16 ? 3 : 2;
    __shared__ __attribute__((aligned(16))) char sin[S16_PATCH];
    __shared__ __attribute__((aligned(16))) float hb[CT * PT * FP];           // horizontal maxima [conv2 row][pooled column][16]
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, kq = lane >> 4;
    int t = lo_xcd_tile();
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int n = t / tiles_y;
    const int iy0 = 6 * PT * ty, ix0 = 6 * PT * tx;                            // first input pixel of the tile (2 x 3 x PT per tile)
    // the patch: every load of the block in flight at once, rows of 1056 contiguous bytes.  Pixels past the image edge are read from
    // the edge instead (clamped): they only reach conv2 outputs outside the map, which no valid pooling window contains.
    const char* const xb = static_cast<const char*>(x) + (size_t)n * H * W * (FP * 2);
    uint4 v[S16_PER];
#pragma unroll
    for (int i = 0; i < S16_PER; ++i) {
        int it = threadIdx.x + 256 * i;
        it = it < S16_ITEMS ? it : S16_ITEMS - 1;
        const int row = it / (2 * IT), rem = it - row * (2 * IT);
        int gy = iy0 + row, gx = ix0 + (rem >> 1);
        gy = gy < H ? gy : H - 1; gx = gx < W ? gx : W - 1;
        v[i] = *reinterpret_cast<const uint4*>(xb + ((size_t)gy * W + gx) * (FP * 2) + (rem & 1) * 16);
    }
    // A fragments: lane (cout j, kq) holds k = 8 kq .. 8 kq + 7 of the pair's 32: tap 2q + (kq >> 1), channels 8 (kq & 1) + e
    i32x4 wa[5][NPART];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int tap = 2 * q + (kq >> 1);
        unsigned short part[NPART][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float w = tap < 9 ? wp[(tap * FP + 8 * (kq & 1) + e) * FP + j] : 0.f;
#pragma unroll
            for (int pp = 0; pp < NPART; ++pp) {
                float back;
                part[pp][e] = (unsigned short)lo_to16<ST>(w, back);
                w -= back;
            }
        }
#pragma unroll
        for (int pp = 0; pp < NPART; ++pp)
#pragma unroll
            for (int dw = 0; dw < 4; ++dw) wa[q][pp][dw] = (int)((unsigned)part[pp][2 * dw] | ((unsigned)part[pp][2 * dw + 1] << 16));
    }
    const f32x4 bias = *reinterpret_cast<const f32x4*>(wp + 9 * FP * FP + 4 * kq);
    // (an opaque use of every piece: without it hipcc SINKS the loads whose only use is a predicated LDS store below into that store's
    // branch -- pieces 5 .. 8 became four dependent round trips, load, vmcnt(0), store, in an 8 us kernel; round 5)
#pragma unroll
    for (int i = 0; i < S16_PER; ++i) {
        i32x4 t = {(int)v[i].x, (int)v[i].y, (int)v[i].z, (int)v[i].w};
        asm volatile("" : "+v"(t));
        v[i] = uint4{(unsigned)t.x, (unsigned)t.y, (unsigned)t.z, (unsigned)t.w};
    }
#pragma unroll
    for (int i = 0; i < S16_PER; ++i) {
        const int it = threadIdx.x + 256 * i;
        if (it < S16_ITEMS) {
            const int row = it / (2 * IT), rem = it - row * (2 * IT), px = rem >> 1;
            *reinterpret_cast<uint4*>(sin + row * S16_ROW + (px & 1) * S16_HALF + (px >> 1) * 32 + (rem & 1) * 16) = v[i];
        }
    }
    __syncthreads();
    int off[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        int tap = 2 * q + (kq >> 1);
        tap = tap < 9 ? tap : 8;                                               // the tenth slot: zero weights, any finite data
        const int ky = tap / 3, kx = tap - 3 * ky;
        off[q] = ky * S16_ROW + (kx & 1) * S16_HALF + (j + (kx >> 1)) * 32 + (kq & 1) * 16;
    }
    f32x4 acc[4] = {bias, bias, bias, bias};
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const i32x4 b = *reinterpret_cast<const i32x4*>(sin + 2 * (wv + 4 * r) * S16_ROW + off[q]);
#pragma unroll
            for (int pp = 0; pp < NPART; ++pp) acc[r] = lo_mfma32<ST>(wa[q][pp], b, acc[r]);
        }
    // D fragment: lane (j, kq) = channels 4 kq .. 4 kq + 3 of conv2 column j.  Window of pooled column p: columns 3p .. 3p + 6.
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        f32x4 m = lo_max_shl<0x101>(acc[r]);                                   // columns j .. j + 1
        m = lo_max_shl<0x102>(m);                                              // j .. j + 3
        m = lo_max_shl<0x103>(m);                                              // j .. j + 6
        if (j == 0 || j == 3 || j == 6 || j == 9)
            *reinterpret_cast<f32x4*>(hb + ((wv + 4 * r) * PT + j / 3) * FP + 4 * kq) = m;
    }
    __syncthreads();
    if (threadIdx.x < PT * PT * 4) {
        const int qq = threadIdx.x & 3, pl = threadIdx.x >> 2;
        const int py = pl / PT, pxx = pl - py * PT;
        const int gy = PT * ty + py, gx = PT * tx + pxx;
        if (gy < H3 && gx < W3) {
            f32x4 m = *reinterpret_cast<const f32x4*>(hb + ((3 * py) * PT + pxx) * FP + qq * 4);
#pragma unroll
            for (int ky = 1; ky < 7; ++ky) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(hb + ((3 * py + ky) * PT + pxx) * FP + qq * 4);
                m.x = fmaxf(m.x, u.x); m.y = fmaxf(m.y, u.y); m.z = fmaxf(m.z, u.z); m.w = fmaxf(m.w, u.w);
            }
            *reinterpret_cast<f32x4*>(y + (((size_t)n * H3 + gy) * W3 + gx) * FP + qq * 4) = m;
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
    __shared__ __attribute__((aligned(16))) float buf[2][PMAX * PMAX * BP];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    const int L = p.n_layers;
    int t = lo_xcd_tile();
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int n = t / p.tiles_y;
    // patch of the pooled map: rows y0 - L .. y0 + OT + L - 1; outside the map = 0 (the first layer's zero padding)
    int S = OT + 2 * L;
    int oy = OT * ty - L, ox = OT * tx - L;                    // map coordinates of the current patch's (0, 0)
    for (int it = threadIdx.x; it < S * S * 4; it += 256) {
        const int qq = it & 3, pl = it >> 2;
        const int ly = pl / S, lx = pl - ly * S;
        const int gy = oy + ly, gx = ox + lx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gy >= 0 && gy < p.H3 && gx >= 0 && gx < p.W3) v = *reinterpret_cast<const f32x4*>(p.x + (((size_t)n * p.H3 + gy) * p.W3 + gx) * FP + qq * 4);
        *reinterpret_cast<f32x4*>(buf[0] + pl * BP + qq * 4) = v;
    }
    __syncthreads();
    int cur = 0;
    for (int l = 0; l < L; ++l) {
        const int act = p.act[l], kind = p.kind[l];
        const bool last = l == L - 1;
        if (kind == 0) {
            // dense 3x3 on the matrix cores: S x S -> (S - 2) x (S - 2), 16 output pixels (raster order) per group
            f32x4 wr[9], bias;
            lo_load_weights<9>(p.w[l], wr, bias, lane);
            const int So = S - 2, npx = So * So;
            for (int grp = wv; grp * 16 < npx; grp += 4) {
                const int pl = grp * 16 + j;
                const bool live = pl < npx;
                const int plc = live ? pl : 0;
                const int ly = plc / So, lx = plc - ly * So;
                const float* in = buf[cur] + (ly * S + lx) * BP + 4 * g;
                f32x4 acc = bias;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        acc = lo_mfma4(wr[ky * 3 + kx], *reinterpret_cast<const f32x4*>(in + (ky * S + kx) * BP), acc);
                const int gy = oy + 1 + ly, gx = ox + 1 + lx;
                const bool inside = gy >= 0 && gy < p.H3 && gx >= 0 && gx < p.W3;
                acc.x = lo_act(acc.x, act); acc.y = lo_act(acc.y, act); acc.z = lo_act(acc.z, act); acc.w = lo_act(acc.w, act);
                if (!inside) acc = f32x4{0.f, 0.f, 0.f, 0.f};                  // the next layer's zero padding
                if (live) {
                    if (last) { if (inside) *reinterpret_cast<f32x4*>(p.y + (((size_t)n * p.H3 + gy) * p.W3 + gx) * FP + 4 * g) = acc; }
                    else *reinterpret_cast<f32x4*>(buf[cur ^ 1] + pl * BP + 4 * g) = acc;
                }
            }
            __syncthreads();
            cur ^= 1;
            S = So;
            ++oy; ++ox;
        } else {
            // BSConvU: pointwise 1x1 on the whole patch (matrix cores), zero outside the map (team18_bsrn.py:82-88 pads the pointwise
            // OUTPUT), then the depthwise 3x3 on the vector units
            {
                f32x4 wr[1], bias;
                lo_load_weights<1>(p.w[l], wr, bias, lane);
                const int npx = S * S;
                for (int grp = wv; grp * 16 < npx; grp += 4) {
                    const int pl = grp * 16 + j;
                    const bool live = pl < npx;
                    const int plc = live ? pl : 0;
                    const int ly = plc / S, lx = plc - ly * S;
                    f32x4 acc = lo_mfma4(wr[0], *reinterpret_cast<const f32x4*>(buf[cur] + plc * BP + 4 * g), bias);
                    const int gy = oy + ly, gx = ox + lx;
                    if (!(gy >= 0 && gy < p.H3 && gx >= 0 && gx < p.W3)) acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (live) *reinterpret_cast<f32x4*>(buf[cur ^ 1] + pl * BP + 4 * g) = acc;
                }
            }
            __syncthreads();
            cur ^= 1;
            const int So = S - 2, cp = p.cp[l];
            const float* __restrict__ dwl = p.wdw[l];
            for (int it = threadIdx.x; it < So * So * 4; it += 256) {
                const int qq = it & 3, pl = it >> 2;
                const int ly = pl / So, lx = pl - ly * So;
                const int gy = oy + 1 + ly, gx = ox + 1 + lx;
                const bool inside = gy >= 0 && gy < p.H3 && gx >= 0 && gx < p.W3;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (inside && qq * 4 < cp) {
                    acc = *reinterpret_cast<const f32x4*>(dwl + 9 * cp + qq * 4);
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
                            acc += *reinterpret_cast<const f32x4*>(buf[cur] + ((ly + ky) * S + lx + kx) * BP + qq * 4) *
                                   *reinterpret_cast<const f32x4*>(dwl + (ky * 3 + kx) * cp + qq * 4);
                    acc.x = lo_act(acc.x, act); acc.y = lo_act(acc.y, act); acc.z = lo_act(acc.z, act); acc.w = lo_act(acc.w, act);
                }
                if (last) { if (inside) *reinterpret_cast<f32x4*>(p.y + (((size_t)n * p.H3 + gy) * p.W3 + gx) * FP + qq * 4) = acc; }
                else *reinterpret_cast<f32x4*>(buf[cur ^ 1] + pl * BP + qq * 4) = acc;
            }
            __syncthreads();
            cur ^= 1;
            S = So;
            ++oy; ++ox;
        }
    }
}

template <int ST>
size_t s2_lds()
{
    // conv2 tile + input patch; > 64 KB for fp32 storage: the attribute is set once per device (as esr_s16.hip does)
    const size_t bytes = CT * CT * FP * sizeof(float) + S2In<ST>::BYTES;
    static std::atomic<unsigned> attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !attr_set[dev].load(std::memory_order_relaxed)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&esa_s2pool_kernel<ST>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess)
            attr_set[dev].store(1u, std::memory_order_relaxed);
    }
    return bytes;
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
        case ESR_STORE_F32: esr_note_kernel("esa_s2pool_kernel<0>"); hipLaunchKernelGGL(esa_s2pool_kernel<ESR_STORE_F32>, grid, dim3(256), s2_lds<ESR_STORE_F32>(), st, d->x.ptr, w0, pooled, d->h, d->w, H2, W2, H3, W3, tx, ty); break;
        case ESR_STORE_BF16: esr_note_kernel("esa_s2pool16_kernel<1>"); hipLaunchKernelGGL(esa_s2pool16_kernel<ESR_STORE_BF16>, grid, dim3(256), 0, st, d->x.ptr, w0, pooled, d->h, d->w, H3, W3, tx, ty); break;
        case ESR_STORE_F16: esr_note_kernel("esa_s2pool16_kernel<2>"); hipLaunchKernelGGL(esa_s2pool16_kernel<ESR_STORE_F16>, grid, dim3(256), 0, st, d->x.ptr, w0, pooled, d->h, d->w, H3, W3, tx, ty); break;
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
