// esr_hip.hip -- gfx950 (MI355X / CDNA4) kernels + C ABI for the NTIRE2022_ESR forward path.
// Interface: include/esr_hip.h.  Design notes: DESIGN.md.
//
// conv_f32_kernel: NHWC implicit-GEMM convolution (k=1|3, stride 1, same padding) in exact fp32
// on v_mfma_f32_16x16x4_f32.
//   GEMM view    D[cout][pixel] = sum_k W[cout][k] * X[k][pixel],  k = (cin chunk, tap, cin in chunk)
//   MFMA A       = weights  (lane l: A[i = l&15][k = l>>4])      16 output channels
//   MFMA B       = pixels   (lane l: B[k = l>>4][j = l&15])      16 pixels of one image row
//   MFMA D       lane l holds D[(l>>4)*4 + r][l&15], r=0..3  -> 4 CONSECUTIVE output channels of ONE
//                pixel per lane: the NHWC store is one dwordx4, and with 16c+4i+j channel order the
//                PixelShuffle(4) store is one dwordx4 of 4 horizontally adjacent HR pixels.
//   block        NW waves (4 or 8), 16 x (4*NW) output pixels x (NT*16) output channels; wave wv owns rows
//                4wv..4wv+3 (4 pixel tiles) x NT channel tiles -> 4*NT accumulators.  NW = 4: two blocks per CU;
//                NW = 8 (large 3x3 launches): one block per CU, eight waves share one weight stage.
//   K loop       input channels in chunks of 8, double-buffered LDS stages, one barrier per chunk: the chunk's
//                weights go global->LDS by DMA (global_load_lds), the halo tile global->VGPR->LDS (zero fill,
//                (pixel, half) scatter), requested two stages ahead; persistent blocks walk the tiles.
//   LDS images   input  [half h][halo pixel][4 ch]   (h = channels 0-3 | 4-7 of the chunk)
//                weight [tap][tile][lane][2]          (lane-linear: conflict-free ds_read_b64)
//                a lane (p, kq) reads channels c0+2kq+{0,1}: k-slot kq of MFMA j <-> channel c0+2kq+j,
//                identical on the A and B side by construction of the packer below.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <type_traits>

#include "esr_hip.h"
#include "esr_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int TILE = 16;      // output tile edge (pixels)
constexpr int CHUNK = 8;      // input channels per K stage
constexpr int THREADS = 256;
constexpr int MAX_RESIDENT_BLOCKS = 512;   // 256 CUs x 2 blocks (LDS: 75 KB per block)

struct ConvK {
    const float* x;
    const float* wp;      // packed weights
    const float* bias;    // NT*16 floats (inside the packed blob)
    const float* res;
    float* y0;
    float* y1;
    int N, H, W;
    int nchunks;          // ceil(cin_phys / 8)
    int cin;              // logical cin (NCHW input only)
    int in_pitch, in_coff;
    int res_pitch, res_coff;
    int y0_pitch, y0_coff, y1_pitch, y1_coff;
    int cout_store;       // round_up4(cout): channels >= this are never stored
    int split;
    int act;
    float slope;
    int res_mode;
    int out_layout;
    int tiles_x, tiles_y;
    // fused 1x1 tail (TNT > 0 kernels): see esr_conv_desc.tail_*
    const float* wp2;     // packed 1x1 weights ([chunk of 8][tile][lane][2]) followed by its bias
    const float* bias2;
    const float* cat;     // first `cat_chunks`*16 input channels of the 1x1
    int cat_pitch, cat_coff, cat_chunks;
    int mid_act;
    // post 1x1 (PNT > 0 kernels): see esr_conv_desc.post_*
    const float* wp3; const float* bias3; float* y2;
    int y2_pitch, y2_coff, y2_cout4, post_act, post_nch8;
    int res_in;           // the (pre-activation) residual IS the conv input: taken from the staged tile, no residual loads
    int out16;            // NCHW head only: the NHWC output is stored as bf16 (1) / fp16 (2) (esr_storage), 0 = fp32
    int y1_blk;           // esr_conv_desc.blocked8 & ESR_BLOCKED_OUT1: y1 is [n][y1_pitch / 8][H][W][8]
    int in_blk;           // ... & ESR_BLOCKED_IN (imdb_tail_kernel only): x is [n][in_pitch / 8][H][W][8]
    int y0_blk, res_blk;  // ... & ESR_BLOCKED_OUT0 / ESR_BLOCKED_RES (imdb_tail_kernel only, ABI v9): y0 / res are [n][pitch / 8][H][W][8]
};

__device__ __forceinline__ float act_any(float v, int act, float slope)
{
    switch (act) {
        case ESR_ACT_LRELU: return fmaxf(v, slope * v);
        case ESR_ACT_RELU: return fmaxf(v, 0.f);
        case ESR_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        default: return v;
    }
}

template <int ACT>
__device__ __forceinline__ f32x4 act4(f32x4 v, float slope)
{
    if (ACT == ESR_ACT_LRELU) {
        // 0 <= slope <= 1  =>  leaky_relu(v) = max(v, slope*v): 2 VALU per element, no compare/select
        const f32x4 m = v * slope;
        v.x = fmaxf(v.x, m.x); v.y = fmaxf(v.y, m.y); v.z = fmaxf(v.z, m.z); v.w = fmaxf(v.w, m.w);
    } else if (ACT == ESR_ACT_RELU) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    } else if (ACT == ESR_ACT_GELU) {
        v.x = 0.5f * v.x * (1.f + erff(v.x * 0.70710678118654752440f));
        v.y = 0.5f * v.y * (1.f + erff(v.y * 0.70710678118654752440f));
        v.z = 0.5f * v.z * (1.f + erff(v.z * 0.70710678118654752440f));
        v.w = 0.5f * v.w * (1.f + erff(v.w * 0.70710678118654752440f));
    }
    return v;
}

// ---- epilogues -------------------------------------------------------------------------------------
// Bias is already in the accumulators.
// (1) NHWC: the MFMA D fragment gives lane (px, kq) the 4 channels 16t+4kq.. of pixel px, i.e. 16 B from
//     each of 16 different pixels per store instruction -- 64 separate 16-byte requests (~2x the issue time
//     of a contiguous store, tools/store_pattern.hip).  Each wave therefore transposes one pixel row at a
//     time through a private LDS scratch ([16 px][64 ch + 4 pad], conflict-free both ways) so that lane l
//     owns the 16-byte chunk (l & 15) of pixel (l >> 4) + 4i: residual loads and stores are 256 contiguous
//     bytes per pixel, 1 KB per instruction for a 64-channel buffer.  All residual loads are issued up
//     front (a load->wait->add->store chain per float4 serialises: vmcnt counts stores too).
// (2) PixelShuffle(4) NCHW output: the native fragment is already ideal (16 lanes x 16 B contiguous).
constexpr int EPI_PITCH = 68;                       // floats per scratch pixel row: 64 + 4 pad
constexpr int EPI_WAVE_FLOATS = 16 * EPI_PITCH;     // one 16-pixel row per wave

// generic (bounds-checked) version: edge tiles and rarely used activation / residual combinations
template <int NT, bool Y1BLK = false>
__device__ __forceinline__ void epilogue_nhwc_checked(const ConvK& p, f32x4 (&acc)[NT][4], float* scr, int n, int x0, int y0,
                                                   int wv, int lane)
{
    const int px = lane & 15, kq = lane >> 4;       // fragment mapping
    const int ch = lane & 15, prow = lane >> 4;     // transposed mapping: chunk ch of pixel prow + 4i
    const int cb = ch * 4;
    const bool ch_ok = cb < p.cout_store;
    const bool to0 = cb < p.split;
    const int c1 = p.y1_coff + cb - p.split;          // channel inside the y1 tensor
    const size_t hw8 = (size_t)p.H * p.W * 8;
    float* const ybase = to0 ? p.y0 + p.y0_coff + cb
                             : (Y1BLK ? p.y1 + (size_t)n * hw8 * (p.y1_pitch / 8 - 1) + (size_t)(c1 >> 3) * hw8 + (c1 & 7) : p.y1 + c1);
    const int ypitch = to0 ? p.y0_pitch : (Y1BLK ? 8 : p.y1_pitch);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gy = y0 + wv * 4 + r;
#pragma unroll
        for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(scr + px * EPI_PITCH + t * 16 + kq * 4) = acc[t][r];
        for (int i = 0; i < 4; ++i) {
            const int gx = x0 + 4 * i + prow;
            f32x4 v = *reinterpret_cast<const f32x4*>(scr + (4 * i + prow) * EPI_PITCH + min(cb, NT * 16 - 4));
            if (!(ch_ok && gy < p.H && gx < p.W)) continue;
            const int pix = (n * p.H + gy) * p.W + gx;
            f32x4 rv = {0.f, 0.f, 0.f, 0.f};
            if (p.res_mode != ESR_RES_NONE) rv = *reinterpret_cast<const f32x4*>(p.res + (size_t)(pix * p.res_pitch + p.res_coff + cb));
            if (p.res_mode == ESR_RES_PRE_ACT) v += rv;
            v.x = act_any(v.x, p.act, p.slope); v.y = act_any(v.y, p.act, p.slope);
            v.z = act_any(v.z, p.act, p.slope); v.w = act_any(v.w, p.act, p.slope);
            if (p.res_mode == ESR_RES_POST_ACT) v += rv;
            *reinterpret_cast<f32x4*>(ybase + (size_t)(pix * ypitch)) = v;
        }
    }
}

// fast path: the 16x16 tile lies inside the image.  Row/pixel-group bases are wave-uniform (scalar), each
// lane adds one precomputed offset; no bounds checks; activation and residual mode are compile-time.
template <int ACT, int RES, int NT, bool Y1BLK = false>
__device__ __forceinline__ void epilogue_nhwc_fast(const ConvK& p, f32x4 (&acc)[NT][4], float* scr, int n, int x0, int y0,
                                                   int wv, int lane)
{
    const int px = lane & 15, kq = lane >> 4;
    const int ch = lane & 15, prow = lane >> 4;
    const int cb = ch * 4;
    const int pix00 = __builtin_amdgcn_readfirstlane((n * p.H + y0 + wv * 4) * p.W + x0);   // wave-uniform
    const bool to0 = cb < p.split;
    const bool to1 = !to0 && cb < p.cout_store;
    const unsigned off0 = (unsigned)(prow * p.y0_pitch + p.y0_coff + cb);
    // blocked y1 ([n][C/8][H][W][8]): element (pixel pu, channel c1) lies at pu * 8 + n * HW8 * (C/8 - 1) + (c1 / 8) * HW8 + c1 % 8
    const int c1 = p.y1_coff + cb - p.split;
    const unsigned hw8 = (unsigned)(p.H * p.W * 8);
    const unsigned off1 = Y1BLK ? (unsigned)(c1 >> 3) * hw8 + (unsigned)(prow * 8 + (c1 & 7)) : (unsigned)(prow * p.y1_pitch + c1);
    float* const y1c = p.y1 + (Y1BLK ? (size_t)n * hw8 * (p.y1_pitch / 8 - 1) : 0);
    const bool has_split = p.split < p.cout_store;                                        // uniform
    const int rd = (NT == 4) ? cb : min(cb, NT * 16 - 4);

    f32x4 rv[4][4];
    if (RES != ESR_RES_NONE) {
        const unsigned offr = (unsigned)(prow * p.res_pitch + p.res_coff + min(cb, p.cout_store - 4));
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rv[r][i] = *reinterpret_cast<const f32x4*>(p.res + (size_t)(pix00 + r * p.W + 4 * i) * p.res_pitch + offr);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(scr + px * EPI_PITCH + t * 16 + kq * 4) = acc[t][r];
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(scr + (4 * i + prow) * EPI_PITCH + rd);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 o;
            if (RES == ESR_RES_PRE_ACT) o = act4<ACT>(v[i] + rv[r][i], p.slope);
            else if (RES == ESR_RES_POST_ACT) o = act4<ACT>(v[i], p.slope) + rv[r][i];
            else o = act4<ACT>(v[i], p.slope);
            const size_t pu = (size_t)(pix00 + r * p.W + 4 * i);                            // uniform
            if (!has_split) {
                if (NT == 4 || to0) *reinterpret_cast<f32x4*>(p.y0 + pu * p.y0_pitch + off0) = o;
            } else {
                // split store (IMDBlock: 16 channels to the concat slice, 48 to the next stage) as ONE instruction with
                // per-lane 64-bit addresses: two lane-masked stores cost twice the issue time of this kernel's scarcest
                // resource, the VMEM issue slot
                float* const dst = to0 ? p.y0 + pu * p.y0_pitch + off0 : (Y1BLK ? y1c + pu * 8 + off1 : p.y1 + pu * p.y1_pitch + off1);
                if (to0 || to1) *reinterpret_cast<f32x4*>(dst) = o;
            }
        }
    }
}

template <int ACT, int NT, bool Y1BLK = false>
__device__ __forceinline__ void epilogue_nhwc_fast_res(const ConvK& p, f32x4 (&acc)[NT][4], float* scr, int n, int x0, int y0,
                                                       int wv, int lane)
{
    if (p.res_mode == ESR_RES_NONE) epilogue_nhwc_fast<ACT, ESR_RES_NONE, NT, Y1BLK>(p, acc, scr, n, x0, y0, wv, lane);
    else if (p.res_mode == ESR_RES_PRE_ACT) epilogue_nhwc_fast<ACT, ESR_RES_PRE_ACT, NT, Y1BLK>(p, acc, scr, n, x0, y0, wv, lane);
    else epilogue_nhwc_fast<ACT, ESR_RES_POST_ACT, NT, Y1BLK>(p, acc, scr, n, x0, y0, wv, lane);
}

template <int NT, bool Y1BLK = false>
__device__ __forceinline__ void epilogue_nhwc(const ConvK& p, f32x4 (&acc)[NT][4], float* scr, int n, int x0, int y0,
                                              int wv, int lane, int tile_h = TILE)
{
    const bool inside = x0 + TILE <= p.W && y0 + tile_h <= p.H && (NT < 4 || p.cout_store == 64);   // uniform
    if (inside && p.act == ESR_ACT_LRELU) epilogue_nhwc_fast_res<ESR_ACT_LRELU, NT, Y1BLK>(p, acc, scr, n, x0, y0, wv, lane);
    else if (inside && p.act == ESR_ACT_NONE) epilogue_nhwc_fast_res<ESR_ACT_NONE, NT, Y1BLK>(p, acc, scr, n, x0, y0, wv, lane);
    else if (inside && p.act == ESR_ACT_GELU && p.res_mode == ESR_RES_NONE)
        epilogue_nhwc_fast<ESR_ACT_GELU, ESR_RES_NONE, NT, Y1BLK>(p, acc, scr, n, x0, y0, wv, lane);
    else epilogue_nhwc_checked<NT, Y1BLK>(p, acc, scr, n, x0, y0, wv, lane);
}

// Channel-blocked out0 ([n][y0_pitch / 8][H][W][8], ESR_BLOCKED_OUT0: the fused IMDB tail's output since round 4), NT = 4, no residual
// (folded into the accumulators), any activation.  Same transposition through the wave's scratch as above; lane (chunk ch, pixel
// prow + 4 i) then writes 16 bytes of plane ch / 2: the four prow lanes of a chunk cover 4 consecutive pixels x 32 bytes = one whole
// 128-byte line per plane and instruction.
template <int NT>
__device__ __forceinline__ void epilogue_blk8(const ConvK& p, f32x4 (&acc)[NT][4], float* scr, int n, int x0, int y0, int wv, int lane, int tile_h)
{
    const int px = lane & 15, kq = lane >> 4;
    const int ch = lane & 15, prow = lane >> 4;
    const int cb = ch * 4;
    const int c0 = p.y0_coff + cb;
    const size_t hw8 = (size_t)p.H * p.W * 8;
    float* const ylane = p.y0 + (size_t)n * hw8 * (p.y0_pitch / 8) + (size_t)(c0 >> 3) * hw8 + (c0 & 7) + prow * 8;
    const bool inside = x0 + TILE <= p.W && y0 + tile_h <= p.H && p.cout_store == 64 && NT == 4;        // uniform
    if (inside && (p.act == ESR_ACT_NONE || p.act == ESR_ACT_LRELU)) {
        // fast path: no bounds checks, one uniform pixel base per row, the activation decided once
        const float slope = p.act == ESR_ACT_NONE ? 1.f : p.slope;
        const int pix00 = __builtin_amdgcn_readfirstlane((y0 + wv * 4) * p.W + x0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(scr + px * EPI_PITCH + t * 16 + kq * 4) = acc[t][r];
            f32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(scr + (4 * i + prow) * EPI_PITCH + cb);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 o = v[i];
                o.x = fmaxf(o.x, slope * o.x); o.y = fmaxf(o.y, slope * o.y); o.z = fmaxf(o.z, slope * o.z); o.w = fmaxf(o.w, slope * o.w);
                *reinterpret_cast<f32x4*>(ylane + (size_t)(pix00 + r * p.W + 4 * i) * 8) = o;
            }
        }
        return;
    }
    const bool ch_ok = cb < p.cout_store;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gy = y0 + wv * 4 + r;
#pragma unroll
        for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(scr + px * EPI_PITCH + t * 16 + kq * 4) = acc[t][r];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gx = x0 + 4 * i + prow;
            f32x4 v = *reinterpret_cast<const f32x4*>(scr + (4 * i + prow) * EPI_PITCH + min(cb, NT * 16 - 4));
            v.x = act_any(v.x, p.act, p.slope); v.y = act_any(v.y, p.act, p.slope);
            v.z = act_any(v.z, p.act, p.slope); v.w = act_any(v.w, p.act, p.slope);
            if (ch_ok && gy < p.H && gx < p.W) *reinterpret_cast<f32x4*>(ylane + ((size_t)gy * p.W + 4 * i + x0) * 8) = v;
        }
    }
}

// NCHW head of a 16-bit-storage network: the native fragment (4 channels of one pixel per lane) goes out as 8 bytes per lane.
// HBM-write-bound and 1 % of a forward: no transposition.
template <int NT>
__device__ __forceinline__ void epilogue_nhwc_store16(const ConvK& p, f32x4 (&acc)[NT][4], int n, int x0, int y0, int wv, int lane)
{
    const int px = lane & 15, kq = lane >> 4;
    const int gx = x0 + px;
    if (gx >= p.W) return;
    const int c8 = (p.cout_store + 7) & ~7;                  // pad channels up to the 16-byte granule are written as act(0 + bias 0) = 0
    unsigned short* const y = reinterpret_cast<unsigned short*>(p.y0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gy = y0 + wv * 4 + r;
        if (gy >= p.H) continue;
        const size_t pix = ((size_t)n * p.H + gy) * p.W + gx;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int cb = t * 16 + kq * 4;
            if (cb >= c8) continue;
            f32x4 v = acc[t][r];
            v.x = act_any(v.x, p.act, p.slope); v.y = act_any(v.y, p.act, p.slope);
            v.z = act_any(v.z, p.act, p.slope); v.w = act_any(v.w, p.act, p.slope);
            unsigned lo, hi;
            if (p.out16 == ESR_STORE_BF16) {
                typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                b2 a, b;
                a[0] = (__bf16)v.x; a[1] = (__bf16)v.y; b[0] = (__bf16)v.z; b[1] = (__bf16)v.w;
                lo = __builtin_bit_cast(unsigned, a); hi = __builtin_bit_cast(unsigned, b);
            } else {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                h2 a, b;
                a[0] = (_Float16)v.x; a[1] = (_Float16)v.y; b[0] = (_Float16)v.z; b[1] = (_Float16)v.w;
                lo = __builtin_bit_cast(unsigned, a); hi = __builtin_bit_cast(unsigned, b);
            }
            uint2 o;
            o.x = lo; o.y = hi;
            *reinterpret_cast<uint2*>(y + pix * p.y0_pitch + p.y0_coff + cb) = o;
        }
    }
}

template <int NT>
__device__ __forceinline__ void epilogue_shuffle(const ConvK& p, f32x4 (&acc)[NT][4], int n, int x0, int y0, int wv, int lane)
{
    const int px = lane & 15, kq = lane >> 4;
    const int gx = x0 + px;
    if (gx >= p.W) return;
    const size_t W4 = (size_t)p.W * 4, H4 = (size_t)p.H * 4;
    const int nco = p.cout_store / 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gy = y0 + wv * 4 + r;
        if (gy >= p.H) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t * 16 + kq * 4 >= p.cout_store) continue;
            f32x4 v = acc[t][r];
            if (p.act != ESR_ACT_NONE || p.res_mode != ESR_RES_NONE) {          // not on the networks' path: generic
                f32x4 rvv = {0.f, 0.f, 0.f, 0.f};
                if (p.res_mode != ESR_RES_NONE) {
                    const int pix = (n * p.H + gy) * p.W + gx;
                    rvv = *reinterpret_cast<const f32x4*>(p.res + (size_t)(pix * p.res_pitch + p.res_coff + t * 16 + kq * 4));
                }
                if (p.res_mode == ESR_RES_PRE_ACT) v += rvv;
                v.x = act_any(v.x, p.act, p.slope); v.y = act_any(v.y, p.act, p.slope);
                v.z = act_any(v.z, p.act, p.slope); v.w = act_any(v.w, p.act, p.slope);
                if (p.res_mode == ESR_RES_POST_ACT) v += rvv;
            }
            // out[n, t, 4gy+kq, 4gx+0..3]  (channel 16t + 4kq + j)
            float* dst = p.y0 + (((size_t)n * nco + t) * H4 + (size_t)gy * 4 + kq) * W4 + (size_t)gx * 4;
            *reinterpret_cast<f32x4*>(dst) = v;
        }
    }
}

// ---- the kernel -------------------------------------------------------------------------------------
// Persistent over tiles: block b walks tiles b, b+G, b+2G, ... (XCD-aware order) and the last K chunk of
// tile i stages chunk 0 of tile i+1, so only the very first tile of a block pays a prologue.
// TNT > 0: fused 1x1 tail (IMDBlock conv4 -> cat -> conv1x1 -> + x, basicblock.py:263-265).  The D fragment of the 3x3
// (lane (px, kq): channels 4kq..4kq+3 of pixel px) IS a B fragment of a 16-channel K chunk of the following 1x1, so the
// 3x3 result goes from accumulator registers straight into the second GEMM; the other K chunks of the 1x1 are read from
// the concat buffer as B fragments (16 bytes per lane, requested a whole chunk ahead), its weights sit in LDS for the
// life of the block ([16-channel chunk][tile][lane][4]) and its result takes the ordinary epilogue.
constexpr int TAIL_C16 = 4;                           // K of the 1x1 <= 64

// PNT > 0: "post" 1x1 -- the activated output tile of this conv (its D fragments, again B fragments of a 1x1 whose K chunks
// are this conv's output-channel tiles) also feeds a second 1x1 whose result goes to another view: RFDB's distillation
// conv c{j+1}_d = lrelu(W . r_j) is computed by the kernel that produces r_j (rfdn_baseline/block.py:150-160) instead of
// by a launch of its own that reads r_j back.
template <int NT, int KS, bool IN_NCHW, int NW, int TNT = 0, int PNT = 0, bool Y1BLK = false>
__global__ __launch_bounds__(64 * NW, 2) void conv_f32_kernel(const ConvK p)
{
    static_assert(!Y1BLK || (NT == 4 && KS == 3 && !IN_NCHW && TNT == 0 && PNT == 0), "blocked y1: the 64-output 3x3 with a split store");
    static_assert(PNT == 0 || (TNT == 0 && KS == 3 && !IN_NCHW && NW == 8), "post 1x1: 8-wave 3x3 NHWC kernels");
    static_assert(TNT == 0 || (NT == 1 && KS == 3 && !IN_NCHW && NW == 4), "tail: 3x3, <= 16 channels, 4-wave blocks");
    constexpr int THREADS = 64 * NW;                  // shadows the file-scope constant: NW waves of 4 rows each
    constexpr int TILE_H = 4 * NW;
    constexpr int HALO = KS / 2;
    constexpr int TH = TILE + 2 * HALO;               // halo tile width (also the LDS row pitch in pixels)
    constexpr int THY = TILE_H + 2 * HALO;
    constexpr int NPX = TH * THY;
    constexpr int TAPS = KS * KS;
    constexpr int IN_ITEMS = 2 * NPX;                 // 16-byte items per stage (input)
    constexpr int IN_BYTES = IN_ITEMS * 16;
    constexpr int W_ITEMS = TAPS * NT * 32;           // 16-byte items per stage (weights)
    constexpr int W_FLOATS = W_ITEMS * 4;
    constexpr int STAGE_BYTES = IN_BYTES + W_ITEMS * 16;
    constexpr int IN_ROUNDS = (IN_ITEMS + THREADS - 1) / THREADS;
    constexpr int W_ROUNDS = (W_ITEMS + THREADS - 1) / THREADS;
    constexpr unsigned OOB = 0x80000000u;             // > any per-image byte offset (host checks < 2 GiB)

    constexpr int TAIL_FLOATS = TNT ? TAIL_C16 * TNT * 256 + TNT * 16 : (PNT ? NT * PNT * 256 + PNT * 16 : 0);   // 1x1 weight image + bias
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES + NW * EPI_WAVE_FLOATS * 4 + TAIL_FLOATS * 4];

    // Issue priority: everything that is not the MFMA stream (staging, barrier, epilogue: a handful of
    // instructions per 288 MFMAs) runs at raised priority so it is issued ahead of the SIMD partner's stream.
    __builtin_amdgcn_s_setprio(3);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int px = lane & 15;
    const int kq = lane >> 4;
    float* const scr = reinterpret_cast<float*>(smem + 2 * STAGE_BYTES) + wv * EPI_WAVE_FLOATS;

    // ---- tile walk -------------------------------------------------------------------------------
    const int ntiles = p.N * p.tiles_y * p.tiles_x;
    const int G = gridDim.x;
    // Within a full group of G tiles, XCD x (blocks with b % 8 == x, as dispatched on gfx950) takes a
    // contiguous run of G/8 tiles so that neighbouring tiles share an L2.  Speed only: any map is correct.
    auto tile_index = [&](int k) -> int {
        const int base = k * G;
        if (base >= ntiles) return -1;
        int off = blockIdx.x;
        if ((G & 7) == 0 && base + G <= ntiles) off = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
        const int t = base + off;
        return t < ntiles ? t : -1;
    };

    struct TileCtx { int n, x0, y0; unsigned voff[IN_ROUNDS]; };
    // Staging descriptors.  NHWC input: one raw buffer per image, a lane's byte offset is chunk invariant,
    // the chunk advances through the SGPR soffset, out-of-image halo items use an out-of-range offset so the
    // hardware returns zeros.  No per-chunk address VALU, no select, no branch (a per-item `if (ok) load`
    // makes hipcc branch around every load and serialise them with vmcnt waits).  Item idx = (pixel, half):
    // adjacent lanes read the two 16-byte halves of one pixel's 32-byte chunk.
    auto setup_tile = [&](int t, TileCtx& c) {
        const int tx = t % p.tiles_x;
        const int tq = t / p.tiles_x;
        const int ty = tq % p.tiles_y;
        c.n = tq / p.tiles_y;
        c.x0 = tx * TILE;
        c.y0 = ty * TILE_H;
#pragma unroll
        for (int r = 0; r < IN_ROUNDS; ++r) {
            const int idx = tid + r * THREADS;
            const int half = idx & 1;
            const int pl = idx >> 1;
            const int ly = pl / TH, lx = pl - ly * TH;
            const int gy = c.y0 - HALO + ly, gx = c.x0 - HALO + lx;
            const bool ok = idx < IN_ITEMS && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            if (IN_NCHW)
                c.voff[r] = (ok && half == 0) ? (unsigned)(gy * p.W + gx) * 4u : OOB;
            else
                c.voff[r] = ok ? (unsigned)((gy * p.W + gx) * p.in_pitch + p.in_coff + 4 * half) * 4u : OOB;
        }
    };
    const size_t img_floats = (size_t)p.H * p.W * (IN_NCHW ? p.cin : p.in_pitch);
    auto image_rsrc = [&](int n) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)n * img_floats), 0, (int)(img_floats * 4), 0x00020000);
    };
    f32x4 in_reg[IN_ROUNDS];

    auto load_input = [&](const TileCtx& tc, int c) {
        const __amdgpu_buffer_rsrc_t xrsrc = image_rsrc(tc.n);
#pragma unroll
        for (int r = 0; r < IN_ROUNDS; ++r) {
            if (IN_NCHW) {
                const unsigned plane = (unsigned)(p.H * p.W) * 4u;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                v.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, tc.voff[r], 0, 0));
                if (p.cin > 1) v.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, tc.voff[r], plane, 0));
                if (p.cin > 2) v.z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, tc.voff[r], 2 * plane, 0));
                if (p.cin > 3) v.w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, tc.voff[r], 3 * plane, 0));
                in_reg[r] = v;
            } else {
                in_reg[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, tc.voff[r], c * (CHUNK * 4), 0));
            }
        }
    };
    // weights go global -> LDS directly (their LDS image is lane-linear: item idx at byte idx*16; W_ITEMS is a multiple
    // of 64, so a wave is entirely inside or outside the range): no staging VGPRs, no ds_write.
    // The DMA is issued from inline asm on purpose: with the builtin, hipcc assumes every later ds_read may alias the
    // DMA destination and puts `s_waitcnt vmcnt(0)` in front of the first fragment read of each chunk, stalling the wave
    // on the weight DMA and on the input loads still in flight.  Hidden from the compiler, its own vmcnt waits can only
    // over-wait (the asm ops add to the count), and stage_barrier() below waits for the DMA explicitly.
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto load_weights = [&](int c, int dstbuf) {
        const float* wsrc = p.wp + (size_t)c * W_FLOATS;
        const unsigned wdst = smem_lds + dstbuf * STAGE_BYTES + IN_BYTES;
#pragma unroll
        for (int r = 0; r < W_ROUNDS; ++r) {
            const int idx = tid + r * THREADS;
            if (W_ITEMS % THREADS == 0 || idx < W_ITEMS) {
                const float* g = wsrc + idx * 4;
                const unsigned dst = __builtin_amdgcn_readfirstlane(wdst + (unsigned)(idx & ~63) * 16u);   // wave-uniform LDS base
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
            }
        }
    };
    auto store_stage = [&](int buf) {
        char* s = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int r = 0; r < IN_ROUNDS; ++r) {
            const int idx = tid + r * THREADS;
            if (IN_ITEMS % THREADS == 0 || idx < IN_ITEMS)
                *reinterpret_cast<f32x4*>(s + (idx & 1) * (NPX * 16) + (idx >> 1) * 16) = in_reg[r];
        }
    };
    // End-of-stage synchronisation.  `ahead` = this wave has just issued the IN_ROUNDS input loads of the stage after
    // next; VMEM loads (LDS-DMA included) return in issue order, so vmcnt(IN_ROUNDS) guarantees the older weight DMA
    // of the next stage has landed while those newest loads stay in flight across the barrier.
    auto stage_barrier = [&](bool ahead) {
        if (ahead) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(IN_ROUNDS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    // lane-constant LDS byte offsets
    const int b_base = (kq >> 1) * (NPX * 16) + ((wv * 4) * TH + px) * 16 + (kq & 1) * 8;
    const int a_base = IN_BYTES + lane * 8;

    // Staging pipeline (per block, stages g = (tile, chunk) in order): at the top of a stage body its LDS buffer is
    // complete and -- when `inflight` -- the INPUT of stage g+1 is already on its way into in_reg (requested before the
    // previous barrier, so it has the barrier gap plus a whole chunk of MFMAs to arrive: tools/dbg/ablate.py showed
    // 5 % of the kernel waiting for these loads with a one-chunk lead).  The body requests the weights of g+1 (they can
    // only start now: they land in the buffer the previous stage was reading), runs the MFMAs, writes in_reg to LDS and
    // requests the input of stage g+2.
    constexpr bool STAGGER = NW == 8;           // implies KS == 3, NHWC input, nchunks >= 2 (host dispatch)
    const bool late = STAGGER && wv >= 4;
    constexpr bool AHEAD = !IN_NCHW;            // the NCHW head issues a cin-dependent number of loads: plain vmcnt(0)
    int k = 0;
    int t = tile_index(0);
    if (t < 0) return;
    TileCtx cur, nxt;
    // bias of this lane's 4 output channels per tile, loaded ONCE: a per-tile reload would make the compiler wait
    // vmcnt(0) at every tile start, i.e. also for the input loads deliberately left in flight across the barrier
    f32x4 biasv[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) biasv[tt] = *reinterpret_cast<const f32x4*>(p.bias + tt * 16 + kq * 4);

    float* const wl = reinterpret_cast<float*>(smem + 2 * STAGE_BYTES + NW * EPI_WAVE_FLOATS * 4);
    if (TNT) {
        // blob order in, (chunk16, tile, lane, j) order out: float4 q = elements (i, j'=0), (i, 1), (i+1, 0), (i+1, 1) of one
        // (chunk8, tile, kq'); channel 8*chunk8 + 2kq' + j' -> chunk16 = chunk8/2, slot (ch16 >> 2), j = ch16 & 3
        const int nch8 = 2 * (p.cat_chunks + 1);
        for (int q = tid; q < nch8 * TNT * 32; q += THREADS) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p.wp2 + (size_t)q * 4);
            const int ip = q & 7, kq8 = (q >> 3) & 3, ct = q >> 5;
            const int tt = ct % TNT, chunk8 = ct / TNT;
            const int ch16 = 8 * (chunk8 & 1) + 2 * kq8;
            float* dst = wl + (((chunk8 >> 1) * TNT + tt) * 64 + (ch16 >> 2) * 16 + 2 * ip) * 4 + (ch16 & 3);
            dst[0] = v.x; dst[1] = v.y; dst[4] = v.z; dst[5] = v.w;
        }
        if (tid < TNT * 16) wl[TAIL_C16 * TNT * 256 + tid] = p.bias2[tid];
    }
    if (PNT) {
        // same blob -> image transform as the tail's; K chunks of 16 = this conv's NT output tiles
        const int nch8 = p.post_nch8;                        // <= 2 * NT; the image beyond it stays zero
        for (int e = tid; e < NT * PNT * 256; e += THREADS) wl[e] = 0.f;
        __syncthreads();
        for (int q = tid; q < nch8 * PNT * 32; q += THREADS) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p.wp3 + (size_t)q * 4);
            const int ip = q & 7, kq8 = (q >> 3) & 3, ct = q >> 5;
            const int tt = ct % PNT, chunk8 = ct / PNT;
            const int ch16 = 8 * (chunk8 & 1) + 2 * kq8;
            float* dst = wl + (((chunk8 >> 1) * PNT + tt) * 64 + (ch16 >> 2) * 16 + 2 * ip) * 4 + (ch16 & 3);
            dst[0] = v.x; dst[1] = v.y; dst[4] = v.z; dst[5] = v.w;
        }
        if (tid < PNT * 16) wl[NT * PNT * 256 + tid] = p.bias3[tid];
    }
    const size_t cat_img_floats = (size_t)p.H * p.W * p.cat_pitch;

    setup_tile(t, cur);
    int tn = tile_index(1);
    if (tn >= 0) setup_tile(tn, nxt);
    load_input(cur, 0);
    load_weights(0, 0);
    store_stage(0);
    bool inflight = false;
    if (AHEAD) {
        if (p.nchunks > 1) { load_input(cur, 1); inflight = true; }
        else if (tn >= 0) { load_input(nxt, 0); inflight = true; }
    }
    stage_barrier(inflight);
    int sbuf = 0;

    for (;;) {
        const bool has_next = tn >= 0;

        // accumulators start at the bias: no bias add in the epilogue
        f32x4 acc[NT][4];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[tt][r] = biasv[tt];

        f32x4 bc[TNT ? TAIL_C16 - 1 : 1][4];            // tail: B fragments of the concat part of the 1x1
        for (int c = 0; c < p.nchunks; ++c) {
            const bool more = c + 1 < p.nchunks;
            if (TNT && !more) {
                // requested a whole chunk before use, and BEFORE this chunk's weight DMA / input requests so that the
                // counted vmcnt of stage_barrier() still sees the input loads as the newest
                const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(p.cat + (size_t)cur.n * cat_img_floats), 0, (int)(cat_img_floats * 4), 0x00020000);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gy = cur.y0 + wv * 4 + r, gx = cur.x0 + px;
                    const unsigned vo = (gy < p.H && gx < p.W) ? (unsigned)((gy * p.W + gx) * p.cat_pitch + p.cat_coff + 4 * kq) * 4u : OOB;
#pragma unroll
                    for (int C = 0; C < TAIL_C16 - 1; ++C)
                        if (C < p.cat_chunks) bc[C][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(crsrc, vo, C * 64, 0));
                }
            }
            // 8-wave blocks: the two waves of a SIMD belong to the same block and would stage (and then compete for the
            // MFMA pipe) in lockstep.  Waves 0-3 therefore do all their staging at the top of the chunk, waves 4-7 in the
            // middle of it; needs nchunks >= 2 so that the input of the next stage is always already in flight.
            auto staging_block = [&]() {
                if (more || has_next) {
                    store_stage(sbuf ^ 1);
                    load_weights(more ? c + 1 : 0, sbuf ^ 1);
                }
                inflight = false;
                if (c + 2 < p.nchunks) { load_input(cur, c + 2); inflight = true; }
                else if (has_next && c + 2 - p.nchunks < p.nchunks) { load_input(nxt, c + 2 - p.nchunks); inflight = true; }
            };
            if (STAGGER) {
                if (!late) staging_block();
            } else if (more) load_weights(c + 1, sbuf ^ 1);
            else if (has_next) load_weights(0, sbuf ^ 1);
            if (!STAGGER && !inflight) {                               // one-chunk lead (head conv, or nothing was requested ahead)
                if (more) load_input(cur, c + 1);
                else if (has_next) load_input(nxt, 0);
            }
            const char* s = smem + sbuf * STAGE_BYTES;
            if (KS == 3 && !IN_NCHW && TNT == 0 && p.res_in) {
                // act(conv(x) + x) (RFDB, rfdn_baseline/block.py:151-157): the residual of output channels 8c..8c+7 is the
                // centre pixel of input chunk c, which is in LDS right now in exactly the layout of the accumulator rows
                // ([half][pixel][4 channels]); 4 ds_reads for half the lanes replace 16 global loads per tile in the epilogue
                const int ttc = c >> 1;
                if ((kq >> 1) == (c & 1)) {
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt)
                        if (tt == ttc) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                acc[tt][r] += *reinterpret_cast<const f32x4*>(s + (kq & 1) * (NPX * 16) + ((wv * 4 + r + 1) * TH + px + 1) * 16);
                        }
                }
            }
            // fragment reads run one tap ahead of the MFMAs that consume them
            f32x2 a[2][NT], b[2][4];
            auto load_frag = [&](int slot, int tap) {
                const int dy = tap / KS, dx = tap - dy * KS;
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
                    a[slot][tt] = *reinterpret_cast<const f32x2*>(s + a_base + (tap * NT + tt) * 512);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    b[slot][r] = *reinterpret_cast<const f32x2*>(s + b_base + ((r + dy) * TH + dx) * 16);
            };
            load_frag(0, 0);
            __builtin_amdgcn_s_setprio(0);             // only the MFMA stream runs at base priority
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int cs = tap & 1;
                if (tap + 1 < TAPS) load_frag(cs ^ 1, tap + 1);
                __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ABOVE this tap's MFMAs
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[tt][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cs][tt][j], b[cs][r][j], acc[tt][r], 0, 0, 0);
                if (STAGGER && tap == TAPS / 2 && late) {
                    __builtin_amdgcn_s_setprio(3);
                    staging_block();
                    __builtin_amdgcn_s_setprio(0);
                }
            }
            __builtin_amdgcn_s_setprio(3);
            if (!STAGGER && (more || has_next)) store_stage(sbuf ^ 1);
            if (!STAGGER) inflight = false;
            if (AHEAD && !STAGGER) {                                   // input of the stage after next
                if (c + 2 < p.nchunks) { load_input(cur, c + 2); inflight = true; }
                else if (has_next && c + 2 - p.nchunks < p.nchunks) { load_input(nxt, c + 2 - p.nchunks); inflight = true; }
            }
            stage_barrier(inflight);
            sbuf ^= 1;
        }

        if (TNT) {
            // second GEMM: K = concat chunks (from global, requested at the top of the last chunk) + the 3x3 result
            f32x4 acc2[TNT ? TNT : 1][4];
#pragma unroll
            for (int tt = 0; tt < TNT; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc2[tt][r] = *reinterpret_cast<const f32x4*>(wl + TAIL_C16 * TNT * 256 + tt * 16 + kq * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x4 v = acc[0][r];
                v.x = act_any(v.x, p.mid_act, p.slope); v.y = act_any(v.y, p.mid_act, p.slope);
                v.z = act_any(v.z, p.mid_act, p.slope); v.w = act_any(v.w, p.mid_act, p.slope);
                acc[0][r] = v;
            }
            auto tail_chunk = [&](int C, const f32x4 (&bf)[4]) {
                f32x4 a2[TNT ? TNT : 1];
#pragma unroll
                for (int tt = 0; tt < TNT; ++tt) a2[tt] = *reinterpret_cast<const f32x4*>(wl + ((C * TNT + tt) * 64 + lane) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int tt = 0; tt < TNT; ++tt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc2[tt][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[tt][j], bf[r][j], acc2[tt][r], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);          // one chunk's A fragments at a time (register budget)
            };
#pragma unroll
            for (int C = 0; C < TAIL_C16 - 1; ++C)
                if (C < p.cat_chunks) tail_chunk(C, bc[C]);
            tail_chunk(p.cat_chunks, acc[0]);
            epilogue_nhwc<(TNT ? TNT : 1)>(p, acc2, scr, cur.n, cur.x0, cur.y0, wv, lane, TILE_H);
        } else if (PNT) {
            // post GEMM from activated copies of the fragments (host: no residual left for the epilogue, so act(acc) IS the
            // conv's output), then the ordinary epilogue of the main output
            f32x4 accd[PNT ? PNT : 1][4];
#pragma unroll
            for (int td = 0; td < PNT; ++td)
#pragma unroll
                for (int r = 0; r < 4; ++r) accd[td][r] = *reinterpret_cast<const f32x4*>(wl + NT * PNT * 256 + td * 16 + kq * 4);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                f32x4 bv[4];                                  // activated copy: the B operand of the post GEMM
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    f32x4 v = acc[tt][r];
                    v.x = act_any(v.x, p.act, p.slope); v.y = act_any(v.y, p.act, p.slope);
                    v.z = act_any(v.z, p.act, p.slope); v.w = act_any(v.w, p.act, p.slope);
                    bv[r] = v;
                }
                f32x4 a3[PNT ? PNT : 1];
#pragma unroll
                for (int td = 0; td < PNT; ++td) a3[td] = *reinterpret_cast<const f32x4*>(wl + ((tt * PNT + td) * 64 + lane) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int td = 0; td < PNT; ++td)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            accd[td][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[td][j], bv[r][j], accd[td][r], 0, 0, 0);
            }
            {
                const int gx = cur.x0 + px;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gy = cur.y0 + wv * 4 + r;
                    if (gy >= p.H || gx >= p.W) continue;
                    float* dst = p.y2 + ((size_t)(cur.n * p.H + gy) * p.W + gx) * p.y2_pitch + p.y2_coff;
#pragma unroll
                    for (int td = 0; td < PNT; ++td) {
                        if (td * 16 + kq * 4 >= p.y2_cout4) continue;
                        f32x4 v = accd[td][r];
                        v.x = act_any(v.x, p.post_act, p.slope); v.y = act_any(v.y, p.post_act, p.slope);
                        v.z = act_any(v.z, p.post_act, p.slope); v.w = act_any(v.w, p.post_act, p.slope);
                        *reinterpret_cast<f32x4*>(dst + td * 16 + kq * 4) = v;
                    }
                }
            }
            epilogue_nhwc<NT>(p, acc, scr, cur.n, cur.x0, cur.y0, wv, lane, TILE_H);
        } else if (p.out_layout == ESR_NCHW_SHUFFLE4) epilogue_shuffle<NT>(p, acc, cur.n, cur.x0, cur.y0, wv, lane);
        else if (IN_NCHW && p.out16) epilogue_nhwc_store16<NT>(p, acc, cur.n, cur.x0, cur.y0, wv, lane);
        else epilogue_nhwc<NT, Y1BLK>(p, acc, scr, cur.n, cur.x0, cur.y0, wv, lane, TILE_H);
        if (!has_next) break;
        cur = nxt;
        ++k;
        tn = tile_index(k + 1);
        if (tn >= 0) setup_tile(tn, nxt);
    }
}

#include "imdb_tail.inc"                   // IMDBlock's fused tail at the network's own shape: DMA ring of three, residual folded

thread_local char g_err[256] = "";
thread_local char g_kname[256] = "";       // device symbol(s) of the op being run (esr_note_kernel)
thread_local bool g_ktrace = false;

inline void set_err(const char* what, hipError_t e) { esr_set_err(what, e); }
inline int round_up(int v, int m) { return esr_round_up(v, m); }

// Which conv_f32_kernel variant a launch takes: 8-wave blocks on 16x32-pixel tiles (one per CU) for large 3x3 NHWC
// launches, else 4-wave blocks on 16x16 tiles (two per CU).  The eight waves share one weight stage, so a SIMD issues a
// third fewer staging instructions per MFMA (they, not the MFMA pipe, bound this kernel: DESIGN.md).
constexpr int TALL_MIN_TILES = 256;        // one 16x32 tile per CU: below that the 4-wave shape fills the chip better
inline int conv_block_waves(int ksize, bool in_nchw, int nt, int nchunks, int n, int h, int w)
{
    const int tall_min = TALL_MIN_TILES;
    if (ksize != 3 || in_nchw || nt < 3 || nchunks < 2) return 4;
    const long ntall = (long)n * ((w + TILE - 1) / TILE) * ((h + 31) / 32);
    return ntall >= tall_min ? 8 : 4;
}

template <int NT, int KS, bool IN_NCHW>
int launch_conv(const ConvK& k, hipStream_t st)
{
    // persistent: at most 2 blocks per CU (LDS-limited), each walks ntiles/grid tiles
    const int ntiles = k.N * k.tiles_x * k.tiles_y;
    constexpr bool CAN_TALL = KS == 3 && !IN_NCHW && NT >= 3;
    if (CAN_TALL && conv_block_waves(KS, IN_NCHW, NT, k.nchunks, k.N, k.H, k.W) == 8) {
        const int tall_y = (k.H + 31) / 32;
        const int ntall = k.N * k.tiles_x * tall_y;
        ConvK kk = k;
        kk.tiles_y = tall_y;
        const int grid = ntall < 256 ? ntall : 256;
        esr_note_kernel("conv_f32_kernel<%d, %d, %s, 8, 0, %d, %s>", NT, KS, esr_tf(IN_NCHW), (NT == 4 && kk.wp3) ? 2 : 0, esr_tf(NT == 4 && !kk.wp3 && kk.y1_blk));
        if (NT == 4 && kk.wp3)
            hipLaunchKernelGGL((conv_f32_kernel<NT, KS, IN_NCHW, CAN_TALL ? 8 : 4, 0, (CAN_TALL && NT == 4) ? 2 : 0>), dim3(grid), dim3(512), 0, st, kk);
        else if (NT == 4 && kk.y1_blk)
            hipLaunchKernelGGL((conv_f32_kernel<NT, KS, IN_NCHW, CAN_TALL ? 8 : 4, 0, 0, CAN_TALL && NT == 4>), dim3(grid), dim3(512), 0, st, kk);
        else
            hipLaunchKernelGGL((conv_f32_kernel<NT, KS, IN_NCHW, CAN_TALL ? 8 : 4>), dim3(grid), dim3(512), 0, st, kk);
        esr_graph_note_io(st, kk.x, offsetof(ConvK, x), kk.y0, offsetof(ConvK, y0));
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            set_err("conv_f32_kernel (tall) launch", e);
            return ESR_ERR_LAUNCH;
        }
        return ESR_OK;
    }
    const int grid = ntiles < MAX_RESIDENT_BLOCKS ? ntiles : MAX_RESIDENT_BLOCKS;
    esr_note_kernel("conv_f32_kernel<%d, %d, %s, 4, 0, 0, %s>", NT, KS, esr_tf(IN_NCHW), esr_tf(CAN_TALL && NT == 4 && k.y1_blk));
    if (CAN_TALL && NT == 4 && k.y1_blk) hipLaunchKernelGGL((conv_f32_kernel<NT, KS, IN_NCHW, 4, 0, 0, CAN_TALL && NT == 4>), dim3(grid), dim3(THREADS), 0, st, k);
    else hipLaunchKernelGGL((conv_f32_kernel<NT, KS, IN_NCHW, 4>), dim3(grid), dim3(THREADS), 0, st, k);
    esr_graph_note_io(st, k.x, offsetof(ConvK, x), k.y0, offsetof(ConvK, y0));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_err("conv_f32_kernel launch", e);
        return ESR_ERR_LAUNCH;
    }
    return ESR_OK;
}

// IMDBlock's own shape (48 -> 16 3x3, 3 x 16 concat channels, 64 outputs, no or a pre-activation residual) takes imdb_tail_kernel
inline bool imdb_tail_shape(const ConvK& k)
{
    if (k.nchunks != IT_NCH || k.cat_chunks != IT_CAT || k.cout_store != 64) return false;
    if (k.res_mode != ESR_RES_NONE && k.res_mode != ESR_RES_PRE_ACT) return false;
    const double px = (double)k.H * k.W * 4.0;
    return px * k.in_pitch < 2147483647.0 && px * k.cat_pitch < 2147483647.0 && (k.res_mode == ESR_RES_NONE || px * k.res_pitch < 2147483647.0);
}

int launch_conv_tail(const ConvK& k, hipStream_t st)
{
    const int ntiles = k.N * k.tiles_x * k.tiles_y;
    const int grid = ntiles < MAX_RESIDENT_BLOCKS ? ntiles : MAX_RESIDENT_BLOCKS;
    if (imdb_tail_shape(k)) {
        ConvK kk = k;
        kk.res_mode = ESR_RES_NONE;                 // folded into the 1x1's accumulators: the epilogue adds nothing
        kk.tiles_y = (k.H + 4 * IT_NW - 1) / (4 * IT_NW);      // 16 x 32 pixel tiles, one 8-wave block per CU
        const int nt32 = k.N * kk.tiles_x * kk.tiles_y;
        const int g32 = nt32 < 256 ? nt32 : 256;
        esr_note_kernel("imdb_tail_kernel<%s>", esr_tf(k.res_mode == ESR_RES_PRE_ACT));
        if (k.res_mode == ESR_RES_PRE_ACT) hipLaunchKernelGGL((imdb_tail_kernel<true>), dim3(g32), dim3(64 * IT_NW), 0, st, kk);
        else hipLaunchKernelGGL((imdb_tail_kernel<false>), dim3(g32), dim3(64 * IT_NW), 0, st, kk);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            set_err("imdb_tail_kernel launch", e);
            return ESR_ERR_LAUNCH;
        }
        return ESR_OK;
    }
    esr_note_kernel("conv_f32_kernel<1, 3, false, 4, 4, 0, false>");
    hipLaunchKernelGGL((conv_f32_kernel<1, 3, false, 4, 4>), dim3(grid), dim3(THREADS), 0, st, k);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_err("conv_f32_kernel (1x1 tail) launch", e);
        return ESR_ERR_LAUNCH;
    }
    return ESR_OK;
}

template <int KS, bool IN_NCHW>
int launch_conv_nt(int nt, const ConvK& k, hipStream_t st)
{
    switch (nt) {
        case 1: return launch_conv<1, KS, IN_NCHW>(k, st);
        case 2: return launch_conv<2, KS, IN_NCHW>(k, st);
        case 3: return launch_conv<3, KS, IN_NCHW>(k, st);
        case 4: return launch_conv<4, KS, IN_NCHW>(k, st);
    }
    return ESR_ERR_UNSUPPORTED;
}

}  // namespace

void esr_set_err(const char* what, hipError_t e)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}

void esr_note_kernel(const char* fmt, ...)
{
    if (!g_ktrace) return;
    size_t n = strlen(g_kname);
    if (n && n + 3 < sizeof(g_kname)) { memcpy(g_kname + n, " + ", 4); n += 3; }      // an op lowered to two launches
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kname + n, sizeof(g_kname) - n, fmt, ap);
    va_end(ap);
}

int esr_check_launch(const char* what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        esr_set_err(what, e);
        return ESR_ERR_LAUNCH;
    }
    return ESR_OK;
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int esr_abi_version(void) { return ESR_ABI_VERSION; }
const char* esr_last_hip_error(void) { return g_err; }
size_t esr_sizeof(int which)
{
    switch (which) {
        case 0: return sizeof(esr_view);
        case 1: return sizeof(esr_conv_desc);
        case 2: return sizeof(esr_esa_desc);
        case 3: return sizeof(esr_bsconv_desc);
        case 4: return sizeof(esr_ca_desc);
        case 5: return sizeof(esr_op);
        case 6: return sizeof(esr_esa_lowres_desc);
        case 7: return sizeof(esr_chain_desc);
        default: return 0;
    }
}

const char* esr_build_info(void) { return "gfx950 f32:v_mfma_f32_16x16x4_f32 (Winograd F(2x2,3x3) / direct, tiles 16x16/16x32, chunk 8) s16:v_mfma_f32_16x16x32_{bf16,f16} (16-bit storage: conv_s16 LDS-DMA ring, chunk 16; conv48r / conv48rp weights in registers, whole-pixel stages) persistent"; }

size_t esr_packed_conv_bytes(int cin_phys, int cout, int ksize)
{
    if (cin_phys <= 0 || cout <= 0 || (ksize != 1 && ksize != 3)) return 0;
    const size_t nt = (size_t)round_up(cout, 16) / 16;
    const size_t nchunks = (size_t)round_up(cin_phys, CHUNK) / CHUNK;
    return (nchunks * ksize * ksize * nt * 128 + nt * 16) * sizeof(float);
}

static int pack_index(int nt, int taps, int slot, int tap, int o, int* j_out)
{
    const int chunk = slot / CHUNK, within = slot % CHUNK;
    const int kq = within / 2, j = within % 2;
    const int t = o / 16, i = o % 16;
    *j_out = j;
    return (((chunk * taps + tap) * nt + t) * 64 + kq * 16 + i) * 2 + j;
}

int esr_pack_conv_f32(const float* w, const float* bias, int cin, int cout, int ksize,
                      const int32_t* cin_map, int cin_phys, void* out, size_t out_bytes)
{
    if (!w || !out || cin <= 0 || cout <= 0 || (ksize != 1 && ksize != 3)) return ESR_ERR_BAD_ARG;
    if (!cin_map && cin_phys < cin) return ESR_ERR_BAD_ARG;
    const size_t need = esr_packed_conv_bytes(cin_phys, cout, ksize);
    if (need == 0 || out_bytes < need) return ESR_ERR_BAD_ARG;
    const int nt = round_up(cout, 16) / 16, taps = ksize * ksize;
    float* o = static_cast<float*>(out);
    memset(o, 0, need);
    for (int s = 0; s < cin_phys; ++s) {
        const int c = cin_map ? cin_map[s] : (s < cin ? s : -1);
        if (c < 0) continue;
        if (c >= cin) return ESR_ERR_BAD_ARG;
        for (int oc = 0; oc < cout; ++oc)
            for (int tap = 0; tap < taps; ++tap) {
                int j;
                o[pack_index(nt, taps, s, tap, oc, &j)] = w[((size_t)oc * cin + c) * taps + tap];
            }
    }
    float* bo = o + (size_t)(round_up(cin_phys, CHUNK) / CHUNK) * taps * nt * 128;
    if (bias)
        for (int oc = 0; oc < cout; ++oc) bo[oc] = bias[oc];
    return ESR_OK;
}

int esr_unpack_conv_f32(const void* packed, size_t bytes, int cin, int cout, int ksize,
                        const int32_t* cin_map, int cin_phys, float* w, float* bias)
{
    if (!packed || !w || cin <= 0 || cout <= 0 || (ksize != 1 && ksize != 3)) return ESR_ERR_BAD_ARG;
    if (bytes < esr_packed_conv_bytes(cin_phys, cout, ksize)) return ESR_ERR_BAD_ARG;
    const int nt = round_up(cout, 16) / 16, taps = ksize * ksize;
    const float* o = static_cast<const float*>(packed);
    memset(w, 0, sizeof(float) * (size_t)cout * cin * taps);
    for (int s = 0; s < cin_phys; ++s) {
        const int c = cin_map ? cin_map[s] : (s < cin ? s : -1);
        if (c < 0) continue;
        for (int oc = 0; oc < cout; ++oc)
            for (int tap = 0; tap < taps; ++tap) {
                int j;
                w[((size_t)oc * cin + c) * taps + tap] = o[pack_index(nt, taps, s, tap, oc, &j)];
            }
    }
    if (bias) {
        const float* bo = o + (size_t)(round_up(cin_phys, CHUNK) / CHUNK) * taps * nt * 128;
        for (int oc = 0; oc < cout; ++oc) bias[oc] = bo[oc];
    }
    return ESR_OK;
}

int esr_conv_block_waves(const esr_conv_desc* d)
{
    if (!d || d->cin <= 0 || d->cout <= 0) return 0;
    if (d->storage != ESR_STORE_F32 && d->in_layout == ESR_NHWC) return esr_s16_block_waves(d);      // conv_s16_kernel: 8-wave blocks, or two of 4
    const bool in_nchw = d->in_layout == ESR_NCHW_IN;
    const int cin_phys = in_nchw ? CHUNK : round_up(d->cin, CHUNK);
    return conv_block_waves(d->ksize, in_nchw, round_up(d->cout, 16) / 16, cin_phys / CHUNK, d->n, d->h, d->w);
}

int esr_conv2d_f32(const esr_conv_desc* d, void* hip_stream)
{
    if (!d || !d->in.ptr || !d->wpacked) return ESR_ERR_BAD_ARG;
    if (!d->out0.ptr && !(d->post_wpacked && d->storage != ESR_STORE_F32)) return ESR_ERR_BAD_ARG;   // 16-bit post chain may consume the result alone
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cin <= 0 || d->cout <= 0) return ESR_ERR_BAD_ARG;
    if (d->ksize != 1 && d->ksize != 3) return ESR_ERR_UNSUPPORTED;
    if (d->cout > 64) return ESR_ERR_UNSUPPORTED;
    const bool in_nchw = d->in_layout == ESR_NCHW_IN;
    if (in_nchw && (d->cin > 4 || d->ksize != 3)) return ESR_ERR_UNSUPPORTED;
    if (!in_nchw && d->in_layout != ESR_NHWC) return ESR_ERR_BAD_ARG;
    const bool store16 = d->storage == ESR_STORE_BF16 || d->storage == ESR_STORE_F16;
    if (d->storage != ESR_STORE_F32 && !store16) return ESR_ERR_BAD_ARG;
    if (store16 && !in_nchw) return esr_conv2d_s16(d, hip_stream);         // 16-bit storage: esr_s16.hip
    if (d->compute != ESR_COMPUTE_F32) return ESR_ERR_BAD_ARG;              // fp32 MFMA from here on (incl. the NCHW head)
    if (d->border_bias) return ESR_ERR_UNSUPPORTED;                        // border table: conv_s16_kernel only
    if (d->hilo) return ESR_ERR_UNSUPPORTED;                               // hi + lo pairs: esr_conv2d_s16 only (a low tensor would be silently ignored here)
    if (d->in_seg_stride != 0) return ESR_ERR_UNSUPPORTED;                 // segmented input: conv_s16_kernel only
    if (store16) {
        // the network head with 16-bit activations downstream: fp32 NCHW input (exact), fp32 MFMA, 16-bit NHWC store
        if (d->out_layout != ESR_NHWC || d->res_mode != ESR_RES_NONE || d->tail_wpacked || d->post_wpacked) return ESR_ERR_UNSUPPORTED;
        if (d->split > 0 && d->split < d->cout) return ESR_ERR_UNSUPPORTED;
        if ((d->out0.pitch & 7) || (d->out0.coff & 7) || d->out0.coff + round_up(d->cout, 8) > d->out0.pitch) return ESR_ERR_BAD_ARG;
    }
    if (!in_nchw && ((d->in.pitch & 3) || (d->in.coff & 3))) return ESR_ERR_BAD_ARG;
    const int cin_phys = in_nchw ? CHUNK : round_up(d->cin, CHUNK);
    if (!in_nchw && d->in.coff + cin_phys > d->in.pitch) return ESR_ERR_BAD_ARG;   // chunk reads stay inside the pixel
    // fused 1x1 tail: the epilogue fields describe the 1x1's output
    const bool tail = d->tail_wpacked != nullptr;
    if (tail) {
        if (d->ksize != 3 || in_nchw || d->out_layout != ESR_NHWC || d->cout > 16) return ESR_ERR_UNSUPPORTED;
        if (d->tail_cat_c <= 0 || (d->tail_cat_c & 15) || d->tail_cat_c + 16 > 16 * TAIL_C16) return ESR_ERR_UNSUPPORTED;
        if (d->tail_cout <= 48 || d->tail_cout > 64) return ESR_ERR_UNSUPPORTED;
        if (!d->tail_cat.ptr || (d->tail_cat.pitch & 3) || (d->tail_cat.coff & 3) || d->tail_cat.coff + d->tail_cat_c > d->tail_cat.pitch)
            return ESR_ERR_BAD_ARG;
    }
    const bool post = d->post_wpacked != nullptr;
    if (post) {
        if (tail || d->ksize != 3 || in_nchw || d->out_layout != ESR_NHWC || d->cout <= 48 || d->cout > 64 ||
            d->post_cout <= 0 || d->post_cout > 32)
            return ESR_ERR_UNSUPPORTED;
        const bool res_is_in = d->res_mode == ESR_RES_PRE_ACT && d->cin == d->cout && d->res.ptr == d->in.ptr &&
                               d->res.pitch == d->in.pitch && d->res.coff == d->in.coff;
        if (d->res_mode != ESR_RES_NONE && !res_is_in) return ESR_ERR_UNSUPPORTED;
        if (d->split > 0 && d->split < d->cout) return ESR_ERR_UNSUPPORTED;
        const int pc4 = round_up(d->post_cout, 4);
        if (!d->post_out.ptr || (d->post_out.pitch & 3) || (d->post_out.coff & 3) || d->post_out.coff + pc4 > d->post_out.pitch)
            return ESR_ERR_BAD_ARG;
        if ((double)d->n * d->h * d->w * d->post_out.pitch >= 2147483647.0) return ESR_ERR_UNSUPPORTED;
        // the 1x1 reads the conv's output as 8-channel chunks when it has to run as a launch of its own (small shapes)
        if (d->out0.coff + round_up(d->cout, CHUNK) > d->out0.pitch) return ESR_ERR_BAD_ARG;
    }
    const int ecout = tail ? d->tail_cout : d->cout;          // channels the epilogue stores
    const int cout4 = round_up(ecout, 4);
    int split = d->split <= 0 ? cout4 : d->split;
    if (split >= ecout) split = cout4;
    if (split & 3) return ESR_ERR_BAD_ARG;
    if (d->out_layout == ESR_NCHW_SHUFFLE4) {
        if (ecout % 16) return ESR_ERR_UNSUPPORTED;
    } else if (d->out_layout == ESR_NHWC) {
        if ((d->out0.pitch & 3) || (d->out0.coff & 3) || d->out0.coff + split > d->out0.pitch) return ESR_ERR_BAD_ARG;
        if (split < cout4) {
            if (!d->out1.ptr || (d->out1.pitch & 3) || (d->out1.coff & 3) ||
                d->out1.coff + (cout4 - split) > d->out1.pitch)
                return ESR_ERR_BAD_ARG;
        }
    } else {
        return ESR_ERR_BAD_ARG;
    }
    if (d->res_mode != ESR_RES_NONE) {
        if (!d->res.ptr || (d->res.pitch & 3) || (d->res.coff & 3) || d->res.coff + cout4 > d->res.pitch)
            return ESR_ERR_BAD_ARG;
    }
    // 32-bit element offsets inside the kernel; per-image raw buffers < 2 GiB (OOB offset 0x80000000)
    {
        const double px_all = (double)d->n * d->h * d->w, px_img = (double)d->h * d->w;
        int maxpitch = in_nchw ? 4 : d->in.pitch;
        if (d->out_layout == ESR_NHWC) maxpitch = maxpitch > d->out0.pitch ? maxpitch : d->out0.pitch;
        if (d->out_layout == ESR_NHWC && d->out1.ptr) maxpitch = maxpitch > d->out1.pitch ? maxpitch : d->out1.pitch;
        if (d->res_mode != ESR_RES_NONE) maxpitch = maxpitch > d->res.pitch ? maxpitch : d->res.pitch;
        if (px_all * maxpitch >= 2147483647.0) return ESR_ERR_UNSUPPORTED;
        if (px_img * (in_nchw ? d->cin : d->in.pitch) * 4.0 >= 2147483647.0) return ESR_ERR_UNSUPPORTED;
    }

    // channel-blocked views: the bits are validated BEFORE the Winograd dispatch (which knows IN / OUT1 only: anything else would run
    // on NHWC addressing where the caller declared a blocked tensor)
    if (d->blocked8 & ~(ESR_BLOCKED_IN | ESR_BLOCKED_OUT1 | ESR_BLOCKED_OUT0 | ESR_BLOCKED_RES)) return ESR_ERR_BAD_ARG;
    if ((d->blocked8 & (ESR_BLOCKED_OUT0 | ESR_BLOCKED_RES)) && (!tail || store16)) return ESR_ERR_UNSUPPORTED;
    if (d->wino_wpacked && !store16 && esr_wino_supported(d)) return esr_conv2d_wino(d, hip_stream);

    const int nt = round_up(d->cout, 16) / 16;
    const int taps = d->ksize * d->ksize;
    ConvK k;
    k.out16 = store16 ? d->storage : 0;
    k.x = static_cast<const float*>(d->in.ptr);
    k.wp = static_cast<const float*>(d->wpacked);
    k.nchunks = cin_phys / CHUNK;
    k.bias = k.wp + (size_t)k.nchunks * taps * nt * 128;
    k.res = static_cast<const float*>(d->res.ptr);
    k.y0 = static_cast<float*>(d->out0.ptr);
    k.y1 = static_cast<float*>(d->out1.ptr);
    k.N = d->n; k.H = d->h; k.W = d->w;
    k.cin = d->cin;
    k.in_pitch = d->in.pitch; k.in_coff = d->in.coff;
    k.res_pitch = d->res.pitch; k.res_coff = d->res.coff;
    k.y0_pitch = d->out0.pitch; k.y0_coff = d->out0.coff;
    k.y1_pitch = d->out1.pitch; k.y1_coff = d->out1.coff;
    k.cout_store = cout4;
    k.split = split;
    k.act = d->act; k.slope = d->slope; k.res_mode = d->res_mode;
    k.res_in = 0;
    // channel-blocked views (ABI v6): out1 of an fp32 NHWC split store; in of the fused IMDB tail at the network's shape
    k.y1_blk = (d->blocked8 & ESR_BLOCKED_OUT1) ? 1 : 0;
    k.in_blk = (d->blocked8 & ESR_BLOCKED_IN) ? 1 : 0;
    k.y0_blk = (d->blocked8 & ESR_BLOCKED_OUT0) ? 1 : 0;
    k.res_blk = (d->blocked8 & ESR_BLOCKED_RES) ? 1 : 0;
    if (d->blocked8 & ~(ESR_BLOCKED_IN | ESR_BLOCKED_OUT1 | ESR_BLOCKED_OUT0 | ESR_BLOCKED_RES)) return ESR_ERR_BAD_ARG;
    // blocked out0 / res: only the fused IMDB tail (checked again where the tail shape is known), whole planes
    if ((k.y0_blk || k.res_blk) && (!tail || store16)) return ESR_ERR_UNSUPPORTED;
    if (k.y0_blk && ((d->out0.pitch & 7) || (d->out0.coff & 7) || (double)d->h * d->w * d->out0.pitch * 4.0 >= 2147483647.0)) return ESR_ERR_UNSUPPORTED;
    if (k.res_blk && (d->res_mode != ESR_RES_PRE_ACT || (d->res.pitch & 7) || (d->res.coff & 7))) return ESR_ERR_UNSUPPORTED;
    if (k.y1_blk) {
        if (d->out_layout != ESR_NHWC || tail || post || store16 || split >= cout4 || d->ksize != 3 || in_nchw || d->cout <= 48 || d->cout > 64 || (d->out1.pitch & 7) || (d->out1.coff & 7) || (split & 7) ||
            (double)d->h * d->w * d->out1.pitch * 4.0 >= 2147483647.0)
            return ESR_ERR_UNSUPPORTED;
    }
    if (k.in_blk && (!tail || in_nchw || (d->in.pitch & 7) || (d->in.coff & 7))) return ESR_ERR_UNSUPPORTED;
    if (!tail && d->ksize == 3 && !in_nchw && d->res_mode == ESR_RES_PRE_ACT && d->cin == d->cout &&
        d->res.ptr == d->in.ptr && d->res.pitch == d->in.pitch && d->res.coff == d->in.coff) {
        k.res_in = 1;                               // residual == input: added from the staged input tile inside the K loop
        k.res_mode = ESR_RES_NONE;
    }
    k.out_layout = d->out_layout;
    k.tiles_x = (d->w + TILE - 1) / TILE;
    k.tiles_y = (d->h + TILE - 1) / TILE;
    k.wp2 = nullptr; k.bias2 = nullptr; k.cat = nullptr; k.cat_pitch = k.cat_coff = k.cat_chunks = 0; k.mid_act = ESR_ACT_NONE;
    k.wp3 = nullptr; k.bias3 = nullptr; k.y2 = nullptr; k.y2_pitch = k.y2_coff = k.y2_cout4 = 0; k.post_act = ESR_ACT_NONE; k.post_nch8 = 0;
    if (post) {
        if (conv_block_waves(3, false, nt, k.nchunks, d->n, d->h, d->w) != 8) {
            // small launch (4-wave shape, no room for the 1x1's weights next to two blocks per CU): two launches
            esr_conv_desc a = *d;
            a.post_wpacked = nullptr;
            const int rc = esr_conv2d_f32(&a, hip_stream);
            if (rc != ESR_OK) return rc;
            esr_conv_desc b;
            memset(&b, 0, sizeof(b));
            b.n = d->n; b.h = d->h; b.w = d->w; b.cin = d->cout; b.cout = d->post_cout; b.ksize = 1;
            b.in_layout = ESR_NHWC; b.out_layout = ESR_NHWC; b.act = d->post_act; b.slope = d->slope;
            b.in = d->out0; b.out0 = d->post_out; b.wpacked = d->post_wpacked; b.compute = ESR_COMPUTE_F32;
            return esr_conv2d_f32(&b, hip_stream);
        }
        k.wp3 = static_cast<const float*>(d->post_wpacked);
        const int pnt = round_up(d->post_cout, 16) / 16;
        if (pnt != 2) return ESR_ERR_UNSUPPORTED;              // post_cout in (16, 32]
        k.bias3 = k.wp3 + (size_t)(round_up(d->cout, CHUNK) / CHUNK) * pnt * 128;
        k.y2 = static_cast<float*>(d->post_out.ptr);
        k.y2_pitch = d->post_out.pitch; k.y2_coff = d->post_out.coff; k.y2_cout4 = round_up(d->post_cout, 4);
        k.post_act = d->post_act;
        k.post_nch8 = round_up(d->cout, CHUNK) / CHUNK;
    }
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if (tail) {
        k.wp2 = static_cast<const float*>(d->tail_wpacked);
        k.cat_chunks = d->tail_cat_c / 16;
        k.bias2 = k.wp2 + (size_t)(2 * (k.cat_chunks + 1)) * 4 * 128;       // [chunk of 8][tap = 1][4 tiles][128]
        k.cat = static_cast<const float*>(d->tail_cat.ptr);
        k.cat_pitch = d->tail_cat.pitch; k.cat_coff = d->tail_cat.coff;
        k.mid_act = d->tail_mid_act;
        {
            const double px_all = (double)d->n * d->h * d->w;
            if (px_all * d->tail_cat.pitch >= 2147483647.0) return ESR_ERR_UNSUPPORTED;
        }
        if ((k.in_blk || k.y0_blk || k.res_blk) && !imdb_tail_shape(k)) return ESR_ERR_UNSUPPORTED;      // only imdb_tail_kernel knows the blocked layouts
        return launch_conv_tail(k, st);
    }
    if (in_nchw) return launch_conv_nt<3, true>(nt, k, st);
    if (d->ksize == 3) return launch_conv_nt<3, false>(nt, k, st);
    return launch_conv_nt<1, false>(nt, k, st);
}

static int run_one(const esr_op& op, void* hip_stream)
{
    switch (op.kind) {
        case ESR_OP_CONV: return esr_conv2d_f32(&op.conv, hip_stream);
        case ESR_OP_CONV3X3S2: return esr_conv3x3s2_f32(&op.esa, hip_stream);
        case ESR_OP_MAXPOOL7S3: return esr_maxpool7s3_f32(&op.esa, hip_stream);
        case ESR_OP_ESA_APPLY: return esr_esa_apply_f32(&op.esa, hip_stream);
        case ESR_OP_DWCONV: return esr_dwconv3x3_f32(&op.conv, hip_stream);
        case ESR_OP_BSCONV: return esr_bsconv_f32(&op.bs, hip_stream);
        case ESR_OP_PACK_INPUT: return esr_pack_input_s16(&op.conv, hip_stream);
        case ESR_OP_ESA_LOWRES: return esr_esa_lowres_f32(&op.lo, hip_stream);
        case ESR_OP_CONV_CHAIN: return esr_conv_chain_s16(&op.chain, hip_stream);
        default: return ESR_ERR_BAD_ARG;
    }
}

struct esr_profiler {
    int n_ops, max_passes, passes;
    hipEvent_t* ev;   // [max_passes][n_ops][2]
    char (*sym)[256]; // [n_ops]: device symbol(s) of each op's launch(es) in the last profiled pass
};

int esr_prof_create(int n_ops, int max_passes, esr_profiler** out)
{
    if (!out || n_ops <= 0 || max_passes <= 0) return ESR_ERR_BAD_ARG;
    esr_profiler* p = new esr_profiler{n_ops, max_passes, 0, nullptr, nullptr};
    p->sym = new char[n_ops][256];
    for (int i = 0; i < n_ops; ++i) p->sym[i][0] = 0;
    const size_t n = (size_t)n_ops * max_passes * 2;
    p->ev = new hipEvent_t[n];
    for (size_t i = 0; i < n; ++i) {
        const hipError_t e = hipEventCreate(&p->ev[i]);
        if (e != hipSuccess) {
            set_err("hipEventCreate", e);
            for (size_t j = 0; j < i; ++j) (void)hipEventDestroy(p->ev[j]);
            delete[] p->ev;
            delete[] p->sym;
            delete p;
            return ESR_ERR_LAUNCH;
        }
    }
    *out = p;
    return ESR_OK;
}

void esr_prof_destroy(esr_profiler* p)
{
    if (!p) return;
    const size_t n = (size_t)p->n_ops * p->max_passes * 2;
    for (size_t i = 0; i < n; ++i) (void)hipEventDestroy(p->ev[i]);
    delete[] p->ev;
    delete[] p->sym;
    delete p;
}

int esr_prof_kernel_symbol(esr_profiler* p, int op, char* buf, size_t n)
{
    if (!p || !buf || n == 0 || op < 0 || op >= p->n_ops) return ESR_ERR_BAD_ARG;
    snprintf(buf, n, "%s", p->sym[op]);
    return ESR_OK;
}

int esr_run_ops_profiled(const esr_op* ops, int n_ops, void* hip_stream, esr_profiler* p)
{
    if (!ops || !p || n_ops != p->n_ops) return ESR_ERR_BAD_ARG;
    if (p->passes >= p->max_passes) return esr_run_ops(ops, n_ops, hip_stream);   // full: run untimed
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    hipEvent_t* ev = p->ev + (size_t)p->passes * n_ops * 2;
    g_ktrace = true;
    for (int i = 0; i < n_ops; ++i) {
        g_kname[0] = 0;
        (void)hipEventRecord(ev[2 * i], st);
        const int rc = run_one(ops[i], hip_stream);
        (void)hipEventRecord(ev[2 * i + 1], st);
        memcpy(p->sym[i], g_kname, sizeof(g_kname));
        if (rc != ESR_OK) { g_ktrace = false; return rc; }
    }
    g_ktrace = false;
    p->passes++;
    return ESR_OK;
}

int esr_prof_collect(esr_profiler* p, double* ms_sum, int n_ops, int* passes)
{
    if (!p || !ms_sum || n_ops != p->n_ops) return ESR_ERR_BAD_ARG;
    for (int i = 0; i < n_ops; ++i) ms_sum[i] = 0.0;
    for (int k = 0; k < p->passes; ++k) {
        hipEvent_t* ev = p->ev + (size_t)k * n_ops * 2;
        for (int i = 0; i < n_ops; ++i) {
            float ms = 0.f;
            const hipError_t e = hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
            if (e != hipSuccess) {
                set_err("hipEventElapsedTime", e);
                return ESR_ERR_LAUNCH;
            }
            ms_sum[i] += ms;
        }
    }
    if (passes) *passes = p->passes;
    p->passes = 0;
    return ESR_OK;
}

int esr_run_ops(const esr_op* ops, int n_ops, void* hip_stream)
{
    if (!ops || n_ops < 0) return ESR_ERR_BAD_ARG;
    for (int i = 0; i < n_ops; ++i) {
        const int rc = run_one(ops[i], hip_stream);
        if (rc != ESR_OK) return rc;
    }
    return ESR_OK;
}

}  // extern "C"
