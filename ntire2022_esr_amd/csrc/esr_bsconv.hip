// esr_bsconv.hip -- BSConvU (models/team18_bsrn.py:44-88) as ONE kernel: pointwise 1x1 (nn.Linear on NHWC) -> depthwise 3x3
// (zero padding on the pointwise OUTPUT, bias) -> + residual -> activation, optionally with the block's distillation 1x1
// (team18_bsrn.py:135-148: c{j}_d reads the same input as c{j}_r) riding along.  Separate kernels move the pointwise
// result through HBM twice (write, then read by the depthwise conv) and read the input twice more; here the input tile
// (16x16 pixels + halo) is read once, the pointwise result lives in LDS only.
//   phase 1  pointwise GEMM on MFMA over the 18x18 halo pixel list (21 pixel tiles of 16): B fragments straight from global
//            memory (lane (px, kq): the 16 bytes of channels 16C+4kq.. of its pixel, hardware zero fill outside the image),
//            weights resident in LDS ([16-channel chunk][tile][lane][4]); the D fragment (4 channels of one pixel per lane)
//            goes to the LDS tile [halo pixel][channel], zeroed where the pixel lies outside the image (the depthwise conv
//            pads the pointwise output with zeros, not with its bias); distillation outputs go to global memory directly.
//   phase 2  depthwise 3x3 on VALU: thread = (pixel, 4 channels), 9 float4 LDS reads, bias, residual, activation, one
//            16-byte store; 192 contiguous bytes per pixel for 48 channels.
// Memory-bound by design: per tile it reads 1.27x the input (halo) + the residual and writes the output once.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include "esr_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// 16-bit storage (ST = ESR_STORE_BF16 / ESR_STORE_F16, BASELINE.json config [4]): the input / residual / outputs are
// 16-bit in HBM (half the bytes of this memory-bound kernel); the pointwise GEMM runs on v_mfma_f32_16x16x32_{bf16,f16}
// with the B fragment loaded straight from global memory as it is stored (16 bytes = 8 channels per lane) and the 1x1
// weights as hi + lo 16-bit pairs (esr_pack_conv_s16 blobs: the blob IS the LDS image); the pointwise result stays fp32
// in LDS, the depthwise conv / residual / GELU are fp32, results are rounded once when stored.

namespace {

constexpr int BT = 16;                 // output tile edge
constexpr int BH = BT + 2;             // halo edge
constexpr int BNPX = BH * BH;          // 324 halo pixels
constexpr int BNPT = (BNPX + 15) / 16; // 21 pixel tiles of 16
constexpr int BPTW = (BNPT + 3) / 4;   // pixel tiles per wave (6)
constexpr int BMAXC16 = 4;             // cin <= 64
constexpr int BMAXC32 = 2;             // 16-bit storage: K slots of 32 channels
constexpr unsigned BOOB = 0x80000000u;

struct BsK {
    const void* x; const void* res; void* y; void* dy;
    const float* pw; const float* pwb; const float* dwp; const float* dpw; const float* dpb;
    int N, H, W;
    int nch8;             // 8-channel chunks of the packed 1x1 blobs
    int cp;               // depthwise channels rounded up to 4
    int d_cout4;
    int x_pitch, x_coff, r_pitch, r_coff, y_pitch, y_coff, dy_pitch, dy_coff;
    int act, res_mode, d_act;
    float slope;
    int tiles_x, tiles_y;
};

// GELU for the 16-bit storage modes: x * Phi(x) with Phi(x) - 0.5 = x * P(x^2), P a degree-7 minimax polynomial on |x| <= 4
// (|error| of Phi <= 2.1e-5, tools/fit_gelu.py), the argument clamped to [-4, 4] and the factor x to [-4, inf): |gelu error| <=
// 1.3e-4 for x <= 4 and 5.3e-5 x beyond, about one fp16 step of the values that matter, far below a bf16 step -- and 11 plain VALU instructions
// (packable two values at a time) instead of libm erff's ~40 or the 16 + v_rcp + v_exp of an erf approximation.  The fp32
// path keeps erff.
__device__ __forceinline__ float gelu16(float x)
{
    const float xc = fminf(fmaxf(x, -4.f), 4.f);
    const float t = xc * xc;
    float p = -1.580786198e-09f;
    p = fmaf(p, t, 1.217111051e-07f);
    p = fmaf(p, t, -4.100866386e-06f);
    p = fmaf(p, t, 8.066739505e-05f);
    p = fmaf(p, t, -1.048204400e-03f);
    p = fmaf(p, t, 9.664874174e-03f);
    p = fmaf(p, t, -6.617537882e-02f);
    p = fmaf(p, t, 3.988475079e-01f);
    return fmaxf(x, -4.f) * fmaf(xc, p, 0.5f);
}

template <int ST>
__device__ __forceinline__ float bs_act(float v, int act, float slope)
{
    if (act == ESR_ACT_GELU) {
        if (ST == ESR_STORE_F32) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        return gelu16(v);
    }
    if (act == ESR_ACT_LRELU) return fmaxf(v, v * slope);
    if (act == ESR_ACT_RELU) return fmaxf(v, 0.f);
    return v;
}

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

// esr_pack_conv_f32(ksize = 1) blob ([chunk of 8][tile][kq'*16+i][j'], channel = 8*chunk + 2kq' + j') -> LDS image
// [chunk of 16][tile][lane = kq*16+i][j], channel = 16*chunk + 4kq + j; an odd chunk count leaves the upper half zero
template <int NTL>
__device__ __forceinline__ void build_image(float* img, const float* blob, int nch8, int tid)
{
    const int nc16 = (nch8 + 1) >> 1;
    if (nch8 & 1)
        for (int e = tid; e < NTL * 128; e += 256) {
            const int tt = e >> 7, l = 32 + ((e >> 2) & 31), j = e & 3;
            img[(((nc16 - 1) * NTL + tt) * 64 + l) * 4 + j] = 0.f;
        }
    for (int q = tid; q < nch8 * NTL * 32; q += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(blob + (size_t)q * 4);
        const int ip = q & 7, kq8 = (q >> 3) & 3, ct = q >> 5;
        const int tt = ct % NTL, chunk8 = ct / NTL;
        const int ch16 = 8 * (chunk8 & 1) + 2 * kq8;
        float* dst = img + (((chunk8 >> 1) * NTL + tt) * 64 + (ch16 >> 2) * 16 + 2 * ip) * 4 + (ch16 & 3);
        dst[0] = v.x; dst[1] = v.y; dst[4] = v.z; dst[5] = v.w;
    }
}

template <int ST>
__device__ __forceinline__ f32x4 mfma16(f32x4 a, f32x4 b, f32x4 c)
{
    if (ST == ESR_STORE_BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int NTP, int NTD, int ST>
__global__ __launch_bounds__(256, 2) void bsconv_kernel(const BsK p)
{
    constexpr bool S16 = ST != ESR_STORE_F32;
    constexpr int ES = S16 ? 2 : 4;                          // bytes per stored element
    extern __shared__ __attribute__((aligned(16))) float bsm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int px = lane & 15, kq = lane >> 4;
    const int nc16 = (p.nch8 + 1) >> 1;
    const int nc32 = (nc16 + 1) >> 1;
    const int wimg = S16 ? nc32 * 512 : nc16 * 256;          // floats of weight image per output tile (16-bit: hi + lo images)
    // pointwise result [BNPX] x tlp bytes: fp32, or fp16 when the storage is fp16 (TL16: half the LDS -- two blocks per CU --
    // and half the depthwise conv's LDS traffic; 11 mantissa bits, the network's own storage precision).  The pixel pitch
    // carries 16 pad bytes: at 256 (128) bytes the 16 pixels of a D fragment hit the same banks.
    constexpr bool TL16 = ST == ESR_STORE_F16;
    const int tlp = p.cp * (TL16 ? 2 : 4) + 16;
    char* const tl = reinterpret_cast<char*>(bsm);
    float* const ipw = bsm + (BNPX * tlp) / 4;               // pointwise weight image
    float* const idp = ipw + wimg * NTP;                     // distillation weight image
    float* const sdw = idp + wimg * NTD;                     // depthwise [tap][cp] + bias[cp]
    float* const sb = sdw + 10 * p.cp;                       // pointwise bias [NTP*16], distillation bias [NTD*16]
    if (S16) {
        // esr_pack_conv_s16(ksize = 1) blobs are [16-channel chunk][tile][lane = (hi|lo, half, i)][8 x 16 bit].  The LDS images
        // are [32-channel chunk][tile][lane = kq*16+i][8 x 16 bit] = channels 32C + 8kq .. +7 of output channel i, a hi
        // image and a lo image: lane (px, kq) of the B operand then holds 16 DISTINCT bytes and the four lanes of a pixel read
        // 64 contiguous bytes (with the blob's own map lanes kq >= 2 re-read the bytes of kq < 2 for the lo weights).
        auto build16 = [&](float* img, const float* blob, int ntl) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            for (int u = tid; u < nc32 * ntl * 128; u += 256) {           // 16-byte units: [C][tt][hi|lo][kq][i]
                const int i = u & 15, kq2 = (u >> 4) & 3, lo = (u >> 6) & 1, ct = u >> 7;
                const int tt = ct % ntl, C = ct / ntl;
                const int c16 = 2 * C + (kq2 >> 1);
                const f32x4 v = c16 < nc16 ? *reinterpret_cast<const f32x4*>(blob + ((size_t)(c16 * ntl + tt) * 64 + lo * 32 + (kq2 & 1) * 16 + i) * 4) : zero;
                *reinterpret_cast<f32x4*>(img + ((size_t)((C * ntl + tt) * 2 + lo) * 64 + kq2 * 16 + i) * 4) = v;
            }
        };
        build16(ipw, p.pw, NTP);
        if (NTD) build16(idp, p.dpw, NTD);
    } else {
        build_image<NTP>(ipw, p.pw, p.nch8, tid);
        if (NTD) build_image<(NTD ? NTD : 1)>(idp, p.dpw, p.nch8, tid);
    }
    for (int i = tid; i < 10 * p.cp; i += 256) sdw[i] = p.dwp[i];
    if (tid < NTP * 16) sb[tid] = p.pwb[tid];
    if (NTD && tid < NTD * 16) sb[NTP * 16 + tid] = p.dpb[tid];
    __syncthreads();

    const int ntiles = p.N * p.tiles_y * p.tiles_x;
    const size_t img_elems = (size_t)p.H * p.W * p.x_pitch;
    const int cin_phys = p.nch8 * 8;
    const int nq = p.cp >> 2;

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = t % p.tiles_x, tq = t / p.tiles_x;
        const int ty = tq % p.tiles_y, n = tq / p.tiles_y;
        const int x0 = tx * BT, y0 = ty * BT;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(static_cast<const char*>(p.x) + (size_t)n * img_elems * ES), 0, (int)(img_elems * ES), 0x00020000);

        // ---- phase 1: all B fragments of this wave's pixel tiles are requested before the first MFMA ----------------
        f32x4 b[BPTW][S16 ? BMAXC32 : BMAXC16];
        bool valid[BPTW];
        int gpix[BPTW];
#pragma unroll
        for (int s = 0; s < BPTW; ++s) {
            const int pl = (wv + 4 * s) * 16 + px;
            const int ly = pl / BH, lx = pl - ly * BH;
            const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
            valid[s] = wv + 4 * s < BNPT && pl < BNPX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            gpix[s] = gy * p.W + gx;
            // fp32: lane (px, kq) holds channels 16C + 4kq .. +3 (k slot of the 16x16x4 MFMAs); 16-bit: channels
            // 32C + 8kq .. +7 (16 bytes; the four lanes of a pixel read 64 contiguous bytes)
            const int chl = S16 ? 8 * kq : 4 * kq;
            const unsigned vo = valid[s] ? (unsigned)(gpix[s] * p.x_pitch + p.x_coff + chl) * (unsigned)ES : BOOB;
#pragma unroll
            for (int C = 0; C < (S16 ? BMAXC32 : BMAXC16); ++C) {
                const bool ok = S16 ? (C < nc32 && 32 * C + chl < cin_phys) : (C < nc16 && 16 * C + chl < cin_phys);
                b[s][C] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? vo : BOOB, C * (S16 ? 64 : 64), 0));
            }
        }
#pragma unroll
        for (int s = 0; s < BPTW; ++s) {
            if (wv + 4 * s >= BNPT) break;
            f32x4 acc[NTP], dacc[NTD ? NTD : 1];
#pragma unroll
            for (int tt = 0; tt < NTP; ++tt) acc[tt] = *reinterpret_cast<const f32x4*>(sb + tt * 16 + kq * 4);
#pragma unroll
            for (int td = 0; td < NTD; ++td) dacc[td] = *reinterpret_cast<const f32x4*>(sb + NTP * 16 + td * 16 + kq * 4);
#pragma unroll
            for (int C = 0; C < (S16 ? BMAXC32 : BMAXC16); ++C) {
                if (C >= (S16 ? nc32 : nc16)) break;
#pragma unroll
                for (int tt = 0; tt < NTP; ++tt) {
                    if (S16) {
                        const float* img = ipw + ((C * NTP + tt) * 128 + lane) * 4;
                        acc[tt] = mfma16<ST>(*reinterpret_cast<const f32x4*>(img), b[s][C], acc[tt]);
                        acc[tt] = mfma16<ST>(*reinterpret_cast<const f32x4*>(img + 256), b[s][C], acc[tt]);       // lo weights
                    } else {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(ipw + ((C * NTP + tt) * 64 + lane) * 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[s][C][j], acc[tt], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int td = 0; td < NTD; ++td) {
                    if (S16) {
                        const float* img = idp + ((C * NTD + td) * 128 + lane) * 4;
                        dacc[td] = mfma16<ST>(*reinterpret_cast<const f32x4*>(img), b[s][C], dacc[td]);
                        dacc[td] = mfma16<ST>(*reinterpret_cast<const f32x4*>(img + 256), b[s][C], dacc[td]);
                    } else {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(idp + ((C * NTD + td) * 64 + lane) * 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) dacc[td] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[s][C][j], dacc[td], 0, 0, 0);
                    }
                }
            }
            const int pl = (wv + 4 * s) * 16 + px;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            if (pl < BNPX) {
#pragma unroll
                for (int tt = 0; tt < NTP; ++tt)
                    if (tt * 16 + kq * 4 < p.cp) {
                        const f32x4 v = valid[s] ? acc[tt] : zero;
                        if (TL16) {
                            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                            h2 lo2, hi2;
                            lo2[0] = (_Float16)v.x; lo2[1] = (_Float16)v.y; hi2[0] = (_Float16)v.z; hi2[1] = (_Float16)v.w;
                            *reinterpret_cast<uint2*>(tl + pl * tlp + (tt * 16 + kq * 4) * 2) = uint2{__builtin_bit_cast(unsigned, lo2), __builtin_bit_cast(unsigned, hi2)};
                        } else {
                            *reinterpret_cast<f32x4*>(tl + pl * tlp + (tt * 16 + kq * 4) * 4) = v;
                        }
                    }
            }
            if (NTD) {
                const int ly = pl / BH, lx = pl - ly * BH;
                if (valid[s] && ly >= 1 && ly <= BT && lx >= 1 && lx <= BT) {
                    const size_t dsti = ((size_t)n * p.H * p.W + gpix[s]) * p.dy_pitch + p.dy_coff;
#pragma unroll
                    for (int td = 0; td < NTD; ++td) {
                        if (td * 16 + kq * 4 >= p.d_cout4) continue;
                        f32x4 v = dacc[td];
                        v.x = bs_act<ST>(v.x, p.d_act, p.slope); v.y = bs_act<ST>(v.y, p.d_act, p.slope);
                        v.z = bs_act<ST>(v.z, p.d_act, p.slope); v.w = bs_act<ST>(v.w, p.d_act, p.slope);
                        st4<ST>(p.dy, dsti + td * 16 + kq * 4, v);
                    }
                }
            }
        }
        __syncthreads();

        // ---- phase 2: depthwise 3x3 + bias (+res) (+act) ---------------------------------------------------------------
        // thread = (channel quad q, pixel slot): its 9 weight vectors stay in registers, all residual loads of its pixels
        // are requested before the first one is used
        {
            const int pp = 256 / nq;                              // pixels per pass
            const int q = tid % nq, ps = tid / nq;
            const bool worker = ps < pp;
            f32x4 wreg[9], bias4;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) wreg[tp] = *reinterpret_cast<const f32x4*>(sdw + tp * p.cp + q * 4);
            bias4 = *reinterpret_cast<const f32x4*>(sdw + 9 * p.cp + q * 4);
            const int npass = (BT * BT + pp - 1) / pp;
            auto res_of = [&](int i) {
                const int pixel = ps + pp * i;
                const int gy = y0 + (pixel >> 4), gx = x0 + (pixel & 15);
                f32x4 r = {0.f, 0.f, 0.f, 0.f};
                if (p.res_mode != ESR_RES_NONE && worker && pixel < BT * BT && gy < p.H && gx < p.W)
                    r = ld4<ST>(p.res, ((size_t)n * p.H * p.W + (size_t)gy * p.W + gx) * p.r_pitch + p.r_coff + q * 4);
                return r;
            };
            f32x4 rv0 = res_of(0), rv1 = res_of(1);               // two passes ahead
#pragma unroll 1
            for (int i = 0; i < npass; ++i) {
                const f32x4 rv = rv0;
                rv0 = rv1;
                rv1 = res_of(i + 2);
                const int pixel = ps + pp * i;
                const int oy = pixel >> 4, ox = pixel & 15;
                const int gy = y0 + oy, gx = x0 + ox;
                if (!(worker && pixel < BT * BT && gy < p.H && gx < p.W)) continue;
                f32x4 a = bias4;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const char* src = tl + ((oy + ky) * BH + ox + kx) * tlp;
                        const f32x4 w = wreg[ky * 3 + kx];
                        if (TL16) {
                            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                            const h4 h = *reinterpret_cast<const h4*>(src + q * 8);
                            a.x = fmaf((float)h[0], w.x, a.x); a.y = fmaf((float)h[1], w.y, a.y);       // v_fma_mix_f32: the conversion is free
                            a.z = fmaf((float)h[2], w.z, a.z); a.w = fmaf((float)h[3], w.w, a.w);
                        } else {
                            a += *reinterpret_cast<const f32x4*>(src + q * 16) * w;
                        }
                    }
                if (p.res_mode == ESR_RES_PRE_ACT) a += rv;
                a.x = bs_act<ST>(a.x, p.act, p.slope); a.y = bs_act<ST>(a.y, p.act, p.slope);
                a.z = bs_act<ST>(a.z, p.act, p.slope); a.w = bs_act<ST>(a.w, p.act, p.slope);
                if (p.res_mode == ESR_RES_POST_ACT) a += rv;
                st4<ST>(p.y, ((size_t)n * p.H * p.W + (size_t)gy * p.W + gx) * p.y_pitch + p.y_coff + q * 4, a);
            }
        }
        __syncthreads();
    }
}

template <int NTP, int NTD, int ST>
int launch_bs(const BsK& k, size_t lds, hipStream_t st)
{
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bsconv_kernel<NTP, NTD, ST>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    const int ntiles = k.N * k.tiles_x * k.tiles_y;
    const int grid = ntiles < 512 ? ntiles : 512;
    esr_note_kernel("bsconv_kernel<%d, %d, %d>", NTP, NTD, ST);
    hipLaunchKernelGGL((bsconv_kernel<NTP, NTD, ST>), dim3(grid), dim3(256), lds, st, k);
    return esr_check_launch("bsconv_kernel launch");
}

template <int NTP, int ST>
int launch_bs_d(int ntd, const BsK& k, size_t lds, hipStream_t st)
{
    switch (ntd) {
        case 0: return launch_bs<NTP, 0, ST>(k, lds, st);
        case 1: return launch_bs<NTP, 1, ST>(k, lds, st);
        case 2: return launch_bs<NTP, 2, ST>(k, lds, st);
    }
    return ESR_ERR_UNSUPPORTED;
}

template <int ST>
int launch_bs_p(int ntp, int ntd, const BsK& k, size_t lds, hipStream_t st)
{
    switch (ntp) {
        case 1: return launch_bs_d<1, ST>(ntd, k, lds, st);
        case 2: return launch_bs_d<2, ST>(ntd, k, lds, st);
        case 3: return launch_bs_d<3, ST>(ntd, k, lds, st);
        case 4: return launch_bs_d<4, ST>(ntd, k, lds, st);
    }
    return ESR_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int esr_bsconv_f32(const esr_bsconv_desc* d, void* hip_stream)
{
    if (!d || !d->in.ptr || !d->out.ptr || !d->pw_packed || !d->dw_packed) return ESR_ERR_BAD_ARG;
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cin <= 0 || d->c <= 0) return ESR_ERR_BAD_ARG;
    if (d->cin > 64 || d->c > 64) return ESR_ERR_UNSUPPORTED;
    const bool s16 = d->storage == ESR_STORE_BF16 || d->storage == ESR_STORE_F16;
    if (d->storage != ESR_STORE_F32 && !s16) return ESR_ERR_BAD_ARG;
    const int cin_phys = esr_round_up(d->cin, 8), cp = esr_round_up(d->c, 4);
    if ((d->in.pitch & 3) || (d->in.coff & 3) || d->in.coff + cin_phys > d->in.pitch) return ESR_ERR_BAD_ARG;
    if (s16 && ((d->in.pitch & 7) || (d->in.coff & 7))) return ESR_ERR_BAD_ARG;          // 16-byte B fragments
    if ((d->out.pitch & 3) || (d->out.coff & 3) || d->out.coff + cp > d->out.pitch) return ESR_ERR_BAD_ARG;
    if (d->res_mode != ESR_RES_NONE && (!d->res.ptr || (d->res.pitch & 3) || (d->res.coff & 3) || d->res.coff + cp > d->res.pitch))
        return ESR_ERR_BAD_ARG;
    const int ntp = esr_round_up(d->c, 16) / 16;
    int ntd = 0, dc4 = 0;
    if (d->d_packed) {
        if (d->d_cout <= 0 || d->d_cout > 32) return ESR_ERR_UNSUPPORTED;
        dc4 = esr_round_up(d->d_cout, 4);
        ntd = esr_round_up(d->d_cout, 16) / 16;
        if (!d->d_out.ptr || (d->d_out.pitch & 3) || (d->d_out.coff & 3) || d->d_out.coff + dc4 > d->d_out.pitch) return ESR_ERR_BAD_ARG;
    }
    {
        const double px_all = (double)d->n * d->h * d->w;
        int maxpitch = d->in.pitch > d->out.pitch ? d->in.pitch : d->out.pitch;
        if (d->res_mode != ESR_RES_NONE && d->res.pitch > maxpitch) maxpitch = d->res.pitch;
        if (d->d_packed && d->d_out.pitch > maxpitch) maxpitch = d->d_out.pitch;
        if (px_all * maxpitch >= 2147483647.0) return ESR_ERR_UNSUPPORTED;
        if ((double)d->h * d->w * d->in.pitch * 4.0 >= 2147483647.0) return ESR_ERR_UNSUPPORTED;
    }
    BsK k;
    k.x = d->in.ptr; k.res = d->res.ptr;
    k.y = d->out.ptr; k.dy = d->d_out.ptr;
    k.nch8 = cin_phys / 8;
    const int nc16w = (k.nch8 + 1) / 2;
    k.pw = static_cast<const float*>(d->pw_packed);
    // fp32: esr_pack_conv_f32 layout (chunks of 8: weights, then bias[nt*16]); 16-bit: esr_pack_conv_s16 (chunks of 16, 1 KB per
    // chunk and tile = 256 floats, then the fp32 bias)
    k.pwb = s16 ? k.pw + (size_t)nc16w * ntp * 256 : k.pw + (size_t)k.nch8 * ntp * 128;
    k.dwp = static_cast<const float*>(d->dw_packed);
    k.dpw = static_cast<const float*>(d->d_packed);
    k.dpb = k.dpw ? (s16 ? k.dpw + (size_t)nc16w * ntd * 256 : k.dpw + (size_t)k.nch8 * ntd * 128) : nullptr;
    k.N = d->n; k.H = d->h; k.W = d->w; k.cp = cp; k.d_cout4 = dc4;
    k.x_pitch = d->in.pitch; k.x_coff = d->in.coff; k.r_pitch = d->res.pitch; k.r_coff = d->res.coff;
    k.y_pitch = d->out.pitch; k.y_coff = d->out.coff; k.dy_pitch = d->d_out.pitch; k.dy_coff = d->d_out.coff;
    k.act = d->act; k.res_mode = d->res_mode; k.d_act = d->d_act; k.slope = d->slope;
    k.tiles_x = (d->w + BT - 1) / BT; k.tiles_y = (d->h + BT - 1) / BT;
    const int nc16 = (k.nch8 + 1) / 2;
    const int tlp = cp * (d->storage == ESR_STORE_F16 ? 2 : 4) + 16;
    const size_t wimg = s16 ? (size_t)((nc16 + 1) / 2) * 512 : (size_t)nc16 * 256;
    const size_t lds = (size_t)BNPX * tlp + (wimg * (ntp + ntd) + 10 * cp + (ntp + ntd) * 16) * sizeof(float);
    if (lds > 160 * 1024) return ESR_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    switch (d->storage) {
        case ESR_STORE_F32: return launch_bs_p<ESR_STORE_F32>(ntp, ntd, k, lds, st);
        case ESR_STORE_BF16: return launch_bs_p<ESR_STORE_BF16>(ntp, ntd, k, lds, st);
        case ESR_STORE_F16: return launch_bs_p<ESR_STORE_F16>(ntp, ntd, k, lds, st);
    }
    return ESR_ERR_BAD_ARG;
}
