// esr_ca.hip -- channel attention: CALayer (models/basicblock.py:333-348) and the contrast-aware CCALayer
// (models/team05_efdn/plainblock.py:106-122; stdv_channels / mean_channels :85-93).  Interface: include/esr_hip.h.
//
//     CA  : y = x * sigmoid(W2 . relu(W1 . mean_hw(x) + b1) + b2)
//     CCA : y = x * sigmoid(W2 . relu(W1 . (std_hw(x) + mean_hw(x)) + b1) + b2)      std = sqrt(mean((x - mean)^2))
//
// Two launches, both HBM-bound (x is read twice, y written once):
//   ca_reduce_kernel   per (image, channel) sum and sum of squares over H x W.  Per-thread partials are fp64 (the variance is
//                      E[x^2] - mean^2: in fp32 that cancels catastrophically for |mean| >> std), reduced through LDS, one fp64
//                      atomicAdd per (block, channel) into `stats` (zeroed by the call on the same stream).
//   ca_apply_kernel    every block recomputes the gate of its image from `stats` (two tiny dense layers: c x cr + cr x c MACs),
//                      then scales its pixels.
// Layouts: NHWC views with pitch / first-channel offset (fp32 or 16-bit storage: the engine's native layout) or plain NCHW
// fp32 (a stand-alone drop-in for the reference's NCHW modules).
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include "esr_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int CA_MAXC = 64;
constexpr int CA_MAXR = 16;

template <int ST>
__device__ __forceinline__ f32x4 ca_ld4(const void* base, size_t idx)
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
__device__ __forceinline__ void ca_st4(void* base, size_t idx, f32x4 v)
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

struct CaK {
    const void* x; void* y;
    const float* w1; const float* w2;     // dense [c][cr] + b1[cr] ; dense [cr][c4] + b2[c4]
    double* stats;                        // [n][2][c4]: sum, sum of squares
    int N, HW, c, c4, cr, contrast;
    int x_pitch, x_coff, y_pitch, y_coff;
    int blocks_per_image;
};

// NHWC: thread = (pixel slot, channel quad); a block strides over its share of one image's pixels
template <int ST>
__global__ __launch_bounds__(256) void ca_reduce_nhwc_kernel(const CaK p)
{
    __shared__ double red[256 * 8];
    const int nq = p.c4 >> 2;
    const int q = threadIdx.x % nq, ps = threadIdx.x / nq, pp = 256 / nq;
    const int n = blockIdx.x / p.blocks_per_image, b = blockIdx.x % p.blocks_per_image;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    if (ps < pp)
        for (int px = b * pp + ps; px < p.HW; px += p.blocks_per_image * pp) {
            const f32x4 v = ca_ld4<ST>(p.x, ((size_t)n * p.HW + px) * p.x_pitch + p.x_coff + q * 4);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            ss[0] += (double)v.x * v.x; ss[1] += (double)v.y * v.y; ss[2] += (double)v.z * v.z; ss[3] += (double)v.w * v.w;
        }
    for (int j = 0; j < 4; ++j) {
        red[threadIdx.x * 8 + j] = s[j];
        red[threadIdx.x * 8 + 4 + j] = ss[j];
    }
    __syncthreads();
    // thread t < 2 * c4 sums component t over the pixel slots
    if (threadIdx.x < 2 * p.c4) {
        const int which = threadIdx.x / p.c4, ch = threadIdx.x % p.c4;
        double a = 0;
        for (int k = 0; k < pp; ++k) a += red[(k * nq + (ch >> 2)) * 8 + which * 4 + (ch & 3)];
        atomicAdd(p.stats + ((size_t)n * 2 + which) * p.c4 + ch, a);
    }
}

// NCHW fp32: block = (image, channel) plane slices
__global__ __launch_bounds__(256) void ca_reduce_nchw_kernel(const CaK p)
{
    __shared__ double red[512];
    const int plane = blockIdx.x / p.blocks_per_image, b = blockIdx.x % p.blocks_per_image;     // plane = n * c + ch
    const float* xp = static_cast<const float*>(p.x) + (size_t)plane * p.HW;
    double s = 0, ss = 0;
    for (int i = b * 256 + threadIdx.x; i < p.HW; i += p.blocks_per_image * 256) {
        const float v = xp[i];
        s += v; ss += (double)v * v;
    }
    red[threadIdx.x] = s; red[256 + threadIdx.x] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { red[threadIdx.x] += red[threadIdx.x + o]; red[256 + threadIdx.x] += red[256 + threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int n = plane / p.c, ch = plane % p.c;
        atomicAdd(p.stats + ((size_t)n * 2 + 0) * p.c4 + ch, red[0]);
        atomicAdd(p.stats + ((size_t)n * 2 + 1) * p.c4 + ch, red[256]);
    }
}

// gate of image n into LDS g[c4]
__device__ __forceinline__ void ca_gate(const CaK& p, int n, float* g, float* hid)
{
    const double inv = 1.0 / (double)p.HW;
    for (int ch = threadIdx.x; ch < p.c4; ch += blockDim.x) {
        float v = 0.f;
        if (ch < p.c) {
            const double mean = p.stats[((size_t)n * 2 + 0) * p.c4 + ch] * inv;
            double val = mean;
            if (p.contrast) {
                double var = p.stats[((size_t)n * 2 + 1) * p.c4 + ch] * inv - mean * mean;
                var = var < 0 ? 0 : var;
                val = sqrt(var) + mean;                  // contrast + average pooling (plainblock.py:119)
            }
            v = (float)val;
        }
        g[ch] = v;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < p.cr; r += blockDim.x) {
        float a = p.w1[p.c * p.cr + r];
        for (int ch = 0; ch < p.c; ++ch) a = fmaf(g[ch], p.w1[ch * p.cr + r], a);
        hid[r] = fmaxf(a, 0.f);
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < p.c4; ch += blockDim.x) {
        float a = 0.f;
        if (ch < p.c) {
            a = p.w2[p.cr * p.c4 + ch];
            for (int r = 0; r < p.cr; ++r) a = fmaf(hid[r], p.w2[r * p.c4 + ch], a);
            a = 1.f / (1.f + expf(-a));
        }
        g[ch] = a;
    }
    __syncthreads();
}

template <int ST>
__global__ __launch_bounds__(256) void ca_apply_nhwc_kernel(const CaK p)
{
    __shared__ __attribute__((aligned(16))) float g[CA_MAXC];
    __shared__ float hid[CA_MAXR];
    const int n = blockIdx.x / p.blocks_per_image, b = blockIdx.x % p.blocks_per_image;
    ca_gate(p, n, g, hid);
    const int nq = p.c4 >> 2;
    const int q = threadIdx.x % nq, ps = threadIdx.x / nq, pp = 256 / nq;
    if (ps >= pp) return;
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + q * 4);
    for (int px = b * pp + ps; px < p.HW; px += p.blocks_per_image * pp) {
        const size_t pix = (size_t)n * p.HW + px;
        const f32x4 v = ca_ld4<ST>(p.x, pix * p.x_pitch + p.x_coff + q * 4);
        ca_st4<ST>(p.y, pix * p.y_pitch + p.y_coff + q * 4, v * gv);
    }
}

__global__ __launch_bounds__(256) void ca_apply_nchw_kernel(const CaK p)
{
    __shared__ float g[CA_MAXC], hid[CA_MAXR];
    const int plane = blockIdx.x / p.blocks_per_image, b = blockIdx.x % p.blocks_per_image;
    const int n = plane / p.c, ch = plane % p.c;
    ca_gate(p, n, g, hid);
    const float gv = g[ch];
    const float* xp = static_cast<const float*>(p.x) + (size_t)plane * p.HW;
    float* yp = static_cast<float*>(p.y) + (size_t)plane * p.HW;
    for (int i = b * 256 + threadIdx.x; i < p.HW; i += p.blocks_per_image * 256) yp[i] = xp[i] * gv;
}

}  // namespace

extern "C" int esr_channel_attention_f32(const esr_ca_desc* d, void* hip_stream)
{
    if (!d || !d->x.ptr || !d->y.ptr || !d->w1 || !d->w2 || !d->stats) return ESR_ERR_BAD_ARG;
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->c <= 0 || d->cr <= 0) return ESR_ERR_BAD_ARG;
    if (d->c > CA_MAXC || d->cr > CA_MAXR) return ESR_ERR_UNSUPPORTED;
    const bool nchw = d->layout == ESR_NCHW_IN;
    if (!nchw && d->layout != ESR_NHWC) return ESR_ERR_BAD_ARG;
    const int c4 = esr_round_up(d->c, 4);
    if (nchw && d->storage != ESR_STORE_F32) return ESR_ERR_UNSUPPORTED;
    if (!nchw) {
        if ((d->x.pitch & 3) || (d->x.coff & 3) || d->x.coff + c4 > d->x.pitch) return ESR_ERR_BAD_ARG;
        if ((d->y.pitch & 3) || (d->y.coff & 3) || d->y.coff + c4 > d->y.pitch) return ESR_ERR_BAD_ARG;
    }
    if ((double)d->n * d->h * d->w * (nchw ? d->c : (d->x.pitch > d->y.pitch ? d->x.pitch : d->y.pitch)) >= 9.0e18) return ESR_ERR_UNSUPPORTED;
    CaK k;
    k.x = d->x.ptr; k.y = d->y.ptr;
    k.w1 = static_cast<const float*>(d->w1); k.w2 = static_cast<const float*>(d->w2);
    k.stats = static_cast<double*>(d->stats);
    k.N = d->n; k.HW = d->h * d->w; k.c = d->c; k.c4 = c4; k.cr = d->cr; k.contrast = d->contrast ? 1 : 0;
    k.x_pitch = d->x.pitch; k.x_coff = d->x.coff; k.y_pitch = d->y.pitch; k.y_coff = d->y.coff;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if (hipMemsetAsync(d->stats, 0, (size_t)d->n * 2 * c4 * sizeof(double), st) != hipSuccess) return ESR_ERR_LAUNCH;
    if (nchw) {
        const int planes = d->n * d->c;
        int bpi = (2048 + planes - 1) / planes;                           // ~2048 blocks in flight
        const int maxb = (k.HW + 255) / 256;
        bpi = bpi < 1 ? 1 : (bpi > maxb ? maxb : bpi);
        k.blocks_per_image = bpi;
        hipLaunchKernelGGL(ca_reduce_nchw_kernel, dim3(planes * bpi), dim3(256), 0, st, k);
        hipLaunchKernelGGL(ca_apply_nchw_kernel, dim3(planes * bpi), dim3(256), 0, st, k);
        return esr_check_launch("ca_nchw kernels launch");
    }
    const int pp = 256 / (c4 / 4);
    int bpi = (2048 + d->n - 1) / d->n;
    const int maxb = (k.HW + pp - 1) / pp;
    bpi = bpi < 1 ? 1 : (bpi > maxb ? maxb : bpi);
    k.blocks_per_image = bpi;
    const dim3 grid(d->n * bpi);
    switch (d->storage) {
        case ESR_STORE_F32:
            hipLaunchKernelGGL(ca_reduce_nhwc_kernel<ESR_STORE_F32>, grid, dim3(256), 0, st, k);
            hipLaunchKernelGGL(ca_apply_nhwc_kernel<ESR_STORE_F32>, grid, dim3(256), 0, st, k);
            break;
        case ESR_STORE_BF16:
            hipLaunchKernelGGL(ca_reduce_nhwc_kernel<ESR_STORE_BF16>, grid, dim3(256), 0, st, k);
            hipLaunchKernelGGL(ca_apply_nhwc_kernel<ESR_STORE_BF16>, grid, dim3(256), 0, st, k);
            break;
        case ESR_STORE_F16:
            hipLaunchKernelGGL(ca_reduce_nhwc_kernel<ESR_STORE_F16>, grid, dim3(256), 0, st, k);
            hipLaunchKernelGGL(ca_apply_nhwc_kernel<ESR_STORE_F16>, grid, dim3(256), 0, st, k);
            break;
        default: return ESR_ERR_BAD_ARG;
    }
    return esr_check_launch("ca_nhwc kernels launch");
}
