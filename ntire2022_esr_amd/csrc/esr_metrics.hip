// esr_metrics.hip -- run()'s per-image metrics on the device: SSIM and the checked tensor2uint.
// Interface: include/esr_hip.h (ABI v9).  Reference: utils/utils_image.py:509-554 (calculate_ssim / ssim), :204-208 (tensor2uint).
//
// calculate_ssim crops `border` pixels, then ssim() filters img1, img2, img1^2, img2^2, img1*img2 (float64) with the 11x11 Gaussian
// window outer(k, k), k = cv2.getGaussianKernel(11, 1.5), keeps the 'valid' region [5:-5, 5:-5] and averages
//     ((2 mu1 mu2 + C1)(2 s12 + C2)) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)),   C1 = (0.01*255)^2, C2 = (0.03*255)^2
// over it.  For an HxWx3 input the reference runs ssim() on the WHOLE array three times (its loop index is unused, :521-527):
// cv2.filter2D filters every channel, the crop is spatial, so the result is the mean of the map over all three channels.
// This kernel does exactly that: separable 11-tap passes (vertical, then horizontal: the order of ntire2022_esr_amd/image_util._ssim)
// in float64 on the uint8 images already on the device, one partial sum per block; the host adds the partials (a fixed-order float64
// sum) and divides -- one scalar leaves the GPU.  Parity: UNPINNED against the reference (its SSIM needs cv2, absent here); pinned to
// image_util.calculate_ssim (<= 1e-9, tests/test_gpu_harness.py).
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "esr_hip.h"
#include "esr_internal.h"

namespace {

constexpr int SS_TY = 16, SS_TX = 32;                 // output tile of a block (one channel)
constexpr int SS_R = 5;                               // window radius
constexpr int SS_PY = SS_TY + 2 * SS_R, SS_PX = SS_TX + 2 * SS_R;

struct SsimK {
    const uint8_t* a; const uint8_t* b;
    int W, C, border;
    int vh, vw;                                       // valid output region (per channel)
    int tiles_x, tiles_y;
    double k[11];
    double* partials;
};

__global__ __launch_bounds__(256) void ssim_u8_kernel(const SsimK p)
{
    __shared__ uint8_t pa[SS_PY][SS_PX], pb[SS_PY][SS_PX];
    __shared__ double v[5][SS_TY][SS_PX];            // vertical pass: a, b, a^2, b^2, a b
    __shared__ double red[4];
    const int c = blockIdx.z;
    const int oy0 = blockIdx.y * SS_TY, ox0 = blockIdx.x * SS_TX;       // tile origin in the valid-output frame
    const int tid = threadIdx.x;
    // patch rows oy0 .. oy0 + TY + 9, columns ox0 .. ox0 + TX + 9 of the border-cropped image (clamped: only feeds masked outputs)
    const int ch = p.vh + 2 * SS_R, cw = p.vw + 2 * SS_R;               // cropped image size
    for (int e = tid; e < SS_PY * SS_PX; e += 256) {
        const int py = e / SS_PX, px = e - py * SS_PX;
        const int y = min(oy0 + py, ch - 1) + p.border, x = min(ox0 + px, cw - 1) + p.border;
        const size_t idx = ((size_t)y * p.W + x) * p.C + c;
        pa[py][px] = p.a[idx];
        pb[py][px] = p.b[idx];
    }
    __syncthreads();
    for (int e = tid; e < SS_TY * SS_PX; e += 256) {
        const int ty = e / SS_PX, px = e - ty * SS_PX;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
#pragma unroll
        for (int t = 0; t < 11; ++t) {
            const double x = (double)pa[ty + t][px], y = (double)pb[ty + t][px], w = p.k[t];
            s0 += w * x; s1 += w * y; s2 += w * (x * x); s3 += w * (y * y); s4 += w * (x * y);
        }
        v[0][ty][px] = s0; v[1][ty][px] = s1; v[2][ty][px] = s2; v[3][ty][px] = s3; v[4][ty][px] = s4;
    }
    __syncthreads();
    const double C1 = (0.01 * 255) * (0.01 * 255), C2 = (0.03 * 255) * (0.03 * 255);
    double acc = 0;
    for (int e = tid; e < SS_TY * SS_TX; e += 256) {
        const int ty = e / SS_TX, tx = e - ty * SS_TX;
        if (oy0 + ty >= p.vh || ox0 + tx >= p.vw) continue;
        double m[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 11; ++t) {
            const double w = p.k[t];
#pragma unroll
            for (int q = 0; q < 5; ++q) m[q] += w * v[q][ty][tx + t];
        }
        const double mu1 = m[0], mu2 = m[1];
        const double mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const double s1 = m[2] - mu1_sq, s2 = m[3] - mu2_sq, s12 = m[4] - mu12;
        acc += ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2));
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) p.partials[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void tensor2uint_chk_kernel(const float* __restrict__ x, uint8_t* __restrict__ y, int C, int H,
                                                              int W, float dr, float scale, int* __restrict__ nonfinite)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;        // output element index (HWC)
    const long long n = (long long)H * W * C;
    bool bad = false;
    if (i < n) {
        const int c = (int)(i % C);
        const long long hw = i / C;
        float v = x[(size_t)c * H * W + hw];
        bad = !(fabsf(v) <= 3.402823466e38f);                             // Inf or NaN
        v = v < 0.f ? 0.f : (v > dr ? dr : v);
        y[i] = (uint8_t)__float2int_rn(__fmul_rn(v, scale));
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(nonfinite, 1);
}

}  // namespace

extern "C" {

static bool ssim_geometry(int h, int w, int c, int border, int* vh, int* vw)
{
    if (h <= 0 || w <= 0 || (c != 1 && c != 3) || border < 0) return false;
    *vh = h - 2 * border - 2 * SS_R;
    *vw = w - 2 * border - 2 * SS_R;
    return *vh > 0 && *vw > 0;
}

size_t esr_ssim_partials(int h, int w, int c, int border)
{
    int vh, vw;
    if (!ssim_geometry(h, w, c, border, &vh, &vw)) return 0;
    return (size_t)((vh + SS_TY - 1) / SS_TY) * ((vw + SS_TX - 1) / SS_TX) * c;
}

int esr_ssim_u8(const uint8_t* a, const uint8_t* b, int h, int w, int c, int border, double* partials, size_t n_partials, void* hip_stream)
{
    int vh, vw;
    if (!a || !b || !partials || !ssim_geometry(h, w, c, border, &vh, &vw)) return ESR_ERR_BAD_ARG;
    if (n_partials < esr_ssim_partials(h, w, c, border)) return ESR_ERR_BAD_ARG;
    SsimK k;
    k.a = a; k.b = b; k.W = w; k.C = c; k.border = border; k.vh = vh; k.vw = vw;
    k.tiles_y = (vh + SS_TY - 1) / SS_TY; k.tiles_x = (vw + SS_TX - 1) / SS_TX;
    if (k.tiles_y > 65535) return ESR_ERR_UNSUPPORTED;
    double s = 0;                                       // cv2.getGaussianKernel(11, 1.5)
    for (int i = 0; i < 11; ++i) { const double x = i - 5.0; k.k[i] = exp(-(x * x) / (2.0 * 1.5 * 1.5)); s += k.k[i]; }
    for (int i = 0; i < 11; ++i) k.k[i] /= s;
    k.partials = partials;
    hipLaunchKernelGGL(ssim_u8_kernel, dim3(k.tiles_x, k.tiles_y, c), dim3(256), 0, static_cast<hipStream_t>(hip_stream), k);
    return esr_check_launch("ssim_u8_kernel launch");
}

int esr_tensor2uint_u8_chk(const float* x, uint8_t* y, int c, int h, int w, float data_range, int* nonfinite, void* hip_stream)
{
    if (!x || !y || !nonfinite || c <= 0 || h <= 0 || w <= 0 || !(data_range > 0.f)) return ESR_ERR_BAD_ARG;
    const long long n = (long long)c * h * w;
    hipLaunchKernelGGL(tensor2uint_chk_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                       x, y, c, h, w, data_range, 255.0f / data_range, nonfinite);
    return esr_check_launch("tensor2uint_chk_kernel launch");
}

}  // extern "C"


// ---- measurement helpers of bench.py (ABI v11) ---------------------------------------------------------------------------------------------
// esr_bw_probe: what a plain streaming kernel reaches on THIS device at a given working set -- one launch copies `bytes` from the first half
// of `buf` to the second half `reps` times, timed by one event pair.  bench.py divides the B = 1 kernels' algorithmic bytes by this (a
// 16.6 MB tensor that the previous launch wrote sits in the 256 MB Infinity Cache: 8 TB/s of HBM is not the roof) and reports the 2 x 1 GiB
// rate as the practical HBM roof.  Round 6: the round-5 probe was one 16-byte load per lane and iteration (4.98 TB/s at
// 2 x 1 GiB, where MI355X_MICROARCH.md measures 6.29 with a float4 copy); the probe now times a small family -- U independent 16-byte loads in
// flight per lane before the first store (U = 1, 4, 8), plain or nontemporal stores, 8 / 16 / 32 blocks per CU -- and reports the BEST: a roof
// must not depend on one kernel's shape.
namespace {
typedef unsigned bw_u4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void bw_probe_kernel(const bw_u4* __restrict__ src, bw_u4* __restrict__ dst, size_t n16, int reps)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (int r = 0; r < reps; ++r) {
        size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
        for (; i + (U - 1) * stride < n16; i += U * stride) {
            bw_u4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                v[u].x += (unsigned)r;          // (a pass must not be optimised into the previous one)
                if (NT) __builtin_nontemporal_store(v[u], &dst[i + u * stride]);
                else dst[i + u * stride] = v[u];
            }
        }
        for (; i < n16; i += stride) {
            bw_u4 v = src[i];
            v.x += (unsigned)r;
            dst[i] = v;
        }
    }
}

using bw_fn = void (*)(const bw_u4*, bw_u4*, size_t, int);
struct bw_variant { bw_fn fn; int blocks; };
}  // namespace

extern "C" int esr_bw_probe(void* buf, size_t bytes, int reps, void* hip_stream, double* gbs_out)
{
    if (!buf || bytes < 4096 || reps <= 0 || !gbs_out) return ESR_ERR_BAD_ARG;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const size_t n16 = bytes / 16;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return ESR_ERR_LAUNCH;
    const bw_u4* src = static_cast<const bw_u4*>(buf);
    bw_u4* dst = static_cast<bw_u4*>(buf) + n16;
    const bw_variant variants[] = {
        {bw_probe_kernel<1, false>, 8 * cus}, {bw_probe_kernel<4, false>, 8 * cus}, {bw_probe_kernel<4, true>, 8 * cus},
        {bw_probe_kernel<8, false>, 8 * cus}, {bw_probe_kernel<8, true>, 8 * cus}, {bw_probe_kernel<4, false>, 16 * cus},
        {bw_probe_kernel<4, true>, 16 * cus}, {bw_probe_kernel<4, true>, 32 * cus}, {bw_probe_kernel<8, true>, 4 * cus},
    };
    int rc = ESR_OK;
    double best = 0.0;
    const bool verbose = std::getenv("ESR_BW_PROBE_VERBOSE") != nullptr;
    hipLaunchKernelGGL(variants[0].fn, dim3(variants[0].blocks), dim3(256), 0, st, src, dst, n16, 2);        // warm the caches
    for (const bw_variant& v : variants) {
        (void)hipEventRecord(e0, st);
        hipLaunchKernelGGL(v.fn, dim3(v.blocks), dim3(256), 0, st, src, dst, n16, reps);
        (void)hipEventRecord(e1, st);
        rc = esr_check_launch("bw_probe_kernel launch");
        if (rc == ESR_OK && hipEventSynchronize(e1) != hipSuccess) rc = ESR_ERR_LAUNCH;
        float ms = 0.f;
        if (rc == ESR_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = ESR_ERR_LAUNCH;
        if (rc != ESR_OK) break;
        const double gbs = 2.0 * (double)(n16 * 16) * reps / (ms * 1e-3) / 1e9;
        if (gbs > best) best = gbs;
        if (verbose) std::fprintf(stderr, "esr_bw_probe: %zu B x %d, variant %d (%d blocks): %.1f GB/s\n", n16 * 16, reps, (int)(&v - variants), v.blocks, gbs);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (rc != ESR_OK) return rc;
    *gbs_out = best;
    return ESR_OK;
}

// esr_event_pair_ms: what per-launch hipEvent brackets add to a launch -- n launches of a ~15 us copy kernel timed (a) by ONE pair around all
// of them and (b) by a pair around each: (sum of (b) - (a)) / n.  That is the inflation of esr_run_ops_profiled's per-op numbers relative to
// the kernels running back to back (~2.5 us: a fifth of a 13 us single-image launch; it agrees with rocprofv3's kernel durations to a few
// tenths of a microsecond, where an EMPTY pair -- 4.8 us -- or a device-stamped probe -- 11 us, it also sees the dispatch latency -- do not).
namespace {
__global__ __launch_bounds__(256) void event_probe_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
}  // namespace

extern "C" int esr_event_pair_ms(void* hip_stream, int n, double* ms_out)
{
    if (n <= 0 || n > 1024 || !ms_out) return ESR_ERR_BAD_ARG;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const size_t bytes = 24u << 20;
    char* buf = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&buf), 2 * bytes) != hipSuccess) return ESR_ERR_LAUNCH;
    hipEvent_t* ev = new hipEvent_t[2 * n + 2];
    int made = 0;
    for (; made < 2 * n + 2; ++made)
        if (hipEventCreate(&ev[made]) != hipSuccess) break;
    int rc = made == 2 * n + 2 ? ESR_OK : ESR_ERR_LAUNCH;
    auto launch = [&]() {
        hipLaunchKernelGGL(event_probe_kernel, dim3(1024), dim3(256), 0, st, reinterpret_cast<const uint4*>(buf), reinterpret_cast<uint4*>(buf + bytes), bytes / 16);
    };
    double together = 0.0, apart = 0.0;
    if (rc == ESR_OK) {
        (void)hipMemsetAsync(buf, 0, 2 * bytes, st);
        for (int i = 0; i < 4; ++i) launch();
        (void)hipEventRecord(ev[2 * n], st);
        for (int i = 0; i < n; ++i) launch();
        (void)hipEventRecord(ev[2 * n + 1], st);
        for (int i = 0; i < n; ++i) {
            (void)hipEventRecord(ev[2 * i], st);
            launch();
            (void)hipEventRecord(ev[2 * i + 1], st);
        }
        rc = esr_check_launch("event_probe_kernel launch");
        if (rc == ESR_OK && hipStreamSynchronize(st) != hipSuccess) rc = ESR_ERR_LAUNCH;
    }
    if (rc == ESR_OK) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[2 * n], ev[2 * n + 1]) != hipSuccess) rc = ESR_ERR_LAUNCH;
        together = ms;
        for (int i = 0; i < n && rc == ESR_OK; ++i) {
            if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) != hipSuccess) rc = ESR_ERR_LAUNCH;
            apart += ms;
        }
        if (rc == ESR_OK) *ms_out = apart > together ? (apart - together) / n : 0.0;
    }
    for (int i = 0; i < made; ++i) (void)hipEventDestroy(ev[i]);
    delete[] ev;
    (void)hipFree(buf);
    return rc;
}
