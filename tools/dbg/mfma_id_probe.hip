// is acc + x through an identity MFMA (one non-zero product per output) bit-identical to the fp32 VALU add?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool BF16>
__global__ void k(const unsigned short* x, const float* c, float* out_mfma, float* out_valu, int n)
{
    const int lane = threadIdx.x & 63, px = lane & 15, kq = lane >> 4;
    for (int it = blockIdx.x; it < n; it += gridDim.x) {
        // B: lane (px, kq) holds 8 values x[it][px][8 (kq & 1) + j] in slots kq < 2, garbage-ish other data in kq >= 2
        i32x4 b; unsigned short bb[8];
        for (int j = 0; j < 8; ++j) bb[j] = x[(size_t)it * 512 + px * 32 + (kq * 8 + j)];
        memcpy(&b, bb, 16);
        // A: identity selector: row i takes k = slot of channel i of the first 16
        unsigned short aa[8]; const int i = px;
        for (int j = 0; j < 8; ++j) aa[j] = (kq == (i >> 3) && j == (i & 7)) ? (BF16 ? 0x3F80 : 0x3C00) : 0;
        i32x4 a; memcpy(&a, aa, 16);
        f32x4 cc = *reinterpret_cast<const f32x4*>(c + (size_t)it * 1024 + px * 16 * 4 + kq * 4);   // D lane (px, kq): channels 4 kq .. 4 kq + 3 of pixel px  -> c[it][px][ch]
        cc = *reinterpret_cast<const f32x4*>(c + ((size_t)it * 16 + px) * 16 + kq * 4);
        f32x4 d;
        if (BF16) d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), cc, 0, 0, 0);
        else d = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), cc, 0, 0, 0);
        f32x4 v = cc;
        for (int r = 0; r < 4; ++r) {
            const unsigned short h = x[(size_t)it * 512 + px * 32 + 4 * kq + r];
            float f;
            if (BF16) { unsigned u = (unsigned)h << 16; memcpy(&f, &u, 4); } else { _Float16 hh; memcpy(&hh, &h, 2); f = (float)hh; }
            v[r] += f;
        }
        *reinterpret_cast<f32x4*>(out_mfma + ((size_t)it * 16 + px) * 16 + kq * 4) = d;
        *reinterpret_cast<f32x4*>(out_valu + ((size_t)it * 16 + px) * 16 + kq * 4) = v;
    }
}
int main()
{
    const int n = 4096;
    unsigned short* hx = (unsigned short*)malloc(n * 512 * 2); float* hc = (float*)malloc(n * 256 * 4);
    for (int bf = 0; bf < 2; ++bf) {
        srand(7 + bf);
        for (int i = 0; i < n * 512; ++i) { float f = ((rand() % 20001) - 10000) / 3000.f * ((rand() & 7) == 0 ? 1e-3f : 1.f);
            if (bf) { unsigned u; memcpy(&u, &f, 4); hx[i] = (unsigned short)(u >> 16); } else { _Float16 h = (_Float16)f; memcpy(&hx[i], &h, 2); } }
        for (int i = 0; i < n * 256; ++i) hc[i] = ((rand() % 20001) - 10000) / 777.f * ((rand() & 3) == 0 ? 1e-4f : 1.f);
        unsigned short* dx; float *dc, *d1, *d2;
        hipMalloc(&dx, n * 512 * 2); hipMalloc(&dc, n * 256 * 4); hipMalloc(&d1, n * 256 * 4); hipMalloc(&d2, n * 256 * 4);
        hipMemcpy(dx, hx, n * 512 * 2, hipMemcpyHostToDevice); hipMemcpy(dc, hc, n * 256 * 4, hipMemcpyHostToDevice);
        if (bf) hipLaunchKernelGGL(k<true>, dim3(256), dim3(64), 0, 0, dx, dc, d1, d2, n); else hipLaunchKernelGGL(k<false>, dim3(256), dim3(64), 0, 0, dx, dc, d1, d2, n);
        float* o1 = (float*)malloc(n * 256 * 4); float* o2 = (float*)malloc(n * 256 * 4);
        hipMemcpy(o1, d1, n * 256 * 4, hipMemcpyDeviceToHost); hipMemcpy(o2, d2, n * 256 * 4, hipMemcpyDeviceToHost);
        long diff = 0; for (int i = 0; i < n * 256; ++i) diff += memcmp(&o1[i], &o2[i], 4) != 0;
        printf("%s: %ld of %d results differ between the identity MFMA and the VALU add\n", bf ? "bf16" : "f16", diff, n * 256);
    }
    return 0;
}
