// esr_s16_dev.h -- device-side primitives shared by the 16-bit-storage kernels (esr_s16.hip, esr_chain.hip): vector typedefs, compile-time
// loops, MFMA / rounding wrappers, the packed GELU, LDS-DMA issue and counted waits.  Everything is force-inlined and lives in an anonymous
// namespace: a translation unit that includes this header gets its own copies.  Design notes: esr_s16.hip (file header), DESIGN.md section 4.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <utility>
#include <type_traits>

#include "esr_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// Kernel parameter block of every 16-bit-storage convolution kernel (esr_s16.hip, esr_c64m.hip), filled by esr_conv2d_s16.
struct S16K {
    const char* x;        // NHWC 16-bit input
    const char* wp;       // esr_pack_conv_s16 blob: weight image, then fp32 bias
    const float* bias;
    const char* res;      // NHWC 16-bit residual
    char* y0;             // NHWC 16-bit output, or NCHW fp32 (ESR_NCHW_SHUFFLE4)
    char* y1;
    int N, H, W;
    int nchunks;          // ceil(cin_phys / 16)
    int ring;             // input stages in LDS
    int in_pitch, in_coff;
    int res_pitch, res_coff;
    int y0_pitch, y0_coff, y1_pitch, y1_coff;
    int cout_store;       // NHWC: round_up8(cout) -- channels >= this are never stored; SHUFFLE4: cout
    int split;
    int act;
    float slope;          // LeakyReLU slope; the kernel evaluates max(v, slope * v): 1 = identity, 0 = ReLU
    int res_mode;         // residual read from HBM (0 = none)
    int res_in;           // pre-activation residual == the conv input: added from the staged tile in LDS, no loads
    int nres;             // residual from HBM staged like input chunks: this many extra 16-channel stages per tile (PNT1 == 0 kernels)
    int out_layout;
    int tiles_x, tiles_y;
    unsigned magic_x, magic_y;   // ceil(2^32 / tiles): t / tiles == umulhi(t, magic) for t * tiles < 2^32 (0: tiles == 1)
    // post chain (PNT1 > 0 kernels): 1x1 convolution(s) of the epilogue result, evaluated in the epilogue (esr_conv_desc.post_*)
    const char* pw1; const char* pw2;      // esr_pack_post_s16 blobs: hi images, lo images, fp32 bias
    char* py1; char* py2;
    int py1_pitch, py1_coff, py2_pitch, py2_coff;
    int p1_cout8, p2_cout8;                // channels stored (multiples of 8 / 4)
    float p1_slope;                        // activation of post 1 as max(v, slope v)
    int p1_gelu;                           // ... or GELU
    int post_lo;                           // the low-part weight images are resident too (w = hi + lo)
    int store_main;                        // 0: the conv's own result is consumed by the post chain only
    const float* border;                   // esr_conv_desc.border_bias, or NULL
    long long seg_stride;                  // segmented input: bytes between the tensors of the concat (else 0)
    int seg_chunks;                        // chunks per input segment (one tensor: nchunks)
    // hi + lo tensors (esr_conv_desc.hilo, HILO kernels): a value is the sum of two 16-bit numbers kept in two dense tensors of the same
    // shape, the low parts `hilo_stride` bytes behind the high parts
    int w_chunks;                          // resident weight chunks: input chunk c multiplies weight chunk c mod w_chunks (hi + lo INPUT: nchunks / 2;
                                           // the input is then a two-segment concat: seg_stride = hilo_stride, seg_chunks = w_chunks)
    int hilo_out;                          // the epilogue stores the low parts too (through y1 = y0 + hilo_stride)
    long long res_lo_stride;               // hi + lo RESIDUAL: 2 NT residual stages per tile, the second NT from res + this many bytes (else 0)
    // esr_c64m.hip (v_mfma_f32_32x32x16 family): the weight image in that MFMA's fragment order (appended to the esr_pack_conv_s16 blob), the
    // post 1x1's images likewise (esr_pack_post_s16 blob) and its fp32 bias
    const char* wm32; const char* pm32; const float* pbias1;
    // rfdb_tail_kernel (esr_conv_desc.tail_* in 16-bit storage): the 1x1's esr_pack_tail_s16 blob, the three concat segments (first one; the
    // others cat_seg_stride bytes apart), their pitch / first channel
    const char* tw; const char* cat; int cat_pitch, cat_coff; long long cat_seg_stride;
};

// esr_c64m.hip: the 64 -> 64 3x3 family on v_mfma_f32_32x32x16 (round 6).  `post`: with one post 1x1 of <= 32 outputs (RFDB c{j}_r + c{j+1}_d);
// `hl` (bf16, no post): a hi + lo residual pair of another tensor added before the activation, hi + lo output (the LR conv behind the long skip)
int esr_launch_conv64m(const S16K& k, bool bf16, bool post, bool hl, hipStream_t st);
// ... and RFDB's c4 -> cat -> c5 -> esa.conv1 in one launch (round 6, ABI v12)
int esr_launch_rfdb_tail(const S16K& k, bool bf16, hipStream_t st);
// esr_r16.hip: the register-resident 48-channel / c4 kernels (weights in accumulation registers, one wave per SIMD)
int esr_launch_conv48rp(const S16K& k, bool bf16, bool lrs, hipStream_t st);
int esr_launch_conv48r(const S16K& k, bool bf16, int nt, bool ext, int rw, hipStream_t st);
int esr_launch_conv48rq(const S16K& k, hipStream_t st);
int esr_launch_conv64r(const S16K& k, bool bf16, hipStream_t st);
// byte offsets of the 32x32x16 weight images inside the esr_pack_conv_s16 / esr_pack_post_s16 blobs (0: this shape carries none), and their writers
size_t esr_m32_conv_offset(int cin_phys, int cout, int ksize);
size_t esr_m32_conv_bytes(int cin_phys, int cout, int ksize);
size_t esr_m32_post_offset(int cin, int cout);
size_t esr_m32_post_bytes(int cin, int cout);

namespace {

// compile-time loops: the body sees its index as a constant expression (std::integral_constant) -- schedules written as `if constexpr`
// chains do not depend on hipcc's unrolling heuristics (a loop it leaves rolled indexes registers dynamically: scratch)
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int TILE = 16;            // output tile width (pixels) = one MFMA's pixel dimension
constexpr int RING_MIN = 3, RING_MAX = 8;   // input stages in LDS (as many as fit next to the resident weights)
constexpr unsigned OOB = 0x80000000u;
constexpr int LDS_LIMIT = 160 * 1024;
constexpr int MAX_DEVICES = 64;     // per-device launch attributes (launch_s16)
constexpr int S16_NW = 8;           // waves per tile

template <bool BF16>
__device__ __forceinline__ f32x4 mfma32(i32x4 a, i32x4 b, f32x4 c)
{
    if (BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// two fp32 -> one dword of two 16-bit values (RNE), and back
template <bool BF16>
__device__ __forceinline__ unsigned pack2(float a, float b)
{
    if (BF16) {
        bf16x2 v;
        v[0] = (__bf16)a; v[1] = (__bf16)b;
        return __builtin_bit_cast(unsigned, v);
    }
    f16x2 v;
    v[0] = (_Float16)a; v[1] = (_Float16)b;
    return __builtin_bit_cast(unsigned, v);
}

template <bool BF16>
__device__ __forceinline__ void unpack2(unsigned u, float& a, float& b)
{
    if (BF16) {
        a = __builtin_bit_cast(float, u << 16);
        b = __builtin_bit_cast(float, u & 0xffff0000u);
    } else {
        const f16x2 v = __builtin_bit_cast(f16x2, u);
        a = (float)v[0]; b = (float)v[1];
    }
}

template <bool BF16>
__device__ __forceinline__ f32x4 unpack4(uint2 u)
{
    float a, b, c, d;
    unpack2<BF16>(u.x, a, b);
    unpack2<BF16>(u.y, c, d);
    return f32x4{a, b, c, d};
}

// GELU for the 16-bit storage modes: x * Phi(x) with Phi(x) - 0.5 = x * P(x^2), P a degree-7 minimax polynomial on |x| <= 4
// (|error| of Phi <= 2.1e-5, tools/fit_gelu.py), the argument clamped to [-4, 4] and the factor x to [-4, inf): |gelu error| <=
// 1.3e-4 for x <= 4 and 5.3e-5 x beyond, about one fp16 step of the values that matter, far below a bf16 step -- and 11 plain VALU instructions
// (packable two values at a time) instead of libm erff's ~40 or the 16 + v_rcp + v_exp of an erf approximation.  The fp32
// path keeps erff.
__device__ __forceinline__ __attribute__((unused)) float gelu16(float x)       // the scalar definition (esr_bsconv.hip uses it as is)
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

// the same arithmetic (bit for bit) on a D fragment with packed fp32 instructions: 16 v_pk_fma + 4 v_pk_mul + 4 v_med3 + 4 v_max
// = 7 VALU instructions per value instead of 13
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu16x2(f32x2 x)
{
    f32x2 xc, xm;
    xc.x = __builtin_amdgcn_fmed3f(x.x, -4.f, 4.f); xc.y = __builtin_amdgcn_fmed3f(x.y, -4.f, 4.f);
    const f32x2 t = xc * xc;
    f32x2 p = {-1.580786198e-09f, -1.580786198e-09f};
    p = __builtin_elementwise_fma(p, t, f32x2{1.217111051e-07f, 1.217111051e-07f});
    p = __builtin_elementwise_fma(p, t, f32x2{-4.100866386e-06f, -4.100866386e-06f});
    p = __builtin_elementwise_fma(p, t, f32x2{8.066739505e-05f, 8.066739505e-05f});
    p = __builtin_elementwise_fma(p, t, f32x2{-1.048204400e-03f, -1.048204400e-03f});
    p = __builtin_elementwise_fma(p, t, f32x2{9.664874174e-03f, 9.664874174e-03f});
    p = __builtin_elementwise_fma(p, t, f32x2{-6.617537882e-02f, -6.617537882e-02f});
    p = __builtin_elementwise_fma(p, t, f32x2{3.988475079e-01f, 3.988475079e-01f});
    const float m4 = -4.f;
    asm("v_max_f32 %0, %1, %2" : "=v"(xm.x) : "v"(x.x), "v"(m4));
    asm("v_max_f32 %0, %1, %2" : "=v"(xm.y) : "v"(x.y), "v"(m4));
    return xm * __builtin_elementwise_fma(xc, p, f32x2{0.5f, 0.5f});
}
__device__ __forceinline__ f32x4 gelu16x4(f32x4 v)
{
    // (round 5: the same polynomial as 12 PLAIN VALU instructions per value -- v_pk_fma_f32 costs a wave ~4x a plain instruction beside MFMAs,
    // tools/r05/mfma_valu_probe.hip -- was measured and LOST: 48 live scalars per fragment push conv48r / conv48rq into scratch (BSRN fp16
    // 2710 -> 1600 images/s), tools/r05/f_gelu.sh)
    // (also measured: the scalar C++ form under -fno-slp-vectorize, conv48rq 0.35 -> 0.39-0.41 ms; -fno-slp-vectorize alone: no difference,
    // tools/r05/h_slp.sh; round 6, on the 4-value blocks of conv64m_kernel / rfdb_tail_kernel's GB forms, four interleaved plain chains
    // under -fno-slp-vectorize: 48 instead of 28 instructions per block, conv64m<.., 3, true> 313 -> 343 us -- the packed form stays)
    // The two halves' Horner chains INTERLEAVED, instruction by instruction (round 5).  A v_pk_fma_f32 that reads the previous one's result needs a
    // wait state, and as gelu16x2(lo) followed by gelu16x2(hi) hipcc filled every one of them with an s_nop 0: 510 s_nop per tile of
    // conv48r_kernel<.., 3, EXT, 8>, each an issue slot of a wave that is bound by exactly those (profiles/r05_instruction_census.txt).
    // Same operations on the same values: bit-identical to gelu16x2 / gelu16.
    f32x2 xa = {v.x, v.y}, xb = {v.z, v.w}, ca, cb, ma, mb;
    ca.x = __builtin_amdgcn_fmed3f(xa.x, -4.f, 4.f); ca.y = __builtin_amdgcn_fmed3f(xa.y, -4.f, 4.f);
    cb.x = __builtin_amdgcn_fmed3f(xb.x, -4.f, 4.f); cb.y = __builtin_amdgcn_fmed3f(xb.y, -4.f, 4.f);
    const f32x2 ta = ca * ca, tb = cb * cb;
    f32x2 pa = {-1.580786198e-09f, -1.580786198e-09f}, pb = pa;
#define ESR_G16_STEP(c) pa = __builtin_elementwise_fma(pa, ta, f32x2{c, c}); pb = __builtin_elementwise_fma(pb, tb, f32x2{c, c});
    ESR_G16_STEP(1.217111051e-07f) ESR_G16_STEP(-4.100866386e-06f) ESR_G16_STEP(8.066739505e-05f) ESR_G16_STEP(-1.048204400e-03f)
    ESR_G16_STEP(9.664874174e-03f) ESR_G16_STEP(-6.617537882e-02f) ESR_G16_STEP(3.988475079e-01f)
#undef ESR_G16_STEP
    const float m4 = -4.f;
    asm("v_max_f32 %0, %1, %2" : "=v"(ma.x) : "v"(xa.x), "v"(m4));
    asm("v_max_f32 %0, %1, %2" : "=v"(ma.y) : "v"(xa.y), "v"(m4));
    asm("v_max_f32 %0, %1, %2" : "=v"(mb.x) : "v"(xb.x), "v"(m4));
    asm("v_max_f32 %0, %1, %2" : "=v"(mb.y) : "v"(xb.y), "v"(m4));
    const f32x2 qa = __builtin_elementwise_fma(ca, pa, f32x2{0.5f, 0.5f}), qb = __builtin_elementwise_fma(cb, pb, f32x2{0.5f, 0.5f});
    const f32x2 ra = ma * qa, rb = mb * qb;
    return f32x4{ra.x, ra.y, rb.x, rb.y};
}

// Epilogue activation: max(v, slope * v); slope carries none (1) / LeakyReLU (s) / ReLU (0).  GELU is applied IN PLACE to the
// accumulators at the end of a tile's last stage (gelu_inplace below), after which the epilogue runs with slope = 1: as a
// second body of the epilogue its polynomials cost every variant ~50 VGPRs (or spills) for an activation only a few
// launches use.  asm: fmaxf() on an MFMA result costs a second v_max (hipcc canonicalises the operand first).
__device__ __forceinline__ float act1(float v, float slope)
{
    float r;
    const float sv = slope * v;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(sv));
    return r;
}

// LDS-DMA: every lane moves 16 bytes from (buffer base + voff + soff) to LDS byte (lds_dst + lane * 16); an out-of-range
// voff writes zeros.  Issued from inline asm so that hipcc does not put vmcnt(0) in front of later ds_reads (it cannot see
// which LDS bytes the DMA touches); completion is tracked by the counted waits of the stage loop.
__device__ __forceinline__ void dma_buf16(unsigned lds_dst, unsigned voff, i32x4 rsrc, unsigned soff)
{
    // wave-uniform by construction; readfirstlane pins them to SGPRs where hipcc's uniformity analysis gives up
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    soff = __builtin_amdgcn_readfirstlane(soff);
    rsrc.x = __builtin_amdgcn_readfirstlane(rsrc.x); rsrc.y = __builtin_amdgcn_readfirstlane(rsrc.y);
    rsrc.z = __builtin_amdgcn_readfirstlane(rsrc.z); rsrc.w = __builtin_amdgcn_readfirstlane(rsrc.w);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

__device__ __forceinline__ void dma_glb16(unsigned lds_dst, const void* g)
{
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(g) : "memory");
}

// s_waitcnt vmcnt(cnt) for a wave-uniform runtime cnt (the immediate has to be a constant): a branch tree over the values
// the stage loop produces; rounding DOWN is always safe (waits for more).
__device__ __forceinline__ void wait_vm_dyn(int cnt)
{
#define ESR_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (cnt < 0 ? 0 : (cnt > 47 ? 47 : cnt)) {
        ESR_W(0) ESR_W(1) ESR_W(2) ESR_W(3) ESR_W(4) ESR_W(5) ESR_W(6) ESR_W(7) ESR_W(8) ESR_W(9) ESR_W(10) ESR_W(11)
        ESR_W(12) ESR_W(13) ESR_W(14) ESR_W(15) ESR_W(16) ESR_W(17) ESR_W(18) ESR_W(19) ESR_W(20) ESR_W(21) ESR_W(22) ESR_W(23)
        ESR_W(24) ESR_W(25) ESR_W(26) ESR_W(27) ESR_W(28) ESR_W(29) ESR_W(30) ESR_W(31) ESR_W(32) ESR_W(33) ESR_W(34) ESR_W(35)
        ESR_W(36) ESR_W(37) ESR_W(38) ESR_W(39) ESR_W(40) ESR_W(41) ESR_W(42) ESR_W(43) ESR_W(44) ESR_W(45) ESR_W(46) ESR_W(47)
    }
#undef ESR_W
}

__device__ __forceinline__ i32x4 make_rsrc(const void* base, size_t bytes)
{
    i32x4 r;
    r.x = (int)(size_t)base;
    r.y = (int)(((size_t)base >> 32) & 0xffff);
    r.z = (int)bytes;
    r.w = 0x00020000;
    return r;
}

// ---- shared by every persistent 16-bit kernel (esr_s16.hip; round 5: one definition instead of one lambda per kernel) ---------------------
// the k-th tile of this block: XCD-aware order (blocks are dealt round-robin to the 8 XCDs: the eight blocks of an XCD take neighbouring tiles,
// whose halos then meet in that XCD's L2), -1 behind the last tile
__device__ __forceinline__ int s16_tile_index(int k, int ntiles)
{
    const int G = gridDim.x;
    const int base = k * G;
    if (base >= ntiles) return -1;
    int off = blockIdx.x;
    if ((G & 7) == 0 && base + G <= ntiles) off = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int t = base + off;
    return t < ntiles ? t : -1;
}
// tile t -> image n, origin (x0, y0): t / tiles by multiplication (magic = ceil(2^32 / tiles), 0 for tiles == 1: the host checks the range)
__device__ __forceinline__ void s16_tile_coords(int t, unsigned mx, unsigned my, int tiles_x, int tiles_y, int tile_h, int& n, int& x0, int& y0)
{
    const int tq = mx ? (int)__umulhi((unsigned)t, mx) : t;
    const int tx = t - tq * tiles_x;
    n = my ? (int)__umulhi((unsigned)tq, my) : tq;
    const int ty = tq - n * tiles_y;
    x0 = tx * TILE;
    y0 = ty * tile_h;
}
// two D-fragment halves (tiles a | b: 4 channels per lane each) -> 8 consecutive channels per lane: the store epilogues' 16-byte pieces
__device__ __forceinline__ i32x4 s16_swap16(uint2 X, uint2 Y)
{
    typedef unsigned s16_u32x2 __attribute__((ext_vector_type(2)));
    const s16_u32x2 a = __builtin_amdgcn_permlane16_swap(X.x, Y.x, false, false);
    const s16_u32x2 b = __builtin_amdgcn_permlane16_swap(X.y, Y.y, false, false);
    return i32x4{(int)a.x, (int)b.x, (int)a.y, (int)b.y};
}

}  // namespace
