// esr_s16.hip -- the 16-bit-STORAGE convolution of libesr_hip.so (BASELINE.json configs [2]-[4]: bf16 / fp16).
// Interface: include/esr_hip.h (esr_conv_desc.storage != ESR_STORE_F32).  Design notes: DESIGN.md section 4.
//
// Activations live in HBM as NHWC bf16 / fp16; the matrix products run on v_mfma_f32_16x16x32_{bf16,f16} with fp32
// accumulation; bias, residual, activation are applied in fp32 and the result is rounded ONCE (RNE) when it is stored.
// At 16x the fp32 matrix rate the layer is HBM-bound, so the kernel is built around the memory pipe:
//   * the input halo tile goes global -> LDS by DMA (`buffer_load_dwordx4 ... lds`): no staging VGPRs, no ds_write, no
//     conversion (the storage type IS the operand type); out-of-image halo pixels use an out-of-range buffer offset and
//     the hardware writes zeros (the convolution's zero padding);
//   * a ring of three K stages (16 channels = 32 bytes per pixel each): while stage g is multiplied, g+1 and g+2 are in
//     flight, across tile boundaries of the persistent block, so ~40 KB per CU are always on the way from HBM;
//   * the layer's whole weight set is resident in LDS for the life of the block (<= 80 KB: 64 -> 64 channels, 3x3);
//   * one barrier per stage, counted `s_waitcnt vmcnt` (loads return in order: the older stage has landed while the
//     newer stays in flight); the epilogue runs AFTER that barrier so its stores get a whole stage to drain before the
//     next counted wait has to look past them.
// GEMM view and fragment maps: D[cout][pixel], A = weights, B = 16 consecutive pixels of one image row, D gives lane
// (px, kq) 4 consecutive output channels of one pixel -- exactly as conv_f32_kernel (esr_hip.hip).  K slots of one MFMA:
// lane (i, kq) holds 8 consecutive k = 8 channels (16 bytes) of ONE tap: kq & 1 selects the channel half of the chunk,
// kq >> 1 the tap of a tap PAIR, so the 9 taps of a chunk take 5 MFMAs (the 10th tap slot holds zero weights).
// For 1x1 convolutions the second tap slot is not wasted: it carries the LOW part of the weights (w = hi + lo, both
// 16-bit), so 1x1 layers see effectively fp32-accurate weights at no cost.  3x3 weights are rounded with error
// diffusion over the 9 taps of each (cout, cin) filter (esr_pack_conv_s16): the filter's DC gain, which dominates the
// response to natural features, keeps fp32 accuracy.  Measured effect on RLFN bf16: tools/emulate_s16.py, DESIGN.md.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "esr_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int TILE = 16;            // output tile width (pixels) = one MFMA's pixel dimension
constexpr int RING = 3;             // input stages in LDS
constexpr int EPI_PITCH = 68;       // floats per scratch pixel row (64 + 4 pad)
constexpr int EPI_PIX = 8;          // pixels per transposed half row
constexpr unsigned OOB = 0x80000000u;
constexpr int LDS_LIMIT = 160 * 1024;

struct S16K {
    const char* x;        // NHWC 16-bit input
    const char* wp;       // esr_pack_conv_s16 blob: weight image, then fp32 bias
    const float* bias;
    const char* res;      // NHWC 16-bit residual
    char* y0;             // NHWC 16-bit output, or NCHW fp32 (ESR_NCHW_SHUFFLE4)
    char* y1;
    int N, H, W;
    int nchunks;          // ceil(cin_phys / 16)
    int in_pitch, in_coff;
    int res_pitch, res_coff;
    int y0_pitch, y0_coff, y1_pitch, y1_coff;
    int cout_store;       // NHWC: round_up8(cout) -- channels >= this are never stored; SHUFFLE4: cout
    int split;
    int act;
    float slope;          // LeakyReLU slope; the kernel evaluates max(v, slope * v): 1 = identity, 0 = ReLU
    int res_mode;
    int out_layout;
    int tiles_x, tiles_y;
};

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

__device__ __forceinline__ float act1(float v, int act, float slope)
{
    // slope carries none (1) / LeakyReLU (s) / ReLU (0); GELU is the only other activation on the path (BSRN)
    if (act == ESR_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    return fmaxf(v, slope * v);
}

// LDS-DMA: every lane moves 16 bytes from (buffer base + voff + soff) to LDS byte (lds_dst + lane * 16); an out-of-range
// voff writes zeros.  Issued from inline asm so that hipcc does not put vmcnt(0) in front of later ds_reads (it cannot see
// which LDS bytes the DMA touches); completion is tracked by the counted waits of the stage loop.
__device__ __forceinline__ void dma_buf16(unsigned lds_dst, unsigned voff, i32x4 rsrc, unsigned soff)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

__device__ __forceinline__ void dma_glb16(unsigned lds_dst, const void* g)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(g) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NT, int KS, int NW, bool BF16>
__global__ __launch_bounds__(64 * NW, 1) void conv_s16_kernel(const S16K p)
{
    constexpr int THREADS = 64 * NW;
    constexpr int HALO = KS / 2;
    constexpr int TH = TILE + 2 * HALO;          // halo tile width = LDS row pitch in pixels
    constexpr int TILE_H = 4 * NW;               // wave wv owns rows 4wv .. 4wv+3
    constexpr int THY = TILE_H + 2 * HALO;
    constexpr int NPX = TH * THY;
    constexpr int PPP = (NPX + 63) / 64;         // 1 KB DMA pieces per channel-half plane
    constexpr int PLANE_BYTES = PPP * 1024;      // multiple of 256: a tap shift moves all lanes of a ds_read_b128 alike
    constexpr int STAGE_BYTES = 2 * PLANE_BYTES; // [half][halo pixel][8 channels]
    constexpr int NPIECES = 2 * PPP;
    constexpr int PPW = (NPIECES + NW - 1) / NW; // pieces per wave and stage (waves >= NPIECES % NW: one fewer)
    constexpr int TAPS = KS * KS;
    constexpr int PAIRS = (TAPS + 1) / 2;
    constexpr int W_CHUNK_BYTES = PAIRS * NT * 1024;   // [pair][tile][lane][16 B]

    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15;
    const int kq = lane >> 4;
    const int w_bytes = p.nchunks * W_CHUNK_BYTES;
    char* const ring = smem + w_bytes;
    float* const scr = reinterpret_cast<float*>(ring + RING * STAGE_BYTES) + wv * (EPI_PIX * EPI_PITCH);
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned ring_lds = smem_lds + (unsigned)w_bytes;

    // ---- tile walk (persistent; XCD-aware order as in conv_f32_kernel) ---------------------------------------------
    const int ntiles = p.N * p.tiles_y * p.tiles_x;
    const int G = gridDim.x;
    auto tile_index = [&](int k) -> int {
        const int base = k * G;
        if (base >= ntiles) return -1;
        int off = blockIdx.x;
        if ((G & 7) == 0 && base + G <= ntiles) off = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
        const int t = base + off;
        return t < ntiles ? t : -1;
    };
    auto tile_coords = [&](int t, int& n, int& x0, int& y0) {
        const int tx = t % p.tiles_x;
        const int tq = t / p.tiles_x;
        const int ty = tq % p.tiles_y;
        n = tq / p.tiles_y;
        x0 = tx * TILE;
        y0 = ty * TILE_H;
    };

    // ---- load cursor: the (tile, chunk) stage requested next ---------------------------------------------------------
    // piece pc = wv + NW * r of a stage: plane pc / PPP, items (pc % PPP) * 64 + lane of that plane
    const int n_my = (NPIECES % NW == 0 || wv < NPIECES % NW) ? PPW : PPW - 1;     // wave-uniform
    const size_t img_bytes = (size_t)p.H * p.W * p.in_pitch * 2;
    int lk = 0;                   // tile iteration of the cursor
    int lc = 0;                   // chunk of the cursor
    int lslot = 0;
    bool lvalid;
    unsigned lvoff[PPW];
    i32x4 lrsrc;
    auto cursor_tile = [&]() {
        const int t = tile_index(lk);
        lvalid = t >= 0;
        if (!lvalid) return;
        int n, x0, y0;
        tile_coords(t, n, x0, y0);
        const char* base = p.x + (size_t)n * img_bytes;
        lrsrc.x = (int)(size_t)base;
        lrsrc.y = (int)(((size_t)base >> 32) & 0xffff);
        lrsrc.z = (int)img_bytes;
        lrsrc.w = 0x00020000;
#pragma unroll
        for (int r = 0; r < PPW; ++r) {
            const int pc = wv + NW * r;
            const int plane = pc / PPP;
            const int pl = (pc - plane * PPP) * 64 + lane;
            const int ly = pl / TH, lx = pl - ly * TH;
            const int gy = y0 - HALO + ly, gx = x0 - HALO + lx;
            const bool ok = pc < NPIECES && pl < NPX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            lvoff[r] = ok ? (unsigned)((gy * p.W + gx) * p.in_pitch + p.in_coff + 8 * plane) * 2u : OOB;
        }
    };
    auto issue_stage = [&]() {          // DMA of the cursor's stage into ring slot lslot, then advance the cursor
        const unsigned dst0 = ring_lds + (unsigned)(lslot * STAGE_BYTES);
        const unsigned soff = (unsigned)lc * 32u;
#pragma unroll
        for (int r = 0; r < PPW; ++r) {
            const int pc = wv + NW * r;
            if (NPIECES % NW == 0 || r < PPW - 1 || pc < NPIECES)       // wave-uniform
                dma_buf16(dst0 + (unsigned)pc * 1024u, lvoff[r], lrsrc, soff);
        }
        lslot = lslot == RING - 1 ? 0 : lslot + 1;
        if (++lc == p.nchunks) {
            lc = 0;
            ++lk;
            cursor_tile();
        }
    };
    // counted wait + barrier: the OLDEST outstanding stage has landed in every wave's view; `newer` = a younger stage
    // of this wave is still in flight and stays so
    auto stage_sync = [&](bool newer) {
        if (!newer) wait_vm<0>();
        else if (n_my == PPW) wait_vm<PPW>();
        else wait_vm<(PPW > 1 ? PPW - 1 : 0)>();
        __builtin_amdgcn_s_barrier();
    };

    // ---- prologue: weights (resident), stages 0 and 1 ---------------------------------------------------------------
    {
        const int wpieces = w_bytes / 1024;
        for (int pc = wv; pc < wpieces; pc += NW)
            dma_glb16(smem_lds + (unsigned)pc * 1024u, p.wp + (size_t)pc * 1024 + lane * 16);
    }
    cursor_tile();
    if (!lvalid) return;                 // block without tiles (grid <= ntiles: does not happen)
    issue_stage();
    bool ahead = lvalid;
    if (ahead) issue_stage();
    stage_sync(ahead);

    f32x4 biasv[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) biasv[tt] = *reinterpret_cast<const f32x4*>(p.bias + tt * 16 + kq * 4);

    // lane-constant LDS offsets of the B fragments: pair q reads tap min(2q + (kq >> 1), TAPS - 1), channel half kq & 1
    int b_off[PAIRS];
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
        const int tap = min(2 * q + (kq >> 1), TAPS - 1);
        b_off[q] = (kq & 1) * PLANE_BYTES + (((wv * 4) + tap / KS) * TH + px + tap % KS) * 16;
    }
    const int a_off = lane * 16;

    int slot = 0;
    for (int k = 0;; ++k) {
        const int t = tile_index(k);
        if (t < 0) break;
        int n, x0, y0;
        tile_coords(t, n, x0, y0);
        f32x4 acc[NT][4];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[tt][r] = biasv[tt];

        for (int c = 0; c < p.nchunks; ++c) {
            // request the stage after next: its slot was last read two stages ago, every wave is past that barrier
            ahead = lvalid;
            if (ahead) issue_stage();
            const char* s = ring + slot * STAGE_BYTES;
            const char* wc = smem + c * W_CHUNK_BYTES + a_off;
            i32x4 a[2][NT], b[2][4];
            auto load_frag = [&](int buf, int q) {
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) a[buf][tt] = *reinterpret_cast<const i32x4*>(wc + (q * NT + tt) * 1024);
#pragma unroll
                for (int r = 0; r < 4; ++r) b[buf][r] = *reinterpret_cast<const i32x4*>(s + b_off[q] + r * (TH * 16));
            };
            load_frag(0, 0);
#pragma unroll
            for (int q = 0; q < PAIRS; ++q) {
                const int cs = q & 1;
                if (q + 1 < PAIRS) load_frag(cs ^ 1, q + 1);
                __builtin_amdgcn_sched_barrier(0);          // keep the prefetch above this pair's MFMAs
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[tt][r] = mfma32<BF16>(a[cs][tt], b[cs][r], acc[tt][r]);
            }
            // the next stage (requested one stage ago) has landed once at most the newest request is outstanding
            stage_sync(ahead);
            slot = slot == RING - 1 ? 0 : slot + 1;
        }

        // ---- epilogue (after the barrier: the other waves are already multiplying the next tile) --------------------
        if (p.out_layout == ESR_NCHW_SHUFFLE4) {
            // out[n, t, 4gy + kq, 4gx + 0..3] = channel 16t + 4kq + j: the D fragment is one dwordx4 of 4 adjacent HR pixels
            const int gx = x0 + px;
            const size_t W4 = (size_t)p.W * 4, H4 = (size_t)p.H * 4;
            const int nco = p.cout_store / 16;
            float* const yo = reinterpret_cast<float*>(p.y0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gy = y0 + wv * 4 + r;
                if (gy >= p.H || gx >= p.W) continue;
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    if (tt * 16 + kq * 4 >= p.cout_store) continue;
                    f32x4 v = acc[tt][r];
                    v.x = act1(v.x, p.act, p.slope); v.y = act1(v.y, p.act, p.slope);
                    v.z = act1(v.z, p.act, p.slope); v.w = act1(v.w, p.act, p.slope);
                    *reinterpret_cast<f32x4*>(yo + (((size_t)n * nco + tt) * H4 + (size_t)gy * 4 + kq) * W4 + (size_t)gx * 4) = v;
                }
            }
        } else {
            // each wave transposes half a pixel row at a time through its private scratch: lane (p8, cg) then owns the 8
            // channels 8cg.. of pixel p8 -- one 16-byte residual load and one 16-byte store per lane, 128 contiguous
            // bytes per pixel, 1 KB per instruction
            const int p8 = lane >> 3, cg = lane & 7;
            const int cb = cg * 8;
            const bool ch_ok = cb < p.cout_store;
            const bool to0 = cb < p.split;
            const int rdc = (NT == 4) ? cb : min(cb, NT * 16 - 8);
            char* const ybase = to0 ? p.y0 + (size_t)(p.y0_coff + cb) * 2 : p.y1 + (size_t)(p.y1_coff + cb - p.split) * 2;
            const int ypitch2 = (to0 ? p.y0_pitch : p.y1_pitch) * 2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gy = y0 + wv * 4 + r;
                i32x4 rv[2];
                bool ok[2];
                size_t pix[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int gx = x0 + 8 * h + p8;
                    ok[h] = ch_ok && gy < p.H && gx < p.W;
                    pix[h] = ((size_t)n * p.H + gy) * p.W + gx;
                    rv[h] = i32x4{0, 0, 0, 0};
                    if (p.res_mode != ESR_RES_NONE && ok[h])
                        rv[h] = *reinterpret_cast<const i32x4*>(p.res + (pix[h] * p.res_pitch + p.res_coff + cb) * 2);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    __builtin_amdgcn_wave_barrier();
                    if ((px >> 3) == h) {
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt) *reinterpret_cast<f32x4*>(scr + (px & 7) * EPI_PITCH + tt * 16 + kq * 4) = acc[tt][r];
                    }
                    __builtin_amdgcn_wave_barrier();
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(scr + p8 * EPI_PITCH + rdc);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(scr + p8 * EPI_PITCH + rdc + 4);
                    float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    float rr[8];
                    unpack2<BF16>((unsigned)rv[h].x, rr[0], rr[1]); unpack2<BF16>((unsigned)rv[h].y, rr[2], rr[3]);
                    unpack2<BF16>((unsigned)rv[h].z, rr[4], rr[5]); unpack2<BF16>((unsigned)rv[h].w, rr[6], rr[7]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float o = v[j];
                        if (p.res_mode == ESR_RES_PRE_ACT) o += rr[j];
                        o = act1(o, p.act, p.slope);
                        if (p.res_mode == ESR_RES_POST_ACT) o += rr[j];
                        v[j] = o;
                    }
                    if (ok[h]) {
                        i32x4 o;
                        o.x = (int)pack2<BF16>(v[0], v[1]); o.y = (int)pack2<BF16>(v[2], v[3]);
                        o.z = (int)pack2<BF16>(v[4], v[5]); o.w = (int)pack2<BF16>(v[6], v[7]);
                        *reinterpret_cast<i32x4*>(ybase + pix[h] * ypitch2) = o;
                    }
                }
            }
        }
    }
}

template <int NT, int KS, int NW, bool BF16>
int launch_s16(const S16K& k, size_t lds, hipStream_t st)
{
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s16_kernel<NT, KS, NW, BF16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
        attr = true;
    }
    const int ntiles = k.N * k.tiles_x * k.tiles_y;
    const int grid = ntiles < 256 ? ntiles : 256;          // one block per CU (LDS), persistent over the tiles
    hipLaunchKernelGGL((conv_s16_kernel<NT, KS, NW, BF16>), dim3(grid), dim3(64 * NW), lds, st, k);
    return esr_check_launch("conv_s16_kernel launch");
}

template <int KS, bool BF16>
int launch_s16_nt(int nt, const S16K& k, size_t lds, hipStream_t st)
{
    switch (nt) {
        case 1: return launch_s16<1, KS, 8, BF16>(k, lds, st);
        case 2: return launch_s16<2, KS, 8, BF16>(k, lds, st);
        case 3: return launch_s16<3, KS, 8, BF16>(k, lds, st);
        case 4: return launch_s16<4, KS, 8, BF16>(k, lds, st);
    }
    return ESR_ERR_UNSUPPORTED;
}

// LDS bytes of a launch: resident weights + input ring + epilogue scratch
size_t s16_lds_bytes(int nchunks, int nt, int ksize, int nw)
{
    const int halo = ksize / 2, th = TILE + 2 * halo, thy = 4 * nw + 2 * halo;
    const int ppp = (th * thy + 63) / 64;
    const int pairs = (ksize * ksize + 1) / 2;
    return (size_t)nchunks * pairs * nt * 1024 + (size_t)RING * 2 * ppp * 1024 + (size_t)nw * EPI_PIX * EPI_PITCH * 4;
}

inline uint16_t f32_to_bf16(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);       // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_to_f32(uint16_t h)
{
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline uint16_t f32_to_f16(float f)
{
    const _Float16 h = (_Float16)f;        // host compiler: IEEE RNE
    uint16_t r;
    memcpy(&r, &h, 2);
    return r;
}
inline float f16_to_f32(uint16_t h)
{
    _Float16 v;
    memcpy(&v, &h, 2);
    return (float)v;
}
inline uint16_t to16(double v, int compute) { return compute == ESR_COMPUTE_BF16 ? f32_to_bf16((float)v) : f32_to_f16((float)v); }
inline double from16(uint16_t h, int compute) { return compute == ESR_COMPUTE_BF16 ? bf16_to_f32(h) : f16_to_f32(h); }

// element index of (physical slot s, tap slot ts in {0..2*pairs-1}, output channel oc) in the weight image
inline size_t s16_index(int nt, int pairs, int s, int ts, int oc)
{
    const int chunk = s / 16, within = s % 16;
    const int q = ts / 2, kq = (ts & 1) * 2 + within / 8, j = within % 8;
    return ((((size_t)chunk * pairs + q) * nt + oc / 16) * 64 + kq * 16 + oc % 16) * 8 + j;
}

}  // namespace

extern "C" {

size_t esr_packed_conv_s16_bytes(int cin_phys, int cout, int ksize)
{
    if (cin_phys <= 0 || cout <= 0 || (ksize != 1 && ksize != 3)) return 0;
    const size_t nt = (size_t)esr_round_up(cout, 16) / 16;
    const size_t nchunks = (size_t)esr_round_up(cin_phys, 16) / 16;
    const size_t pairs = (size_t)(ksize * ksize + 1) / 2;
    return nchunks * pairs * nt * 1024 + nt * 16 * sizeof(float);
}

int esr_pack_conv_s16(const float* w, const float* bias, int cin, int cout, int ksize, const int32_t* cin_map, int cin_phys,
                      int compute, void* out, size_t out_bytes)
{
    if (!w || !out || cin <= 0 || cout <= 0 || (ksize != 1 && ksize != 3)) return ESR_ERR_BAD_ARG;
    if (compute != ESR_COMPUTE_BF16 && compute != ESR_COMPUTE_F16) return ESR_ERR_BAD_ARG;
    if (!cin_map && cin_phys < cin) return ESR_ERR_BAD_ARG;
    const size_t need = esr_packed_conv_s16_bytes(cin_phys, cout, ksize);
    if (need == 0 || out_bytes < need) return ESR_ERR_BAD_ARG;
    const int nt = esr_round_up(cout, 16) / 16, taps = ksize * ksize, pairs = (taps + 1) / 2;
    const int nchunks = esr_round_up(cin_phys, 16) / 16;
    memset(out, 0, need);
    uint16_t* o = static_cast<uint16_t*>(out);
    for (int s = 0; s < cin_phys; ++s) {
        const int c = cin_map ? cin_map[s] : (s < cin ? s : -1);
        if (c < 0) continue;
        if (c >= cin) return ESR_ERR_BAD_ARG;
        for (int oc = 0; oc < cout; ++oc) {
            const float* wf = w + ((size_t)oc * cin + c) * taps;
            if (ksize == 1) {
                // w = hi + lo: the second tap slot of the pair carries the rounding residual of the first
                const uint16_t hi = to16(wf[0], compute);
                const uint16_t lo = to16((double)wf[0] - from16(hi, compute), compute);
                o[s16_index(nt, pairs, s, 0, oc)] = hi;
                o[s16_index(nt, pairs, s, 1, oc)] = lo;
            } else {
                // error diffusion over the 9 taps: tap k is rounded after adding the rounding error carried from tap k-1,
                // so the SUM of the filter's taps (its DC gain) is exact to one rounding of the last tap
                double e = 0.0;
                for (int tap = 0; tap < taps; ++tap) {
                    const double t = (double)wf[tap] + e;
                    const uint16_t q = to16(t, compute);
                    e = t - from16(q, compute);
                    o[s16_index(nt, pairs, s, tap, oc)] = q;
                }
            }
        }
    }
    float* bo = reinterpret_cast<float*>(static_cast<char*>(out) + (size_t)nchunks * pairs * nt * 1024);
    if (bias)
        for (int oc = 0; oc < cout; ++oc) bo[oc] = bias[oc];
    return ESR_OK;
}

int esr_unpack_conv_s16(const void* packed, size_t bytes, int cin, int cout, int ksize, const int32_t* cin_map, int cin_phys,
                        int compute, float* w, float* bias)
{
    if (!packed || !w || cin <= 0 || cout <= 0 || (ksize != 1 && ksize != 3)) return ESR_ERR_BAD_ARG;
    if (compute != ESR_COMPUTE_BF16 && compute != ESR_COMPUTE_F16) return ESR_ERR_BAD_ARG;
    if (bytes < esr_packed_conv_s16_bytes(cin_phys, cout, ksize)) return ESR_ERR_BAD_ARG;
    const int nt = esr_round_up(cout, 16) / 16, taps = ksize * ksize, pairs = (taps + 1) / 2;
    const int nchunks = esr_round_up(cin_phys, 16) / 16;
    const uint16_t* o = static_cast<const uint16_t*>(packed);
    memset(w, 0, sizeof(float) * (size_t)cout * cin * taps);
    for (int s = 0; s < cin_phys; ++s) {
        const int c = cin_map ? cin_map[s] : (s < cin ? s : -1);
        if (c < 0) continue;
        for (int oc = 0; oc < cout; ++oc)
            for (int tap = 0; tap < taps; ++tap) {
                double v = from16(o[s16_index(nt, pairs, s, tap, oc)], compute);
                if (ksize == 1) v += from16(o[s16_index(nt, pairs, s, 1, oc)], compute);
                w[((size_t)oc * cin + c) * taps + tap] = (float)v;       // the EFFECTIVE weight the kernel multiplies by
            }
    }
    if (bias) {
        const float* bo = reinterpret_cast<const float*>(static_cast<const char*>(packed) + (size_t)nchunks * pairs * nt * 1024);
        for (int oc = 0; oc < cout; ++oc) bias[oc] = bo[oc];
    }
    return ESR_OK;
}

}  // extern "C"

// called by esr_conv2d_f32 (esr_hip.hip) for descriptors with 16-bit storage
int esr_conv2d_s16(const esr_conv_desc* d, void* hip_stream)
{
    const bool bf16 = d->storage == ESR_STORE_BF16;
    if (d->storage != ESR_STORE_BF16 && d->storage != ESR_STORE_F16) return ESR_ERR_BAD_ARG;
    if (d->compute != (bf16 ? ESR_COMPUTE_BF16 : ESR_COMPUTE_F16)) return ESR_ERR_BAD_ARG;   // operand type = storage type
    if (d->in_layout != ESR_NHWC) return ESR_ERR_UNSUPPORTED;                                  // the NCHW head runs on conv_f32_kernel
    if (d->tail_wpacked || d->post_wpacked) return ESR_ERR_UNSUPPORTED;
    if ((d->in.pitch & 7) || (d->in.coff & 7)) return ESR_ERR_BAD_ARG;                         // 16-byte granules
    const int cin_phys = esr_round_up(d->cin, 16);
    if (d->in.coff + cin_phys > d->in.pitch) return ESR_ERR_BAD_ARG;                           // chunk reads stay inside the pixel
    const int nt = esr_round_up(d->cout, 16) / 16;
    const bool shuffle = d->out_layout == ESR_NCHW_SHUFFLE4;
    const int cout8 = esr_round_up(d->cout, 8);
    int split = d->split <= 0 ? cout8 : d->split;
    if (split >= d->cout) split = cout8;
    if (split & 7) return ESR_ERR_BAD_ARG;
    if (shuffle) {
        if (d->cout % 16 || d->res_mode != ESR_RES_NONE) return ESR_ERR_UNSUPPORTED;
    } else if (d->out_layout == ESR_NHWC) {
        if ((d->out0.pitch & 7) || (d->out0.coff & 7) || d->out0.coff + split > d->out0.pitch) return ESR_ERR_BAD_ARG;
        if (split < cout8 && (!d->out1.ptr || (d->out1.pitch & 7) || (d->out1.coff & 7) || d->out1.coff + (cout8 - split) > d->out1.pitch))
            return ESR_ERR_BAD_ARG;
    } else {
        return ESR_ERR_BAD_ARG;
    }
    if (d->res_mode != ESR_RES_NONE && (!d->res.ptr || (d->res.pitch & 7) || (d->res.coff & 7) || d->res.coff + cout8 > d->res.pitch))
        return ESR_ERR_BAD_ARG;
    if ((double)d->h * d->w * d->in.pitch * 2.0 >= 2147483647.0) return ESR_ERR_UNSUPPORTED;   // per-image raw buffer < 2 GiB
    const int nchunks = cin_phys / 16;
    const size_t lds = s16_lds_bytes(nchunks, nt, d->ksize, 8);
    if (lds > (size_t)LDS_LIMIT) return ESR_ERR_UNSUPPORTED;                                     // weight set too large to stay resident
    const int pairs = (d->ksize * d->ksize + 1) / 2;

    S16K k;
    k.x = static_cast<const char*>(d->in.ptr);
    k.wp = static_cast<const char*>(d->wpacked);
    k.bias = reinterpret_cast<const float*>(k.wp + (size_t)nchunks * pairs * nt * 1024);
    k.res = static_cast<const char*>(d->res.ptr);
    k.y0 = static_cast<char*>(d->out0.ptr);
    k.y1 = static_cast<char*>(d->out1.ptr);
    k.N = d->n; k.H = d->h; k.W = d->w;
    k.nchunks = nchunks;
    k.in_pitch = d->in.pitch; k.in_coff = d->in.coff;
    k.res_pitch = d->res.pitch; k.res_coff = d->res.coff;
    k.y0_pitch = d->out0.pitch; k.y0_coff = d->out0.coff;
    k.y1_pitch = d->out1.pitch; k.y1_coff = d->out1.coff;
    k.cout_store = shuffle ? d->cout : cout8;
    k.split = split;
    k.act = d->act;
    k.slope = d->act == ESR_ACT_LRELU ? d->slope : (d->act == ESR_ACT_RELU ? 0.f : 1.f);
    k.res_mode = d->res_mode;
    k.out_layout = d->out_layout;
    k.tiles_x = (d->w + TILE - 1) / TILE;
    k.tiles_y = (d->h + 31) / 32;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if (d->ksize == 3) return bf16 ? launch_s16_nt<3, true>(nt, k, lds, st) : launch_s16_nt<3, false>(nt, k, lds, st);
    return bf16 ? launch_s16_nt<1, true>(nt, k, lds, st) : launch_s16_nt<1, false>(nt, k, lds, st);
}
