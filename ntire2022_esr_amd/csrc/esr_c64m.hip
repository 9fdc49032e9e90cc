// esr_c64m.hip -- the 3x3 convolutions over 64 physical channels with 64 outputs on v_mfma_f32_32x32x16_{bf16,f16} (round 6).
// Interface: esr_conv2d_f32 with 16-bit storage (include/esr_hip.h) -> esr_conv2d_s16 (esr_s16.hip) -> esr_launch_conv64m.
// Reference semantics: RFDB's c1_r / c2_r / c3_r = lrelu(conv3x3(x) + x) (rfdn_baseline/block.py:150-158) and, POST, the next distillation
// conv c{j+1}_d + LeakyReLU of the activated result in the same launch (:152-156).
//
// Why a second kernel family (tools/r06/mfma_shape_probe.hip): conv64r / conv64rq_kernel issue 3.3-3.8 non-MFMA
// instructions per v_mfma_f32_16x16x32 from ONE wave per SIMD; that MFMA occupies the pipe for 16 cycles = 4 issue slots, two fillers are
// free and every further one costs ~4.5 cycles: 27 ns per 16384 MACs measured at that mix.  v_mfma_f32_32x32x16 holds the pipe 32 cycles
// for the same 16384 MACs, hides five fillers (16.9 ns at 5, 21.2 ns at 8 per MFMA) and even alone runs 15.7 ns against 2 x 9.3.
//
// GEMM view: D[cout][pixel] as everywhere in the library; A = weights (32 output channels x 16 k), B = 32 pixels x 16 k, k = the 16
// channels of ONE chunk at ONE tap (nine taps, no tenth tap slot: 10 % fewer MACs than the tap-pair form).  Lane l: A row / B column
// l & 31, k = 8 (l >> 5) + j.  A block = 4 waves = one 16 x 16 pixel tile, a wave = 4 rows = two row PAIRS; the 32 pixels of an MFMA are
// a row pair (pixel n = l & 31: row n >> 4, column n & 15).  Per row pair: 4 chunks x 9 taps x 2 output halves = 72 MFMAs, + 2 that
// load the bias (A = [b_hi b_mid b_lo 0 ..], B = ones: C-operand registers would cost 32 VGPRs) + 4 that add the residual == input
// (A = a 0 / 1 selection matrix against the centre tap's B fragment, which is exact: no centre-pixel reads, no VALU adds).
// D fragment: register r of lane l holds output channel 32 half + 8 (r >> 2) + 4 (l >> 5) + (r & 3) of pixel l & 31; two 4-channel
// blocks of lanes l and l ^ 32 become one 16-byte store through v_permlane32_swap.
//   * input halo tile global -> LDS by DMA into two stages (18 x 18 pixels, 160-byte pixels: conv64r_kernel's conflict-free pitch) with a
//     row pitch of 181 slots: the two rows of an MFMA's pixel set then fall on even / odd 16-byte columns (a ds_read_b128 is served in
//     groups of 16 lanes that hold eight pixels of each row);
//   * weights: 72 fragments of 1 KB.  Plain kernel: all in registers (64 in accumulation registers + 8 in VGPRs); POST: chunks 0 .. 2 in
//     accumulation registers, chunk 3 through a ring from LDS next to the post images;
//   * B fragments: one ds_read_b128 per tap and chunk with an immediate offset (no address arithmetic), a ring of four read three steps ahead;
//   * the finished row pair's epilogue (activation, rounding, hi | lo split for the post 1x1, its MFMAs, swaps, stores) as micro-operations
//     of ~4 instructions behind each MFMA of the next pair, as conv64rq_kernel.
// Same packed 16-bit weights (error-diffused taps) as conv_s16_kernel, laid out for this MFMA (esr_pack_conv_s16 appends the image);
// results agree with conv_s16_kernel to fp32 accumulation order (tests: tests/test_gpu_c64m.py against the fp64 reference).
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <utility>
#include <type_traits>

#include "esr_internal.h"
#include "esr_s16_dev.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// research builds (tools/r06/build_variant.sh ... "-DC64M_ABL=n"): bit 0 no stores, bit 1 no DMA behind the first tile, bit 2 no epilogue,
// bit 3 stores to LINEAR addresses (1 KB contiguous per instruction: wrong placement, the cost of perfectly coalesced stores), bit 4 the DMA
// re-reads the current tile (L2 hits)
#ifndef C64M_ABL
#define C64M_ABL 0
#endif
// the next tile's DMA piece i (of every wave) is issued behind k step C64M_SPREAD * i of the tile's 72 (pieces every step, every 5th step, all 13
// at the top of the tile, one wave per step in turn, waves skewed in time: none faster -- what the DMA costs, ~18 % of the tile loop's cycles,
// comes with the data's arrival, not with the instructions: profiles/r06_c64m_block_ticks.txt, LAB_NOTES 11.2)
#ifndef C64M_SPREAD
#define C64M_SPREAD 3
#endif

#ifdef C64M_TRACE4
// research builds: every block's tile-loop time and its tile-end waits (wave 0), in s_memtime ticks: [block][0 = loop, 1 = sum of the tile-end waits, 2 = tiles]
__device__ unsigned long long c64m_trace4[2][256][4];
extern "C" int esr_c64m_trace4_read(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(c64m_trace4), sizeof(c64m_trace4)) == hipSuccess ? 0 : -1; }
#endif

namespace {

constexpr int M_NCH = 4, M_TAPS = 9, M_TH = 18, M_RW = 4, M_THY = 4 * M_RW + 2;
constexpr int M_LSL = 10;                                 // 16-byte slots of a pixel in LDS (8 + 2 pad)
constexpr int M_PIXB = M_LSL * 16;                        // 160
constexpr int M_ROWSL = M_TH * M_LSL + 1;                 // 181 slots per staged row: odd, see the file header
constexpr int M_ROWB = M_ROWSL * 16;                      // 2896
constexpr int M_NSLOT = M_THY * M_ROWSL;                  // 3258
constexpr int M_NPIECES = (M_NSLOT + 63) / 64;            // 51
constexpr int M_STAGE = M_NPIECES * 1024;                 // 52224
constexpr int M_PPW = (M_NPIECES + 3) / 4;                // 13 pieces per wave, the last wave one fewer
constexpr int M_NG = M_NCH * M_TAPS;                      // 36 k steps per row pair
constexpr int M_NFRAG = M_NG * 2;                         // 72 weight fragments
constexpr int M_SLOTS = 2 + 2 * M_NG + M_NCH;             // 78 MFMAs per row pair
constexpr int M_W3 = 2 * M_STAGE;                         // POST: chunk 3's 18 fragments
constexpr int M_OFF_POST = M_W3 + M_TAPS * 2 * 1024;      // POST: the post images, 8 steps x (hi, lo)
constexpr int M_POST_IMG = 16 * 1024;
constexpr int M_LDS_PLAIN = 2 * M_STAGE;
constexpr int M_LDS_POST = M_OFF_POST + M_POST_IMG;
static_assert(M_LDS_POST <= LDS_LIMIT, "LDS map");
static_assert(C64M_SPREAD >= 1 && (M_PPW - 1) * C64M_SPREAD < 2 * M_NG, "the DMA pieces fit the tile's k steps");

// slot of k step g's LAST MFMA within its row pair (2 bias slots, two per step, one more behind the centre tap of each chunk)
constexpr int m_slot_of_step(int g) { return 2 + 2 * g + (g > 4) + (g > 13) + (g > 22) + (g > 31) + 1 + (g % M_TAPS == 4); }
// slots whose micro-operation issues a store (see `op` in the kernel): POST q = 14, 29, 44, 59, 71, 72 at slot q + 4; plain q = 6, 13, 20, 27 at 2 q + 4
constexpr int m_stores_behind(bool post, int slot)
{
    int n = 0;
    if (post) { for (int s : {18, 33, 48, 63, 75, 76}) n += s > slot; }
    else { for (int s : {16, 30, 44, 58}) n += s > slot; }
    return n;
}
// vector-memory instructions a wave has issued behind its last DMA piece by the end of a tile: the tile-end wait
constexpr int m_tail_stores(bool post, int spread)
{
    const int last = spread * (M_PPW - 1), spp = post ? 6 : 4;
    return last < M_NG ? m_stores_behind(post, m_slot_of_step(last)) + spp : m_stores_behind(post, m_slot_of_step(last - M_NG));
}

// the three-chunk forms (NCH = 3, GB: ESDB's c{j}_r, 27 k steps): store slots of a pair's stream -- POST q = 8, 17, 26, 36, 37 at slot q + 4,
// plain q = 6, 13, 20 at 2 q + 4 -- and the tile-end wait: the last DMA piece (L = 36) rides behind k step 9 of the SECOND pair
constexpr int m3_tail_stores(bool post, int spread)
{
    const int slot = m_slot_of_step(spread * (M_PPW - 1) - 3 * M_TAPS);
    int n = 0;
    if (post) { for (int s : {12, 21, 30, 40, 41}) n += s > slot; }
    else { for (int s : {16, 30, 44}) n += s > slot; }
    return n;
}

template <bool BF16, bool AG>
__device__ __forceinline__ void mfma_m(f32x16& acc, const i32x4& a, const i32x4& b)
{
    if (BF16) {
        if (AG) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
        else asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    } else {
        if (AG) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
        else asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    }
}
// acc = A B (C = 0): the first MFMA of a row pair's chain
template <bool BF16>
__device__ __forceinline__ void mfma_m0(f32x16& acc, const i32x4& a, const i32x4& b)
{
    if (BF16) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b));
    else asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b));
}
template <bool BF16>
__device__ __forceinline__ f32x16 mfma_b(i32x4 a, i32x4 b, f32x16 c)
{
    if (BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// fp32 -> three 16-bit parts (b = hi + mid + lo to 24 bits): the bias as k slots 0 .. 2 of an A fragment against a B fragment of ones
template <bool BF16>
__device__ __forceinline__ i32x4 bias_frag(float b, bool first_half)
{
    float h0, m0, dummy;
    unpack2<BF16>(pack2<BF16>(b, 0.f), h0, dummy);
    const float r1 = b - h0;
    unpack2<BF16>(pack2<BF16>(r1, 0.f), m0, dummy);
    const float r2 = r1 - m0;
    const unsigned x = pack2<BF16>(h0, m0), y = pack2<BF16>(r2, 0.f);
    return first_half ? i32x4{(int)x, (int)y, 0, 0} : i32x4{0, 0, 0, 0};
}

// NCH = 3, GB (round 6, last): ESDB's c{j}_r (team18_bsrn.py:150-163) -- a BSConvU run as a dense 3x3 over 48 physical channels with 33 .. 48
// outputs (+ input, GELU; the merged pointwise bias's border table added in front of the GELU), plain or with the next distillation Linear + GELU
// behind it (fp16).  27 k steps per pair, all 54 weight fragments in accumulation registers, six 4-channel blocks per pixel (the fourth block
// pair is padding: no operations, no store); the LDS pixel keeps its 10 slots (6 used).
template <bool BF16, bool POST, bool HL, int NCH, bool GB>
__global__ __launch_bounds__(256, 1) void conv64m_kernel(const S16K p)
{
    static_assert(!HL || (BF16 && !POST), "hi + lo residual / output: bf16, no post chain");
    static_assert((NCH == 4 && !GB) || (NCH == 3 && GB && !HL && !(POST && BF16)), "64 channels | ESDB's 48 with the border table and GELU");
    constexpr int RW = M_RW, TAPS = M_TAPS, NG = NCH * TAPS, STAGE = M_STAGE, ROWB = M_ROWB, PIXB = M_PIXB, PPW = M_PPW;
    constexpr int NREG = (POST && NCH == 4) ? 3 * TAPS * 2 : 2 * NG;      // fragments in registers; the first min(NREG, 64) in accumulation registers
    constexpr int SLOTS = 2 + 2 * NG + NCH;                // MFMAs per pair (not HL)
    constexpr int OFF_BT = POST ? M_LDS_POST : M_LDS_PLAIN;               // GB: the border table [16][48] fp32
    constexpr int NAG = NREG < 64 ? NREG : 64;
    constexpr int NVG = NREG - NAG;
    constexpr int SPP = POST ? 6 : (HL ? 8 : 4);           // stores per row pair
    constexpr int S_MAIN = HL ? 10 : 2;                    // slot of the first k step's first MFMA (HL: 2 bias + 8 residual MFMAs in front)
    constexpr unsigned ONE = BF16 ? 0x3f80u : 0x3c00u;
    constexpr bool PLO = BF16;                             // the post 1x1 sees hi + lo activations and weights (bf16), hi only (fp16)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pn = lane & 31, hh = lane >> 5;              // MFMA column (pixel of the row pair) / k half
    const int px = pn & 15, pe = pn >> 4;
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
#ifdef C64M_TRACE4
    const unsigned long long t4_entry = __builtin_readcyclecounter();
#endif

    // ---- prologue: weights.  POST: chunk 3's fragments and the post images go to LDS by DMA; everything else straight into registers
    if (POST) {
        if (NREG < 2 * NG)
            for (int pc = wv; pc < TAPS * 2; pc += 4) dma_glb16(smem_lds + (unsigned)(M_W3 + pc * 1024), p.wm32 + (size_t)(NREG + pc) * 1024 + lane * 16);
        for (int pc = wv; pc < M_POST_IMG / 1024; pc += 4) dma_glb16(smem_lds + (unsigned)(M_OFF_POST + pc * 1024), p.pm32 + (size_t)pc * 1024 + lane * 16);
    }
    if (GB && wv < 3) dma_glb16(smem_lds + (unsigned)(OFF_BT + wv * 1024), reinterpret_cast<const char*>(p.border) + (size_t)wv * 1024 + lane * 16);
    i32x4 wa[NAG];
    i32x4 wx[NVG > 0 ? NVG : 1];
#pragma unroll
    for (int f = 0; f < NAG; ++f) wa[f] = *reinterpret_cast<const i32x4*>(p.wm32 + (size_t)f * 1024 + lane * 16);
#pragma unroll
    for (int f = 0; f < NVG; ++f) wx[f] = *reinterpret_cast<const i32x4*>(p.wm32 + (size_t)(NAG + f) * 1024 + lane * 16);
    // bias fragments (A) against a fragment of ones (B); the residual's selection matrices
    i32x4 a_bias[2], a_id[2], a_pb = {0, 0, 0, 0};
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) a_bias[hf] = bias_frag<BF16>(p.bias[32 * hf + pn], hh == 0);
    if (POST) a_pb = bias_frag<BF16>(p.pbias1[pn], hh == 0);
    const i32x4 b_ones = hh == 0 ? i32x4{(int)(ONE | (ONE << 16)), (int)ONE, 0, 0} : i32x4{0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j0 = pn - 16 * q - 8 * hh;               // k slot of this lane's half that carries output row pn
        const bool on = (HL || p.res_in != 0) && j0 >= 0 && j0 < 8;
        const unsigned v = ONE << ((j0 & 1) * 16);
        a_id[q] = i32x4{(on && (j0 >> 1) == 0) ? (int)v : 0, (on && (j0 >> 1) == 1) ? (int)v : 0, (on && (j0 >> 1) == 2) ? (int)v : 0, (on && (j0 >> 1) == 3) ? (int)v : 0};
    }

    const int ntiles = p.N * p.tiles_y * p.tiles_x;
    auto tile_index = [&](int k) -> int { return s16_tile_index(k, ntiles); };
    auto tile_coords = [&](int t, int& n, int& x0, int& y0) __attribute__((always_inline)) { s16_tile_coords(t, p.magic_x, p.magic_y, p.tiles_x, p.tiles_y, (4 * RW), n, x0, y0); };
    const size_t img_bytes = (size_t)p.H * p.W * p.in_pitch * 2;
    // DMA pieces.  Piece i of this wave covers 64 consecutive 16-byte slots of the stage; slot -> (row, pixel, part) is lane-constant, so the
    // offset of the slot's bytes RELATIVE to the tile's first halo pixel is computed once (rel[i]; pad slots: OOB) and the tile's origin goes
    // into the buffer descriptor (base = the first halo pixel, range = what is left of the image behind it).  What lies outside the image:
    //   * halo rows BELOW it fall outside the descriptor's range: the hardware writes zeros;
    //   * the halo row ABOVE (tiles with y0 == 0), the halo column LEFT (x0 == 0) and columns at / beyond the image's right edge are
    //     lane-constant sets of slots: `edge` holds "row 0" (bit i) and "column 0" (bit 13 + i) of piece i, `lxp` the slots' columns
    //     (5 bits each); once per tile they become `bad` (bit i: piece i's slot is outside), and a piece is v_bfe_i32 + v_or (all ones = out of
    //     range) + s_add into m0 + the load for EVERY tile.  (Until this form the border tiles took a 22-instruction piece with per-lane
    //     compares under exec masks; a block keeps its tile position from image to image when an image is a multiple of 256 tiles, so the 60
    //     border blocks of a 256 x 256 batch -- 2400 cycles per tile slower -- set the launch's time: 180 -> 1xx us, profiles/r06_c64m_*.)
    // (tight pitch, round 6: a 16-byte part that lies behind the pixel -- channels 56..63 of a 56-channel pitch -- is not requested: zero fill,
    // like the stage's pad slots, instead of the next pixel's first bytes)
    unsigned rel[PPW], edge = 0u, lxp[3] = {0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const unsigned sl = (unsigned)((wv + 4 * i) * 64 + lane);
        const unsigned row = sl / (unsigned)M_ROWSL, rem = sl - row * (unsigned)M_ROWSL;
        const unsigned lx = rem / (unsigned)M_LSL, part = rem - lx * (unsigned)M_LSL;
        const bool real = part < (unsigned)(2 * NCH) && (unsigned)p.in_coff + 8u * part < (unsigned)p.in_pitch && lx < (unsigned)M_TH && row < (unsigned)M_THY && (i < PPW - 1 || wv + 4 * i < M_NPIECES);
        rel[i] = real ? (row * (unsigned)p.W + lx) * (unsigned)p.in_pitch * 2u + part * 16u : OOB;
        edge |= (row == 0u ? 1u << i : 0u) | (lx == 0u ? 1u << (13 + i) : 0u);
        lxp[i / 6] |= (lx < 31u ? lx : 31u) << (5 * (i % 6));
    }
    // (m0 is not restored: hipcc keeps nothing in it on gfx950 outside s_set_gpr_idx / s_movrel sequences, and this kernel indexes no register
    // dynamically; dma_buf16 saves it for kernels that might)
    auto dma_piece_fast = [&](auto i_, unsigned bad, i32x4 rsrc, unsigned lds0) __attribute__((always_inline)) {
        constexpr int i = decltype(i_)::value;
        if (i < PPW - 1 || wv + 4 * i < M_NPIECES) {                     // wave-uniform
            const unsigned voff = rel[i] | (unsigned)__builtin_amdgcn_sbfe((int)bad, (unsigned)i, 1u);
            asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds"
                         :: "s"(lds0), "n"(i * 4096), "v"(voff), "s"(rsrc) : "memory");
        }
    };
    // the general form: piece i of this wave of the tile (n, x0, y0) into stage `slot`; nothing valid (behind the last tile): zeros
    auto dma_piece = [&](int i, bool valid, int n, int x0, int y0, int slot) __attribute__((always_inline)) {
        const int pc = wv + 4 * i;
        if (i < PPW - 1 || pc < M_NPIECES) {                             // wave-uniform
            const unsigned sl = (unsigned)(pc * 64 + lane);               // 16-byte slot of the stage: row sl / 181, then pixel / part of the row
            const unsigned row = sl / (unsigned)M_ROWSL, rem = sl - row * (unsigned)M_ROWSL;
            const unsigned lx = rem / (unsigned)M_LSL, part = rem - lx * (unsigned)M_LSL;
            const int gy = y0 - 1 + (int)row, gx = x0 - 1 + (int)lx;
            const bool ok = valid && part < (unsigned)(2 * NCH) && (unsigned)p.in_coff + 8u * part < (unsigned)p.in_pitch && lx < (unsigned)M_TH && row < (unsigned)M_THY && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            const unsigned voff = ok ? (unsigned)((gy * p.W + gx) * p.in_pitch + p.in_coff) * 2u + part * 16u : OOB;
            dma_buf16(smem_lds + (unsigned)(slot * STAGE + pc * 1024), voff, make_rsrc(p.x + (size_t)(valid ? n : 0) * img_bytes, img_bytes), 0u);
        }
    };

    int n, x0, y0;
    {
        const int t0 = tile_index(0);
        if (t0 < 0) return;
        tile_coords(t0, n, x0, y0);
#pragma unroll
        for (int i = 0; i < PPW; ++i) dma_piece(i, true, n, x0, y0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const unsigned b_base = (unsigned)((wv * RW + pe) * ROWB + px * PIXB + hh * 16);       // this lane's B fragments: + stage, + row / tap / chunk immediates
    // (laundered through an empty asm: hipcc otherwise re-adds the > 64 KB constant in front of every read instead of using the immediate offset)
    unsigned w3_off = (unsigned)(M_W3 + lane * 16), img1_off = (unsigned)(M_OFF_POST + lane * 16);
    asm volatile("" : "+v"(w3_off), "+v"(img1_off));
    const char* const w3 = smem + w3_off;
    const char* const img1 = smem + img1_off;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const float slope = p.slope, p1s = p.p1_slope;
    const size_t y_img = (size_t)p.H * p.W * p.y0_pitch * 2;
    const unsigned rowb = (unsigned)p.W * (unsigned)p.y0_pitch * 2u;
    const size_t p1_img = POST ? (size_t)p.H * p.W * p.py1_pitch * 2 : 0;
    const unsigned rowb1 = POST ? (unsigned)p.W * (unsigned)p.py1_pitch * 2u : 0u;

    // HL: the residual is a hi + lo pair of ANOTHER tensor (the long skip: LR_conv's `fea`).  A lane's B operand of the selection MFMA of chunk c
    // is 16 bytes of ITS pixel (channels 16 c + 8 h ..) -- one buffer_load_dwordx4 per chunk and part straight into registers, issued behind k
    // steps 4 .. 11 of the pair's own stream, consumed in front of the next pair's first k step behind an exact s_waitcnt.  (asm loads: hipcc
    // would guard builtin loads with waits that do not count the DMA pieces, i.e. wait for the next tile's pieces as well)
    const size_t r_img = HL ? (size_t)p.H * p.W * p.res_pitch * 2 : 0;
    const unsigned rowbr = HL ? (unsigned)p.W * (unsigned)p.res_pitch * 2u : 0u;
    i32x4 rq[8];                         // [chunk][hi | lo]
#pragma unroll
    for (int i = 0; i < 8; ++i) rq[i] = i32x4{0, 0, 0, 0};
    unsigned r_v = OOB;
    int r_n = 0;
    auto res_offsets = [&](int nn_, int x0_, int y0_) __attribute__((always_inline)) {
        const bool inx = x0_ + px < p.W;
        const unsigned pix = (unsigned)((y0_ + wv * RW + pe) * p.W + x0_ + px);
        r_v = inx ? (pix * (unsigned)p.res_pitch + (unsigned)p.res_coff) * 2u + (unsigned)hh * 16u : OOB;
        r_n = nn_;
    };
    auto load_res = [&](auto rp_, auto i_) __attribute__((always_inline)) {
        constexpr int rp = decltype(rp_)::value, i = decltype(i_)::value, c = i >> 1, part = i & 1;
        // (rows below the image fall outside the descriptor's range: zeros; the part's offset goes into the base -- an soffset is range-checked)
        const i32x4 rr = make_rsrc(p.res + (size_t)r_n * r_img + (part ? (size_t)p.res_lo_stride : (size_t)0), r_img);
        const unsigned voff = r_v + (unsigned)(2 * rp) * rowbr + (unsigned)(c * 32);
        i32x4 t;                         // (asm operands of a lambda must be its own locals; the assignment is a renaming)
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(t) : "v"(voff), "s"(rr) : "memory");
        rq[i] = t;
    };

    f32x16 acc[2][2];                    // [row pair & 1][output half]
    f32x16 d1;                           // the post 1x1's accumulators (32 outputs x the pair's 32 pixels)
#pragma unroll
    for (int j = 0; j < 16; ++j) { d1[j] = 0.f; acc[0][0][j] = 0.f; acc[0][1][j] = 0.f; acc[1][0][j] = 0.f; acc[1][1][j] = 0.f; }
    i32x4 bs[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};     // [block & 1]: the block's four values rounded (x, y) and their low parts (z, w)
    i32x4 pa[2][2];                      // post images of a block: [block & 1][hi, lo]
    uint2 pq[4] = {{0u, 0u}, {0u, 0u}, {0u, 0u}, {0u, 0u}};      // the post result rounded, per 4-channel block
    unsigned bt_o[2] = {(unsigned)OFF_BT, (unsigned)OFF_BT};       // GB: [pair & 1] the lane's row of the border table (+ 16 h bytes)
    f32x4 btv = {0.f, 0.f, 0.f, 0.f};
    float tv0 = 0.f, tv1 = 0.f, tv2 = 0.f, tv3 = 0.f, lv0 = 0.f, lv1 = 0.f, lv2 = 0.f, lv3 = 0.f;
    unsigned e_v[4], e_vP[2];            // store offsets of the tile whose epilogue is in flight: (half, block pair) / post block pair
#pragma unroll
    for (int j = 0; j < 4; ++j) e_v[j] = OOB;
    e_vP[0] = e_vP[1] = OOB;
    int e_n = 0;
    unsigned e_lin = OOB;                // (C64M_ABL & 8)
    auto store_offsets = [&](int nn_, int x0_, int y0_) __attribute__((always_inline)) {
        const bool inx = x0_ + px < p.W;
        const unsigned pix = (unsigned)((y0_ + wv * RW + pe) * p.W + x0_ + px);
        const unsigned base = (pix * (unsigned)p.y0_pitch + (unsigned)p.y0_coff) * 2u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = 16 * j + 8 * hh;               // (half, block pair) j = 2 half + bp: channels 32 half + 16 bp + 8 h .. + 7
            e_v[j] = (inx && ch < p.cout_store && !(C64M_ABL & 1)) ? base + (unsigned)ch * 2u : OOB;
        }
        if (POST) {
            const unsigned base1 = (pix * (unsigned)p.py1_pitch + (unsigned)p.py1_coff) * 2u;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ch = 16 * j + 8 * hh;
                e_vP[j] = (inx && ch < p.p1_cout8 && !(C64M_ABL & 1)) ? base1 + (unsigned)ch * 2u : OOB;
            }
        }
        e_n = nn_;
        if ((C64M_ABL & 8) != 0) e_lin = (unsigned)(((y0_ + wv * RW) * p.W + x0_ * 4) * p.y0_pitch * 2) + (unsigned)lane * 16u;    // the wave's 8 KB of a 16-row band, linear
    };
    auto load_pa = [&](int blk) __attribute__((always_inline)) {
        pa[blk & 1][0] = *reinterpret_cast<const i32x4*>(img1 + (blk * 2) * 1024);
        if (PLO) pa[blk & 1][1] = *reinterpret_cast<const i32x4*>(img1 + (blk * 2 + 1) * 1024);
    };
    // one block (4 channels of the lane's pixel) of the finished pair: step u of its list
    auto block_op = [&](auto par_, auto blk_, auto u_) __attribute__((always_inline)) {
        constexpr int par = decltype(par_)::value, blk = decltype(blk_)::value, u = decltype(u_)::value;
        constexpr int hf = blk >> 2, b = blk & 3, sl = blk & 1;
        f32x16& A = acc[par][hf];
        if constexpr (u == 0) {
            // (GB: channels 32 hf + 8 b + 4 h .. + 3 of the pixel's table row)
            if constexpr (GB) btv = *reinterpret_cast<const f32x4*>(smem + bt_o[par] + hf * 128 + b * 32);
            else { A[4 * b] = act1(A[4 * b], slope); A[4 * b + 1] = act1(A[4 * b + 1], slope); }
            if constexpr (POST && blk == 0) load_pa(0);
        } else if constexpr (u == 1) {
            if constexpr (GB) {
                const f32x4 t = gelu16x4(f32x4{A[4 * b] + btv.x, A[4 * b + 1] + btv.y, A[4 * b + 2] + btv.z, A[4 * b + 3] + btv.w});
                A[4 * b] = t.x; A[4 * b + 1] = t.y; A[4 * b + 2] = t.z; A[4 * b + 3] = t.w;
            } else { A[4 * b + 2] = act1(A[4 * b + 2], slope); A[4 * b + 3] = act1(A[4 * b + 3], slope); }
            if constexpr (POST && blk == 0) d1 = mfma_b<BF16>(a_pb, b_ones, f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
        } else if constexpr (u == 2) {
            bs[sl].x = (int)pack2<BF16>(A[4 * b], A[4 * b + 1]); bs[sl].y = (int)pack2<BF16>(A[4 * b + 2], A[4 * b + 3]);
            if constexpr ((POST || HL) && PLO) unpack2<BF16>((unsigned)bs[sl].x, tv0, tv1);
            if constexpr (POST && !PLO) { bs[sl].z = 0; bs[sl].w = 0; }
        } else if constexpr (u == 3) {
            if constexpr (PLO) {
                unpack2<BF16>((unsigned)bs[sl].y, tv2, tv3);
                lv0 = A[4 * b] - tv0; lv1 = A[4 * b + 1] - tv1;
            }
        } else if constexpr (u == 4) {
            if constexpr (PLO) {
                lv2 = A[4 * b + 2] - tv2; lv3 = A[4 * b + 3] - tv3;
                bs[sl].z = (int)pack2<BF16>(lv0, lv1); bs[sl].w = (int)pack2<BF16>(lv2, lv3);
            }
        } else if constexpr (u == 5) {
            d1 = mfma_b<BF16>(pa[sl][0], bs[sl], d1);
            if constexpr (blk < 2 * NCH - 1) load_pa(blk + 1);
        } else if constexpr (u == 6) {
            if constexpr (PLO) d1 = mfma_b<BF16>(pa[sl][1], bs[sl], d1);
        }
    };
    // the conv's store of (half, block pair) P of the finished pair whose first row is r
    auto store_op = [&](auto P_, auto r_) __attribute__((always_inline)) {
        constexpr int P = decltype(P_)::value, r = decltype(r_)::value;
        const u32x2 s0 = __builtin_amdgcn_permlane32_swap((unsigned)bs[0].x, (unsigned)bs[1].x, false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane32_swap((unsigned)bs[0].y, (unsigned)bs[1].y, false, false);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y0 + (size_t)e_n * y_img, 0, (int)y_img, 0x00020000);
        if constexpr ((C64M_ABL & 8) != 0)
            __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)s0.x, (int)s1.x, (int)s0.y, (int)s1.y}, yr, e_lin + (unsigned)((P * 2 + r / 2) * 1024), 0, 0);
        else
        __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)s0.x, (int)s1.x, (int)s0.y, (int)s1.y}, yr, e_v[P] + (unsigned)r * rowb, 0, 0);
    };
    // HL: the low parts of the same block pair, `hilo_stride` bytes behind (S16K.y1)
    auto store_lo_op = [&](auto P_, auto r_) __attribute__((always_inline)) {
        constexpr int P = decltype(P_)::value, r = decltype(r_)::value;
        const u32x2 s0 = __builtin_amdgcn_permlane32_swap((unsigned)bs[0].z, (unsigned)bs[1].z, false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane32_swap((unsigned)bs[0].w, (unsigned)bs[1].w, false, false);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y1 + (size_t)e_n * y_img, 0, (int)y_img, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)s0.x, (int)s1.x, (int)s0.y, (int)s1.y}, yr, e_v[P] + (unsigned)r * rowb, 0, 0);
    };
    auto post_act_op = [&](auto pb_, auto m_) __attribute__((always_inline)) {
        constexpr int pb = decltype(pb_)::value, m = decltype(m_)::value;
        if constexpr (GB) {                                  // the next distillation Linear's GELU
            if constexpr (m == 0) {
                const f32x4 t = gelu16x4(f32x4{d1[4 * pb], d1[4 * pb + 1], d1[4 * pb + 2], d1[4 * pb + 3]});
                d1[4 * pb] = t.x; d1[4 * pb + 1] = t.y; d1[4 * pb + 2] = t.z; d1[4 * pb + 3] = t.w;
            } else {
                pq[pb].x = pack2<BF16>(d1[4 * pb], d1[4 * pb + 1]); pq[pb].y = pack2<BF16>(d1[4 * pb + 2], d1[4 * pb + 3]);
            }
        } else if constexpr (m == 0) {
            d1[4 * pb] = act1(d1[4 * pb], p1s); d1[4 * pb + 1] = act1(d1[4 * pb + 1], p1s);
        } else {
            d1[4 * pb + 2] = act1(d1[4 * pb + 2], p1s); d1[4 * pb + 3] = act1(d1[4 * pb + 3], p1s);
            pq[pb].x = pack2<BF16>(d1[4 * pb], d1[4 * pb + 1]); pq[pb].y = pack2<BF16>(d1[4 * pb + 2], d1[4 * pb + 3]);
        }
    };
    auto post_store_op = [&](auto j_, auto r_) __attribute__((always_inline)) {
        constexpr int j = decltype(j_)::value, r = decltype(r_)::value;
        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(pq[2 * j].x, pq[2 * j + 1].x, false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(pq[2 * j].y, pq[2 * j + 1].y, false, false);
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(p.py1 + (size_t)e_n * p1_img, 0, (int)p1_img, 0x00020000);
        if constexpr ((C64M_ABL & 8) != 0)
            __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)s0.x, (int)s1.x, (int)s0.y, (int)s1.y}, r1, e_lin / 2 + (unsigned)((j * 2 + r / 2) * 1024), 0, 0);
        else
        __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)s0.x, (int)s1.x, (int)s0.y, (int)s1.y}, r1, e_vP[j] + (unsigned)r * rowb1, 0, 0);
    };
    // operation q of the finished pair's epilogue.  POST: block pairs of 15 operations (7 + 7 + the pair's store), three idle slots while
    // the last post MFMA drains, then the post result (8 + 2).  Plain: 3 + 3 + 1 per block pair.
    auto op = [&](auto par_, auto r_, auto q_) __attribute__((always_inline)) {
        constexpr int q = decltype(q_)::value;
        if constexpr (NCH == 3 && POST) {
            // three block pairs of 4 + 4 operations (table row | + row, GELU | rounding | the post MFMA) + the pair's store, three idle slots
            // while the last post MFMA drains, the post result's three blocks (GELU | rounding), its two stores
            if constexpr (q >= 0 && q < 27) {
                constexpr int P = q / 9, w = q % 9;
                if constexpr (w < 8) block_op(par_, std::integral_constant<int, 2 * P + w / 4>{}, std::integral_constant<int, (w % 4) < 3 ? (w % 4) : 5>{});
                else store_op(std::integral_constant<int, P>{}, r_);
            } else if constexpr (q >= 30 && q < 36) {
                post_act_op(std::integral_constant<int, (q - 30) / 2>{}, std::integral_constant<int, (q - 30) % 2>{});
            } else if constexpr (q == 36 || q == 37) {
                post_store_op(std::integral_constant<int, q - 36>{}, r_);
            }
        } else if constexpr (NCH == 3) {
            if constexpr (q >= 0 && q < 21) {
                constexpr int P = q / 7, w = q % 7;
                if constexpr (w < 3) block_op(par_, std::integral_constant<int, 2 * P>{}, std::integral_constant<int, w>{});
                else if constexpr (w < 6) block_op(par_, std::integral_constant<int, 2 * P + 1>{}, std::integral_constant<int, w - 3>{});
                else store_op(std::integral_constant<int, P>{}, r_);
            }
        } else if constexpr (POST) {
            if constexpr (q >= 0 && q < 60) {
                constexpr int P = q / 15, w = q % 15;
                if constexpr (w < 7) block_op(par_, std::integral_constant<int, 2 * P>{}, std::integral_constant<int, w>{});
                else if constexpr (w < 14) block_op(par_, std::integral_constant<int, 2 * P + 1>{}, std::integral_constant<int, w - 7>{});
                else store_op(std::integral_constant<int, P>{}, r_);
            } else if constexpr (q >= 63 && q < 71) {
                post_act_op(std::integral_constant<int, (q - 63) / 2>{}, std::integral_constant<int, (q - 63) % 2>{});
            } else if constexpr (q == 71 || q == 72) {
                post_store_op(std::integral_constant<int, q - 71>{}, r_);
            }
        } else if constexpr (HL) {
            // 5 + 5 operations per block pair (activation, rounding, the low parts), its two stores
            if constexpr (q >= 0 && q < 48) {
                constexpr int P = q / 12, w = q % 12;
                if constexpr (w < 5) block_op(par_, std::integral_constant<int, 2 * P>{}, std::integral_constant<int, w>{});
                else if constexpr (w < 10) block_op(par_, std::integral_constant<int, 2 * P + 1>{}, std::integral_constant<int, w - 5>{});
                else if constexpr (w == 10) store_op(std::integral_constant<int, P>{}, r_);
                else store_lo_op(std::integral_constant<int, P>{}, r_);
            }
        } else {
            if constexpr (q >= 0 && q < 28) {
                constexpr int P = q / 7, w = q % 7;
                if constexpr (w < 3) block_op(par_, std::integral_constant<int, 2 * P>{}, std::integral_constant<int, w>{});
                else if constexpr (w < 6) block_op(par_, std::integral_constant<int, 2 * P + 1>{}, std::integral_constant<int, w - 3>{});
                else store_op(std::integral_constant<int, P>{}, r_);
            }
        }
    };
    // slot s of a pair (behind its MFMA s, 0 .. 77)
    auto micro = [&](auto par_, auto r_, auto s_) __attribute__((always_inline)) {
        constexpr int s = decltype(s_)::value;
        if constexpr ((C64M_ABL & 4) != 0) {
        } else if constexpr (POST) {
            if constexpr (s >= 4) op(par_, r_, std::integral_constant<int, s - 4>{});
        } else if constexpr (HL) {
            if constexpr (s >= 10) op(par_, r_, std::integral_constant<int, s - 10>{});        // (behind the residual's MFMAs)
        } else {
            if constexpr (s >= 4 && ((s - 4) & 1) == 0) op(par_, r_, std::integral_constant<int, (s - 4) / 2>{});
        }
    };

#ifdef C64M_TRACE4
    const unsigned long long t4_start = __builtin_readcyclecounter();
    unsigned long long t4_wait = 0, t4_tiles = 0;
#endif
    for (int k = 0;; ++k) {
        const int tn = tile_index(k + 1);
        const bool more = tn >= 0;
        int nn = 0, nx0 = 0, ny0 = 0;
        if (more) tile_coords(tn, nn, nx0, ny0);
        const char* const bb = smem + b_base + (unsigned)((k & 1) * STAGE);
        // the next tile's DMA: origin and range into the descriptor, the slots outside the image into `bad` (nothing follows: every slot)
        // (C64M_ABL & 16: the DMA re-reads the CURRENT tile -- L2 hits -- to tell memory latency / bandwidth from issue cost)
        const int dx0 = (C64M_ABL & 16) ? x0 : nx0, dy0 = (C64M_ABL & 16) ? y0 : ny0, dn = (C64M_ABL & 16) ? n : nn;
        const unsigned tbase = (unsigned)(((dy0 - 1) * p.W + (dx0 - 1)) * p.in_pitch + p.in_coff) * 2u;
        const i32x4 nrsrc = make_rsrc(p.x + (size_t)dn * img_bytes + (size_t)(int)tbase, img_bytes - (size_t)(int)tbase);
        unsigned bad;
        {
            const unsigned sel = (dy0 == 0 ? 0x1fffu : 0u) | (dx0 == 0 ? 0x1fffu << 13 : 0u);
            const unsigned t = edge & sel;
            bad = more ? ((t | (t >> 13)) & 0x1fffu) : 0xffffffffu;
            if (dx0 + TILE + 1 > p.W) {                                   // the tile's right halo column (or more) is outside: per-slot column test
                const unsigned lim = (unsigned)(p.W - dx0 + 1);
#pragma unroll
                for (int i = 0; i < PPW; ++i) bad |= (((lxp[i / 6] >> (5 * (i % 6))) & 31u) >= lim ? 1u : 0u) << i;
            }
        }
        const unsigned lds0 = smem_lds + (unsigned)(((k + 1) & 1) * STAGE + wv * 1024);
        // B fragments: a ring of four, read THREE k steps ahead of their MFMAs; chunk 3's A fragments (POST): a ring of three pairs, read
        // TWO steps ahead.  Linear step index L = 36 rp + g over the tile's 72 steps
        constexpr int AHEAD = 3;
        i32x4 b[4];
        i32x4 a3[3][2];
        auto read_b = [&](auto L_) __attribute__((always_inline)) {
            constexpr int L = decltype(L_)::value;
            constexpr int rp_ = L / NG, g_ = L % NG, c_ = g_ / TAPS, t_ = g_ % TAPS;
            b[L & 3] = *reinterpret_cast<const i32x4*>(bb + (2 * rp_ + t_ / 3) * ROWB + (t_ % 3) * PIXB + c_ * 32);
        };
        auto read_a = [&](auto L_) __attribute__((always_inline)) {
            constexpr int L = decltype(L_)::value;
            constexpr int g_ = L % NG;
            if constexpr (2 * g_ >= NREG) {
                a3[L % 3][0] = *reinterpret_cast<const i32x4*>(w3 + (2 * g_ - NREG) * 1024);
                a3[L % 3][1] = *reinterpret_cast<const i32x4*>(w3 + (2 * g_ + 1 - NREG) * 1024);
            }
        };
        static_for<AHEAD>([&](auto L_) __attribute__((always_inline)) { read_b(L_); });
        auto run_pair = [&](auto rp_tag) __attribute__((always_inline)) {
            constexpr int rp = decltype(rp_tag)::value;
            constexpr int par = rp & 1;
            using PrevPar = std::integral_constant<int, par ^ 1>;
            using PrevRow = std::integral_constant<int, (rp == 0 ? RW - 2 : 2 * rp - 2)>;
            if constexpr (rp == 1) store_offsets(n, x0, y0);       // behind the carried epilogue's last store, ahead of this tile's first
            if constexpr (GB) {                                    // this pair's table row (read by its epilogue, a pair later)
                const int gx = x0 + px, gy = y0 + wv * RW + 2 * rp + pe;
                const int m = (gx == 0 ? 1 : 0) | (gx == p.W - 1 ? 2 : 0) | (gy == 0 ? 4 : 0) | (gy == p.H - 1 ? 8 : 0);
                bt_o[par] = (unsigned)(OFF_BT + m * 192 + hh * 16);
            }
            // slots 0, 1: the bias (the finished pair's accumulators are the other set)
            mfma_m0<BF16>(acc[par][0], a_bias[0], b_ones);
            micro(PrevPar{}, PrevRow{}, std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            mfma_m0<BF16>(acc[par][1], a_bias[1], b_ones);
            micro(PrevPar{}, PrevRow{}, std::integral_constant<int, 1>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (HL) {
                // slots 2 .. 9: the FINISHED pair's residual (chunk c, hi | lo) onto its output channels 16 c .. 16 c + 15.  Younger than its last
                // load (behind k step 11 of its stream) are that stream's stores behind slot 33 (4) and, in a tile's first pair, DMA pieces 4 .. 11
                i32x4 q0 = rq[0], q1 = rq[1], q2 = rq[2], q3 = rq[3], q4 = rq[4], q5 = rq[5], q6 = rq[6], q7 = rq[7];
                asm volatile("s_waitcnt vmcnt(%8)" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "n"(rp == 1 ? 12 : 4) : "memory");
                rq[0] = q0; rq[1] = q1; rq[2] = q2; rq[3] = q3; rq[4] = q4; rq[5] = q5; rq[6] = q6; rq[7] = q7;
                static_for<8>([&](auto i_) __attribute__((always_inline)) {
                    constexpr int i = decltype(i_)::value, c = i >> 1;
                    mfma_m<BF16, false>(acc[par ^ 1][c >> 1], a_id[c & 1], rq[i]);
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (rp == 0) res_offsets(n, x0, y0);         // (this tile's pixels: both pairs' loads)
            }
            static_for<NG>([&](auto g_) __attribute__((always_inline)) {
                constexpr int g = decltype(g_)::value;
                constexpr int c = g / TAPS, t = g % TAPS, L = rp * NG + g, cs = L & 3;
                constexpr int s0 = HL ? S_MAIN + 2 * g : 2 + 2 * g + (g > 4) + (g > 13) + (g > 22) + (g > 31);
                if constexpr (L + AHEAD < 2 * NG) read_b(std::integral_constant<int, L + AHEAD>{});
                if constexpr (L + 2 < 2 * NG) read_a(std::integral_constant<int, L + 2>{});
                __builtin_amdgcn_sched_barrier(0);
                static_for<2>([&](auto hf_) __attribute__((always_inline)) {
                    constexpr int hf = decltype(hf_)::value, f = 2 * g + hf;
                    if constexpr (f < NAG) mfma_m<BF16, true>(acc[par][hf], wa[f], b[cs]);
                    else if constexpr (f < NREG) mfma_m<BF16, false>(acc[par][hf], wx[f - NAG < 0 ? 0 : f - NAG], b[cs]);
                    else mfma_m<BF16, false>(acc[par][hf], a3[L % 3][hf], b[cs]);
                    micro(PrevPar{}, PrevRow{}, std::integral_constant<int, s0 + hf>{});
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (HL && g >= 4 && g < 12) load_res(rp_tag, std::integral_constant<int, g - 4>{});
                if constexpr (t == 4 && !HL) {
                    // the residual == input: this chunk's 16 channels of the centre pixel onto output channels 16 c .. 16 c + 15
                    mfma_m<BF16, false>(acc[par][c >> 1], a_id[c & 1], b[cs]);
                    micro(PrevPar{}, PrevRow{}, std::integral_constant<int, s0 + 2>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (L % C64M_SPREAD == 0 && L / C64M_SPREAD < PPW && !(C64M_ABL & 2)) {       // the next tile's DMA, in the shadow of the matrix pipe
                    dma_piece_fast(std::integral_constant<int, L / C64M_SPREAD>{}, bad, nrsrc, lds0);
                }
            });
        };
        run_pair(std::integral_constant<int, 0>{});
        run_pair(std::integral_constant<int, 1>{});
        // the next tile has landed: younger than the wave's last DMA piece are exactly m_tail_stores() stores
#ifdef C64M_TRACE4
        const unsigned long long t4_a = __builtin_readcyclecounter();
#endif
        // (HL: the second pair's 8 residual loads and 8 stores)
        if constexpr ((C64M_ABL & 32) == 0)
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(HL ? 16 : (NCH == 3 ? m3_tail_stores(POST, C64M_SPREAD) : m_tail_stores(POST, C64M_SPREAD))) : "memory");      // (C64M_ABL & 32: no wait -- timing only)
        if constexpr ((C64M_ABL & 64) == 0) __builtin_amdgcn_s_barrier();                                   // (C64M_ABL & 64: no barrier -- timing only)
#ifdef C64M_TRACE4
        t4_wait += __builtin_readcyclecounter() - t4_a;
        t4_tiles += 1;
#endif
        if (!more) break;
        n = nn; x0 = nx0; y0 = ny0;
    }
#ifdef C64M_TRACE4
    if (tid == 0 && blockIdx.x < 256) {
        unsigned long long* o = c64m_trace4[POST ? 1 : 0][blockIdx.x];
        o[0] = __builtin_readcyclecounter() - t4_start; o[1] = t4_wait; o[2] = t4_tiles; o[3] = __builtin_readcyclecounter() - t4_entry;
    }
#endif
    // the last tile's last row pair: the same operations, back to back
    // (asm MFMAs: hipcc does not pad MFMA -> VALU reads of their results.  The wait states carry the accumulators as operands: a volatile asm with
    // a memory clobber orders nothing that lives in registers, and hipcc did hoist the reads above it in rfdb_tail_kernel<fp16>)
    if constexpr (HL) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rq[0]), "+v"(rq[1]), "+v"(rq[2]), "+v"(rq[3]), "+v"(rq[4]), "+v"(rq[5]), "+v"(rq[6]), "+v"(rq[7]) :: "memory");
        static_for<8>([&](auto i_) __attribute__((always_inline)) {
            constexpr int i = decltype(i_)::value, c = i >> 1;
            mfma_m<BF16, false>(acc[1][c >> 1], a_id[c & 1], rq[i]);
        });
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc[1][0]), "+v"(acc[1][1]) :: "memory");
    static_for<(HL ? M_SLOTS : SLOTS)>([&](auto s_) __attribute__((always_inline)) {
        micro(std::integral_constant<int, 1>{}, std::integral_constant<int, RW - 2>{}, s_);
        __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the trailing zero-fill DMA must not outlive the block)
}

template <bool BF16, bool POST, bool HL, int NCH, bool GB>
int launch_conv64m(const S16K& k, hipStream_t st)
{
    constexpr int LDS = (POST ? M_LDS_POST : M_LDS_PLAIN) + (GB ? 3072 : 0);
    static_assert(LDS <= LDS_LIMIT, "LDS map");
    static std::atomic<unsigned> attr_set[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return ESR_ERR_LAUNCH;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv64m_kernel<BF16, POST, HL, NCH, GB>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            esr_set_err("hipFuncSetAttribute(conv64m_kernel, MaxDynamicSharedMemorySize)", e);
            return ESR_ERR_LAUNCH;
        }
        attr_set[dev].store(1u, std::memory_order_relaxed);
    }
    const int ntiles = k.N * k.tiles_x * k.tiles_y;
    const int grid = ntiles < 256 ? ntiles : 256;
    // (the symbol as rocprofv3 prints it -- every template argument, defaulted ones included: tools/pmc_traffic.py joins on it)
    esr_note_kernel("conv64m_kernel<%s, %s, %s, %d, %s>", esr_tf(BF16), esr_tf(POST), esr_tf(HL), NCH, esr_tf(GB));
    hipLaunchKernelGGL((conv64m_kernel<BF16, POST, HL, NCH, GB>), dim3(grid), dim3(256), LDS, st, k);
    return esr_check_launch("conv64m_kernel launch");
}


// ---- rfdb_tail_kernel: RFDB's c4 -> cat(d1, d2, d3, r4) -> c5 -> esa.conv1 in ONE launch (round 6, ABI v12; rfdn_baseline/block.py:161-164, :117) -------
//   r4  = round16(act(conv3x3_c4(r3) + b4))          32 physical channels, never stored -- the tensor the separate launches kept in `cat`
//   v   = W5 . [d1 | d2 | d3 | r4] + b5               fp32; stored rounded (64 physical channels)
//   c1  = round16(Wc1 . v + bc1)                      on the UNROUNDED v as hi | lo B operands (bf16; fp16: the rounded v), 16 channels
// conv64m_kernel's frame: 16 x 16 tiles, a wave = two row pairs, the c4 convolution as the MAIN stream (one output half: 1 bias + 36 MFMAs per
// pair, all 36 weight fragments in accumulation registers), everything else as the finished pair's epilogue behind it:
//   * d1 .. d3 never pass through LDS: a lane's B operand of k step (segment s, half-segment u) is 16 bytes of ITS pixel -- one
//     buffer_load_dwordx4 per step straight into registers, issued at the top of the pair's own main stream (two register sets), consumed a
//     pair later behind an exact s_waitcnt;
//   * r4 goes from the D fragment to the B operand inside the lane (two 4-channel blocks per k step; the packer orders c5's rows to match);
//   * c5's weights as hi + lo (conv_s16_kernel's 1x1 form): the 16 high fragments in accumulation registers, the low ones in LDS next to
//     esa.conv1's images; v's hi | lo split, conv1's MFMAs, swaps and stores as in conv64m_kernel<.., POST>.
// 87 MFMAs per pair (37 + 34 + 16; fp16: 63), five stores, six loads; the separate launches moved 608 bytes per pixel, this one 480.
constexpr int T_OFF_W5LO = 2 * M_STAGE;                   // c5's low-part fragments [ks][half]: 16 KB
constexpr int T_OFF_C1 = T_OFF_W5LO + 16 * 1024;          // esa.conv1's images [step][hi | lo]: 16 KB
constexpr int T_OFF_BT = T_OFF_C1 + M_POST_IMG;           // GB: c4's border-bias table [16][32] fp32: 2 KB
constexpr int T_LDS = T_OFF_BT + 2048;
constexpr int T_FIRST = 8;                                // the next tile's DMA piece i behind k step T_FIRST + t_spread(NCH) * i of the FIRST pair
constexpr int t_spread(int nch) { return nch == 4 ? 2 : 1; }
constexpr int t_ops(int nch) { return nch == 4 ? 4 : 6; }                 // epilogue operations per k step (36 | 27 k steps per pair)
static_assert(T_LDS <= LDS_LIMIT && T_FIRST + t_spread(4) * (M_PPW - 1) < 4 * M_TAPS && T_FIRST + t_spread(3) * (M_PPW - 1) < 3 * M_TAPS, "LDS map / the pieces fit the first pair");
// the epilogue's operation list (index q); see `top` in the kernel
constexpr int TQ_WAIT = 0, TQ_LOAD = 1, TQ_R4 = 7, TQ_B5 = 19, TQ_D5 = 21, TQ_R5 = 51, TQ_GAP1 = 61, TQ_V = 73, TQ_GAP2 = 125, TQ_C1 = 133, TQ_END = 136;
static_assert(TQ_END <= t_ops(4) * 4 * M_TAPS && TQ_END <= t_ops(3) * 3 * M_TAPS, "the epilogue fits the main stream's k steps");
// stores: v's four at the end of every second block of the V phase, conv1's at TQ_C1 + 2
constexpr int t_store_q(int i) { return i < 4 ? TQ_V + 13 * i + 12 : TQ_C1 + 2; }
constexpr int t_stores_behind_step(int g, int tops) { int n = 0; for (int i = 0; i < 5; ++i) n += t_store_q(i) / tops > g; return n; }

// NCH = 3, GB (round 6, last): ESDB's tail (team18_bsrn.py:165-171) -- c4 is a BSConvU run as a dense 3x3 over 48 channels (three chunks: 27 k
// steps per pair, six epilogue operations behind each; the LDS pixel keeps its 10 slots, 6 used) with the border-bias table of the merged
// BSConvU added to r4 in front of its GELU (the table in LDS, row = which sides of the pixel lie outside, row 0 = zeros: no branch).
template <bool BF16, int NCH, bool GB>
__global__ __launch_bounds__(256, 1) void rfdb_tail_kernel(const S16K p)
{
    static_assert((NCH == 4 && !GB) || (NCH == 3 && GB), "RFDB's tail | ESDB's tail");
    constexpr int RW = M_RW, TAPS = M_TAPS, NG = NCH * TAPS, STAGE = M_STAGE, ROWB = M_ROWB, PIXB = M_PIXB, PPW = M_PPW;
    constexpr int T_OPS = t_ops(NCH), T_SPREAD = t_spread(NCH);
    constexpr unsigned ONE = BF16 ? 0x3f80u : 0x3c00u;
    constexpr bool PLO = BF16;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pn = lane & 31, hh = lane >> 5;
    const int px = pn & 15, pe = pn >> 4;
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- prologue: c5's low parts and conv1's images to LDS; c4's 36 fragments and c5's 16 high fragments into accumulation registers
    for (int pc = wv; pc < 16; pc += 4) {
        const int ks = pc >> 1, half = pc & 1;
        dma_glb16(smem_lds + (unsigned)(T_OFF_W5LO + pc * 1024), p.tw + (size_t)((ks * 2 + half) * 2 + 1) * 1024 + lane * 16);
    }
    for (int pc = wv; pc < M_POST_IMG / 1024; pc += 4) dma_glb16(smem_lds + (unsigned)(T_OFF_C1 + pc * 1024), p.pm32 + (size_t)pc * 1024 + lane * 16);
    if (GB && wv < 2) dma_glb16(smem_lds + (unsigned)(T_OFF_BT + wv * 1024), reinterpret_cast<const char*>(p.border) + (size_t)wv * 1024 + lane * 16);
    i32x4 wa4[NG], w5h[16];
#pragma unroll
    for (int f = 0; f < NG; ++f) wa4[f] = *reinterpret_cast<const i32x4*>(p.wm32 + (size_t)f * 1024 + lane * 16);
#pragma unroll
    for (int f = 0; f < 16; ++f) w5h[f] = *reinterpret_cast<const i32x4*>(p.tw + (size_t)(f * 2) * 1024 + lane * 16);
    const i32x4 a_b4 = bias_frag<BF16>(p.bias[pn], hh == 0);
    i32x4 a_b5[2];
    const float* const b5 = reinterpret_cast<const float*>(p.tw + (size_t)32 * 1024);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) a_b5[hf] = bias_frag<BF16>(b5[32 * hf + pn], hh == 0);
    const i32x4 a_bc = bias_frag<BF16>(p.pbias1[pn], hh == 0);
    const i32x4 b_ones = hh == 0 ? i32x4{(int)(ONE | (ONE << 16)), (int)ONE, 0, 0} : i32x4{0, 0, 0, 0};

    const int ntiles = p.N * p.tiles_y * p.tiles_x;
    auto tile_index = [&](int k) -> int { return s16_tile_index(k, ntiles); };
    auto tile_coords = [&](int t, int& n, int& x0, int& y0) __attribute__((always_inline)) { s16_tile_coords(t, p.magic_x, p.magic_y, p.tiles_x, p.tiles_y, (4 * RW), n, x0, y0); };
    const size_t img_bytes = (size_t)p.H * p.W * p.in_pitch * 2;
    // DMA pieces of the input halo tile: conv64m_kernel's (lane-constant offsets and edge masks, the origin in the descriptor)
    unsigned rel[PPW], edge = 0u, lxp[3] = {0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const unsigned sl = (unsigned)((wv + 4 * i) * 64 + lane);
        const unsigned row = sl / (unsigned)M_ROWSL, rem = sl - row * (unsigned)M_ROWSL;
        const unsigned lx = rem / (unsigned)M_LSL, part = rem - lx * (unsigned)M_LSL;
        const bool real = part < (unsigned)(2 * NCH) && (unsigned)p.in_coff + 8u * part < (unsigned)p.in_pitch && lx < (unsigned)M_TH && row < (unsigned)M_THY && (i < PPW - 1 || wv + 4 * i < M_NPIECES);
        rel[i] = real ? (row * (unsigned)p.W + lx) * (unsigned)p.in_pitch * 2u + part * 16u : OOB;
        edge |= (row == 0u ? 1u << i : 0u) | (lx == 0u ? 1u << (13 + i) : 0u);
        lxp[i / 6] |= (lx < 31u ? lx : 31u) << (5 * (i % 6));
    }
    auto dma_piece_fast = [&](auto i_, unsigned bad, i32x4 rsrc, unsigned lds0) __attribute__((always_inline)) {
        constexpr int i = decltype(i_)::value;
        if (i < PPW - 1 || wv + 4 * i < M_NPIECES) {
            const unsigned voff = rel[i] | (unsigned)__builtin_amdgcn_sbfe((int)bad, (unsigned)i, 1u);
            asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds"
                         :: "s"(lds0), "n"(i * 4096), "v"(voff), "s"(rsrc) : "memory");
        }
    };
    auto dma_piece = [&](int i, int n, int x0, int y0) __attribute__((always_inline)) {          // the first tile, into stage 0
        const int pc = wv + 4 * i;
        if (i < PPW - 1 || pc < M_NPIECES) {
            const unsigned sl = (unsigned)(pc * 64 + lane);
            const unsigned row = sl / (unsigned)M_ROWSL, rem = sl - row * (unsigned)M_ROWSL;
            const unsigned lx = rem / (unsigned)M_LSL, part = rem - lx * (unsigned)M_LSL;
            const int gy = y0 - 1 + (int)row, gx = x0 - 1 + (int)lx;
            const bool ok = part < (unsigned)(2 * NCH) && (unsigned)p.in_coff + 8u * part < (unsigned)p.in_pitch && lx < (unsigned)M_TH && row < (unsigned)M_THY && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            const unsigned voff = ok ? (unsigned)((gy * p.W + gx) * p.in_pitch + p.in_coff) * 2u + part * 16u : OOB;
            dma_buf16(smem_lds + (unsigned)(pc * 1024), voff, make_rsrc(p.x + (size_t)n * img_bytes, img_bytes), 0u);
        }
    };

    int n, x0, y0;
    {
        const int t0 = tile_index(0);
        if (t0 < 0) return;
        tile_coords(t0, n, x0, y0);
#pragma unroll
        for (int i = 0; i < PPW; ++i) dma_piece(i, n, x0, y0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const unsigned b_base = (unsigned)((wv * RW + pe) * ROWB + px * PIXB + hh * 16);
    unsigned w5lo_off = (unsigned)(T_OFF_W5LO + lane * 16), c1img_off = (unsigned)(T_OFF_C1 + lane * 16);
    asm volatile("" : "+v"(w5lo_off), "+v"(c1img_off));
    const char* const w5lo = smem + w5lo_off;
    const char* const c1img = smem + c1img_off;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const float slope = p.slope, p1s = p.p1_slope;
    const size_t y_img = (size_t)p.H * p.W * p.y0_pitch * 2, p1_img = (size_t)p.H * p.W * p.py1_pitch * 2, cat_img = (size_t)p.H * p.W * p.cat_pitch * 2;
    const unsigned rowb = (unsigned)p.W * (unsigned)p.y0_pitch * 2u, rowb1 = (unsigned)p.W * (unsigned)p.py1_pitch * 2u, rowbc = (unsigned)p.W * (unsigned)p.cat_pitch * 2u;

    f32x16 acc4[2], acc5[2], d1;
#pragma unroll
    for (int j = 0; j < 16; ++j) { acc4[0][j] = 0.f; acc4[1][j] = 0.f; acc5[0][j] = 0.f; acc5[1][j] = 0.f; d1[j] = 0.f; }
    i32x4 dq[2][6];                      // [pair & 1][k step]: the lane's 16 bytes of d1 .. d3 (two half-segments each)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b_ = 0; b_ < 6; ++b_) dq[a][b_] = i32x4{0, 0, 0, 0};
    i32x4 rbv[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};     // r4 rounded: k steps 6, 7 of c5
    unsigned bt_o[2] = {(unsigned)T_OFF_BT, (unsigned)T_OFF_BT};       // GB: [pair & 1] the lane's row of the border table (+ 16 h bytes)
    f32x4 btv = {0.f, 0.f, 0.f, 0.f};
    i32x4 bs[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};      // v's blocks: rounded (x, y) | low parts (z, w)
    i32x4 lo5[2], pa[2][2];
    uint2 pq[2];
    float tv0 = 0.f, tv1 = 0.f, tv2 = 0.f, tv3 = 0.f, lv0 = 0.f, lv1 = 0.f, lv2 = 0.f, lv3 = 0.f;
    unsigned e_v[4], e_vP = OOB, d_v = OOB;
#pragma unroll
    for (int j = 0; j < 4; ++j) e_v[j] = OOB;
    int e_n = 0, d_n = 0;
    auto store_offsets = [&](int nn_, int x0_, int y0_) __attribute__((always_inline)) {
        const bool inx = x0_ + px < p.W;
        const unsigned pix = (unsigned)((y0_ + wv * RW + pe) * p.W + x0_ + px);
        const unsigned base = (pix * (unsigned)p.y0_pitch + (unsigned)p.y0_coff) * 2u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = 16 * j + 8 * hh;
            e_v[j] = (inx && ch < p.cout_store) ? base + (unsigned)ch * 2u : OOB;
        }
        e_vP = (inx && 8 * hh < p.p1_cout8) ? (pix * (unsigned)p.py1_pitch + (unsigned)p.py1_coff) * 2u + (unsigned)(8 * hh) * 2u : OOB;
        e_n = nn_;
    };
    auto load_offsets = [&](int nn_, int x0_, int y0_) __attribute__((always_inline)) {       // d1 .. d3 of the lane's pixel (first pair's row)
        const bool inx = x0_ + px < p.W;
        const unsigned pix = (unsigned)((y0_ + wv * RW + pe) * p.W + x0_ + px);
        d_v = inx ? (pix * (unsigned)p.cat_pitch + (unsigned)p.cat_coff) * 2u + (unsigned)hh * 16u : OOB;
        d_n = nn_;
    };
    auto load_d = [&](auto rp_, auto ks_) __attribute__((always_inline)) {
        constexpr int rp = decltype(rp_)::value, ks = decltype(ks_)::value, sg = ks >> 1, u = ks & 1;
        // (the segment's offset goes into the descriptor's base: an soffset takes part in the range check on this part -- segments 1 / 2 read zeros)
        const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.cat) + (size_t)d_n * cat_img + (size_t)sg * (size_t)p.cat_seg_stride, 0, (int)cat_img, 0x00020000);
        dq[rp & 1][ks] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(cr, d_v + (unsigned)(2 * rp) * rowbc + (unsigned)(u * 32), 0, 0));
    };
    auto load_lo5 = [&](int ks) __attribute__((always_inline)) {
        lo5[0] = *reinterpret_cast<const i32x4*>(w5lo + (ks * 2) * 1024);
        lo5[1] = *reinterpret_cast<const i32x4*>(w5lo + (ks * 2 + 1) * 1024);
    };
    auto load_pa = [&](int blk) __attribute__((always_inline)) {
        pa[blk & 1][0] = *reinterpret_cast<const i32x4*>(c1img + (blk * 2) * 1024);
        if (PLO) pa[blk & 1][1] = *reinterpret_cast<const i32x4*>(c1img + (blk * 2 + 1) * 1024);
    };
    // operation q of the finished pair's epilogue (par: its c4 accumulators and d registers; r: its first row; FLUSH: behind the last tile)
    auto top = [&](auto par_, auto r_, auto q_, auto flush_) __attribute__((always_inline)) {
        constexpr int par = decltype(par_)::value, r = decltype(r_)::value, q = decltype(q_)::value;
        constexpr bool FLUSH = decltype(flush_)::value;
        if constexpr (q >= TQ_R4 && q < TQ_B5) {                                   // r4: activation, rounding -> rbv
            constexpr int b = (q - TQ_R4) / 3, m = (q - TQ_R4) % 3;
            f32x16& A = acc4[par];
            if constexpr (GB) {
                // block b = channels 8 b + 4 h .. + 3 of the lane's pixel: + the table row, GELU (the polynomial of the other 16-bit kernels)
                if constexpr (m == 0) btv = *reinterpret_cast<const f32x4*>(smem + bt_o[par] + b * 32);
                else if constexpr (m == 1) {
                    const f32x4 t = gelu16x4(f32x4{A[4 * b] + btv.x, A[4 * b + 1] + btv.y, A[4 * b + 2] + btv.z, A[4 * b + 3] + btv.w});
                    A[4 * b] = t.x; A[4 * b + 1] = t.y; A[4 * b + 2] = t.z; A[4 * b + 3] = t.w;
                }
            } else if constexpr (m == 0) { A[4 * b] = act1(A[4 * b], slope); A[4 * b + 1] = act1(A[4 * b + 1], slope); }
            else if constexpr (m == 1) { A[4 * b + 2] = act1(A[4 * b + 2], slope); A[4 * b + 3] = act1(A[4 * b + 3], slope); }
            if constexpr (m == 2) {
                const unsigned x = pack2<BF16>(A[4 * b], A[4 * b + 1]), y = pack2<BF16>(A[4 * b + 2], A[4 * b + 3]);
                if constexpr ((b & 1) == 0) { rbv[b >> 1].x = (int)x; rbv[b >> 1].y = (int)y; }
                else { rbv[b >> 1].z = (int)x; rbv[b >> 1].w = (int)y; }
            }
        } else if constexpr (q >= TQ_B5 && q < TQ_D5) {                            // c5's bias
            // (hipcc gives acc5 the registers of the c4 accumulator that has just been packed.  The fp16 flush once had
            // `v_cvt_pk_f16_f32 v39, v2, v3` directly in front of `v_mfma .. v[0:15], .., .., 0` and packed the MFMA's result in every lane;
            // the sequence alone is clean -- tools/r06/mfma_war_probe.hip -- so the cause is not pinned; with two wait states here and the
            // scheduling barriers between the flush's operations the kernel is right)
            asm volatile("s_nop 1" : "+v"(rbv[1]));
            mfma_m0<BF16>(acc5[q - TQ_B5], a_b5[q - TQ_B5], b_ones);
        } else if constexpr (q >= TQ_D5 && q < TQ_GAP1) {                          // c5: k steps 0 .. 5 on d1 .. d3, 6 / 7 on r4
            constexpr int ks = (q - TQ_D5) / 5, m = (q - TQ_D5) % 5;
            const i32x4& B = ks < 6 ? dq[par][ks < 6 ? ks : 0] : rbv[ks >= 6 ? ks - 6 : 0];
            // (w5 = hi + lo in fp16 as well: the 1x1 launch this replaces carries the residual in its second tap slot)
            if constexpr (m == 0) load_lo5(ks);
            else if constexpr (m == 1) mfma_m<BF16, true>(acc5[0], w5h[2 * ks], B);
            else if constexpr (m == 2) mfma_m<BF16, true>(acc5[1], w5h[2 * ks + 1], B);
            else if constexpr (m == 3) mfma_m<BF16, false>(acc5[0], lo5[0], B);
            else mfma_m<BF16, false>(acc5[1], lo5[1], B);
        } else if constexpr ((q >= TQ_GAP1 && q < TQ_V) || (q >= TQ_GAP2 && q < TQ_C1)) {
            // (asm MFMAs: hipcc pads no read of their results; in the stream the main MFMAs of three k steps lie in between)
            if constexpr (FLUSH && q == TQ_GAP1) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc5[0]), "+v"(acc5[1]) :: "memory");
            if constexpr (q == TQ_GAP1 + 1) load_pa(0);
            if constexpr (q == TQ_GAP1 + 2) d1 = mfma_b<BF16>(a_bc, b_ones, f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
        } else if constexpr (q >= TQ_V && q < TQ_GAP2) {                           // v: 8 blocks; 13 operations per block pair
            constexpr int P = (q - TQ_V) / 13, w = (q - TQ_V) % 13;
            if constexpr (w < 12) {
                constexpr int blk = 2 * P + w / 6, u = w % 6, hf = blk >> 2, b = blk & 3, sl = blk & 1;
                f32x16& A = acc5[hf];
                if constexpr (u == 0) {
                    bs[sl].x = (int)pack2<BF16>(A[4 * b], A[4 * b + 1]); bs[sl].y = (int)pack2<BF16>(A[4 * b + 2], A[4 * b + 3]);
                    if constexpr (PLO) unpack2<BF16>((unsigned)bs[sl].x, tv0, tv1);
                    else { bs[sl].z = 0; bs[sl].w = 0; }
                } else if constexpr (u == 1) {
                    if constexpr (PLO) { unpack2<BF16>((unsigned)bs[sl].y, tv2, tv3); lv0 = A[4 * b] - tv0; lv1 = A[4 * b + 1] - tv1; }
                } else if constexpr (u == 2) {
                    if constexpr (PLO) {
                        lv2 = A[4 * b + 2] - tv2; lv3 = A[4 * b + 3] - tv3;
                        bs[sl].z = (int)pack2<BF16>(lv0, lv1); bs[sl].w = (int)pack2<BF16>(lv2, lv3);
                    }
                } else if constexpr (u == 3) {
                    d1 = mfma_b<BF16>(pa[sl][0], bs[sl], d1);
                    if constexpr (blk < 7) load_pa(blk + 1);
                } else if constexpr (u == 4) {
                    if constexpr (PLO) d1 = mfma_b<BF16>(pa[sl][1], bs[sl], d1);
                }
            } else {                                                               // the pair's store: channels 16 P + 8 h .. + 7
                const u32x2 s0 = __builtin_amdgcn_permlane32_swap((unsigned)bs[0].x, (unsigned)bs[1].x, false, false);
                const u32x2 s1 = __builtin_amdgcn_permlane32_swap((unsigned)bs[0].y, (unsigned)bs[1].y, false, false);
                const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y0 + (size_t)e_n * y_img, 0, (int)y_img, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)s0.x, (int)s1.x, (int)s0.y, (int)s1.y}, yr, e_v[P] + (unsigned)r * rowb, 0, 0);
            }
        } else if constexpr (q >= TQ_C1 && q < TQ_END) {                           // esa.conv1's 16 channels: blocks 0, 1 of d1
            constexpr int m = q - TQ_C1;
            if constexpr (m < 2) {
                d1[4 * m] = act1(d1[4 * m], p1s); d1[4 * m + 1] = act1(d1[4 * m + 1], p1s); d1[4 * m + 2] = act1(d1[4 * m + 2], p1s); d1[4 * m + 3] = act1(d1[4 * m + 3], p1s);
                pq[m].x = pack2<BF16>(d1[4 * m], d1[4 * m + 1]); pq[m].y = pack2<BF16>(d1[4 * m + 2], d1[4 * m + 3]);
            } else {
                const u32x2 s0 = __builtin_amdgcn_permlane32_swap(pq[0].x, pq[1].x, false, false);
                const u32x2 s1 = __builtin_amdgcn_permlane32_swap(pq[0].y, pq[1].y, false, false);
                const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(p.py1 + (size_t)e_n * p1_img, 0, (int)p1_img, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)s0.x, (int)s1.x, (int)s0.y, (int)s1.y}, r1, e_vP + (unsigned)r * rowb1, 0, 0);
            }
        }
    };

    for (int k = 0;; ++k) {
        const int tn = tile_index(k + 1);
        const bool more = tn >= 0;
        int nn = 0, nx0 = 0, ny0 = 0;
        if (more) tile_coords(tn, nn, nx0, ny0);
        const char* const bb = smem + b_base + (unsigned)((k & 1) * STAGE);
        const unsigned tbase = (unsigned)(((ny0 - 1) * p.W + (nx0 - 1)) * p.in_pitch + p.in_coff) * 2u;
        const i32x4 nrsrc = make_rsrc(p.x + (size_t)nn * img_bytes + (size_t)(int)tbase, img_bytes - (size_t)(int)tbase);
        unsigned bad;
        {
            const unsigned sel = (ny0 == 0 ? 0x1fffu : 0u) | (nx0 == 0 ? 0x1fffu << 13 : 0u);
            const unsigned t = edge & sel;
            bad = more ? ((t | (t >> 13)) & 0x1fffu) : 0xffffffffu;
            if (nx0 + TILE + 1 > p.W) {
                const unsigned lim = (unsigned)(p.W - nx0 + 1);
#pragma unroll
                for (int i = 0; i < PPW; ++i) bad |= (((lxp[i / 6] >> (5 * (i % 6))) & 31u) >= lim ? 1u : 0u) << i;
            }
        }
        const unsigned lds0 = smem_lds + (unsigned)(((k + 1) & 1) * STAGE + wv * 1024);
        constexpr int AHEAD = 3;
        i32x4 b[4];
        auto read_b = [&](auto L_) __attribute__((always_inline)) {
            constexpr int L = decltype(L_)::value;
            constexpr int rp_ = L / NG, g_ = L % NG, c_ = g_ / TAPS, t_ = g_ % TAPS;
            b[L & 3] = *reinterpret_cast<const i32x4*>(bb + (2 * rp_ + t_ / 3) * ROWB + (t_ % 3) * PIXB + c_ * 32);
        };
        static_for<AHEAD>([&](auto L_) __attribute__((always_inline)) { read_b(L_); });
        auto run_pair = [&](auto rp_tag) __attribute__((always_inline)) {
            constexpr int rp = decltype(rp_tag)::value;
            constexpr int par = rp & 1;
            using PrevPar = std::integral_constant<int, par ^ 1>;
            using PrevRow = std::integral_constant<int, (rp == 0 ? RW - 2 : 2 * rp - 2)>;
            // the finished pair's d registers: younger than its loads are the pieces issued in ITS main stream (first pair: at least 12)
            // and that stream's five stores
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((rp == 1 ? PPW - 1 : 0) + 5) : "memory");
            if constexpr (rp == 0) load_offsets(n, x0, y0);        // (this tile's pixels: both pairs' loads)
            if constexpr (GB) {                                    // this pair's table row (read by its epilogue, a pair later)
                const int gx = x0 + px, gy = y0 + wv * RW + 2 * rp + pe;
                const int m = (gx == 0 ? 1 : 0) | (gx == p.W - 1 ? 2 : 0) | (gy == 0 ? 4 : 0) | (gy == p.H - 1 ? 8 : 0);
                bt_o[par] = (unsigned)(T_OFF_BT + m * 128 + hh * 16);
            }
            mfma_m0<BF16>(acc4[par], a_b4, b_ones);
            __builtin_amdgcn_sched_barrier(0);
            static_for<NG>([&](auto g_) __attribute__((always_inline)) {
                constexpr int g = decltype(g_)::value;
                constexpr int L = rp * NG + g, cs = L & 3;
                if constexpr (L + AHEAD < 2 * NG) read_b(std::integral_constant<int, L + AHEAD>{});
                __builtin_amdgcn_sched_barrier(0);
                mfma_m<BF16, true>(acc4[par], wa4[g], b[cs]);
                __builtin_amdgcn_sched_barrier(0);
                // this pair's own d loads first (k steps 1 .. 6), then the epilogue of the pair before
                if constexpr (g >= 1 && g <= 6) load_d(rp_tag, std::integral_constant<int, g - 1>{});
                if constexpr (rp == 1 && g == 7) store_offsets(n, x0, y0);     // behind the carried epilogue's... see below
                static_for<T_OPS>([&](auto o_) __attribute__((always_inline)) {
                    constexpr int q = T_OPS * g + decltype(o_)::value;
                    if constexpr (q >= TQ_R4 && q < TQ_END) top(PrevPar{}, PrevRow{}, std::integral_constant<int, q>{}, std::false_type{});
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (rp == 0 && g >= T_FIRST && (g - T_FIRST) % T_SPREAD == 0 && (g - T_FIRST) / T_SPREAD < PPW)
                    dma_piece_fast(std::integral_constant<int, (g - T_FIRST) / T_SPREAD>{}, bad, nrsrc, lds0);
            });
        };
        run_pair(std::integral_constant<int, 0>{});
        run_pair(std::integral_constant<int, 1>{});
        // the next tile has landed: younger than the last DMA piece are the first pair's stores behind it, the second pair's loads and stores
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(t_stores_behind_step(T_FIRST + T_SPREAD * (PPW - 1), T_OPS) + 6 + 5) : "memory");
        __builtin_amdgcn_s_barrier();
        if (!more) break;
        n = nn; x0 = nx0; y0 = ny0;
    }
    // the last tile's last row pair
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc4[1]) :: "memory");     // (operands: see conv64m_kernel's flush)
    static_for<TQ_END - TQ_R4>([&](auto i_) __attribute__((always_inline)) {
        top(std::integral_constant<int, 1>{}, std::integral_constant<int, RW - 2>{}, std::integral_constant<int, TQ_R4 + decltype(i_)::value>{}, std::true_type{});
        __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool BF16, int NCH, bool GB>
int launch_rfdb_tail(const S16K& k, hipStream_t st)
{
    static std::atomic<unsigned> attr_set[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return ESR_ERR_LAUNCH;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rfdb_tail_kernel<BF16, NCH, GB>), hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
        if (e != hipSuccess) {
            esr_set_err("hipFuncSetAttribute(rfdb_tail_kernel, MaxDynamicSharedMemorySize)", e);
            return ESR_ERR_LAUNCH;
        }
        attr_set[dev].store(1u, std::memory_order_relaxed);
    }
    const int ntiles = k.N * k.tiles_x * k.tiles_y;
    const int grid = ntiles < 256 ? ntiles : 256;
    esr_note_kernel("rfdb_tail_kernel<%s, %d, %s>", esr_tf(BF16), NCH, esr_tf(GB));
    hipLaunchKernelGGL((rfdb_tail_kernel<BF16, NCH, GB>), dim3(grid), dim3(256), T_LDS, st, k);
    return esr_check_launch("rfdb_tail_kernel launch");
}

}  // namespace

int esr_launch_rfdb_tail(const S16K& k, bool bf16, hipStream_t st)
{
    if (!k.wm32 || !k.pm32 || !k.pbias1 || !k.tw || !k.cat) return ESR_ERR_BAD_ARG;
    if (k.nchunks == 3) {                          // ESDB: 48 input channels, border table + GELU on r4
        if (!k.border || k.act != ESR_ACT_GELU) return ESR_ERR_UNSUPPORTED;
        return bf16 ? launch_rfdb_tail<true, 3, true>(k, st) : launch_rfdb_tail<false, 3, true>(k, st);
    }
    if (k.nchunks != 4 || k.border || k.act == ESR_ACT_GELU) return ESR_ERR_UNSUPPORTED;
    return bf16 ? launch_rfdb_tail<true, 4, false>(k, st) : launch_rfdb_tail<false, 4, false>(k, st);
}

namespace {
}  // namespace

int esr_launch_conv64m(const S16K& k, bool bf16, bool post, bool hl, hipStream_t st)
{
    if (!k.wm32 || (post && (!k.pm32 || !k.pbias1))) return ESR_ERR_BAD_ARG;
    if (hl) {
        if (!bf16 || post || !k.res || k.res_lo_stride <= 0 || !k.y1 || k.nchunks != 4) return ESR_ERR_BAD_ARG;
        return launch_conv64m<true, false, true, 4, false>(k, st);
    }
    if (k.nchunks == 3) {                          // ESDB's c{j}_r: border table + GELU (post: GELU too, fp16 only)
        if (!k.border || k.act != ESR_ACT_GELU || (post && (bf16 || !k.p1_gelu))) return ESR_ERR_UNSUPPORTED;
        if (post) return launch_conv64m<false, true, false, 3, true>(k, st);
        return bf16 ? launch_conv64m<true, false, false, 3, true>(k, st) : launch_conv64m<false, false, false, 3, true>(k, st);
    }
    if (k.border || k.act == ESR_ACT_GELU || (post && k.p1_gelu)) return ESR_ERR_UNSUPPORTED;
    if (bf16) return post ? launch_conv64m<true, true, false, 4, false>(k, st) : launch_conv64m<true, false, false, 4, false>(k, st);
    return post ? launch_conv64m<false, true, false, 4, false>(k, st) : launch_conv64m<false, false, false, 4, false>(k, st);
}

// ---- host side: where the 32x32x16 images live inside the packed blobs ----------------------------------------------------------------------
// esr_pack_conv_s16 blob of a 3x3 over 64 physical input channels with 17 .. 32 or 49 .. 64 outputs: [tap-pair image][bias][this image];
// NH = 1 or 2 output halves; fragment f = (chunk * 9 + tap) * NH + half, lane l = 32 h + i, element j: the 16-bit weight of output channel
// 32 half + i, input slot 16 chunk + 8 h + j at that tap (the same error-diffused values as the tap-pair image)
size_t esr_m32_conv_bytes(int cin_phys, int cout, int ksize)
{
    const int nch = esr_round_up(cin_phys, 16) / 16, nt = esr_round_up(cout, 16) / 16;
    if (ksize != 3) return 0;
    if (nch == 3) return nt == 2 ? (size_t)3 * M_TAPS * 1024 : (nt == 3 ? (size_t)3 * M_TAPS * 2 * 1024 : 0);      // ESDB's c4 | c{j}_r
    if (nch != 4) return 0;
    return nt == 4 ? (size_t)M_NFRAG * 1024 : (nt == 2 ? (size_t)M_NG * 1024 : 0);
}
size_t esr_m32_conv_offset(int cin_phys, int cout, int ksize)
{
    if (!esr_m32_conv_bytes(cin_phys, cout, ksize)) return 0;
    const size_t nt = (size_t)esr_round_up(cout, 16) / 16, nch = (size_t)esr_round_up(cin_phys, 16) / 16;
    return nch * 5 * nt * 1024 + nt * 16 * sizeof(float);
}
// esr_pack_post_s16 blob of a 1x1 from 49 .. 64 to 1 .. 32 channels: [hi images][lo images][bias][this image 16 KB]: fragment (step, hi | lo),
// step = 4 half + block: the eight input channels 32 half + 8 block + 4 h + (j & 3); hi image: the weight's high part in all eight k slots
// (slots 0 .. 3 meet the activations' high parts, 4 .. 7 their low parts), lo image: its low part in slots 0 .. 3 only; rows >= cout zero
size_t esr_m32_post_bytes(int cin, int cout)
{
    // (48 inputs -- ESDB's esa.conv1 behind rfdb_tail_kernel<.., 3, true> -- : the same eight steps, channels 48 .. 63 zero)
    const int kt = esr_round_up(cin, 16) / 16;
    return ((kt == 4 || kt == 3) && cout >= 1 && cout <= 32) ? (size_t)M_POST_IMG : 0;
}
size_t esr_m32_post_offset(int cin, int cout)
{
    if (!esr_m32_post_bytes(cin, cout)) return 0;
    const size_t ot = (size_t)esr_round_up(cout, 16) / 16, kt = (size_t)esr_round_up(cin, 16) / 16;
    return (size_t)2 * kt * ot * 1024 + ot * 16 * sizeof(float);
}

// ---- esr_pack_tail_s16 (ABI v12): the 1x1 of a 16-bit tail, K = three 32-slot segments + the 3x3's 32 channels, for rfdb_tail_kernel --------
// blob: 8 k steps x 2 output halves x (hi, lo) fragments of 1 KB, then 64 fp32 biases.  Fragment ((ks * 2 + half) * 2 + lo), lane l = 32 h + i
// (output channel 32 half + i), element j:
//   ks = 2 s + u (s = 0 .. 2, u = 0 | 1): slot 16 u + 8 h + j of segment s -- the B operand is 16 bytes of the segment's pixel as stored;
//   ks = 6 + t (t = 0 | 1): channel 8 (2 t + (j >> 2)) + 4 h + (j & 3) of the 3x3's result -- the B operand is two of its D blocks, rounded
namespace {
inline uint16_t m_to16(double v, int compute)
{
    if (compute == ESR_COMPUTE_BF16) {
        const float f = (float)v;
        uint32_t u;
        memcpy(&u, &f, 4);
        if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    }
    const _Float16 h = (_Float16)(float)v;
    uint16_t r;
    memcpy(&r, &h, 2);
    return r;
}
inline double m_from16(uint16_t h, int compute)
{
    if (compute == ESR_COMPUTE_BF16) {
        const uint32_t u = (uint32_t)h << 16;
        float f;
        memcpy(&f, &u, 4);
        return f;
    }
    _Float16 v;
    memcpy(&v, &h, 2);
    return (double)(float)v;
}
}  // namespace

extern "C" size_t esr_packed_tail_s16_bytes(int nseg, int seg_c, int mid_c, int cout)
{
    if (nseg != 3 || seg_c <= 0 || seg_c > 32 || mid_c <= 0 || mid_c > 32 || cout <= 0 || cout > 64) return 0;
    return (size_t)32 * 1024 + 64 * sizeof(float);
}

extern "C" int esr_pack_tail_s16(const float* w, const float* bias, int nseg, int seg_c, int mid_c, int cout, int compute, void* out, size_t out_bytes)
{
    const size_t need = esr_packed_tail_s16_bytes(nseg, seg_c, mid_c, cout);
    if (!w || !out || need == 0 || out_bytes < need) return ESR_ERR_BAD_ARG;
    if (compute != ESR_COMPUTE_BF16 && compute != ESR_COMPUTE_F16) return ESR_ERR_BAD_ARG;
    memset(out, 0, need);
    uint16_t* o = static_cast<uint16_t*>(out);
    const int kin = nseg * seg_c + mid_c;
    for (int ks = 0; ks < 8; ++ks)
        for (int half = 0; half < 2; ++half)
            for (int h = 0; h < 2; ++h)
                for (int i = 0; i < 32; ++i)
                    for (int j = 0; j < 8; ++j) {
                        const int oc = 32 * half + i;
                        int col = -1;                              // column of w (the reference's concat order), -1: a pad slot
                        if (ks < 6) {
                            const int sg = ks / 2, slot = 16 * (ks & 1) + 8 * h + j;
                            if (slot < seg_c) col = sg * seg_c + slot;
                        } else {
                            const int ch = 8 * (2 * (ks - 6) + (j >> 2)) + 4 * h + (j & 3);
                            if (ch < mid_c) col = nseg * seg_c + ch;
                        }
                        if (col < 0 || oc >= cout) continue;
                        const double wv = w[(size_t)oc * kin + col];
                        const uint16_t hi = m_to16(wv, compute);
                        const uint16_t lo = m_to16(wv - m_from16(hi, compute), compute);
                        const size_t e = (size_t)(32 * h + i) * 8 + j;
                        o[(size_t)((ks * 2 + half) * 2 + 0) * 512 + e] = hi;
                        o[(size_t)((ks * 2 + half) * 2 + 1) * 512 + e] = lo;
                    }
    float* bo = reinterpret_cast<float*>(static_cast<char*>(out) + (size_t)32 * 1024);
    if (bias)
        for (int oc = 0; oc < cout; ++oc) bo[oc] = bias[oc];
    return ESR_OK;
}
