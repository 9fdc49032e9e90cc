// esr_r16.hip -- the register-resident 16-bit-storage 3x3 kernels of libesr_hip.so: one wave per SIMD, weights in accumulation registers,
// whole-pixel stages, row pairs, the finished pair's epilogue as micro-steps behind the next pair's MFMAs (LAB_NOTES 9.5, 10.8):
//   conv48r_kernel<bf16|f16, NT, EXT, RW, FX>    3x3 over 48 channels (RLFB c1_r / c2_r, ESDB's BSConvU layers as dense 3x3s)
//   conv48rq_kernel<f16, FX>                     ... + residual == input, border table, GELU and ONE post 1x1 (ESDB c{j}_r + the next distillation conv)
//   conv48rp_kernel<bf16|f16, LRS>               ... + residual from HBM staged per wave + RLFB's 1x1 chain; LRS: the LR conv on hi + lo pairs
//   conv64r_kernel<bf16|f16, 2, EXT>             3x3 over 64 channels with two output tiles (RFDB c4)
// Split out of esr_s16.hip in round 6; the descriptors each kernel takes are decided in esr_s16.hip (esr_conv2d_s16),
// which calls the esr_launch_* entry points at the end of this file.  Same packed weights, fragment maps, operation order and rounding as
// conv_s16_kernel: bit-identical results where the shapes overlap.  Interface: include/esr_hip.h; reference semantics: team04_rlfn.py:109-122,
// team18_bsrn.py:150-172, rfdn_baseline/block.py:148-166.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <utility>
#include <type_traits>

#include "esr_internal.h"
#include "esr_s16_dev.h"

namespace {

// ---- conv48r_kernel: 3x3 convolutions over 48 input channels with their WEIGHTS IN REGISTERS (end of round 3) ---------------------
// conv_s16_kernel is bound by the length of a wave's own instruction stream per MFMA: per 60 MFMAs a wave issues 35
// ds_read_b128 (15 of them weight fragments), a stage barrier and the cursor's bookkeeping, three times per tile.  For the 48-channel
// 3x3s RLFN and BSRN spend a third to a half of their time in (RLFB c1_r / c2_r, team04_rlfn.py:109-116; ESDB c{j}_r / c4 as dense
// BSConvU, team18_bsrn.py:150-163) everything that repeats per K chunk can go:
//   * ONE 4-wave block per CU, one wave per SIMD, up to 512 registers per lane: the layer's 45 (NT = 2: 30) MFMA weight fragments
//     -- 3 chunks x 5 tap pairs x NT output tiles, 180 registers -- are loaded ONCE per block and stay in ACCUMULATION registers as
//     the A operands (asm MFMAs with an "a" constraint: left to itself hipcc parks them in AGPRs and copies 40 fragments back per tile);
//   * the LDS holds nothing but input: two WHOLE-PIXEL halo tiles (18 x 34 pixels x 96 bytes = 57.4 KB each).  A tile is one stage:
//     one barrier per tile instead of three, 96 contiguous bytes per pixel and DMA lane group instead of 32.  Pixel pitch 96 B: the 16
//     lanes of an LDS read group cover 16 different 16-byte slots (6 px + 2 c + h mod 16 is a permutation) -- conflict-free, no padding;
//   * ROW PAIRS are the outer loop of a tile (a wave owns 8 rows of the 16 x 32 tile): walking the 15 tap-pair groups four times costs
//     nothing with the weights in registers, and (a) a pair's 6 NT accumulators are finished after its 15 groups -- activation,
//     rounding, post 1x1 and stores run piecewise between the MFMA groups of the NEXT pair (the last pair's: of the next tile's first),
//     nothing of the epilogue is exposed; (b) the next tile's 58 DMA pieces are all issued during the FIRST pair, three quarters of a
//     tile ahead of their wait; (c) B fragments (one ds_read_b128 feeds NT MFMAs) are read three groups ahead through a ring of four.
// EXT adds what ESDB's dense BSConvU needs, in conv_s16_kernel's order of operations: the residual == input from the staged tile behind
// its chunk's groups, the border-bias table and GELU.  (A post-chain instantiation -- ESDB c{j}_r + the next distillation 1x1, two GELUs per
// pixel -- was written and measured slower than conv_s16_kernel at every size: 0.396 against 0.368 ms at 32 x 270 x 480, 32.3 against 31.4 us on
// one image; it is not part of the kernel any more and those launches stay on conv_s16_kernel.)
// Same packed weights, fragment maps, operation order and rounding as conv_s16_kernel: results are bit-identical (a batch takes this
// kernel, a single small image conv_s16_kernel: test_16bit_batch_equals_per_image).
// FX >= 0 (round 5): the kernel's three run-time switches as COMPILE-TIME constants -- bit 0 GELU, bit 1 border table, bit 2 residual == input.
// As wave-uniform branches inside the micro-step schedule they cost the EXT instantiation 120 s_cbranch + 140 v_mov (phi copies) per tile on top of
// the work itself (4231 against 2427 instructions for the same 360 MFMAs), and a wave that is alone on its SIMD pays ~4 cycles for every one of
// them (profiles/r05_instruction_census.txt).  The host launches the specialisation when a descriptor's switches match one that exists (ESDB:
// 7 = c{j}_r, 3 = c4), FX = -1 (run-time switches) otherwise.
template <bool BF16, int NT, bool EXT, int RW = 8, int FX = -1>
__global__ __launch_bounds__(256, 1) void conv48r_kernel(const S16K p)
{
    // RW = rows per wave: 8 (16 x 32 tiles) or 4 (16 x 16 tiles: small launches -- one DIV2K image is 352 large tiles on 256 CUs, two rounds of
    // which the second fills 37 % of the chip, but 704 small ones; the next tile's DMA then has less time to land, which is why batches keep 8)
    constexpr int NCH = 3, PAIRS = 5, TH = 18, THY = 4 * RW + 2;
    constexpr int PIXB = NCH * 32;                 // 96 bytes per staged pixel
    constexpr int NSLOT = TH * THY * (PIXB / 16);  // 3672 16-byte slots
    constexpr int NPIECES = (NSLOT + 63) / 64;     // 58 (RW = 4: 31) DMA pieces of 1 KB
    constexpr int STAGE = NPIECES * 1024;
    constexpr int PPW = (NPIECES + 3) / 4;         // 15 (8) per wave, the last waves one fewer
    constexpr int NG = NCH * PAIRS;                // tap-pair groups per row pair
    constexpr int SPP = NT == 3 ? 3 : 2;             // stores per row pair
    static_assert(PPW <= NG && (RW == 8 || RW == 4), "at most one DMA piece per tap-pair group of the first row pair");
    static_assert(NT == 2 || NT == 3, "shapes");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int WSTAGE = RW == 8 ? STAGE : 2 * STAGE;                        // where the weight blob (<= 45 KB) is staged before the first tile
    constexpr int BT_OFF = RW == 8 ? 2 * STAGE : 2 * STAGE + NCH * PAIRS * NT * 1024;   // RW = 8: the blob is staged in input stage 1; RW = 4: behind both stages
    float* const btab = reinterpret_cast<float*>(smem + BT_OFF);               // border bias table [16][NT * 16] (EXT && p.border)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, kq = lane >> 4;
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const bool has_border = FX >= 0 ? (FX & 2) != 0 : (EXT && p.border != nullptr), res_in = FX >= 0 ? (FX & 4) != 0 : (EXT && p.res_in != 0),
               gelu = FX >= 0 ? (FX & 1) != 0 : (EXT && p.act == ESR_ACT_GELU);

    // ---- the weights: registers for the life of the block ---------------------------------------------------------------------
    // (the blob goes global -> LDS ONCE per block -- stage 1 is free until the first tile's DMA issue -- and from there into each wave's
    // registers: read straight from global by all four waves it was 180 KB per block, 46 MB per launch on a single DIV2K image)
    constexpr int WPIECES = NCH * PAIRS * NT;      // 1 KB fragments
#pragma unroll
    for (int i = 0; i < (WPIECES + 3) / 4; ++i) {
        const int pc = wv + 4 * i;
        if (pc < WPIECES) dma_glb16(smem_lds + (unsigned)(WSTAGE + pc * 1024), p.wp + (size_t)pc * 1024 + lane * 16);
    }
    i32x4 wr[NCH][PAIRS][NT];
    f32x4 bia[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bia[t] = *reinterpret_cast<const f32x4*>(p.bias + t * 16 + kq * 4);
    if (has_border)             // (LDS-DMA, NT pieces of 1 KB: as a load / wait / ds_write loop it was three dependent round trips in front of the first tile's DMA)
        for (int pc = wv; pc < NT; pc += 4)
            dma_glb16(smem_lds + (unsigned)(BT_OFF + pc * 1024), reinterpret_cast<const char*>(p.border) + (size_t)pc * 1024 + lane * 16);

    const int ntiles = p.N * p.tiles_y * p.tiles_x;
    const int G = gridDim.x;
    auto tile_index = [&](int k) -> int { return s16_tile_index(k, ntiles); };
    auto tile_coords = [&](int t, int& n, int& x0, int& y0) __attribute__((always_inline)) { s16_tile_coords(t, p.magic_x, p.magic_y, p.tiles_x, p.tiles_y, (4 * RW), n, x0, y0); };
    const size_t img_bytes = (size_t)p.H * p.W * p.in_pitch * 2;
    // piece i of this wave of the tile (n, x0, y0) into stage `slot`; nothing valid (behind the last tile): zeros
    auto dma_piece = [&](int i, bool valid, int n, int x0, int y0, int slot) __attribute__((always_inline)) {
        const int pc = wv + 4 * i;
        if (i < PPW - 1 || pc < NPIECES) {                             // wave-uniform
            const unsigned sl = (unsigned)(pc * 64 + lane);             // 16-byte slot of the stage: pixel sl / 6, part sl % 6
            const unsigned pixel = sl / 6u, part = sl - pixel * 6u;
            const unsigned ly = pixel / (unsigned)TH, lx = pixel - ly * (unsigned)TH;
            const int gy = y0 - 1 + (int)ly, gx = x0 - 1 + (int)lx;
            const bool ok = valid && sl < (unsigned)NSLOT && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
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
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int q = 0; q < PAIRS; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                wr[c][q][t] = *reinterpret_cast<const i32x4*>(smem + WSTAGE + ((c * PAIRS + q) * NT + t) * 1024 + lane * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // every wave holds its fragments: stage 1 may be overwritten

    // lane-constant offsets of the B fragments: pair q reads tap min(2q + (kq >> 1), 8), channel half kq & 1 of the chunk
    int b_off[PAIRS];
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
        const int tap = min(2 * q + (kq >> 1), 8);
        b_off[q] = ((wv * RW + tap / 3) * TH + px + tap % 3) * PIXB + (kq & 1) * 16;
    }
    const int c_off = ((wv * RW + 1) * TH + px + 1) * PIXB + kq * 8;      // centre pixel of row 0 of the wave: channels 16 c + 4 kq .. +3 at + 32 c
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto swap16 = [&](uint2 X, uint2 Y) __attribute__((always_inline)) -> i32x4 { return s16_swap16(X, Y); };
    const float slope = gelu ? 1.f : p.slope;
    const size_t y_img = (size_t)p.H * p.W * p.y0_pitch * 2;
    const unsigned rowb = (unsigned)p.W * (unsigned)p.y0_pitch * 2u;

    f32x4 acc[2][NT][2];                 // [row pair & 1][channel tile][row of the pair]
    uint2 pk[NT][2];                     // the finished row pair, rounded
    unsigned e_vA = OOB, e_vB = OOB;     // store offsets (row 0 of the wave) of the tile whose epilogue is in flight
    int e_n = 0;
    // The epilogue of a finished row pair runs in MICRO-STEPS, one behind each MFMA of the next pair's groups (round 4): the wave is alone on
    // its SIMD and issues in order, so VALU work placed as a clump behind a group's last MFMA runs while the matrix pipe idles (an MFMA
    // occupies the pipe for 16 cycles, an independent VALU instruction issues in 4).  Step m of group g: fragment f = g - 1 (g = 1 .. 2 NT)
    // is activated in steps 0 / 1 and rounded in 2 / 3; store i = g - 9 swaps in steps 0 / 1 and leaves in step 2.
    f32x4 ev = {0.f, 0.f, 0.f, 0.f};
    u32x2 es0 = {0u, 0u}, es1 = {0u, 0u};
    auto epi_pack_step = [&](int par, int f, int m) __attribute__((always_inline)) {          // fragment f = 2 t + e of the finished pair
        const int t = f >> 1, e = f & 1;
        if (m == 0) {
            ev = acc[par][t][e];
            if (gelu) ev = gelu16x4(ev);
            else { ev.x = act1(ev.x, slope); ev.y = act1(ev.y, slope); }
        } else if (m == 1) {
            if (!gelu) { ev.z = act1(ev.z, slope); ev.w = act1(ev.w, slope); }
        } else if (m == 2) {
            pk[t][e].x = pack2<BF16>(ev.x, ev.y);
        } else if (m == 3) {
            pk[t][e].y = pack2<BF16>(ev.z, ev.w);
        }
    };
    auto epi_store_step = [&](int i, int r, int m) __attribute__((always_inline)) {           // store i of the pair whose first row is r
        // i = 0 / 1: tiles 0, 1 of row r / r + 1 (64 bytes per pixel); i = 2: the odd last tile of both rows (32 bytes per pixel and row)
        const int ta = i < 2 ? 0 : NT - 1, tb = i < 2 ? 1 : NT - 1, ea = i < 2 ? i : 0, eb = i < 2 ? i : 1;
        if (m == 0) es0 = __builtin_amdgcn_permlane16_swap(pk[ta][ea].x, pk[tb][eb].x, false, false);
        else if (m == 1) es1 = __builtin_amdgcn_permlane16_swap(pk[ta][ea].y, pk[tb][eb].y, false, false);
        else if (m == 2) {
            const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y0 + (size_t)e_n * y_img, 0, (int)y_img, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)es0.x, (int)es1.x, (int)es0.y, (int)es1.y}, yr, (i < 2 ? e_vA + (unsigned)(r + i) * rowb : e_vB + (unsigned)r * rowb), 0, 0);
        }
    };
    auto epi_pack = [&](int par, int f) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; ++m) epi_pack_step(par, f, m);
    };
    auto epi_store = [&](int i, int r) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 3; ++m) epi_store_step(i, r, m);
    };
    auto store_offsets = [&](int nn_, int x0_, int y0_) __attribute__((always_inline)) {
        const bool inx = x0_ + px < p.W;
        const unsigned pix = (unsigned)((y0_ + wv * RW) * p.W + x0_ + px);
        const unsigned base = (pix * (unsigned)p.y0_pitch + (unsigned)p.y0_coff) * 2u;
        const int chA = (kq & 1) * 16 + (kq >> 1) * 8, chB = 32 + (kq >> 1) * 8;
        e_vA = (inx && chA < p.cout_store) ? base + (unsigned)chA * 2u : OOB;
        e_vB = (inx && chB < p.cout_store) ? base + (unsigned)chB * 2u + ((kq & 1) ? rowb : 0u) : OOB;
        e_n = nn_;
    };
    for (int k = 0;; ++k) {
        const int tn = tile_index(k + 1);
        const bool more = tn >= 0;
        int nn = 0, nx0 = 0, ny0 = 0;
        if (more) tile_coords(tn, nn, nx0, ny0);
        const char* sb = smem + (k & 1) * STAGE;
        const bool on_border = has_border && (x0 == 0 || x0 + TILE >= p.W || y0 == 0 || y0 + 4 * RW >= p.H);
        // B fragments: a ring of four (two rows each), read THREE groups ahead of their MFMAs (a group is 2 NT MFMAs = ~100 cycles, an
        // LDS read returns after ~130): linear group index L = 15 rp + g over the tile's 60 groups
        constexpr int AHEAD = 3;
        i32x4 b[4][2];
        auto read_b = [&](int L) __attribute__((always_inline)) {
            const int rp_ = L / NG, g_ = L % NG, c_ = g_ / PAIRS, q_ = g_ % PAIRS;
#pragma unroll
            for (int e = 0; e < 2; ++e) b[L & 3][e] = *reinterpret_cast<const i32x4*>(sb + b_off[q_] + c_ * 32 + (2 * rp_ + e) * (TH * PIXB));
        };
#pragma unroll
        for (int L = 0; L < AHEAD; ++L) read_b(L);
#pragma unroll
        for (int rp = 0; rp < RW / 2; ++rp) {
            const int par = rp & 1;
            uint2 cen[NT][2];            // residual == input: the centre pixels of this pair's rows, 4 channels per tile
            if (EXT && res_in) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int e = 0; e < 2; ++e) cen[t][e] = *reinterpret_cast<const uint2*>(sb + c_off + t * 32 + (2 * rp + e) * (TH * PIXB));
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int c = g / PAIRS, q = g % PAIRS, L = rp * NG + g, cs = L & 3;
                if (L + AHEAD < (RW / 2) * NG) read_b(L + AHEAD);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        if (g == 0) {
                            if (BF16) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(b[cs][e]), "v"(bia[t]));
                            else asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(b[cs][e]), "v"(bia[t]));
                        } else {
                            if (BF16) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(b[cs][e]));
                            else asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(b[cs][e]));
                        }
                        // the previous row pair's epilogue (rp == 0: the previous TILE's last pair), one micro-step behind each MFMA: its
                        // accumulators were last written 15 groups ago.  (The block's first tile: nothing is waiting, the steps run on
                        // whatever the registers hold and their stores are out of range.)
                        {
                            const int m = 2 * t + e, r_prev = rp == 0 ? RW - 2 : 2 * rp - 2;
                            if (g >= 1 && g <= 2 * NT) epi_pack_step(par ^ 1, g - 1, m);
                            if (g >= 9 && g < 9 + SPP) epi_store_step(g - 9, r_prev, m);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                if (rp == 0 && g < PPW) dma_piece(g, more, nn, nx0, ny0, (k + 1) & 1);      // the next tile's DMA, in the shadow of the matrix pipe
                if (rp == 0 && g == NG - 1) store_offsets(n, x0, y0);              // (behind the previous tile's last store)
                if (EXT && q == PAIRS - 1 && (res_in || (on_border && c == NCH - 1))) {
                    // conv_s16_kernel's order: act(conv(x) + x) adds the centre pixels of chunk c to channel tile c BEHIND chunk c's
                    // groups; the border table follows the last chunk.  The MFMAs above are asm: hipcc pads neither the read of their
                    // results (XDL write -> VALU read) nor the next group's read of what is written here
                    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
                    if (res_in && c < NT) {
#pragma unroll
                        for (int e = 0; e < 2; ++e) acc[par][c < NT ? c : 0][e] += unpack4<BF16>(cen[c < NT ? c : 0][e]);
                    }
                    if (on_border && c == NCH - 1) {
                        const int gx = x0 + px;
                        const int cm = (gx == 0 ? 1 : 0) | (gx == p.W - 1 ? 2 : 0);
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int gy = y0 + wv * RW + 2 * rp + e;
                            const int m = cm | (gy == 0 ? 4 : 0) | (gy == p.H - 1 ? 8 : 0);
#pragma unroll
                            for (int t = 0; t < NT; ++t) acc[par][t][e] += *reinterpret_cast<const f32x4*>(btab + m * (NT * 16) + t * 16 + kq * 4);
                        }
                    }
                    asm volatile("s_nop 3" ::: "memory");
                }
            }
        }
        // the next tile has landed: younger than its DMA are the stores of this tile's row pairs but the last
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((RW / 2 - 1) * SPP) : "memory");
        __builtin_amdgcn_s_barrier();
        if (!more) break;
        n = nn; x0 = nx0; y0 = ny0;
    }
    // the last tile's last row pair
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");     // (asm MFMAs: hipcc does not pad MFMA -> VALU reads of their results)
#pragma unroll
    for (int f = 0; f < 2 * NT; ++f) epi_pack(1, f);
#pragma unroll
    for (int i = 0; i < SPP; ++i) epi_store(i, RW - 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the trailing zero-fill DMA must not outlive the block)
}

// ---- conv48rq_kernel: conv48r_kernel<.., 3, EXT, 4> + ONE post 1x1 of two output tiles (round 4) ------------------------------------------
// ESDB's c{j}_r as a dense BSConvU (+ input, GELU) with the next distillation Linear + GELU in its epilogue (team18_bsrn.py:150-163): 74 % of
// BSRN's fp16 step ran on conv_s16_kernel<3, 3, 8, .., 2, 0> at 0.32 of the HBM peak.  A first post-chain instantiation of conv48r_kernel (round
// 3) lost against it: its chain was a clump of ~130 VALU instructions + 12 MFMAs behind one group.  Here the finished pair's whole epilogue is a
// list of 87 micro-operations -- activation halves, roundings, lane swaps, stores, the B operands of the 1x1, its MFMAs one by one, the second
// activation -- and operation k runs behind convolution MFMA k + 3 of the next pair (compile-time schedule, static_for).  fp32 values of the
// activated main result go back into the pair's accumulators, where the 1x1 reads them (conv_s16_kernel: the post chain sees the unrounded tile).
// Same order of operations per accumulator as conv_s16_kernel: bit-identical.  Post images: high parts only (the host sends fp16 plans here,
// post_lo == 0) or high + low (bf16).
// FX: as conv48r_kernel's, + bit 3 = the post 1x1's activation is GELU (ESDB: 15)
template <bool BF16, int FX = -1>
__global__ __launch_bounds__(256, 1) void conv48rq_kernel(const S16K p)
{
    constexpr int NT = 3, RW = 4, PNT1 = 2;
    constexpr bool EXT = true, plo = BF16;
    constexpr int NCH = 3, PAIRS = 5, TH = 18, THY = 4 * RW + 2;
    constexpr int PIXB = NCH * 32;                 // 96 bytes per staged pixel
    constexpr int NSLOT = TH * THY * (PIXB / 16);  // 3672 16-byte slots
    constexpr int NPIECES = (NSLOT + 63) / 64;     // 58 (RW = 4: 31) DMA pieces of 1 KB
    constexpr int STAGE = NPIECES * 1024;
    constexpr int PPW = (NPIECES + 3) / 4;         // 15 (8) per wave, the last waves one fewer
    constexpr int NG = NCH * PAIRS;                // tap-pair groups per row pair
    constexpr int SPP = 3 + 2;                       // stores per row pair: the conv's three, the post's two
    static_assert(PPW <= NG && (RW == 8 || RW == 4), "at most one DMA piece per tap-pair group of the first row pair");
    static_assert(NT == 2 || NT == 3, "shapes");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int WSTAGE = RW == 8 ? STAGE : 2 * STAGE;                        // where the weight blob (<= 45 KB) is staged before the first tile
    constexpr int BT_OFF = 2 * STAGE + NCH * PAIRS * NT * 1024;                // border table behind the staged blob
    constexpr int P1_IMG = NT * PNT1 * 1024;                                   // post images [k tile][out tile] (hi, then lo)
    constexpr int OFF_POST = BT_OFF + NT * 1024;
    float* const btab = reinterpret_cast<float*>(smem + BT_OFF);               // border bias table [16][NT * 16] (EXT && p.border)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, kq = lane >> 4;
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const bool has_border = FX >= 0 ? (FX & 2) != 0 : (EXT && p.border != nullptr), res_in = FX >= 0 ? (FX & 4) != 0 : (EXT && p.res_in != 0),
               gelu = FX >= 0 ? (FX & 1) != 0 : (EXT && p.act == ESR_ACT_GELU);

    // ---- the weights: registers for the life of the block ---------------------------------------------------------------------
    // (the blob goes global -> LDS ONCE per block -- stage 1 is free until the first tile's DMA issue -- and from there into each wave's
    // registers: read straight from global by all four waves it was 180 KB per block, 46 MB per launch on a single DIV2K image)
    constexpr int WPIECES = NCH * PAIRS * NT;      // 1 KB fragments
#pragma unroll
    for (int i = 0; i < (WPIECES + 3) / 4; ++i) {
        const int pc = wv + 4 * i;
        if (pc < WPIECES) dma_glb16(smem_lds + (unsigned)(WSTAGE + pc * 1024), p.wp + (size_t)pc * 1024 + lane * 16);
    }
    for (int pc = wv; pc < (plo ? 2 : 1) * (P1_IMG / 1024); pc += 4) dma_glb16(smem_lds + (unsigned)(OFF_POST + pc * 1024), p.pw1 + (size_t)pc * 1024 + lane * 16);
    i32x4 wr[NCH][PAIRS][NT];
    f32x4 bia[NT], pb1[PNT1];
#pragma unroll
    for (int t = 0; t < PNT1; ++t) pb1[t] = *reinterpret_cast<const f32x4*>(p.pw1 + (size_t)2 * P1_IMG + (t * 16 + kq * 4) * 4);
    const char* const img1 = smem + OFF_POST + lane * 16;
#pragma unroll
    for (int t = 0; t < NT; ++t) bia[t] = *reinterpret_cast<const f32x4*>(p.bias + t * 16 + kq * 4);
    if (has_border)             // (LDS-DMA, NT pieces of 1 KB: as a load / wait / ds_write loop it was three dependent round trips in front of the first tile's DMA)
        for (int pc = wv; pc < NT; pc += 4)
            dma_glb16(smem_lds + (unsigned)(BT_OFF + pc * 1024), reinterpret_cast<const char*>(p.border) + (size_t)pc * 1024 + lane * 16);

    const int ntiles = p.N * p.tiles_y * p.tiles_x;
    const int G = gridDim.x;
    auto tile_index = [&](int k) -> int { return s16_tile_index(k, ntiles); };
    auto tile_coords = [&](int t, int& n, int& x0, int& y0) __attribute__((always_inline)) { s16_tile_coords(t, p.magic_x, p.magic_y, p.tiles_x, p.tiles_y, (4 * RW), n, x0, y0); };
    const size_t img_bytes = (size_t)p.H * p.W * p.in_pitch * 2;
    // piece i of this wave of the tile (n, x0, y0) into stage `slot`; nothing valid (behind the last tile): zeros
    auto dma_piece = [&](int i, bool valid, int n, int x0, int y0, int slot) __attribute__((always_inline)) {
        const int pc = wv + 4 * i;
        if (i < PPW - 1 || pc < NPIECES) {                             // wave-uniform
            const unsigned sl = (unsigned)(pc * 64 + lane);             // 16-byte slot of the stage: pixel sl / 6, part sl % 6
            const unsigned pixel = sl / 6u, part = sl - pixel * 6u;
            const unsigned ly = pixel / (unsigned)TH, lx = pixel - ly * (unsigned)TH;
            const int gy = y0 - 1 + (int)ly, gx = x0 - 1 + (int)lx;
            const bool ok = valid && sl < (unsigned)NSLOT && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
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
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int q = 0; q < PAIRS; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                wr[c][q][t] = *reinterpret_cast<const i32x4*>(smem + WSTAGE + ((c * PAIRS + q) * NT + t) * 1024 + lane * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // every wave holds its fragments: stage 1 may be overwritten

    // lane-constant offsets of the B fragments: pair q reads tap min(2q + (kq >> 1), 8), channel half kq & 1 of the chunk
    int b_off[PAIRS];
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
        const int tap = min(2 * q + (kq >> 1), 8);
        b_off[q] = ((wv * RW + tap / 3) * TH + px + tap % 3) * PIXB + (kq & 1) * 16;
    }
    const int c_off = ((wv * RW + 1) * TH + px + 1) * PIXB + kq * 8;      // centre pixel of row 0 of the wave: channels 16 c + 4 kq .. +3 at + 32 c
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto swap16 = [&](uint2 X, uint2 Y) __attribute__((always_inline)) -> i32x4 { return s16_swap16(X, Y); };
    const float slope = gelu ? 1.f : p.slope;
    const size_t y_img = (size_t)p.H * p.W * p.y0_pitch * 2;
    const unsigned rowb = (unsigned)p.W * (unsigned)p.y0_pitch * 2u;

    f32x4 acc[2][NT][2];                 // [row pair & 1][channel tile][row of the pair]
    uint2 pk[NT][2], pk1[PNT1][2];       // the finished row pair, rounded: the conv's result, the post 1x1's
    unsigned e_vA = OOB, e_vB = OOB, e_vP = OOB;     // store offsets (row 0 of the wave) of the tile whose epilogue is in flight
    int e_n = 0;
    const float p1s = p.p1_slope;
    const bool g1 = FX >= 0 ? (FX & 8) != 0 : p.p1_gelu != 0;
    const size_t p1_img = (size_t)p.H * p.W * p.py1_pitch * 2;
    const unsigned rowb1 = (unsigned)p.W * (unsigned)p.py1_pitch * 2u;
    f32x4 ev = {0.f, 0.f, 0.f, 0.f};
    u32x2 es0 = {0u, 0u}, es1 = {0u, 0u};
    i32x4 bsv[2][2];                     // [k tile & 1][row]: the fp32 fragment as the 1x1's B operand (hi parts | lo parts)
    i32x4 pa[2][PNT1 * 2];               // [k tile & 1][2 ot + lo]
    f32x4 d1[PNT1][2];
    auto actf = [&](f32x4& v, int h, bool ge, float sl) __attribute__((always_inline)) {     // activation of a fragment in two halves (GELU: all in the first)
        if (h == 0) {
            if (ge) v = gelu16x4(v);
            else { v.x = act1(v.x, sl); v.y = act1(v.y, sl); }
        } else if (!ge) { v.z = act1(v.z, sl); v.w = act1(v.w, sl); }
    };
    auto hl = [&](i32x4& o, f32x4 v, int h) __attribute__((always_inline)) {
        if (h == 0) {
            o.x = (int)pack2<BF16>(v.x, v.y); o.y = (int)pack2<BF16>(v.z, v.w);
            if (!BF16) { o.z = 0; o.w = 0; }
        } else if (BF16) {
            float a_, b_, c_, d_;
            unpack2<BF16>((unsigned)o.x, a_, b_);
            unpack2<BF16>((unsigned)o.y, c_, d_);
            o.z = (int)pack2<BF16>(v.x - a_, v.y - b_); o.w = (int)pack2<BF16>(v.z - c_, v.w - d_);
        }
    };
    auto load_p1 = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int ot = 0; ot < PNT1; ++ot) {
            pa[kt & 1][2 * ot] = *reinterpret_cast<const i32x4*>(img1 + (kt * PNT1 + ot) * 1024);
            if (plo) pa[kt & 1][2 * ot + 1] = *reinterpret_cast<const i32x4*>(img1 + P1_IMG + (kt * PNT1 + ot) * 1024);
        }
    };
    auto pm1 = [&](int kt, int i) __attribute__((always_inline)) {               // post MFMA i of k tile kt: 0 .. 3 high images (ot, e), 4 .. 7 low images
        const int lo = i >> 2, ot = (i & 3) >> 1, e = i & 1;
        if (lo && !plo) return;
        d1[ot][e] = mfma32<BF16>(pa[kt & 1][2 * ot + lo], bsv[kt & 1][e], (kt == 0 && !lo) ? pb1[ot] : d1[ot][e]);
    };
    // operation k of the finished pair's epilogue (par = its accumulators, r = its first row); see the kernel's header
    auto op = [&](auto par_, auto r_, auto k_) __attribute__((always_inline)) {
        constexpr int par = decltype(par_)::value, r = decltype(r_)::value, k = decltype(k_)::value;
        if constexpr (k >= 0 && k < 24) {                                      // the conv's fragments: activation (fp32 back into acc), rounding
            constexpr int f = k >> 2, m = k & 3, t = f >> 1, e = f & 1;
            if constexpr (m == 0) { ev = acc[par][t][e]; actf(ev, 0, gelu, slope); }
            else if constexpr (m == 1) { actf(ev, 1, gelu, slope); acc[par][t][e] = ev; }
            else if constexpr (m == 2) pk[t][e].x = pack2<BF16>(ev.x, ev.y);
            else pk[t][e].y = pack2<BF16>(ev.z, ev.w);
        } else if constexpr (k < 33) {                                         // the conv's three stores
            constexpr int i = (k - 24) / 3, m = (k - 24) % 3;
            constexpr int ta = i < 2 ? 0 : NT - 1, tb = i < 2 ? 1 : NT - 1, ea = i < 2 ? i : 0, eb = i < 2 ? i : 1;
            if constexpr (m == 0) es0 = __builtin_amdgcn_permlane16_swap(pk[ta][ea].x, pk[tb][eb].x, false, false);
            else if constexpr (m == 1) es1 = __builtin_amdgcn_permlane16_swap(pk[ta][ea].y, pk[tb][eb].y, false, false);
            else {
                const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y0 + (size_t)e_n * y_img, 0, (int)y_img, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)es0.x, (int)es1.x, (int)es0.y, (int)es1.y}, yr, (i < 2 ? e_vA + (unsigned)(r + i) * rowb : e_vB + (unsigned)r * rowb), 0, 0);
            }
        } else if constexpr (k < 69) {                                         // the 1x1, k tile by k tile: B operands (4 ops), MFMAs (8 ops; fp16: 4)
            constexpr int kt = (k - 33) / 12, j = (k - 33) % 12;
            if constexpr (j < 4) hl(bsv[kt & 1][j >> 1], acc[par][kt][j >> 1], j & 1);
            else pm1(kt, j - 4);
        } else if constexpr (k < 81) {                                         // the 1x1's result: activation, rounding; row 0's two tiles first
            constexpr int q = (k - 69) / 3, m = (k - 69) % 3, e = q >> 1, ot = q & 1;
            if constexpr (m < 2) actf(d1[ot][e], m, g1, p1s);
            else { pk1[ot][e].x = pack2<BF16>(d1[ot][e].x, d1[ot][e].y); pk1[ot][e].y = pack2<BF16>(d1[ot][e].z, d1[ot][e].w); }
        } else if constexpr (k < 87) {                                         // its two stores (one per row)
            constexpr int e = (k - 81) / 3, m = (k - 81) % 3;
            if constexpr (m == 0) es0 = __builtin_amdgcn_permlane16_swap(pk1[0][e].x, pk1[1][e].x, false, false);
            else if constexpr (m == 1) es1 = __builtin_amdgcn_permlane16_swap(pk1[0][e].y, pk1[1][e].y, false, false);
            else {
                const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(p.py1 + (size_t)e_n * p1_img, 0, (int)p1_img, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)es0.x, (int)es1.x, (int)es0.y, (int)es1.y}, r1, e_vP + (unsigned)(r + e) * rowb1, 0, 0);
            }
        }
    };
    // slot s of a pair (behind its convolution MFMA s): operation s - 3, and the post images of k tile kt ten slots ahead of its MFMAs
    auto micro = [&](auto par_, auto r_, auto s_) __attribute__((always_inline)) {
        constexpr int s = decltype(s_)::value;
        if constexpr (s >= 3) op(par_, r_, std::integral_constant<int, s - 3>{});
        if constexpr (s == 30) load_p1(0);
        if constexpr (s == 42) load_p1(1);
        if constexpr (s == 54) load_p1(2);
    };
    auto store_offsets = [&](int nn_, int x0_, int y0_) __attribute__((always_inline)) {
        const bool inx = x0_ + px < p.W;
        const unsigned pix = (unsigned)((y0_ + wv * RW) * p.W + x0_ + px);
        const unsigned base = (pix * (unsigned)p.y0_pitch + (unsigned)p.y0_coff) * 2u;
        const int chA = (kq & 1) * 16 + (kq >> 1) * 8, chB = 32 + (kq >> 1) * 8;
        e_vA = (inx && chA < p.cout_store) ? base + (unsigned)chA * 2u : OOB;
        e_vB = (inx && chB < p.cout_store) ? base + (unsigned)chB * 2u + ((kq & 1) ? rowb : 0u) : OOB;
        e_vP = (inx && chA < p.p1_cout8) ? (pix * (unsigned)p.py1_pitch + (unsigned)p.py1_coff) * 2u + (unsigned)chA * 2u : OOB;
        e_n = nn_;
    };
    for (int k = 0;; ++k) {
        const int tn = tile_index(k + 1);
        const bool more = tn >= 0;
        int nn = 0, nx0 = 0, ny0 = 0;
        if (more) tile_coords(tn, nn, nx0, ny0);
        const char* sb = smem + (k & 1) * STAGE;
        const bool on_border = has_border && (x0 == 0 || x0 + TILE >= p.W || y0 == 0 || y0 + 4 * RW >= p.H);
        // B fragments: a ring of four (two rows each), read THREE groups ahead of their MFMAs (a group is 2 NT MFMAs = ~100 cycles, an
        // LDS read returns after ~130): linear group index L = 15 rp + g over the tile's 60 groups
        constexpr int AHEAD = 3;
        i32x4 b[4][2];
        auto read_b = [&](int L) __attribute__((always_inline)) {
            const int rp_ = L / NG, g_ = L % NG, c_ = g_ / PAIRS, q_ = g_ % PAIRS;
#pragma unroll
            for (int e = 0; e < 2; ++e) b[L & 3][e] = *reinterpret_cast<const i32x4*>(sb + b_off[q_] + c_ * 32 + (2 * rp_ + e) * (TH * PIXB));
        };
#pragma unroll
        for (int L = 0; L < AHEAD; ++L) read_b(L);
        auto run_pair = [&](auto rp_tag) __attribute__((always_inline)) {
            constexpr int rp = decltype(rp_tag)::value;
            constexpr int par = rp & 1;
            if constexpr (rp == 1) store_offsets(n, x0, y0);       // behind the carried epilogue's last store (first pair, slot 89), ahead of this tile's first
            uint2 cen[NT][2];            // residual == input: the centre pixels of this pair's rows, 4 channels per tile
            if (EXT && res_in) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int e = 0; e < 2; ++e) cen[t][e] = *reinterpret_cast<const uint2*>(sb + c_off + t * 32 + (2 * rp + e) * (TH * PIXB));
            }
            static_for<NG>([&](auto g_) __attribute__((always_inline)) {
                constexpr int g = decltype(g_)::value;
                constexpr int c = g / PAIRS, q = g % PAIRS, L = rp * NG + g, cs = L & 3;
                if constexpr (L + AHEAD < (RW / 2) * NG) read_b(L + AHEAD);
                __builtin_amdgcn_sched_barrier(0);
                static_for<2 * NT>([&](auto m_) __attribute__((always_inline)) {
                    constexpr int t = decltype(m_)::value >> 1, e = decltype(m_)::value & 1;
                    {
                        if (g == 0) {
                            if (BF16) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(b[cs][e]), "v"(bia[t]));
                            else asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(b[cs][e]), "v"(bia[t]));
                        } else {
                            if (BF16) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(b[cs][e]));
                            else asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(b[cs][e]));
                        }
                        // the finished pair's epilogue (rp == 0: the previous TILE's last pair; the block's first tile: whatever the registers
                        // hold, stores out of range)
                        micro(std::integral_constant<int, par ^ 1>{}, std::integral_constant<int, (rp == 0 ? RW - 2 : 2 * rp - 2)>{}, std::integral_constant<int, 6 * g + 2 * t + e>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                if (rp == 0 && g < PPW) dma_piece(g, more, nn, nx0, ny0, (k + 1) & 1);      // the next tile's DMA, in the shadow of the matrix pipe
                if (EXT && q == PAIRS - 1 && (res_in || (on_border && c == NCH - 1))) {
                    // conv_s16_kernel's order: act(conv(x) + x) adds the centre pixels of chunk c to channel tile c BEHIND chunk c's
                    // groups; the border table follows the last chunk.  The MFMAs above are asm: hipcc pads neither the read of their
                    // results (XDL write -> VALU read) nor the next group's read of what is written here
                    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
                    if (res_in && c < NT) {
#pragma unroll
                        for (int e = 0; e < 2; ++e) acc[par][c < NT ? c : 0][e] += unpack4<BF16>(cen[c < NT ? c : 0][e]);
                    }
                    if (on_border && c == NCH - 1) {
                        const int gx = x0 + px;
                        const int cm = (gx == 0 ? 1 : 0) | (gx == p.W - 1 ? 2 : 0);
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int gy = y0 + wv * RW + 2 * rp + e;
                            const int m = cm | (gy == 0 ? 4 : 0) | (gy == p.H - 1 ? 8 : 0);
#pragma unroll
                            for (int t = 0; t < NT; ++t) acc[par][t][e] += *reinterpret_cast<const f32x4*>(btab + m * (NT * 16) + t * 16 + kq * 4);
                        }
                    }
                    asm volatile("s_nop 3" ::: "memory");
                }
            });
        };
        run_pair(std::integral_constant<int, 0>{});
        run_pair(std::integral_constant<int, 1>{});
        // the next tile has landed: younger than its DMA are the stores of this tile's row pairs but the last
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((RW / 2 - 1) * SPP) : "memory");
        __builtin_amdgcn_s_barrier();
        if (!more) break;
        n = nn; x0 = nx0; y0 = ny0;
    }
    // the last tile's last row pair: the same operations, back to back
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");     // (asm MFMAs: hipcc does not pad MFMA -> VALU reads of their results)
    static_for<90>([&](auto s_) __attribute__((always_inline)) { micro(std::integral_constant<int, 1>{}, std::integral_constant<int, RW - 2>{}, s_); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the trailing zero-fill DMA must not outlive the block)
}


// ---- conv64r_kernel: conv48r_kernel's plan for 64 physical input channels (round 4) ------------------------------------------------
// RFDB's c3_r / c4 (rfdn_baseline/block.py:157-161) and every other plain 3x3 over 49..64 channels with 2 or 4 output tiles ran on
// conv_s16_kernel at 0.32-0.38 of the HBM peak: a wave issues one ds_read_b128 per two MFMAs there (four weight + four pixel fragments per
// 16 MFMAs) and the eight waves ask for them in lockstep -- the LDS pipe and the matrix pipe each need a stage's whole time.  What changes
// against conv48r_kernel:
//   * the layer's weights are 80 fragments x 4 registers = 320 for NT = 4: more than the 256 accumulation registers.  Chunks 0..2 (240)
//     stay there; chunk 3 (20 KB) stays in LDS where the blob was staged and its fragments are read one group ahead through a ring of two
//     -- on average 3 LDS reads per 8 MFMAs instead of 4 per 8.  NT = 2: all 160 in registers;
//   * a staged pixel is 128 bytes in memory and 160 in LDS (two unused 16-byte slots).  A ds_read_b128 is served in four groups of 16
//     lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS): half of a group reads channel half
//     0 of eight pixels, the other half channel half 1 of the other eight.  With a 128-byte pitch the 16 lanes meet in 2 of the 16 slot
//     columns, with 144 bytes (9 px mod 16) the two halves of a group collide in 7; 10 px mod 16 puts half 0 on the even and half 1 on
//     the odd columns, eight different ones each: conflict-free.  The pad slots are part of the DMA pieces (their lanes fetch nothing:
//     out-of-range offset), 51 pieces of 1 KB per 18 x 18 tile;
//   * 16 x 16 tiles only (a 16 x 32 tile's two stages would not fit); LDS: two stages + the blob behind stage 0 = 131 KB.
// Same packed weights, fragment maps, operation order and rounding as conv_s16_kernel: results are bit-identical.
template <bool BF16, int NT, bool EXT>
__global__ __launch_bounds__(256, 1) void conv64r_kernel(const S16K p)
{
    constexpr int NCH = 4, PAIRS = 5, TH = 18, RW = 4, THY = 4 * RW + 2;
    constexpr int GSL = NCH * 2;                   // 16-byte slots of a pixel in memory
    constexpr int LSL = GSL + 2;                   // ... in LDS
    constexpr int PIXB = LSL * 16;                 // 160
    constexpr int NSLOT = TH * THY * LSL;          // 3240
    constexpr int NPIECES = (NSLOT + 63) / 64;     // 51
    constexpr int STAGE = NPIECES * 1024;
    constexpr int PPW = (NPIECES + 3) / 4;         // 13 per wave, the last wave one fewer
    constexpr int NG = NCH * PAIRS;                // 20 tap-pair groups per row pair
    constexpr int NCR = NT == 4 ? 3 : 4;           // chunks whose weights live in registers
    constexpr int SPP = NT;                        // stores per row pair: NT / 2 tile pairs x 2 rows
    constexpr int WSTAGE = STAGE;                  // the blob is staged behind stage 0 (stage 1 is free until the second tile's DMA) ...
    constexpr int W3 = WSTAGE + NCR * PAIRS * NT * 1024;      // ... and chunk 3 stays where it landed
    static_assert(PPW <= NG, "at most one DMA piece per tap-pair group of the first row pair");
    static_assert(NT == 2 || NT == 4, "shapes");
    static_assert(NCR == NCH || W3 >= 2 * STAGE, "the resident chunk lies behind stage 1");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, kq = lane >> 4;
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const bool res_in = EXT && p.res_in != 0;       // (no GELU here: the 64-channel layers of the path are RFDN's, LeakyReLU)

    constexpr int WPIECES = NCH * PAIRS * NT;      // 1 KB fragments
#pragma unroll
    for (int i = 0; i < (WPIECES + 3) / 4; ++i) {
        const int pc = wv + 4 * i;
        if (pc < WPIECES) dma_glb16(smem_lds + (unsigned)(WSTAGE + pc * 1024), p.wp + (size_t)pc * 1024 + lane * 16);
    }
    i32x4 wr[NCR][PAIRS][NT];
    f32x4 bia[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bia[t] = *reinterpret_cast<const f32x4*>(p.bias + t * 16 + kq * 4);

    const int ntiles = p.N * p.tiles_y * p.tiles_x;
    const int G = gridDim.x;
    auto tile_index = [&](int k) -> int { return s16_tile_index(k, ntiles); };
    auto tile_coords = [&](int t, int& n, int& x0, int& y0) __attribute__((always_inline)) { s16_tile_coords(t, p.magic_x, p.magic_y, p.tiles_x, p.tiles_y, (4 * RW), n, x0, y0); };
    const size_t img_bytes = (size_t)p.H * p.W * p.in_pitch * 2;
    // piece i of this wave of the tile (n, x0, y0) into stage `slot`; nothing valid (behind the last tile): zeros
    auto dma_piece = [&](int i, bool valid, int n, int x0, int y0, int slot) __attribute__((always_inline)) {
        const int pc = wv + 4 * i;
        if (i < PPW - 1 || pc < NPIECES) {                             // wave-uniform
            const unsigned sl = (unsigned)(pc * 64 + lane);             // 16-byte slot of the stage: pixel sl / 10, part sl % 10 (parts 8, 9: the pad)
            const unsigned pixel = sl / (unsigned)LSL, part = sl - pixel * (unsigned)LSL;
            const unsigned ly = pixel / (unsigned)TH, lx = pixel - ly * (unsigned)TH;
            const int gy = y0 - 1 + (int)ly, gx = x0 - 1 + (int)lx;
            const bool ok = valid && part < (unsigned)GSL && sl < (unsigned)NSLOT && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
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
#pragma unroll
    for (int c = 0; c < NCR; ++c)
#pragma unroll
        for (int q = 0; q < PAIRS; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                wr[c][q][t] = *reinterpret_cast<const i32x4*>(smem + WSTAGE + ((c * PAIRS + q) * NT + t) * 1024 + lane * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // every wave holds its fragments: stage 1 may be overwritten

    // lane-constant offsets of the B fragments: pair q reads tap min(2q + (kq >> 1), 8), channel half kq & 1 of the chunk
    int b_off[PAIRS];
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
        const int tap = min(2 * q + (kq >> 1), 8);
        b_off[q] = ((wv * RW + tap / 3) * TH + px + tap % 3) * PIXB + (kq & 1) * 16;
    }
    const int c_off = ((wv * RW + 1) * TH + px + 1) * PIXB + kq * 8;      // centre pixel of row 0 of the wave: channels 16 c + 4 kq .. +3 at + 32 c
    const char* const w3 = smem + W3 + lane * 16;                           // chunk 3's fragments: + (q * NT + t) KB
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto swap16 = [&](uint2 X, uint2 Y) __attribute__((always_inline)) -> i32x4 { return s16_swap16(X, Y); };
    const float slope = p.slope;
    const size_t y_img = (size_t)p.H * p.W * p.y0_pitch * 2;
    const unsigned rowb = (unsigned)p.W * (unsigned)p.y0_pitch * 2u;

    f32x4 acc[2][NT][2];                 // [row pair & 1][channel tile][row of the pair]
    uint2 pk[NT][2];                     // the finished row pair, rounded
    unsigned e_v[NT / 2];                // store offsets (row 0 of the wave, tile pair j) of the tile whose epilogue is in flight
#pragma unroll
    for (int j = 0; j < NT / 2; ++j) e_v[j] = OOB;
    int e_n = 0;
    // The epilogue of a finished row pair runs in MICRO-STEPS, one behind each MFMA of the next pair's groups: this wave is alone on its
    // SIMD and issues in order, so VALU work placed behind a group's last MFMA runs while the matrix pipe idles (an MFMA occupies the pipe
    // for 16 cycles, a dependent-free VALU instruction issues in 4) -- as a clump behind each group the epilogue cost a third of the launch
    // (tools/abl/c64_abl.py: 128 us without it, 213 us with).  step m of group g: fragment f = g - 1 (g = 1 .. 2 NT) is activated in steps
    // 0 / 1 and rounded in 2 / 3; store i = g - 2 NT - 1 swaps in steps 0 / 1 and leaves in step 2.
    f32x4 ev = {0.f, 0.f, 0.f, 0.f};
    u32x2 es0 = {0u, 0u}, es1 = {0u, 0u};
    auto epi_pack_step = [&](int par, int f, int m) __attribute__((always_inline)) {          // fragment f = 2 t + e of the finished pair
        const int t = f >> 1, e = f & 1;
        if (m == 0) {
            ev = acc[par][t][e];
            ev.x = act1(ev.x, slope); ev.y = act1(ev.y, slope);
        } else if (m == 1) {
            ev.z = act1(ev.z, slope); ev.w = act1(ev.w, slope);
        } else if (m == 2) {
            pk[t][e].x = pack2<BF16>(ev.x, ev.y);
        } else if (m == 3) {
            pk[t][e].y = pack2<BF16>(ev.z, ev.w);
        }
    };
    auto epi_store_step = [&](int i, int r, int m) __attribute__((always_inline)) {           // store i = 2 j + e of the pair whose first row is r
        const int j = i >> 1, e = i & 1;
        if (m == 0) es0 = __builtin_amdgcn_permlane16_swap(pk[2 * j][e].x, pk[2 * j + 1][e].x, false, false);
        else if (m == 1) es1 = __builtin_amdgcn_permlane16_swap(pk[2 * j][e].y, pk[2 * j + 1][e].y, false, false);
        else if (m == 2) {
            const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y0 + (size_t)e_n * y_img, 0, (int)y_img, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)es0.x, (int)es1.x, (int)es0.y, (int)es1.y}, yr, e_v[j] + (unsigned)(r + e) * rowb, 0, 0);
        }
    };
    auto epi_pack = [&](int par, int f) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; ++m) epi_pack_step(par, f, m);
    };
    auto epi_store = [&](int i, int r) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 3; ++m) epi_store_step(i, r, m);
    };
    auto store_offsets = [&](int nn_, int x0_, int y0_) __attribute__((always_inline)) {
        const bool inx = x0_ + px < p.W;
        const unsigned pix = (unsigned)((y0_ + wv * RW) * p.W + x0_ + px);
        const unsigned base = (pix * (unsigned)p.y0_pitch + (unsigned)p.y0_coff) * 2u;
#pragma unroll
        for (int j = 0; j < NT / 2; ++j) {
            const int ch = (2 * j + (kq & 1)) * 16 + (kq >> 1) * 8;
            e_v[j] = (inx && ch < p.cout_store) ? base + (unsigned)ch * 2u : OOB;
        }
        e_n = nn_;
    };
    for (int k = 0;; ++k) {
        const int tn = tile_index(k + 1);
        const bool more = tn >= 0;
        int nn = 0, nx0 = 0, ny0 = 0;
        if (more) tile_coords(tn, nn, nx0, ny0);
        const char* sb = smem + (k & 1) * STAGE;
        // B fragments: a ring of four (two rows each), read THREE groups ahead of their MFMAs; chunk 3's A fragments (NT = 4): a ring of three,
        // read TWO groups ahead.  Linear group index L = 20 rp + g over the tile's 40 groups
        constexpr int AHEAD = 3;
        i32x4 b[4][2];
        i32x4 a3[3][NCR == NCH ? 1 : NT];
        auto read_b = [&](int L) __attribute__((always_inline)) {
            const int rp_ = L / NG, g_ = L % NG, c_ = g_ / PAIRS, q_ = g_ % PAIRS;
#pragma unroll
            for (int e = 0; e < 2; ++e) b[L & 3][e] = *reinterpret_cast<const i32x4*>(sb + b_off[q_] + c_ * 32 + (2 * rp_ + e) * (TH * PIXB));
        };
        auto read_a = [&](int L) __attribute__((always_inline)) {
            const int g_ = L % NG, c_ = g_ / PAIRS, q_ = g_ % PAIRS;
            if (NCR < NCH && c_ >= NCR) {
#pragma unroll
                for (int t = 0; t < NT; ++t) a3[L % 3][NCR == NCH ? 0 : t] = *reinterpret_cast<const i32x4*>(w3 + (q_ * NT + t) * 1024);
            }
        };
#pragma unroll
        for (int L = 0; L < AHEAD; ++L) read_b(L);
        read_a(0); read_a(1);                  // (no-ops: the first chunk-3 group is L = 15)
#pragma unroll
        for (int rp = 0; rp < RW / 2; ++rp) {
            const int par = rp & 1;
            uint2 cen[NT][2];            // residual == input: the centre pixels of this pair's rows, 4 channels per tile
            if (EXT && res_in) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int e = 0; e < 2; ++e) cen[t][e] = *reinterpret_cast<const uint2*>(sb + c_off + t * 32 + (2 * rp + e) * (TH * PIXB));
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int c = g / PAIRS, q = g % PAIRS, L = rp * NG + g, cs = L & 3;
                if (L + AHEAD < (RW / 2) * NG) read_b(L + AHEAD);
                if (L + 2 < (RW / 2) * NG) read_a(L + 2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        if (c < NCR) {
                            if (g == 0) {
                                if (BF16) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=v"(acc[par][t][e]) : "a"(wr[c < NCR ? c : 0][q][t]), "v"(b[cs][e]), "v"(bia[t]));
                                else asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=v"(acc[par][t][e]) : "a"(wr[c < NCR ? c : 0][q][t]), "v"(b[cs][e]), "v"(bia[t]));
                            } else {
                                if (BF16) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[par][t][e]) : "a"(wr[c < NCR ? c : 0][q][t]), "v"(b[cs][e]));
                                else asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[par][t][e]) : "a"(wr[c < NCR ? c : 0][q][t]), "v"(b[cs][e]));
                            }
                        } else {
                            if (BF16) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[par][t][e]) : "v"(a3[L % 3][NCR == NCH ? 0 : t]), "v"(b[cs][e]));
                            else asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[par][t][e]) : "v"(a3[L % 3][NCR == NCH ? 0 : t]), "v"(b[cs][e]));
                        }
                        // the previous row pair's epilogue (rp == 0: the previous TILE's last pair), one micro-step behind each MFMA: its
                        // accumulators were last written 20 groups ago
                        // (the block's first tile: nothing is waiting, the steps run on whatever the registers hold and their stores are out of range)
                        {
                            const int m = 2 * t + e, r_prev = rp == 0 ? RW - 2 : 2 * rp - 2;
                            if (g >= 1 && g <= 2 * NT) epi_pack_step(par ^ 1, g - 1, m);
                            if (g >= 2 * NT + 1 && g < 2 * NT + 1 + SPP) epi_store_step(g - 2 * NT - 1, r_prev, m);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                if (rp == 0 && g < PPW) dma_piece(g, more, nn, nx0, ny0, (k + 1) & 1);      // the next tile's DMA, in the shadow of the matrix pipe
                if (rp == 0 && g == NG - 1) store_offsets(n, x0, y0);              // (behind the previous tile's last store)
                if (EXT && q == PAIRS - 1 && res_in && c < NT) {
                    // conv_s16_kernel's order: act(conv(x) + x) adds the centre pixels of chunk c to channel tile c BEHIND chunk c's
                    // groups.  The MFMAs above are asm: hipcc pads neither the read of their results (XDL write -> VALU read) nor the
                    // next group's read of what is written here
                    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
                    for (int e = 0; e < 2; ++e) acc[par][c < NT ? c : 0][e] += unpack4<BF16>(cen[c < NT ? c : 0][e]);
                    asm volatile("s_nop 3" ::: "memory");
                }
            }
        }
        // the next tile has landed: younger than its DMA are the stores of this tile's row pairs but the last
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((RW / 2 - 1) * SPP) : "memory");
        __builtin_amdgcn_s_barrier();
        if (!more) break;
        n = nn; x0 = nx0; y0 = ny0;
    }
    // the last tile's last row pair
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");     // (asm MFMAs: hipcc does not pad MFMA -> VALU reads of their results)
#pragma unroll
    for (int f = 0; f < 2 * NT; ++f) epi_pack(1, f);
#pragma unroll
    for (int i = 0; i < SPP; ++i) epi_store(i, RW - 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the trailing zero-fill DMA must not outlive the block)
}

// ---- conv48rp_kernel: RLFB's c3_r -- a 48 -> 48 3x3 + LeakyReLU + the block input (residual from HBM, post-activation), whose result
// only feeds a chain of two 1x1 convolutions (c5 48 -> 48, esa.conv1 48 -> 16: team04_rlfn.py:117-121, 76) -- on conv48r_kernel's plan:
// weights in accumulation registers, whole-pixel stages, row pairs, the finished pair's epilogue between the next pair's MFMA groups.
// What the shape adds:
//   * 16 x 16 tiles (a wave owns 4 rows = two pairs): input stage 18 x 18 x 96 B = 31 KB, and a RESIDUAL stage of the tile's own 16 x 16
//     pixels (24 KB) next to it, both double-buffered.  A wave stages exactly its own four residual rows (6 pieces) and is their only
//     reader, so the residual needs no barrier of its own; its DMA for the next tile is issued behind the groups in which the carried
//     epilogue (the previous tile's last pair) reads the slot it overwrites;
//   * the post images (18 + 6 KB, hi + lo for bf16) are resident in LDS; the chain runs row by row on the fp32 values exactly as
//     conv_s16_kernel's swap_epi_act does (same order of MFMAs: results are bit-identical);
//   * per tile and wave 8 + 6 DMA pieces in the first pair's groups, four stores per pair; the tile closes with ONE counted wait.
// LRS (round 4): the same kernel as the LR conv of a 48-channel network in bf16 -- `out_lr = LR_conv(body) + out_fea` (team04_rlfn.py:149) with
// `out_fea` and `out_lr` as hi + lo pairs (esr_conv_desc.hilo = RES | OUT): the residual stage holds the wave's own rows of BOTH tensors
// (2 x 24 KB, 12 pieces per wave), (conv + hi) + lo in conv_s16_kernel's order, then the activation, the result rounded to hi and
// bf16(v - hi) and stored as six stores per row pair; no post chain.  On conv_s16_kernel the residual pair was six extra stages per tile
// (0.26 ms at batch 32, 42 us on one image -- more than any other launch of RLFN).
template <bool BF16, bool LRS = false>
__global__ __launch_bounds__(256, 1) void conv48rp_kernel(const S16K p)
{
    constexpr int NT = 3, NCH = 3, PAIRS = 5, TH = 18, THY = 18, RW = 4, PNT1 = 3;
    constexpr int PIXB = NCH * 32;
    constexpr int NSLOT = TH * THY * (PIXB / 16);  // 1944
    constexpr int NPIECES = (NSLOT + 63) / 64;     // 31
    constexpr int STAGE = NPIECES * 1024;          // 31 744
    constexpr int RTEN = 16 * 16 * PIXB;           // 24 576: the tile's own pixels of a residual tensor, [row][px][96 B]
    constexpr int RSTAGE = LRS ? 2 * RTEN : RTEN;  // LRS: high parts, then low parts
    constexpr int RPT = RTEN / 4 / 1024;           // 6 pieces per wave and tensor: its own four rows
    constexpr int RPW = LRS ? 2 * RPT : RPT;
    constexpr int IPW = (NPIECES + 3) / 4;         // 8 input pieces per wave (wave 3: 7)
    constexpr int NG = NCH * PAIRS;
    constexpr int P1_IMG = NT * PNT1 * 1024, P2_IMG = PNT1 * 1024;
    constexpr int SLOT = STAGE + RSTAGE, OFF_POST = 2 * SLOT;                 // LDS map: [input 0][residual 0][input 1][residual 1][P1 hi, lo][P2 hi, lo]
    static_assert(IPW / 2 <= 7 && 7 + RPT <= NG, "DMA pieces fit the first pair's groups");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, kq = lane >> 4;
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    constexpr bool plo = BF16;                    // bf16: hi + lo post images (the host takes this kernel only when conv_s16_kernel would use them too)
    constexpr bool res_post = !LRS;               // RLFB: act(conv) + residual (team04_rlfn.py:117-119); LRS: act(conv + residual)

    // ---- prologue: conv weights through the (still unused) residual stages into registers, post images to their place ----------
    constexpr int WPIECES = NCH * PAIRS * NT;      // 45 KB <= slot 1 (55 KB), free until the first tile's DMA issue for the second tile
    static_assert(WPIECES * 1024 <= SLOT, "weights fit slot 1");
#pragma unroll
    for (int i = 0; i < (WPIECES + 3) / 4; ++i) {
        const int pc = wv + 4 * i;
        if (pc < WPIECES) dma_glb16(smem_lds + (unsigned)(SLOT + pc * 1024), p.wp + (size_t)pc * 1024 + lane * 16);
    }
    if (!LRS) {
        for (int pc = wv; pc < 2 * (P1_IMG / 1024); pc += 4) dma_glb16(smem_lds + (unsigned)(OFF_POST + pc * 1024), p.pw1 + (size_t)pc * 1024 + lane * 16);
        for (int pc = wv; pc < 2 * (P2_IMG / 1024); pc += 4) dma_glb16(smem_lds + (unsigned)(OFF_POST + 2 * P1_IMG + pc * 1024), p.pw2 + (size_t)pc * 1024 + lane * 16);
    }
    i32x4 wr[NCH][PAIRS][NT];
    f32x4 bia[NT], pb1[PNT1], pb2;
#pragma unroll
    for (int t = 0; t < NT; ++t) bia[t] = *reinterpret_cast<const f32x4*>(p.bias + t * 16 + kq * 4);
    if (!LRS) {
#pragma unroll
        for (int t = 0; t < PNT1; ++t) pb1[t] = *reinterpret_cast<const f32x4*>(p.pw1 + (size_t)2 * P1_IMG + (t * 16 + kq * 4) * 4);
        pb2 = *reinterpret_cast<const f32x4*>(p.pw2 + (size_t)2 * P2_IMG + (kq * 4) * 4);
    }
    const char* const img1 = smem + OFF_POST + lane * 16;                     // hi [k tile][out tile], lo at + P1_IMG
    const char* const img2 = smem + OFF_POST + 2 * P1_IMG + lane * 16;        // hi [k tile], lo at + P2_IMG

    const int ntiles = p.N * p.tiles_y * p.tiles_x;
    const int G = gridDim.x;
    auto tile_index = [&](int k) -> int { return s16_tile_index(k, ntiles); };
    auto tile_coords = [&](int t, int& n, int& x0, int& y0) __attribute__((always_inline)) { s16_tile_coords(t, p.magic_x, p.magic_y, p.tiles_x, p.tiles_y, 16, n, x0, y0); };
    const size_t img_bytes = (size_t)p.H * p.W * p.in_pitch * 2, res_bytes = (size_t)p.H * p.W * p.res_pitch * 2;
    auto dma_in = [&](int i, bool valid, int n, int x0, int y0, int slot) __attribute__((always_inline)) {
        const int pc = wv + 4 * i;
        if (i < IPW - 1 || pc < NPIECES) {                             // wave-uniform
            const unsigned sl = (unsigned)(pc * 64 + lane);
            const unsigned pixel = sl / 6u, part = sl - pixel * 6u;
            const unsigned ly = pixel / (unsigned)TH, lx = pixel - ly * (unsigned)TH;
            const int gy = y0 - 1 + (int)ly, gx = x0 - 1 + (int)lx;
            const bool ok = valid && sl < (unsigned)NSLOT && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            const unsigned voff = ok ? (unsigned)((gy * p.W + gx) * p.in_pitch + p.in_coff) * 2u + part * 16u : OOB;
            dma_buf16(smem_lds + (unsigned)(slot * SLOT + pc * 1024), voff, make_rsrc(p.x + (size_t)(valid ? n : 0) * img_bytes, img_bytes), 0u);
        }
    };
    auto dma_res = [&](int i, bool valid, int n, int x0, int y0, int slot) __attribute__((always_inline)) {      // piece i of this wave's own rows (LRS: 6 .. 11 = the low parts)
        const int ten = i / RPT, j = i - ten * RPT;
        const unsigned sl = (unsigned)((wv * RPT + j) * 64 + lane);     // slot of the residual stage: pixel sl / 6 = 16 row + col
        const unsigned pixel = sl / 6u, part = sl - pixel * 6u;
        const int gy = y0 + (int)(pixel >> 4), gx = x0 + (int)(pixel & 15u);
        const bool ok = valid && gy < p.H && gx < p.W;
        const unsigned voff = ok ? (unsigned)((gy * p.W + gx) * p.res_pitch + p.res_coff) * 2u + part * 16u : OOB;
        dma_buf16(smem_lds + (unsigned)(slot * SLOT + STAGE + ten * RTEN + (wv * RPT + j) * 1024), voff,
                  make_rsrc(p.res + (size_t)(ten ? p.res_lo_stride : 0) + (size_t)(valid ? n : 0) * res_bytes, res_bytes), 0u);
    };

    int n, x0, y0;
    {
        const int t0 = tile_index(0);
        if (t0 < 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
        tile_coords(t0, n, x0, y0);
#pragma unroll
        for (int i = 0; i < IPW; ++i) dma_in(i, true, n, x0, y0, 0);
#pragma unroll
        for (int i = 0; i < RPW; ++i) dma_res(i, true, n, x0, y0, 0);         // the first tile's residual too: nothing else waits for it before its first use
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int q = 0; q < PAIRS; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                wr[c][q][t] = *reinterpret_cast<const i32x4*>(smem + SLOT + ((c * PAIRS + q) * NT + t) * 1024 + lane * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // every wave holds its fragments: slot 1 may be written

    int b_off[PAIRS];
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
        const int tap = min(2 * q + (kq >> 1), 8);
        b_off[q] = ((wv * RW + tap / 3) * TH + px + tap % 3) * PIXB + (kq & 1) * 16;
    }
    const int r_off = ((wv * RW) * 16 + px) * PIXB + kq * 8;                 // residual of row 0 of the wave: channels 16 t + 4 kq .. at + 32 t
    const float slope = p.slope, p1s = p.p1_slope;
    // (LRS: the "post 1" output is the conv's own hi + lo pair: y0 and y0 + the pair's stride)
    const size_t p1_img = (size_t)p.H * p.W * (LRS ? p.y0_pitch : p.py1_pitch) * 2, p2_img = (size_t)p.H * p.W * p.py2_pitch * 2;
    const unsigned rowb1 = (unsigned)p.W * (unsigned)(LRS ? p.y0_pitch : p.py1_pitch) * 2u, rowb2 = (unsigned)p.W * (unsigned)p.py2_pitch * 2u;

    f32x4 acc[2][NT][2];
    // the finished pair's fp32 values (act(conv) + residual, then c5's result) live in ITS accumulators -- free until the pair after next
    // starts; the rounded post results in named registers
    uint2 q00, q01, q10, q11, q20, q21, z0, z1;
    uint2 l00, l01, l10, l11, l20, l21;      // LRS: the low parts of the rounded pair
    auto PKL = [&](int t, int e) __attribute__((always_inline)) -> uint2& { return t == 0 ? (e ? l01 : l00) : (t == 1 ? (e ? l11 : l10) : (e ? l21 : l20)); };
    auto PK1 = [&](int t, int e) __attribute__((always_inline)) -> uint2& { return t == 0 ? (e ? q01 : q00) : (t == 1 ? (e ? q11 : q10) : (e ? q21 : q20)); };
    auto PK2 = [&](int e) __attribute__((always_inline)) -> uint2& { return e ? z1 : z0; };
    unsigned e_vA = OOB, e_vB = OOB, e_v2 = OOB;
    int e_n = 0, e_slot = 0;             // image / residual stage of the tile whose epilogue is in flight
    // ---- the finished pair's epilogue as MICRO-STEPS (round 4) ---------------------------------------------------------------------
    // One wave per SIMD issues in order: VALU work placed as a clump behind a group's MFMAs runs while the matrix pipe idles, and this
    // epilogue is ~350 VALU instructions + 36 (fp16: 18) post MFMAs per row pair against the pair's 90 convolution MFMAs.  It is cut
    // into steps of <= ~10 VALU instructions (or one LDS read set, or one post MFMA), one or two behind EACH convolution MFMA of the next
    // pair (slot s = 6 g + 2 t + e, 0 .. 89): an MFMA occupies the pipe for 16 cycles, an independent VALU instruction issues in 4.
    // Per accumulator the order of operations is unchanged (conv_s16_kernel's: k tiles ascending, hi then lo): results stay bit-identical.
    uint2 rraw[4], rlo[4];               // residual fragments on their way from LDS (ring of four: fragment f + 3 is read while f is applied); LRS: their low parts
    i32x4 bsv[2][2];                     // [k tile & 1][row]: the fp32 fragment as the post 1x1's B operand (hi parts | lo parts)
    i32x4 pa[2][6];                      // post A fragments: [buffer][2 ot + lo] (post 1) / [lo][kt] (post 2)
    f32x4 d1[PNT1][2], d2[2];
    auto rd = [&](int f, int r) __attribute__((always_inline)) {
        const int t = f >> 1, e = f & 1;
        rraw[f & 3] = *reinterpret_cast<const uint2*>(smem + e_slot * SLOT + STAGE + r_off + t * 32 + (r + e) * (16 * PIXB));
        if (LRS) rlo[f & 3] = *reinterpret_cast<const uint2*>(smem + e_slot * SLOT + STAGE + RTEN + r_off + t * 32 + (r + e) * (16 * PIXB));
    };
    auto ra = [&](int par, int f, int h) __attribute__((always_inline)) {        // half h of fragment f: + residual, activation (in the pair's accumulators)
        const int t = f >> 1, e = f & 1;
        float ra_, rb_;
        unpack2<BF16>(h ? rraw[f & 3].y : rraw[f & 3].x, ra_, rb_);
        float va = h ? acc[par][t][e].z : acc[par][t][e].x, vb = h ? acc[par][t][e].w : acc[par][t][e].y;
        if (!res_post) { va += ra_; vb += rb_; }
        if (LRS) {                                                               // (conv + hi) + lo: conv_s16_kernel's order (its residual stages NT .. 2 NT - 1)
            float la_, lb_;
            unpack2<BF16>(h ? rlo[f & 3].y : rlo[f & 3].x, la_, lb_);
            va += la_; vb += lb_;
        }
        va = act1(va, slope); vb = act1(vb, slope);
        if (res_post) { va += ra_; vb += rb_; }
        if (h) { acc[par][t][e].z = va; acc[par][t][e].w = vb; } else { acc[par][t][e].x = va; acc[par][t][e].y = vb; }
    };
    auto hl = [&](i32x4& o, f32x4 v, int h) __attribute__((always_inline)) {     // the fp32 fragment as a B operand: h = 0 high parts, h = 1 low parts (bf16)
        if (h == 0) {
            o.x = (int)pack2<BF16>(v.x, v.y); o.y = (int)pack2<BF16>(v.z, v.w);
            if (!BF16) { o.z = 0; o.w = 0; }
        } else if (BF16) {
            float a, b, c, d;
            unpack2<BF16>((unsigned)o.x, a, b);
            unpack2<BF16>((unsigned)o.y, c, d);
            o.z = (int)pack2<BF16>(v.x - a, v.y - b); o.w = (int)pack2<BF16>(v.z - c, v.w - d);
        }
    };
    auto load_p1 = [&](int kt, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int ot = 0; ot < PNT1; ++ot) {
            pa[buf][2 * ot] = *reinterpret_cast<const i32x4*>(img1 + (kt * PNT1 + ot) * 1024);
            if (plo) pa[buf][2 * ot + 1] = *reinterpret_cast<const i32x4*>(img1 + P1_IMG + (kt * PNT1 + ot) * 1024);
        }
    };
    auto pm1 = [&](int kt, int i) __attribute__((always_inline)) {               // post-1 MFMA i of k tile kt: i = 0 .. 5 high images (ot, e), 6 .. 11 low images
        const int lo = i / 6, ot = (i % 6) >> 1, e = i & 1;
        if (lo && !plo) return;
        d1[ot][e] = mfma32<BF16>(pa[kt & 1][2 * ot + lo], bsv[kt & 1][e], (kt == 0 && !lo) ? pb1[ot] : d1[ot][e]);
    };
    auto fin = [&](int par, int ot, int e, int h) __attribute__((always_inline)) {   // c5's result: activation, fp32 back into the pair's accumulators, rounded into PK1
        if (h == 0) {
            f32x4 v = d1[ot][e];
            v.x = act1(v.x, p1s); v.y = act1(v.y, p1s); v.z = act1(v.z, p1s); v.w = act1(v.w, p1s);
            acc[par][ot][e] = v;
        } else {
            const f32x4 v = acc[par][ot][e];
            PK1(ot, e).x = pack2<BF16>(v.x, v.y);
            PK1(ot, e).y = pack2<BF16>(v.z, v.w);
        }
    };
    auto load_p2 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int kt = 0; kt < PNT1; ++kt) {
            pa[0][kt] = *reinterpret_cast<const i32x4*>(img2 + kt * 1024);
            if (plo) pa[1][kt] = *reinterpret_cast<const i32x4*>(img2 + P2_IMG + kt * 1024);
        }
    };
    auto pm2 = [&](int kt, int i) __attribute__((always_inline)) {               // post-2 MFMA i of k tile kt: (e, lo) = (i >> 1, i & 1)
        const int e = i >> 1, lo = i & 1;
        if (lo && !plo) return;
        d2[e] = mfma32<BF16>(pa[lo][kt], bsv[kt & 1][e], (kt == 0 && !lo) ? pb2 : d2[e]);
    };
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 es0 = {0u, 0u}, es1 = {0u, 0u};
    auto st = [&](int i, int r, int h) __attribute__((always_inline)) {          // store i of the pair (first row r): two swaps, then the store
        uint2 X = i == 0 ? q00 : (i == 1 ? q01 : (i == 2 ? q20 : z0)), Y = i == 0 ? q10 : (i == 1 ? q11 : (i == 2 ? q21 : z1));
        if (h == 0) es0 = __builtin_amdgcn_permlane16_swap(X.x, Y.x, false, false);
        else if (h == 1) es1 = __builtin_amdgcn_permlane16_swap(X.y, Y.y, false, false);
        else {
            const i32x4 o = i32x4{(int)es0.x, (int)es1.x, (int)es0.y, (int)es1.y};
            if (i < 3) {
                const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(p.py1 + (size_t)e_n * p1_img, 0, (int)p1_img, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(o, r1, (i < 2 ? e_vA + (unsigned)(r + i) * rowb1 : e_vB + (unsigned)r * rowb1), 0, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(o, __builtin_amdgcn_make_buffer_rsrc(p.py2 + (size_t)e_n * p2_img, 0, (int)p2_img, 0x00020000),
                                                       e_v2 + (unsigned)r * rowb2, 0, 0);
            }
        }
    };
    auto pkl = [&](int par, int f, int h) __attribute__((always_inline)) {      // LRS: fragment f rounded to its high (h = 0) and low (h = 1) parts
        const int t = f >> 1, e = f & 1;
        const f32x4 v = acc[par][t][e];
        if (h == 0) {
            PK1(t, e).x = pack2<BF16>(v.x, v.y);
            PK1(t, e).y = pack2<BF16>(v.z, v.w);
        } else {
            float a, b, c, d;
            unpack2<BF16>(PK1(t, e).x, a, b);
            unpack2<BF16>(PK1(t, e).y, c, d);
            PKL(t, e).x = pack2<BF16>(v.x - a, v.y - b);
            PKL(t, e).y = pack2<BF16>(v.z - c, v.w - d);
        }
    };
    auto stl = [&](int i, int r, int h) __attribute__((always_inline)) {         // LRS: store i = 0 .. 2 of the high parts, 3 .. 5 of the low parts
        const int lo = i / 3, k3 = i - 3 * lo;
        uint2 X, Y;
        if (lo == 0) { X = k3 == 0 ? q00 : (k3 == 1 ? q01 : q20); Y = k3 == 0 ? q10 : (k3 == 1 ? q11 : q21); }
        else { X = k3 == 0 ? l00 : (k3 == 1 ? l01 : l20); Y = k3 == 0 ? l10 : (k3 == 1 ? l11 : l21); }
        if (h == 0) es0 = __builtin_amdgcn_permlane16_swap(X.x, Y.x, false, false);
        else if (h == 1) es1 = __builtin_amdgcn_permlane16_swap(X.y, Y.y, false, false);
        else {
            const i32x4 o = i32x4{(int)es0.x, (int)es1.x, (int)es0.y, (int)es1.y};
            const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(p.y0 + (size_t)(lo ? p.res_lo_stride : 0) + (size_t)e_n * p1_img, 0, (int)p1_img, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(o, r1, (k3 < 2 ? e_vA + (unsigned)(r + k3) * rowb1 : e_vB + (unsigned)r * rowb1), 0, 0);
        }
    };
    // the schedule: what runs behind convolution MFMA s of the next pair (par = the FINISHED pair's accumulators, first row r)
    auto micro = [&](auto par_, auto r_, auto s_) __attribute__((always_inline)) {
        constexpr int par = decltype(par_)::value, r = decltype(r_)::value, s = decltype(s_)::value;
        if constexpr (LRS) {
            // residual as below (slots 0 .. 17); rounding of fragment f in slots 18 + 2 f, + 1; the six stores from slot 30 on, three slots each
            if constexpr (s >= 6 && s < 18) ra(par, (s - 6) >> 1, (s - 6) & 1);
            if constexpr (s < 12 && (s & 1) == 0) rd(s >> 1, r);
            if constexpr (s >= 18 && s < 30) pkl(par, (s - 18) >> 1, (s - 18) & 1);
            if constexpr (s >= 30 && s < 48) stl((s - 30) / 3, r, (s - 30) % 3);
            return;
        }
        // 0 .. 17: the residual: fragment f is read at slot 2 f and applied in slots 2 f + 6, 2 f + 7
        if constexpr (s >= 6 && s < 18) ra(par, (s - 6) >> 1, (s - 6) & 1);
        if constexpr (s < 12 && (s & 1) == 0) rd(s >> 1, r);
        if constexpr (s == 12) load_p1(0, 0);
        // 18 .. 59: post 1: the B operands of k tile 0 in slots 18 .. 21, of k tile kt > 0 beside the MFMAs of k tile kt - 1 (26 + 12 (kt - 1) ..);
        // the 12 MFMAs of k tile kt in slots 24 + 12 kt ..
        static_for<PNT1>([&](auto kt_) __attribute__((always_inline)) {
            constexpr int kt = decltype(kt_)::value;
            constexpr int h0 = kt == 0 ? 18 : 24 + 12 * (kt - 1) + 2;
            if constexpr (s >= h0 && s < h0 + 4) hl(bsv[kt & 1][(s - h0) >> 1], acc[par][kt][(s - h0) >> 1], (s - h0) & 1);
            constexpr int m0 = 24 + 12 * kt;
            if constexpr (s >= m0 && s < m0 + 12) pm1(kt, s - m0);
            if constexpr (kt + 1 < PNT1 && s == m0 + 6) load_p1(kt + 1, (kt + 1) & 1);     // (buffer (kt + 1) & 1 was last read by k tile kt - 1)
        });
        if constexpr (s == 60) load_p2();
        // 60 .. 71: c5's result (ot, e) in slots 60 + 4 ot + 2 e, + 1;  its B operand for post 2 one out tile later;  post 2's MFMAs
        // (k tile kt, 4 each) in 72 + 4 kt ..;  stores: PK1's from slot 76 on (three slots each), PK2's at the end
        if constexpr (s >= 60 && s < 72) fin(par, (s - 60) >> 2, ((s - 60) >> 1) & 1, (s - 60) & 1);
        // (k tile 2 shares its B buffer with k tile 0: its operands follow k tile 0's MFMAs, slots 76 .. 79)
        if constexpr (s >= 64 && s < 72) hl(bsv[((s - 64) >> 2) & 1][((s - 64) >> 1) & 1], acc[par][(s - 64) >> 2][((s - 64) >> 1) & 1], (s - 64) & 1);
        if constexpr (s >= 76 && s < 80) hl(bsv[0][((s - 76) >> 1) & 1], acc[par][2][((s - 76) >> 1) & 1], (s - 76) & 1);
        if constexpr (s >= 72 && s < 84) pm2((s - 72) >> 2, (s - 72) & 3);
        if constexpr (s >= 76 && s < 85) st((s - 76) / 3, r, (s - 76) % 3);
        if constexpr (s == 85) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                PK2(e).x = pack2<BF16>(d2[e].x, d2[e].y);
                PK2(e).y = pack2<BF16>(d2[e].z, d2[e].w);
            }
        }
        if constexpr (s >= 87 && s < 90) st(3, r, s - 87);
    };
    auto store_offsets = [&](int nn_, int x0_, int y0_, int slot_) __attribute__((always_inline)) {
        const bool inx = x0_ + px < p.W;
        const unsigned pix = (unsigned)((y0_ + wv * RW) * p.W + x0_ + px);
        const unsigned b1 = (pix * (unsigned)(LRS ? p.y0_pitch : p.py1_pitch) + (unsigned)(LRS ? p.y0_coff : p.py1_coff)) * 2u;
        const unsigned b2 = (pix * (unsigned)p.py2_pitch + (unsigned)p.py2_coff) * 2u;
        const int chA = (kq & 1) * 16 + (kq >> 1) * 8, chB = 32 + (kq >> 1) * 8, ch2 = (kq >> 1) * 8;
        const int c1max = LRS ? p.cout_store : p.p1_cout8;
        e_vA = (inx && chA < c1max) ? b1 + (unsigned)chA * 2u : OOB;
        e_vB = (inx && chB < c1max) ? b1 + (unsigned)chB * 2u + ((kq & 1) ? rowb1 : 0u) : OOB;
        e_v2 = (!LRS && inx && ch2 < p.p2_cout8) ? b2 + (unsigned)ch2 * 2u + ((kq & 1) ? rowb2 : 0u) : OOB;
        e_n = nn_; e_slot = slot_;
    };
    for (int k = 0;; ++k) {
        const int tn = tile_index(k + 1);
        const bool more = tn >= 0;
        int nn = 0, nx0 = 0, ny0 = 0;
        if (more) tile_coords(tn, nn, nx0, ny0);
        const char* sb = smem + (k & 1) * SLOT;
        constexpr int AHEAD = 3;
        i32x4 b[4][2];
        auto read_b = [&](int L) __attribute__((always_inline)) {
            const int rp_ = L / NG, g_ = L % NG, c_ = g_ / PAIRS, q_ = g_ % PAIRS;
#pragma unroll
            for (int e = 0; e < 2; ++e) b[L & 3][e] = *reinterpret_cast<const i32x4*>(sb + b_off[q_] + c_ * 32 + (2 * rp_ + e) * (TH * PIXB));
        };
#pragma unroll
        for (int L = 0; L < AHEAD; ++L) read_b(L);
        // (one lambda instance per row pair: as ONE doubly unrolled loop the body exceeded hipcc's full-unroll budget, the loops stayed
        // rolled and accumulators / fragment ring were indexed dynamically -- through scratch)
        auto run_pair = [&](auto rp_tag) __attribute__((always_inline)) {
            constexpr int rp = decltype(rp_tag)::value;
            constexpr int par = rp & 1;
            // this tile's store offsets and residual stage: behind the carried epilogue's last store (first pair, slot 89), ahead of the
            // first step of this tile's own epilogue (the residual reads of slot 0)
            if constexpr (rp == 1) store_offsets(n, x0, y0, k & 1);
            static_for<NG>([&](auto g_) __attribute__((always_inline)) {
                constexpr int g = decltype(g_)::value;
                constexpr int c = g / PAIRS, q = g % PAIRS, L = rp * NG + g, cs = L & 3;
                if constexpr (L + AHEAD < (RW / 2) * NG) read_b(L + AHEAD);
                __builtin_amdgcn_sched_barrier(0);
                static_for<2 * NT>([&](auto m_) __attribute__((always_inline)) {
                    constexpr int t = decltype(m_)::value >> 1, e = decltype(m_)::value & 1;
                    {
                        if (g == 0) {
                            if (BF16) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(b[cs][e]), "v"(bia[t]));
                            else asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(b[cs][e]), "v"(bia[t]));
                        } else {
                            if (BF16) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(b[cs][e]));
                            else asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(b[cs][e]));
                        }
                        // the finished pair's epilogue (rp == 0: the previous TILE's last pair), one step behind each MFMA.  (The block's first
                        // tile: nothing is waiting, the steps run on whatever the registers hold and their stores are out of range.)
                        micro(std::integral_constant<int, par ^ 1>{}, std::integral_constant<int, (rp == 0 ? RW - 2 : 2 * rp - 2)>{}, std::integral_constant<int, 6 * g + 2 * t + e>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                // the next tile's DMA in the first pair: input pieces first, the wave's residual rows behind the groups (1 .. 6) in which the
                // carried epilogue reads the residual stage they overwrite
                if (rp == 0 && 2 * g < IPW) {
                    dma_in(2 * g, more, nn, nx0, ny0, (k + 1) & 1);
                    if (2 * g + 1 < IPW) dma_in(2 * g + 1, more, nn, nx0, ny0, (k + 1) & 1);
                }
                if (rp == 0 && g >= 7 && g < 7 + RPT) {
                    dma_res(g - 7, more, nn, nx0, ny0, (k + 1) & 1);
                    if (LRS) dma_res(g - 7 + RPT, more, nn, nx0, ny0, (k + 1) & 1);
                }
            });
        };
        run_pair(std::integral_constant<int, 0>{});
        run_pair(std::integral_constant<int, 1>{});
        // the next tile's stages have landed: younger than their last DMA piece (group 12 of the first pair) are the carried epilogue's
        // four stores (groups 12 - 14) and the four stores of this tile's first pair
        // (LRS: the carried epilogue's stores leave in groups 5 - 7, AHEAD of the last DMA piece: only the first pair's six stores are younger)
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LRS ? 6 : 8) : "memory");
        __builtin_amdgcn_s_barrier();
        if (!more) break;
        n = nn; x0 = nx0; y0 = ny0;
    }
    // the last tile's last pair (its residual stage: e_slot): the same steps, back to back
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    static_for<90>([&](auto s_) __attribute__((always_inline)) { micro(std::integral_constant<int, 1>{}, std::integral_constant<int, RW - 2>{}, s_); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool BF16, bool LRS = false>
int launch_conv48rp(const S16K& k, hipStream_t st)
{
    constexpr int LDS = LRS ? 2 * (31 * 1024 + 2 * 24576) : 2 * 31 * 1024 + 2 * 24576 + 2 * 9 * 1024 + 2 * 3 * 1024;
    static std::atomic<unsigned> attr_set[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return ESR_ERR_LAUNCH;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv48rp_kernel<BF16, LRS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            esr_set_err("hipFuncSetAttribute(conv48rp_kernel, MaxDynamicSharedMemorySize)", e);
            return ESR_ERR_LAUNCH;
        }
        attr_set[dev].store(1u, std::memory_order_relaxed);
    }
    const int ntiles = k.N * k.tiles_x * k.tiles_y;
    const int grid = ntiles < 256 ? ntiles : 256;
    esr_note_kernel("conv48rp_kernel<%s, %s>", esr_tf(BF16), esr_tf(LRS));
    hipLaunchKernelGGL((conv48rp_kernel<BF16, LRS>), dim3(grid), dim3(256), LDS, st, k);
    return esr_check_launch("conv48rp_kernel launch");
}

template <bool BF16, int NT, bool EXT, int RW = 8, int FX = -1>
int launch_conv48r_fx(const S16K& k, hipStream_t st)
{
    // [two input stages][RW = 4: 45 KB where the weight blob is staged][border table]
    constexpr int STAGES = RW == 8 ? 2 * 58 * 1024 : 2 * 31 * 1024 + 15 * NT * 1024;
    const int LDS = STAGES + ((EXT && k.border) ? NT * 1024 : 0);
    static std::atomic<unsigned> attr_set[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return ESR_ERR_LAUNCH;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv48r_kernel<BF16, NT, EXT, RW, FX>), hipFuncAttributeMaxDynamicSharedMemorySize, STAGES + NT * 1024);
        if (e != hipSuccess) {
            esr_set_err("hipFuncSetAttribute(conv48r_kernel, MaxDynamicSharedMemorySize)", e);
            return ESR_ERR_LAUNCH;
        }
        attr_set[dev].store(1u, std::memory_order_relaxed);
    }
    const int ntiles = k.N * k.tiles_x * k.tiles_y;
    const int grid = ntiles < 256 ? ntiles : 256;
    esr_note_kernel("conv48r_kernel<%s, %d, %s, %d, %d>", esr_tf(BF16), NT, esr_tf(EXT), RW, FX);
    hipLaunchKernelGGL((conv48r_kernel<BF16, NT, EXT, RW, FX>), dim3(grid), dim3(256), LDS, st, k);
    return esr_check_launch("conv48r_kernel launch");
}

// the descriptor's switches -> the specialisation that has them compiled in, if there is one (ESDB's two shapes), else the run-time kernel
template <bool BF16, int NT, bool EXT, int RW = 8>
int launch_conv48r(const S16K& k, hipStream_t st)
{
    if constexpr (EXT) {
        const int fx = (k.act == ESR_ACT_GELU ? 1 : 0) | (k.border != nullptr ? 2 : 0) | (k.res_in ? 4 : 0);
        if (NT == 3 && fx == 7) return launch_conv48r_fx<BF16, NT, EXT, RW, (NT == 3 ? 7 : -1)>(k, st);
        if (NT == 2 && fx == 3) return launch_conv48r_fx<BF16, NT, EXT, RW, (NT == 2 ? 3 : -1)>(k, st);
    }
    return launch_conv48r_fx<BF16, NT, EXT, RW, -1>(k, st);
}

template <bool BF16, int FX = -1>
int launch_conv48rq_fx(const S16K& k, hipStream_t st)
{
    // [two input stages][45 KB where the weight blob is staged][border table][post images: hi (+ lo)]
    constexpr int LDS = 2 * 31 * 1024 + 45 * 1024 + 3 * 1024 + (BF16 ? 2 : 1) * 6 * 1024;
    static std::atomic<unsigned> attr_set[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return ESR_ERR_LAUNCH;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv48rq_kernel<BF16, FX>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            esr_set_err("hipFuncSetAttribute(conv48rq_kernel, MaxDynamicSharedMemorySize)", e);
            return ESR_ERR_LAUNCH;
        }
        attr_set[dev].store(1u, std::memory_order_relaxed);
    }
    const int ntiles = k.N * k.tiles_x * k.tiles_y;
    const int grid = ntiles < 256 ? ntiles : 256;
    esr_note_kernel("conv48rq_kernel<%s, %d>", esr_tf(BF16), FX);
    hipLaunchKernelGGL((conv48rq_kernel<BF16, FX>), dim3(grid), dim3(256), LDS, st, k);
    return esr_check_launch("conv48rq_kernel launch");
}

template <bool BF16>
int launch_conv48rq(const S16K& k, hipStream_t st)
{
    const int fx = (k.act == ESR_ACT_GELU ? 1 : 0) | (k.border != nullptr ? 2 : 0) | (k.res_in ? 4 : 0) | (k.p1_gelu ? 8 : 0);
    if (fx == 15) return launch_conv48rq_fx<BF16, 15>(k, st);
    return launch_conv48rq_fx<BF16, -1>(k, st);
}

template <bool BF16, int NT, bool EXT>
int launch_conv64r(const S16K& k, hipStream_t st)
{
    // [stage 0][stage 1 | the weight blob as staged, chunk 3 (NT = 4) resident behind stage 1]
    constexpr int STAGE = 51 * 1024, BLOB = 4 * 5 * NT * 1024;
    constexpr int LDS = STAGE + (BLOB > STAGE ? BLOB : STAGE);
    static std::atomic<unsigned> attr_set[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return ESR_ERR_LAUNCH;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv64r_kernel<BF16, NT, EXT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            esr_set_err("hipFuncSetAttribute(conv64r_kernel, MaxDynamicSharedMemorySize)", e);
            return ESR_ERR_LAUNCH;
        }
        attr_set[dev].store(1u, std::memory_order_relaxed);
    }
    const int ntiles = k.N * k.tiles_x * k.tiles_y;
    const int grid = ntiles < 256 ? ntiles : 256;
    esr_note_kernel("conv64r_kernel<%s, %d, %s>", esr_tf(BF16), NT, esr_tf(EXT));
    hipLaunchKernelGGL((conv64r_kernel<BF16, NT, EXT>), dim3(grid), dim3(256), LDS, st, k);
    return esr_check_launch("conv64r_kernel launch");
}

}  // namespace

// ---- entry points for esr_conv2d_s16 (esr_s16.hip) ------------------------------------------------------------------------------------------
int esr_launch_conv48rp(const S16K& k, bool bf16, bool lrs, hipStream_t st)
{
    if (lrs) return launch_conv48rp<true, true>(k, st);                      // (bf16 only: esr_conv_desc.hilo)
    return bf16 ? launch_conv48rp<true>(k, st) : launch_conv48rp<false>(k, st);
}

int esr_launch_conv48r(const S16K& k, bool bf16, int nt, bool ext, int rw, hipStream_t st)
{
    if (rw == 4) {
        if (nt == 2) return bf16 ? launch_conv48r<true, 2, true, 4>(k, st) : launch_conv48r<false, 2, true, 4>(k, st);
        if (ext) return bf16 ? launch_conv48r<true, 3, true, 4>(k, st) : launch_conv48r<false, 3, true, 4>(k, st);
        return bf16 ? launch_conv48r<true, 3, false, 4>(k, st) : launch_conv48r<false, 3, false, 4>(k, st);
    }
    if (nt == 2) return bf16 ? launch_conv48r<true, 2, true>(k, st) : launch_conv48r<false, 2, true>(k, st);
    if (ext) return bf16 ? launch_conv48r<true, 3, true>(k, st) : launch_conv48r<false, 3, true>(k, st);
    return bf16 ? launch_conv48r<true, 3, false>(k, st) : launch_conv48r<false, 3, false>(k, st);
}

int esr_launch_conv48rq(const S16K& k, hipStream_t st) { return launch_conv48rq<false>(k, st); }      // (fp16 storage only: conv48rq_takes)

int esr_launch_conv64r(const S16K& k, bool bf16, hipStream_t st)
{
    return bf16 ? launch_conv64r<true, 2, true>(k, st) : launch_conv64r<false, 2, true>(k, st);
}
