// esr_s16.hip -- the 16-bit-STORAGE convolution of libesr_hip.so (BASELINE.json configs [2]-[4]: bf16 / fp16).
// Interface: include/esr_hip.h (esr_conv_desc.storage != ESR_STORE_F32).  Design notes: DESIGN.md section 4.
//
// Activations live in HBM as NHWC bf16 / fp16; the matrix products run on v_mfma_f32_16x16x32_{bf16,f16} with fp32
// accumulation; bias, residual, activation are applied in fp32 and the result is rounded ONCE (RNE) when it is stored.
// At 16x the fp32 matrix rate the layer is HBM-bound, so the kernel is built around the memory pipe:
//   * the input halo tile goes global -> LDS by DMA (`buffer_load_dwordx4 ... lds`): no staging VGPRs, no ds_write, no
//     conversion (the storage type IS the operand type); out-of-image halo pixels use an out-of-range buffer offset and
//     the hardware writes zeros (the convolution's zero padding);
//   * a ring of R = 3..8 K stages (16 channels = 32 bytes per pixel each, as many as fit next to the weights): while
//     stage s is multiplied, s+1 .. s+R-1 are in flight, across tile boundaries of the persistent block;
//   * the layer's whole weight set is resident in LDS for the life of the block (<= 80 KB: 64 -> 64 channels, 3x3);
//   * one barrier per stage with an EXACT `s_waitcnt vmcnt(N)`: N counts every vector-memory instruction the wave issued
//     after the DMA of the stage it needs (younger stages, the previous tile's stores, residual loads), so nothing but
//     the needed stage is waited for; the previous tile's epilogue runs behind the DMA issue of the next tile's stage.
// GEMM view and fragment maps: D[cout][pixel], A = weights, B = 16 consecutive pixels of one image row, D gives lane
// (px, kq) 4 consecutive output channels of one pixel -- exactly as conv_f32_kernel (esr_hip.hip).  K slots of one MFMA:
// lane (i, kq) holds 8 consecutive k = 8 channels (16 bytes) of ONE tap: kq & 1 selects the channel half of the chunk,
// kq >> 1 the tap of a tap PAIR, so the 9 taps of a chunk take 5 MFMAs (the 10th tap slot holds zero weights).
// For 1x1 convolutions the second tap slot is not wasted: it carries the LOW part of the weights (w = hi + lo, both
// 16-bit), so 1x1 layers see effectively fp32-accurate weights at no cost.  3x3 weights are rounded with error
// diffusion over the 9 taps of each (cout, cin) filter (esr_pack_conv_s16): the filter's DC gain, which dominates the
// response to natural features, keeps fp32 accuracy.  Measured effect on RLFN bf16: tools/emulate_s16.py, DESIGN.md.
//
// Kernels of this file (all share the packed weights, fragment maps, order of operations and rounding: their results are bit-identical
// where their shapes overlap, which the tests use -- a batch and its single images take different kernels):
//   conv_s16_kernel<NT, KS, NW, bf16|f16, GRES, PNT1, PNT2, HILO>   the general one (ring of 16-channel stages, weights in LDS, 2 waves per SIMD);
//       PNT1 / PNT2: 1x1 post chain on the fp32 tile; HILO (bf16): hi + lo pairs for the long skip (esr_conv_desc.hilo, LAB_NOTES 9.4)
//   conv48r_kernel<bf16|f16, NT, EXT, RW>        3x3 over 48 channels, weights in registers, one wave per SIMD, whole-pixel stages, row pairs
//   conv48rp_kernel<bf16|f16, LRS>               ... + residual from HBM staged per wave + RLFB's 1x1 chain; LRS: the LR conv on hi + lo pairs
//   conv48rq_kernel<f16>                         ... + residual == input, border table, GELU and ONE post 1x1 (ESDB c{j}_r + the next distillation conv)
//   conv64r_kernel<bf16|f16, 2, EXT>             ... over 64 channels with two output tiles (RFDB c4; 160-byte LDS pixels)
//   (the 64 -> 64 shapes, with and without RFDB's post 1x1, moved to esr_c64m.hip in round 6: v_mfma_f32_32x32x16; conv64rq_kernel and
//   conv64r_kernel<.., 4, ..> of rounds 4 / 5 are gone)
// The one-wave-per-SIMD kernels run a finished row pair's epilogue as micro-steps behind each MFMA of the next pair (LAB_NOTES 9.5).
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

// (struct S16K: esr_s16_dev.h -- shared with esr_c64m.hip)


// Pipeline (per block, stages s = (tile, 16-channel chunk) in order; R = ring slots):
//   top of stage s   stage s has landed (previous sync).  DMA of stage s+R-1 into the slot stage s-1 just released; then the
//                    EPILOGUE of the previous tile if s is a tile's first stage (its stores are issued behind the DMA, so the
//                    memory pipe never waits for them); then, if s is the tile's last stage, the residual loads of THIS tile
//   compute(s)       5 tap-pair MFMA groups per chunk (1 for 1x1) from ring slot s % R and the resident weights
//   sync             s_waitcnt vmcnt(N) + s_barrier with N = the exact number of vector-memory instructions this wave has
//                    issued AFTER the DMA of stage s+1 (younger stages, epilogue stores, residual loads): loads and stores
//                    retire in issue order, so stage s+1 has landed while everything younger stays in flight -- R-1 stages
//                    (20 KB each for a 3x3 on 16x32 tiles) are on their way from HBM at any time.
// Every vector-memory instruction is issued unconditionally (invalid lanes use out-of-range buffer offsets: loads return
// zero, stores are dropped), which is what makes the count exact.
// GRES: the launch reads a residual from HBM (its 8 NT registers exist only in these variants, which in exchange keep a
// single set of MFMA operand fragments: they are memory-bound twice over).
// PNT1 / PNT2: output tiles of a chain of 1x1 convolutions evaluated in the epilogue on the fp32 result tile (RLFB: c3_r -> c5 ->
// esa.conv1, team04_rlfn.py:117-121 / :76; RFDB: c{j}_r -> c{j+1}_d, rfdn_baseline/block.py:150-160).  The D fragment of the
// producing GEMM (lane (px, kq): 4 channels of one pixel, fp32) becomes the B operand of the next WITHOUT leaving the lane and
// without being rounded: k slots (kq, 0..3) carry the 16-bit high parts of the four values, (kq, 4..7) their low parts, so the
// intermediate tensor (RLFB's u, which nothing else reads) is neither stored nor quantised.
// HILO (bf16, the long skip head -> (+) -> upsampler; LAB_NOTES 9.4): tensors stored as hi + lo pairs.  Input: the K loop runs over both
// halves of the pixel against the same resident weights (w (hi + lo) = w hi + w lo); residual: 2 NT staged chunks, both added; output:
// a second set of stores with the low parts bf16(v - hi).
template <int NT, int KS, int NW, bool BF16, bool GRES, int PNT1 = 0, int PNT2 = 0, bool HILO = false>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void conv_s16_kernel(const S16K p)
{
    static_assert(PNT2 == 0 || PNT1 > 0, "post 2 needs post 1");
    static_assert(!HILO || (PNT2 == 0 && !GRES && KS == 3), "hi + lo tensors: the 3x3, at most one post 1x1 (the head of RFDN / BSRN with block 1's first distillation conv)");
    constexpr int HALO = KS / 2;
    constexpr int TH = TILE + 2 * HALO;          // halo tile width = LDS row pitch in pixels
    // NW = 4: 16 x 16 tiles and TWO independent blocks per CU (each with its own copy of the weights: only where that fits 80 KB) --
    // the two waves of a SIMD then belong to different blocks and do not share a stage barrier
    constexpr int TILE_H = NW == 4 ? 16 : 32;    // tile height; wave wv owns rows RW wv .. RW wv + RW-1
    constexpr int RW = TILE_H / NW;              // rows per wave: 4 (4 / 8 waves) or 2 (16 waves: 4 per SIMD hide each other's LDS / issue stalls)
    static_assert(NW == 4 || NW == 8 || NW == 16, "4, 8 or 16 waves per tile");
    constexpr int THY = TILE_H + 2 * HALO;
    constexpr int NPX = TH * THY;
    constexpr int NPIECES = (NPX + 31) / 32;     // 1 KB DMA pieces: 32 halo pixels x 32 bytes (lane pair = the 16 channels of a pixel)
    constexpr int STAGE_BYTES = NPIECES * 1024;  // [halo pixel][half][8 channels]
    constexpr int PPW = (NPIECES + NW - 1) / NW; // pieces per wave and stage (waves >= NPIECES % NW: one fewer)
    // 1x1: no halo, so a wave can stage exactly the pixels it computes (pieces PPW wv ..): nothing staged is shared between waves,
    // the stage loop needs NO barrier and the eight waves drift freely (the weights, biases and tables in LDS are read-only)
    constexpr bool OWN_PIECES = KS == 1 && NPIECES == NW * PPW && PPW * 32 == RW * TH;
    static_assert(KS != 1 || OWN_PIECES, "1x1: pieces = the wave's own rows");
    constexpr int TAPS = KS * KS;
    constexpr int PAIRS = (TAPS + 1) / 2;
    constexpr int W_CHUNK_BYTES = PAIRS * NT * 1024;   // [pair][tile][lane][16 B]
    constexpr int RES_LOADS = NT * RW;           // residual loads per wave and tile (8 bytes per lane each)

    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15;
    const int kq = lane >> 4;
    // The launch parameters are re-read from the kernarg segment where they are used (tile set-up, epilogue) instead of living
    // in SGPRs across the stage loop: with ~40 parameters + loop state hipcc spilled 70-160 SGPRs to VGPR lanes and the
    // v_readlane / v_writelane traffic was a third of the kernel's VALU instructions.  The asm makes the pointer opaque, so the
    // scalar loads (a few s_load_dwordx8 per tile) cannot be hoisted back out of the loop.
    typedef const __attribute__((address_space(4))) S16K* kparg_t;
    const kparg_t kp0 = (kparg_t)__builtin_amdgcn_kernarg_segment_ptr();
    auto KP = [&]() __attribute__((always_inline)) -> kparg_t {
        kparg_t q = kp0;
        asm volatile("" : "+s"(q));
        return q;
    };
    const int R = p.ring;
    const int w_main = (HILO ? p.w_chunks : p.nchunks) * W_CHUNK_BYTES;
    // post images: [post 1: NT k-tiles x PNT1 tiles, hi (then lo)][post 2: PNT1 k-tiles x PNT2 tiles, hi (then lo)][biases, 1 KB]
    constexpr int P1_IMG = NT * PNT1 * 1024, P2_IMG = PNT1 * PNT2 * 1024;
    const int plo = (PNT1 > 0 && p.post_lo) ? 2 : 1;
    // [weights][post images (PNT1 > 0)][bias KB: post biases, the conv's own bias in the upper half][border table NT KB (p.border)][ring]
    const int bias_at = w_main + (PNT1 > 0 ? plo * (P1_IMG + P2_IMG) : 0);
    const int w_bytes = bias_at + 1024 + (p.border ? NT * 1024 : 0);
    float* const sbias = reinterpret_cast<float*>(smem + bias_at + 512);
    float* const btab = reinterpret_cast<float*>(smem + bias_at + 1024);  // border bias table [16][NT * 16]
    const char* const pimg1 = smem + w_main;
    const char* const pimg2 = pimg1 + plo * P1_IMG;
    float* const pbias = reinterpret_cast<float*>(smem + w_main + plo * (P1_IMG + P2_IMG));
    char* const ring = smem + w_bytes;
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned ring_lds = smem_lds + (unsigned)w_bytes;

    // ---- tile walk (persistent; XCD-aware order as in conv_f32_kernel) ---------------------------------------------
    const int ntiles = p.N * p.tiles_y * p.tiles_x;
    const int G = gridDim.x;
    auto tile_index = [&](int k) -> int { return s16_tile_index(k, ntiles); };
    auto tile_coords = [&](int t, int& n, int& x0, int& y0) __attribute__((always_inline)) {
        const kparg_t q = KP();
        const unsigned mx = q->magic_x, my = q->magic_y;
        const int tsx = q->tiles_x, tsy = q->tiles_y;
        const int tq = mx ? (int)__umulhi((unsigned)t, mx) : t;          // t / tiles_x without the 25-instruction division
        const int tx = t - tq * tsx;
        n = my ? (int)__umulhi((unsigned)tq, my) : tq;
        const int ty = tq - n * tsy;
        x0 = tx * TILE;
        y0 = ty * TILE_H;
    };

    // ---- load cursor: the (tile, chunk) stage requested next ---------------------------------------------------------
    // piece pc = wv + NW * i of a stage: halo pixels 32 pc + (lane >> 1), channel half lane & 1 -- a lane PAIR reads the 32
    // contiguous bytes of a pixel's chunk (32-byte runs cost the memory pipe 12 % less than 16-byte ones: tools/abl nomfma_r*).
    // Behind the block's last tile the cursor keeps issuing (out-of-range offsets: zeros into a ring slot nobody reads), so
    // every stage carries the same number of DMA instructions and the vmcnt arithmetic has no special cases.
    const int n_my = (NPIECES % NW == 0 || wv < NPIECES % NW) ? PPW : PPW - 1;     // wave-uniform
    int lk = 0;                   // tile iteration of the cursor
    int lc = 0;                   // chunk of the cursor
    int lcc = 0;                  // ... within its input segment
    unsigned lsoff = 0;           // ... as the DMA's scalar byte offset
    int lslot = 0;
    bool lvalid;
    static_assert(PPW <= 3, "lvr0..2");
    unsigned lvoff[PPW];                   // input
    unsigned lvr0 = OOB, lvr1 = OOB, lvr2 = OOB;     // residual (p.nres > 0: staged as extra chunks, added from LDS -- no registers in flight); scalars: as an array hipcc kept it in scratch
    auto LVR = [&](int i) __attribute__((always_inline)) -> unsigned& { return i == 0 ? lvr0 : (i == 1 ? lvr1 : lvr2); };
    i32x4 lrsrc, lrsrcr = {0, 0, 0, 0};
    const int nstages = p.nchunks + p.nres;       // stages per tile
    auto cursor_tile = [&]() __attribute__((always_inline)) {
        const int t = tile_index(lk);
        lvalid = t >= 0;
        if (!lvalid) {
#pragma unroll
            for (int r = 0; r < PPW; ++r) { lvoff[r] = OOB; LVR(r) = OOB; }
            return;
        }
        int n, x0, y0;
        tile_coords(t, n, x0, y0);
        const kparg_t q = KP();
        const int qH = q->H, qW = q->W, qpitch = q->in_pitch, qcoff = q->in_coff;
        const size_t img_bytes = (size_t)qH * qW * qpitch * 2;
        lrsrc = make_rsrc(q->x + (size_t)n * img_bytes, img_bytes);
        const bool withres = q->nres > 0;
        const int qrp = q->res_pitch, qrc = q->res_coff;
        if (withres) {
            const size_t res_bytes = (size_t)qH * qW * qrp * 2;
            lrsrcr = make_rsrc(q->res + (size_t)n * res_bytes, res_bytes);
        }
#pragma unroll
        for (int r = 0; r < PPW; ++r) {
            const int pc = OWN_PIECES ? wv * PPW + r : wv + NW * r;
            const int plane = lane & 1;                       // channel half
            const int pl = pc * 32 + (lane >> 1);
            const int ly = pl / TH, lx = pl - ly * TH;
            const int gy = y0 - HALO + ly, gx = x0 - HALO + lx;
            const bool ok = pc < NPIECES && pl < NPX && (unsigned)gy < (unsigned)qH && (unsigned)gx < (unsigned)qW;
            lvoff[r] = ok ? (unsigned)((gy * qW + gx) * qpitch + qcoff + 8 * plane) * 2u : OOB;
            LVR(r) = (ok && withres) ? (unsigned)((gy * qW + gx) * qrp + qrc + 8 * plane) * 2u : OOB;
        }
    };
    auto dma_piece = [&](int i) __attribute__((always_inline)) {       // piece i of this wave of the cursor's stage, into ring slot lslot
        const int pc = OWN_PIECES ? wv * PPW + i : wv + NW * i;
        if (NPIECES % NW == 0 || i < PPW - 1 || pc < NPIECES) {         // wave-uniform
            const unsigned dst = ring_lds + (unsigned)(lslot * STAGE_BYTES) + (unsigned)pc * 1024u;
            if (lc >= p.nchunks) {                                      // a residual chunk
                int rc = lc - p.nchunks;
                i32x4 rs = lrsrcr;
                if (HILO && rc >= NT) {                                  // ... of the low-part tensor: the buffer base moves on (see cursor_advance)
                    rc -= NT;
                    const unsigned long long b = ((unsigned long long)(unsigned)rs.x | ((unsigned long long)((unsigned)rs.y & 0xffffu) << 32)) + (unsigned long long)p.res_lo_stride;
                    rs.x = (int)(unsigned)b;
                    rs.y = (int)(((unsigned)rs.y & 0xffff0000u) | ((unsigned)(b >> 32) & 0xffffu));
                }
                dma_buf16(dst, LVR(i), rs, (unsigned)rc * 32u);
            } else {
                dma_buf16(dst, lvoff[i], lrsrc, lsoff);
            }
        }
    };
    auto cursor_advance = [&]() __attribute__((always_inline)) {
        lslot = lslot == R - 1 ? 0 : lslot + 1;
        if (++lcc == p.seg_chunks) {                // (esr_conv_desc.in_seg_*: the next chunk lies in the next tensor of the concat)
            // the buffer BASE moves on: the hardware's range check covers the scalar offset too, so a segment stride in soffset
            // would put every later segment out of range (num_records = one tensor's image)
            lcc = 0;
            lsoff = 0;
            const unsigned long long b = ((unsigned long long)(unsigned)lrsrc.x | ((unsigned long long)((unsigned)lrsrc.y & 0xffffu) << 32)) + (unsigned long long)p.seg_stride;
            lrsrc.x = (int)(unsigned)b;
            lrsrc.y = (int)((unsigned)(b >> 32) & 0xffffu);
        } else {
            lsoff += 32u;
        }
        if (++lc == nstages) {
            lc = 0;
            lcc = 0;
            lsoff = 0;
            ++lk;
            cursor_tile();
        }
    };

    // ---- prologue: weights (resident), the first R-1 stages ---------------------------------------------------------
    {
        const int wpieces = w_main / 1024;
        for (int pc = wv; pc < wpieces; pc += NW)
            dma_glb16(smem_lds + (unsigned)pc * 1024u, p.wp + (size_t)pc * 1024 + lane * 16);
        if (PNT1 > 0) {
            // blob: hi images, lo images, bias; resident: hi (then lo when post_lo)
            const int n1 = plo * P1_IMG / 1024, n2 = plo * P2_IMG / 1024;
            for (int pc = wv; pc < n1; pc += NW)
                dma_glb16(smem_lds + (unsigned)(w_main + pc * 1024), p.pw1 + (size_t)pc * 1024 + lane * 16);
            for (int pc = wv; pc < n2; pc += NW)
                dma_glb16(smem_lds + (unsigned)(w_main + plo * P1_IMG + pc * 1024), p.pw2 + (size_t)pc * 1024 + lane * 16);
        }
        // The biases and the border table travel as LDS-DMA pieces too (16 bytes per lane, lanes past the end masked off).  As per-thread loads +
        // LDS writes (until round 5) each of them was a dependent round trip -- load, s_waitcnt vmcnt(0), ds_write -- that also drained the
        // weights' DMA queue in front of the first tile's requests: three to four L2 latencies, 10 % of a single image's 1x1 launch.
        if (wv == 0) {
            const unsigned pb_lds = smem_lds + (unsigned)(w_main + plo * (P1_IMG + P2_IMG));
            if (PNT1 > 0 && lane < PNT1 * 4) dma_glb16(pb_lds, p.pw1 + 2 * P1_IMG + lane * 16);
            if (PNT2 > 0 && lane < PNT2 * 4) dma_glb16(pb_lds + PNT1 * 64, p.pw2 + 2 * P2_IMG + lane * 16);
            if (lane < NT * 4) dma_glb16(smem_lds + (unsigned)(bias_at + 512), reinterpret_cast<const char*>(p.bias) + lane * 16);
        }
        if (p.border)
            for (int pc = wv; pc < NT; pc += NW)
                dma_glb16(smem_lds + (unsigned)(bias_at + 1024 + pc * 1024), reinterpret_cast<const char*>(p.border) + (size_t)pc * 1024 + lane * 16);
    }
    cursor_tile();
    if (!lvalid) return;                 // block without tiles (grid <= ntiles: does not happen)
    for (int i = 0; i < R - 1; ++i) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) dma_piece(j);
        cursor_advance();
    }
    wait_vm_dyn((R - 2) * n_my);         // the weights and stage 0 have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the biases / border table written above
    __builtin_amdgcn_s_barrier();

    // (the bias is re-read from LDS by each tile's first MFMA group: NT * 4 registers less across the whole loop)

    // lane-constant LDS offsets of the B fragments: pair q reads tap min(2q + (kq >> 1), TAPS - 1), channel half kq & 1
    int b_off[PAIRS];
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
        const int tap = min(2 * q + (kq >> 1), TAPS - 1);
        b_off[q] = (((wv * RW) + tap / KS) * TH + px + tap % KS) * 32 + (kq & 1) * 16;
    }
    const int a_off = lane * 16;
    // the centre pixel of this lane's accumulator rows in the staged tile: channels 16c + 4kq .. +3 of chunk c
    const int c_off = ((wv * RW + HALO) * TH + px + HALO) * 32 + (kq >> 1) * 16 + (kq & 1) * 8;

    // The plain NHWC epilogue (no post chain) lives INSIDE the first MFMA group of the next tile's first stage (swap_epi):
    // row by row, activation / rounding of the finished tile's accumulators right before the MFMAs that overwrite them, the
    // D fragments made store-shaped by v_permlane16_swap (no LDS, no waits), their stores in the shadow of the matrix pipe.
    // The post chain (PNT1 / PNT2) runs there too, row by row on the activated fp32 fragments.  Only the pixel-shuffle epilogue
    // of the network's last convolution is a phase of its own in front of the stage's compute.
    const bool swap_epi = p.out_layout != ESR_NCHW_SHUFFLE4;
    constexpr int SWAP_STORES = (NT / 2) * RW + (NT & 1) * (RW / 2);
    constexpr int P1_STORES = (PNT1 / 2) * RW + (PNT1 & 1) * (RW / 2), P2_STORES = PNT2 > 0 ? RW / 2 : 0;
    static_assert(PNT2 <= 1, "post 2: one tile");
    const int epi_stores = p.out_layout == ESR_NCHW_SHUFFLE4 ? RW * NT
                           : ((PNT1 == 0 || p.store_main) ? ((p.split < p.cout_store || (HILO && p.hilo_out)) ? 2 : 1) * SWAP_STORES : 0) + P1_STORES + P2_STORES;   // stores per wave and tile
    const unsigned hmask = (1u << (R - 2)) - 1u;
    unsigned hist_rs = 0, hist_st = 0;   // bit i: stage s - i was a tile's first stage (residual loads) / carried an epilogue's stores

    f32x4 acc[NT][RW];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[tt][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint2 rv[GRES ? NT : 1][RW];         // residual of the current tile in D-fragment layout (hidden asm loads)
#pragma unroll
    for (int tt = 0; tt < (GRES ? NT : 1); ++tt)
#pragma unroll
        for (int r = 0; r < RW; ++r) rv[tt][r] = uint2{0u, 0u};

    // Issued in a tile's FIRST stage, behind the previous tile's epilogue (which frees rv) and in front of the stage's DMA: by
    // the time the tile's own epilogue wants them, nchunks stages of DMA are younger and stay in flight.  hipcc believes the
    // asm's outputs are valid at once, so NOTHING may make it copy these registers before the wait: there is exactly ONE load
    // site per kernel (two sites feeding one consumer meet in a phi, and the copies of the losing site run before the data
    // has arrived -- seen with cin = 16), it lies behind the last use of the previous values (no interference, the loop-carried
    // registers coalesce), and the GRES variants stay clear of spills (tools/dbg/s16_shape_probe.py, test_s16_conv_more_tiles_*).
    auto load_residual = [&](int n, int x0, int y0, bool have) __attribute__((always_inline)) {
        if (!GRES) return;
        const kparg_t q = KP();
        const int qH = q->H, qW = q->W, qrp = q->res_pitch, qrc = q->res_coff, qcs = q->cout_store;
        const size_t res_img = (size_t)qH * qW * qrp * 2;
        const i32x4 rr = make_rsrc(q->res + (size_t)n * res_img, res_img);
        i32x4 rru;
        rru.x = __builtin_amdgcn_readfirstlane(rr.x); rru.y = __builtin_amdgcn_readfirstlane(rr.y);
        rru.z = __builtin_amdgcn_readfirstlane(rr.z); rru.w = __builtin_amdgcn_readfirstlane(rr.w);
        // rows below the image fall past num_records and read zeros; one add per row, one per channel tile
        const unsigned rbase = (unsigned)((y0 + wv * RW) * qW + x0) * (unsigned)qrp * 2u + (__umul24(px, qrp) + (unsigned)(qrc + kq * 4)) * 2u;
        const unsigned rowb = (unsigned)qW * (unsigned)qrp * 2u;
        const bool inx = have && x0 + px < qW;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const int cb = tt * 16 + kq * 4;
                const unsigned vo = (inx && cb < qcs) ? rbase + (unsigned)r * rowb + (unsigned)tt * 32u : OOB;
                // "+v": the destination is TIED to the loop-carried register of rv, so the value never has to be copied into it
                // at the loop latch (with "=v" hipcc gave the asm fresh registers and moved them over before the data was there)
                asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "+v"(rv[GRES ? tt : 0][r]) : "v"(vo), "s"(rru) : "memory");
            }
        }
    };

    const bool act_gelu = p.act == ESR_ACT_GELU;
    // waits for the finished tile's residual (loaded in its first stage: the tile's nchunks stages of DMA are younger)
    auto wait_residual = [&]() __attribute__((always_inline)) {
        if (!GRES) return;
        wait_vm_dyn(p.nchunks * n_my);
#pragma unroll
        for (int tt = 0; tt < (GRES ? NT : 1); ++tt)
#pragma unroll
            for (int r = 0; r < RW; ++r) asm volatile("" : "+v"(rv[tt][r]));      // uses below stay behind the wait
    };

    // GELU, applied to the accumulators at the end of the tile's last stage; the epilogue then sees an identity activation.
    // One fragment at a time (sched_barrier): register pressure stays flat.  A residual loaded from HBM can only follow the
    // GELU (post-activation): its registers are in flight here and must not be touched -- not even by an empty asm, whose
    // re-definition makes hipcc copy them on the paths that skip it (the host rejects GELU + pre-activation residual from HBM;
    // the residual == input case comes from the staged tile and is already in the accumulators).
    auto gelu_inplace = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                acc[tt][r] = gelu16x4(acc[tt][r]);
                __builtin_amdgcn_sched_barrier(0);
            }
    };

    // esr_conv_desc.border_bias: tiles on the image border add the table row of each pixel's outside-mask (row 0 = zeros for
    // the interior pixels of such a tile), at the end of the tile's last stage -- in front of residual, GELU and the epilogue
    auto border_fix = [&](int x0, int y0) __attribute__((always_inline)) {
        const kparg_t q = KP();
        const int qH = q->H, qW = q->W;
        if (!(x0 == 0 || x0 + TILE >= qW || y0 == 0 || y0 + TILE_H >= qH)) return;
        const int gx = x0 + px;
        const int cm = (gx == 0 ? 1 : 0) | (gx == qW - 1 ? 2 : 0);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int gy = y0 + wv * RW + r;
            const int m = cm | (gy == 0 ? 4 : 0) | (gy == qH - 1 ? 8 : 0);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) acc[tt][r] += *reinterpret_cast<const f32x4*>(btab + m * (NT * 16) + tt * 16 + kq * 4);
        }
    };

    // epilogue as a phase: the pixel-shuffle output (fp32 NCHW) of the network's last convolution
    auto epilogue = [&](int n, int x0, int y0) __attribute__((always_inline)) {
        const kparg_t q = KP();
        const int qH = q->H, qW = q->W, qcs = q->cout_store;
        const float qslope = act_gelu ? 1.f : q->slope;
        if constexpr (PNT1 == 0) {
            // out[n, t, 4gy + kq, 4gx + 0..3] = channel 16t + 4kq + j: the D fragment is one dwordx4 of 4 adjacent HR pixels
            const int gx = x0 + px;
            const unsigned W4 = (unsigned)qW * 4u, H4 = (unsigned)qH * 4u;
            const size_t y0_img = (size_t)qcs * qH * qW * 4;                 // NCHW fp32: cout / 16 planes of 4H x 4W
            const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(q->y0 + (size_t)n * y0_img, 0, (int)y0_img, 0x00020000);
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const int gy = y0 + wv * RW + r;
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const bool ok = gy < qH && gx < qW && tt * 16 + kq * 4 < qcs;
                    f32x4 v = acc[tt][r];
                    v.x = act1(v.x, qslope); v.y = act1(v.y, qslope);
                    v.z = act1(v.z, qslope); v.w = act1(v.w, qslope);
                    const unsigned vo = ok ? (((unsigned)tt * H4 + (unsigned)gy * 4u + (unsigned)kq) * W4 + (unsigned)gx * 4u) * 4u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), yr, vo, 0, 0);
                }
            }
            return;
        }
    };

    int slot = 0;
    bool pend = false;                   // a finished tile waits for its epilogue
    int pn = 0, px0 = 0, py0 = 0;
    bool have = false;
    int n = 0, x0 = 0, y0 = 0;

    // ---- swap epilogue: set-up per tile, then one call per accumulator row (inside the first MFMA group) ---------------
    // v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of another: for two D fragments X, Y
    // (lane (px, kq): channels 4kq..4kq+3 of pixel px, 8 bytes) two swaps leave lane (px, kq) with 16 CONTIGUOUS bytes --
    // channels 8(kq >> 1) .. +7 of X (kq even) or of Y (kq odd).  X, Y = channel tiles 2j, 2j+1 of one row (shape A: 64
    // contiguous bytes per pixel and instruction), or the odd last tile of rows r, r+1 (shape B: 32 bytes per pixel).
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    constexpr int NPAIR = NT / 2;
    unsigned vbA0[NPAIR > 0 ? NPAIR : 1], vbA1[NPAIR > 0 ? NPAIR : 1], vbB0 = OOB, vbB1 = OOB;
    unsigned e_rowb0 = 0, e_rowb1 = 0;
    char* e_y0 = nullptr; char* e_y1 = nullptr;
    int e_y0n = 0, e_y1n = 0;
    float e_slope = 0.f;
    int e_res_mode = 0;
    bool e_split = false, e_main = true, e_hilo = false;
    // post outputs: post 1 = PNT1 tiles (pairs + an odd last tile), post 2 = one tile (rows paired)
    constexpr int NPAIR1 = PNT1 / 2;
    unsigned vp1A[NPAIR1 > 0 ? NPAIR1 : 1], vp1B = OOB, vp2B = OOB, e_rowbp1 = 0, e_rowbp2 = 0;
    char* e_p1 = nullptr; char* e_p2 = nullptr;
    int e_p1n = 0, e_p2n = 0;
    float e_s1 = 1.f;
    bool e_g1 = false;
    auto swap_epi_setup = [&]() __attribute__((always_inline)) {
        const kparg_t q = KP();
        const int qH = q->H, qW = q->W, qcs = q->cout_store, qsplit = q->split;
        const int qy0p = q->y0_pitch, qy0c = q->y0_coff, qy1p = q->y1_pitch, qy1c = q->y1_coff;
        e_slope = (act_gelu || (q->nres > 0 && q->res_mode == ESR_RES_POST_ACT)) ? 1.f : q->slope;     // applied in place already
        e_res_mode = q->res_mode;
        e_split = qsplit < qcs;
        e_hilo = HILO && q->hilo_out != 0;
        const size_t y0_img = (size_t)qH * qW * qy0p * 2, y1_img = (size_t)qH * qW * qy1p * 2;
        e_y0 = q->y0 + (size_t)pn * y0_img; e_y0n = (int)y0_img;
        e_y1 = q->y1 + (size_t)pn * y1_img; e_y1n = (int)y1_img;
        e_rowb0 = (unsigned)qW * (unsigned)qy0p * 2u;
        e_rowb1 = (unsigned)qW * (unsigned)qy1p * 2u;
        const unsigned srow = (unsigned)((py0 + wv * RW) * qW + px0);            // wave-uniform: pixel (row 0, px = 0) of this wave
        const unsigned s0 = srow * (unsigned)qy0p * 2u, s1 = srow * (unsigned)qy1p * 2u;
        const unsigned l0 = (__umul24(px, qy0p) + (unsigned)qy0c) * 2u, l1 = (__umul24(px, qy1p) + (unsigned)(qy1c - qsplit)) * 2u;
        const bool inx = pend && px0 + px < qW;       // nothing pending (GRES: the block's first stage): every store out of range
        // rows below the image fall past num_records (= the image's bytes): dropped by the hardware
#pragma unroll
        for (int j = 0; j < NPAIR; ++j) {
            const int ch = (2 * j + (kq & 1)) * 16 + (kq >> 1) * 8;
            vbA0[j] = (inx && ch < qsplit) ? s0 + l0 + (unsigned)ch * 2u : OOB;
            vbA1[j] = (inx && ch >= qsplit && ch < qcs) ? s1 + l1 + (unsigned)ch * 2u : OOB;
            if (HILO && e_hilo) vbA1[j] = (inx && ch < qcs) ? s1 + l1 + (unsigned)ch * 2u : OOB;        // (split == cout_store: l1 counts from y1_coff - cout_store)
        }
        if (NT & 1) {
            const int ch = (NT - 1) * 16 + (kq >> 1) * 8;
            vbB0 = (inx && ch < qsplit) ? s0 + l0 + (unsigned)ch * 2u + ((kq & 1) ? e_rowb0 : 0u) : OOB;
            vbB1 = (inx && ch >= qsplit && ch < qcs) ? s1 + l1 + (unsigned)ch * 2u + ((kq & 1) ? e_rowb1 : 0u) : OOB;
            if (HILO && e_hilo) vbB1 = (inx && ch < qcs) ? s1 + l1 + (unsigned)ch * 2u + ((kq & 1) ? e_rowb1 : 0u) : OOB;
        }
        if (PNT1 > 0) {
            e_main = q->store_main != 0;
            e_s1 = q->p1_slope;
            e_g1 = q->p1_gelu != 0;
            const int qp1p = q->py1_pitch, qp1c = q->py1_coff, qp1n = q->p1_cout8;
            const size_t p1_img = (size_t)qH * qW * qp1p * 2;
            e_p1 = q->py1 + (size_t)pn * p1_img; e_p1n = (int)p1_img;
            e_rowbp1 = (unsigned)qW * (unsigned)qp1p * 2u;
            const unsigned sp = srow * (unsigned)qp1p * 2u + (__umul24(px, qp1p) + (unsigned)qp1c) * 2u;
#pragma unroll
            for (int j = 0; j < NPAIR1; ++j) {
                const int ch = (2 * j + (kq & 1)) * 16 + (kq >> 1) * 8;
                vp1A[j] = (inx && ch < qp1n) ? sp + (unsigned)ch * 2u : OOB;
            }
            if (PNT1 & 1) {
                const int ch = (PNT1 - 1) * 16 + (kq >> 1) * 8;
                vp1B = (inx && ch < qp1n) ? sp + (unsigned)ch * 2u + ((kq & 1) ? e_rowbp1 : 0u) : OOB;
            }
            if (PNT2 > 0) {
                const int qp2p = q->py2_pitch, qp2c = q->py2_coff, qp2n = q->p2_cout8;
                const size_t p2_img = (size_t)qH * qW * qp2p * 2;
                e_p2 = q->py2 + (size_t)pn * p2_img; e_p2n = (int)p2_img;
                e_rowbp2 = (unsigned)qW * (unsigned)qp2p * 2u;
                const int ch = (kq >> 1) * 8;
                vp2B = (inx && ch < qp2n) ? srow * (unsigned)qp2p * 2u + (__umul24(px, qp2p) + (unsigned)(qp2c + ch)) * 2u + ((kq & 1) ? e_rowbp2 : 0u) : OOB;
            }
        }
    };
    uint2 pk[NT][2];                     // rounded rows r - 1 (even), r (odd) of the finished tile
    uint2 pkl[HILO ? NT : 1][2];         // HILO: their low parts
    uint2 pk1[PNT1 > 0 ? PNT1 : 1][2], pk2[2];          // ... of the post chain's results
    auto swap16 = [&](uint2 X, uint2 Y) __attribute__((always_inline)) -> i32x4 { return s16_swap16(X, Y); };
    auto store16 = [&](uint2 X, uint2 Y, unsigned v0, unsigned v1, int r) __attribute__((always_inline)) {
        const i32x4 o = swap16(X, Y);
        __builtin_amdgcn_raw_buffer_store_b128(o, __builtin_amdgcn_make_buffer_rsrc(e_y0, 0, e_y0n, 0x00020000), v0 + (unsigned)r * e_rowb0, 0, 0);
        if (e_split) __builtin_amdgcn_raw_buffer_store_b128(o, __builtin_amdgcn_make_buffer_rsrc(e_y1, 0, e_y1n, 0x00020000), v1 + (unsigned)r * e_rowb1, 0, 0);
    };
    auto store16lo = [&](uint2 X, uint2 Y, unsigned v1, int r) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_buffer_store_b128(swap16(X, Y), __builtin_amdgcn_make_buffer_rsrc(e_y1, 0, e_y1n, 0x00020000), v1 + (unsigned)r * e_rowb1, 0, 0);
    };
    // the fp32 fragment as the B operand of the post 1x1: k slots 0..3 = the 16-bit high parts, 4..7 = the low parts
    auto hilo = [&](f32x4 v) __attribute__((always_inline)) -> i32x4 {
        const unsigned h0 = pack2<BF16>(v.x, v.y), h1 = pack2<BF16>(v.z, v.w);
        if (!BF16) return i32x4{(int)h0, (int)h1, 0, 0};           // fp16: the high parts carry 11 bits, as much as anything stored
        float a, b, c, d;
        unpack2<BF16>(h0, a, b);
        unpack2<BF16>(h1, c, d);
        return i32x4{(int)h0, (int)h1, (int)pack2<BF16>(v.x - a, v.y - b), (int)pack2<BF16>(v.z - c, v.w - d)};
    };
    auto swap_epi_act = [&](int r) __attribute__((always_inline)) {       // reads acc[.][r]
        f32x4 u[PNT1 > 0 ? NT : 1];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            f32x4 v = acc[tt][r];
            f32x4 rf = {0.f, 0.f, 0.f, 0.f};
            if (GRES) rf = unpack4<BF16>(rv[GRES ? tt : 0][r]);
            if (GRES && e_res_mode == ESR_RES_PRE_ACT) v += rf;
            v.x = act1(v.x, e_slope); v.y = act1(v.y, e_slope);
            v.z = act1(v.z, e_slope); v.w = act1(v.w, e_slope);
            if (GRES && e_res_mode == ESR_RES_POST_ACT) v += rf;
            pk[tt][r & 1].x = pack2<BF16>(v.x, v.y);
            pk[tt][r & 1].y = pack2<BF16>(v.z, v.w);
            if (HILO) {
                float a, b, c, d;
                unpack2<BF16>(pk[tt][r & 1].x, a, b);
                unpack2<BF16>(pk[tt][r & 1].y, c, d);
                pkl[HILO ? tt : 0][r & 1].x = pack2<BF16>(v.x - a, v.y - b);
                pkl[HILO ? tt : 0][r & 1].y = pack2<BF16>(v.z - c, v.w - d);
            }
            if (PNT1 > 0) u[tt] = v;
        }
        if constexpr (PNT1 > 0) {
            // ---- post chain on this row's fp32 result (RLFB: c3_r -> c5 -> esa.conv1; RFDB / ESDB: the next distillation conv) ------
            f32x4 d1[PNT1];
#pragma unroll
            for (int ot = 0; ot < PNT1; ++ot) d1[ot] = *reinterpret_cast<const f32x4*>(pbias + ot * 16 + kq * 4);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const i32x4 bsv = hilo(u[kt]);
#pragma unroll
                for (int ot = 0; ot < PNT1; ++ot) {
                    d1[ot] = mfma32<BF16>(*reinterpret_cast<const i32x4*>(pimg1 + (kt * PNT1 + ot) * 1024 + a_off), bsv, d1[ot]);
                    if (plo == 2)
                        d1[ot] = mfma32<BF16>(*reinterpret_cast<const i32x4*>(pimg1 + P1_IMG + (kt * PNT1 + ot) * 1024 + a_off), bsv, d1[ot]);
                }
            }
#pragma unroll
            for (int ot = 0; ot < PNT1; ++ot) {
                f32x4 v = d1[ot];
                if (e_g1) v = gelu16x4(v);
                else { v.x = act1(v.x, e_s1); v.y = act1(v.y, e_s1); v.z = act1(v.z, e_s1); v.w = act1(v.w, e_s1); }
                d1[ot] = v;
                pk1[ot][r & 1].x = pack2<BF16>(v.x, v.y);
                pk1[ot][r & 1].y = pack2<BF16>(v.z, v.w);
            }
            if (PNT2 > 0) {
                f32x4 d2 = *reinterpret_cast<const f32x4*>(pbias + PNT1 * 16 + kq * 4);
#pragma unroll
                for (int kt = 0; kt < PNT1; ++kt) {
                    const i32x4 bsv = hilo(d1[kt]);
                    d2 = mfma32<BF16>(*reinterpret_cast<const i32x4*>(pimg2 + kt * PNT2 * 1024 + a_off), bsv, d2);
                    if (plo == 2) d2 = mfma32<BF16>(*reinterpret_cast<const i32x4*>(pimg2 + P2_IMG + kt * PNT2 * 1024 + a_off), bsv, d2);
                }
                pk2[r & 1].x = pack2<BF16>(d2.x, d2.y);
                pk2[r & 1].y = pack2<BF16>(d2.z, d2.w);
            }
        }
    };
    auto swap_epi_store = [&](int r) __attribute__((always_inline)) {                    // rows r - 1, r (r odd)
        if (PNT1 == 0 || e_main) {
#pragma unroll
            for (int j = 0; j < NPAIR; ++j) {
                store16(pk[2 * j][0], pk[2 * j + 1][0], vbA0[j], vbA1[j], r - 1);
                store16(pk[2 * j][1], pk[2 * j + 1][1], vbA0[j], vbA1[j], r);
            }
            if (NT & 1) store16(pk[NT - 1][0], pk[NT - 1][1], vbB0, vbB1, r - 1);
            if (HILO && e_hilo) {
#pragma unroll
                for (int j = 0; j < NPAIR; ++j) {
                    store16lo(pkl[HILO ? 2 * j : 0][0], pkl[HILO ? 2 * j + 1 : 0][0], vbA1[j], r - 1);
                    store16lo(pkl[HILO ? 2 * j : 0][1], pkl[HILO ? 2 * j + 1 : 0][1], vbA1[j], r);
                }
                if (NT & 1) store16lo(pkl[HILO ? NT - 1 : 0][0], pkl[HILO ? NT - 1 : 0][1], vbB1, r - 1);
            }
        }
        if constexpr (PNT1 > 0) {
            const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(e_p1, 0, e_p1n, 0x00020000);
#pragma unroll
            for (int j = 0; j < NPAIR1; ++j) {
                __builtin_amdgcn_raw_buffer_store_b128(swap16(pk1[2 * j][0], pk1[2 * j + 1][0]), r1, vp1A[j] + (unsigned)(r - 1) * e_rowbp1, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(swap16(pk1[2 * j][1], pk1[2 * j + 1][1]), r1, vp1A[j] + (unsigned)r * e_rowbp1, 0, 0);
            }
            if (PNT1 & 1)
                __builtin_amdgcn_raw_buffer_store_b128(swap16(pk1[PNT1 - 1][0], pk1[PNT1 - 1][1]), r1, vp1B + (unsigned)(r - 1) * e_rowbp1, 0, 0);
            if (PNT2 > 0)
                __builtin_amdgcn_raw_buffer_store_b128(swap16(pk2[0], pk2[1]), __builtin_amdgcn_make_buffer_rsrc(e_p2, 0, e_p2n, 0x00020000),
                                                       vp2B + (unsigned)(r - 1) * e_rowbp2, 0, 0);
        }
    };

    // ---- one stage: MFMA groups from ring slot `slot`, the cursor's DMA pieces between them ------------------------------
    auto compute = [&](auto epi_tag, int c, bool last) __attribute__((always_inline)) {
        constexpr bool EPI = decltype(epi_tag)::value;           // first stage of a tile with the swap epilogue inside
        const bool first = EPI || c == 0;
        const char* sb = ring + slot * STAGE_BYTES;
        const char* wc = smem + ((HILO && c >= p.w_chunks) ? c - p.w_chunks : c) * W_CHUNK_BYTES + a_off;      // HILO input: the low half meets the same weights

        constexpr int NBUF = NW == 16 ? 1 : 2;          // fragment sets: the read of pair q+1 runs under the MFMAs of pair q
        i32x4 a[NBUF][NT], b[NBUF][RW];
        auto load_frag = [&](int buf, int q) __attribute__((always_inline)) {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) a[buf][tt] = *reinterpret_cast<const i32x4*>(wc + (q * NT + tt) * 1024);
#pragma unroll
            for (int r = 0; r < RW; ++r) b[buf][r] = *reinterpret_cast<const i32x4*>(sb + b_off[q] + r * (TH * 32));
        };
        if (NBUF == 2) load_frag(0, 0);
        if (EPI) {
            wait_residual();
            swap_epi_setup();
        }
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) {
            const int cs = NBUF == 2 ? (q & 1) : 0;
            if (NBUF == 2) {
                if (q + 1 < PAIRS) load_frag(cs ^ 1, q + 1);
                __builtin_amdgcn_sched_barrier(0);      // keep the prefetch above this pair's MFMAs
            } else {
                load_frag(0, q);
            }
            if (q == 0 && EPI) {
                // the finished tile's rows leave just in front of the MFMAs that overwrite their accumulators (C input = bias)
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    swap_epi_act(r);
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) acc[tt][r] = mfma32<BF16>(a[cs][tt], b[cs][r], *reinterpret_cast<const f32x4*>(sbias + tt * 16 + kq * 4));
                    if (r & 1) swap_epi_store(r);
                }
            } else if (q == 0 && first) {
                // first MFMA group of a tile: the accumulator input is the bias
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int r = 0; r < RW; ++r)
                        acc[tt][r] = mfma32<BF16>(a[cs][tt], b[cs][r], *reinterpret_cast<const f32x4*>(sbias + tt * 16 + kq * 4));
            } else {
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int r = 0; r < RW; ++r) acc[tt][r] = mfma32<BF16>(a[cs][tt], b[cs][r], acc[tt][r]);
            }
            if (q == 0 && EPI) load_residual(n, x0, y0, have);   // this tile's residual (the ONE load site): behind the epilogue, in front of the DMA
            if (q < PPW) dma_piece(q);                               // the DMA issue rides in the shadow of the matrix pipe
        }
#pragma unroll
        for (int i = PAIRS; i < PPW; ++i) dma_piece(i);
        if (KS == 3 && p.res_in) {
            // act(conv(x) + x): the residual of output channels 16c .. 16c+15 is the centre pixel of input chunk c, still in
            // this stage's ring slot (read behind the MFMAs: nothing waits for it)
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                int cc = c;
                asm volatile("" : "+s"(cc));      // opaque per tile: hipcc otherwise folds the NT tests into acc[c] and the accumulators go to scratch
                if (tt == cc) {
#pragma unroll
                    for (int r = 0; r < RW; ++r)
                        acc[tt][r] += unpack4<BF16>(*reinterpret_cast<const uint2*>(sb + c_off + r * (TH * 32)));
                }
            }
        }
        if (c == p.nchunks - 1 && p.border) border_fix(x0, y0);
        if (last && act_gelu) gelu_inplace();        // (`last` is a residual stage when there are any: residual_stage applies it)
    };

    // ---- a residual stage (p.nres > 0): the residual tensor's channels 16t .. 16t+15 were staged like an input chunk; the centre
    // pixels are added to accumulator tile t from LDS.  Same DMA type, same ring, same counted waits as the input: nothing rides
    // in registers while in flight (asm loads into VGPRs did, and hipcc copied those registers before the data was there).
    auto residual_stage = [&](int c, bool last) __attribute__((always_inline)) {
        const char* sb = ring + slot * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < PPW; ++i) dma_piece(i);
        const bool post = KP()->res_mode == ESR_RES_POST_ACT;
        if (post && c == p.nchunks) {
            // act(conv) + res: the activation goes first, on the accumulators (the epilogue then sees slope 1)
            if (act_gelu) {
                gelu_inplace();
            } else {
                const float sl = KP()->slope;
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int r = 0; r < RW; ++r) {
                        f32x4 v = acc[tt][r];
                        v.x = act1(v.x, sl); v.y = act1(v.y, sl); v.z = act1(v.z, sl); v.w = act1(v.w, sl);
                        acc[tt][r] = v;
                    }
            }
        }
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            int cc = c - p.nchunks;
            asm volatile("" : "+s"(cc));      // opaque per tile (see the res_in block of compute)
            if (tt == cc || (HILO && tt + NT == cc)) {             // HILO: stages NT .. 2 NT - 1 carry the residual's low parts
#pragma unroll
                for (int r = 0; r < RW; ++r)
                    acc[tt][r] += unpack4<BF16>(*reinterpret_cast<const uint2*>(sb + c_off + r * (TH * 32)));
            }
        }
        if (last && act_gelu && !post) gelu_inplace();
    };

    for (int k = 0;; ++k) {
        // the iteration behind the block's last tile drains the pending epilogue through the same code (its MFMAs run on
        // whatever the ring holds and its DMA / residual loads are out of range)
        const int t = tile_index(k);
        have = t >= 0;
        if (!have && !pend) break;
        if (have) tile_coords(t, n, x0, y0);
        const int nst = have ? nstages : 1;
        for (int c = 0; c < nst; ++c) {
            const bool last = c == nst - 1;
            hist_rs = (hist_rs << 1) | (c == 0 ? 1u : 0u);
            hist_st <<= 1;
            if (c == 0 && swap_epi && have && (pend || GRES)) {
                // the previous tile's epilogue inside this tile's first MFMA group (GRES: also for the block's first tile, nothing
                // pending and every store out of range -- the residual loads have their one site in there)
                hist_st |= 1u;
                compute(std::true_type{}, 0, last);
            } else if (c == 0 && swap_epi && pend) {
                // behind the block's last tile: the epilogue alone (with one or two tiles per block -- single images -- a whole
                // stage of MFMAs on stale data would cost a quarter of the block's time)
                wait_residual();
                swap_epi_setup();
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    swap_epi_act(r);
                    if (r & 1) swap_epi_store(r);
                }
                break;
            } else {
                if (c == 0 && pend) {
                    hist_st |= 1u;
                    epilogue(pn, px0, py0);
                }
                if (!have) break;                               // behind the block's last tile: the epilogue was all
                if (c >= p.nchunks) residual_stage(c, last);
                else compute(std::false_type{}, c, last);
            }
            cursor_advance();
            // ---- sync: stage s+1 has landed; everything issued after its DMA may stay in flight -------------------------
            // its DMA was issued R-2 stages ago, behind that stage's own stores / residual loads: younger are the DMA of the
            // R-2 stages since and the first-stage instructions of those among them that opened a tile
            wait_vm_dyn((R - 2) * n_my + epi_stores * __builtin_popcount(hist_st & hmask) + (GRES ? RES_LOADS : 0) * __builtin_popcount(hist_rs & hmask));
            if (!OWN_PIECES) __builtin_amdgcn_s_barrier();
            slot = slot == R - 1 ? 0 : slot + 1;
        }
        if (!have) break;
        pend = true;
        pn = n; px0 = x0; py0 = y0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the trailing (zero-fill) DMA writes LDS: it must not outlive the block
}

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

template <int NT, int KS, int NW, bool BF16, bool GRES, int PNT1 = 0, int PNT2 = 0, bool HILO = false>
int launch_s16(const S16K& k, size_t lds, hipStream_t st)
{
    // the attribute belongs to the (device, instantiation) pair: one process may drive several GPUs (engine contexts are keyed by
    // device).  Relaxed atomics: a racing thread at worst sets the same value twice.
    static std::atomic<unsigned> attr_set[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return ESR_ERR_LAUNCH;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s16_kernel<NT, KS, NW, BF16, GRES, PNT1, PNT2, HILO>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
        if (e != hipSuccess) {
            esr_set_err("hipFuncSetAttribute(conv_s16_kernel, MaxDynamicSharedMemorySize)", e);
            return ESR_ERR_LAUNCH;
        }
        attr_set[dev].store(1u, std::memory_order_relaxed);
    }
    const int ntiles = k.N * k.tiles_x * k.tiles_y;
    const int cap = NW == 4 ? 512 : 256;                   // one block per CU (LDS; NW = 4: two), persistent over the tiles
    const int grid = ntiles < cap ? ntiles : cap;
    // (rocprofv3 prints every template argument, defaulted ones included)
    esr_note_kernel("conv_s16_kernel<%d, %d, %d, %s, %s, %d, %d, %s>", NT, KS, NW, esr_tf(BF16), esr_tf(GRES), PNT1, PNT2, esr_tf(HILO));
    hipLaunchKernelGGL((conv_s16_kernel<NT, KS, NW, BF16, GRES, PNT1, PNT2, HILO>), dim3(grid), dim3(64 * NW), lds, st, k);
    esr_graph_note_io(st, k.x, offsetof(S16K, x), k.y0, offsetof(S16K, y0));
    return esr_check_launch("conv_s16_kernel launch");
}

template <int KS, bool BF16, bool GRES>
int launch_s16_nt(int nt, const S16K& k, size_t lds, hipStream_t st)
{
    switch (nt) {
        case 1: return launch_s16<1, KS, S16_NW, BF16, GRES>(k, lds, st);
        case 2: return launch_s16<2, KS, S16_NW, BF16, GRES>(k, lds, st);
        case 3: return launch_s16<3, KS, S16_NW, BF16, GRES>(k, lds, st);
        case 4: return launch_s16<4, KS, S16_NW, BF16, GRES>(k, lds, st);
    }
    return ESR_ERR_UNSUPPORTED;
}

template <int KS, bool BF16>
int launch_s16_res(int nt, const S16K& k, size_t lds, hipStream_t st)
{
    return launch_s16_nt<KS, BF16, false>(nt, k, lds, st);        // (a residual from HBM is staged through LDS: S16K.nres)
}

// the post-chain variants that exist: (kernel size, main tiles, residual from HBM, post-1 tiles, post-2 tiles)
//   (3, 3, yes, 3, 1)  RLFB  c3_r (+ block input, after the activation) -> c5 -> esa.conv1     nf = 46
//   (3, 4, no,  2, 0)  RFDB  c{j}_r (residual = its input, from LDS) -> c{j+1}_d               nf = 50
//   (3, 3, no,  2, 0)  RFDB                                                                   nf = 40
//   (1, 3 | 4, no, 1, 0)  c5 (the 1x1 over the distillation concat) -> esa.conv1: BSRN / RFDN (team18_bsrn.py:167,110;
//                         rfdn_baseline/block.py:164,118)
//   (1, 3, any, 2, 0)     ESDB conv_out (+ block input) -> the NEXT block's c1_d + GELU (team18_bsrn.py:170-172 -> :150)
inline bool post_variant_exists(int ks, int nt, bool gres, int pnt1, int pnt2)
{
    if (ks == 1) return pnt2 == 0 && (((nt == 3 || nt == 4) && !gres && pnt1 == 1) || (nt == 3 && pnt1 == 2));
    return (nt == 3 && gres && pnt1 == 3 && pnt2 == 1) || (nt == 4 && !gres && pnt1 == 2 && pnt2 == 0) ||
           (nt == 3 && !gres && pnt1 == 2 && pnt2 == 0);
}

template <bool BF16>
int launch_s16_post(int ks, int nt, bool gres, int pnt1, int pnt2, const S16K& k, size_t lds, hipStream_t st)
{
    if (ks == 1) {
        if (nt == 3 && !gres && pnt1 == 1 && pnt2 == 0) return launch_s16<3, 1, S16_NW, BF16, false, 1, 0>(k, lds, st);
        if (nt == 4 && !gres && pnt1 == 1 && pnt2 == 0) return launch_s16<4, 1, S16_NW, BF16, false, 1, 0>(k, lds, st);
        if (nt == 3 && pnt1 == 2 && pnt2 == 0) return launch_s16<3, 1, S16_NW, BF16, false, 2, 0>(k, lds, st);    // (residual, if any, staged through LDS)
        return ESR_ERR_UNSUPPORTED;
    }
    if (nt == 3 && gres && pnt1 == 3 && pnt2 == 1) return launch_s16<3, 3, S16_NW, BF16, false, 3, 1>(k, lds, st);    // (the residual is staged through LDS: S16K.nres)
    if (nt == 4 && !gres && pnt1 == 2 && pnt2 == 0) return launch_s16<4, 3, S16_NW, BF16, false, 2, 0>(k, lds, st);
    if (nt == 3 && !gres && pnt1 == 2 && pnt2 == 0) return launch_s16<3, 3, S16_NW, BF16, false, 2, 0>(k, lds, st);
    return ESR_ERR_UNSUPPORTED;
}

// LDS bytes of a launch: resident weights + `ring` input stages + epilogue scratch
size_t s16_lds_bytes(int nchunks, int nt, int ksize, int nw, int ring, size_t post_bytes = 0)
{
    const int halo = ksize / 2, th = TILE + 2 * halo, thy = (nw == 4 ? 16 : 32) + 2 * halo;
    const int npieces = (th * thy + 31) / 32;
    const int pairs = (ksize * ksize + 1) / 2;
    return (size_t)nchunks * pairs * nt * 1024 + post_bytes + (size_t)ring * npieces * 1024;
}

// residual == input of a 3x3 with as many output as input chunks: taken from the staged tile (S16K.res_in), not from HBM
static bool s16_res_is_input(const esr_conv_desc* d)
{
    return d->ksize == 3 && d->res_mode == ESR_RES_PRE_ACT && esr_round_up(d->cin, 16) == esr_round_up(d->cout, 16) && d->res.ptr == d->in.ptr &&
           d->res.pitch == d->in.pitch && d->res.coff == d->in.coff;
}

// conv48r_kernel's descriptors: a 3x3 over 48 physical input channels with 2 or 3 output tiles, at least one 16 x 32 tile per CU, no
// residual from HBM, no split, no post chain (measured slower there), NHWC, one input tensor
static bool conv48r_takes(const esr_conv_desc* d)
{
    const int nt = esr_round_up(d->cout, 16) / 16, nchunks = esr_round_up(d->cin, 16) / 16;
    if (d->ksize != 3 || nchunks != 3 || (nt != 2 && nt != 3) || d->out_layout != ESR_NHWC || d->in_seg_stride != 0 || d->post_wpacked || d->hilo) return false;
    if (d->res_mode != ESR_RES_NONE && !s16_res_is_input(d)) return false;
    if (d->split > 0 && d->split < d->cout) return false;
    return (long)d->n * ((d->w + TILE - 1) / TILE) * ((d->h + 31) / 32) >= 256;
}

// conv64r_kernel's descriptors: a 3x3 over 64 physical input channels with 2 or 4 output tiles, at least one 16 x 16 tile per CU, no
// residual from HBM, no split, no post chain, no border table, NHWC, one input tensor
static bool conv64r_takes(const esr_conv_desc* d)
{
    const int nt = esr_round_up(d->cout, 16) / 16, nchunks = esr_round_up(d->cin, 16) / 16;
    if (d->ksize != 3 || nchunks != 4 || (nt != 2 && nt != 4) || d->out_layout != ESR_NHWC || d->in_seg_stride != 0 || d->post_wpacked || d->hilo || d->border_bias) return false;
    if (d->act == ESR_ACT_GELU) return false;
    if (d->res_mode != ESR_RES_NONE && !s16_res_is_input(d)) return false;
    if (d->split > 0 && d->split < d->cout) return false;
    return (long)d->n * ((d->w + TILE - 1) / TILE) * ((d->h + 15) / 16) >= 256;
}

int s16_post_plan(const esr_conv_desc* d, int nt, int nchunks, int* pnt1, int* pnt2, int* post_lo, int* ring, size_t* lds);

// conv48rp_kernel's descriptors: RLFB's c3_r -- 48 -> 48 (3 chunks, 3 tiles) with a residual from HBM that is not the input, the conv's
// own result not stored, a post chain of 3 + 1 tiles without GELU -- from 256 tiles of 16 x 16.  Measured (tools/gpu_c48p.sh): one
// 339 x 510 image 26 against conv_s16_kernel's 38.6 us, 0.219 against 0.265 ms at 32 x 256 x 256.
static bool conv48rp_takes(const esr_conv_desc* d)
{
    const int nt = esr_round_up(d->cout, 16) / 16, nchunks = esr_round_up(d->cin, 16) / 16;
    if (d->ksize != 3 || nchunks != 3 || nt != 3 || d->out_layout != ESR_NHWC || d->in_seg_stride != 0 || !d->post_wpacked || !d->post2_wpacked) return false;
    if (d->out0.ptr || d->border_bias || d->act == ESR_ACT_GELU || d->post_act == ESR_ACT_GELU) return false;
    if (d->res_mode != ESR_RES_POST_ACT || s16_res_is_input(d)) return false;
    int pnt1 = 0, pnt2 = 0, post_lo = 0, ring = 0;
    size_t lds = 0;
    if (s16_post_plan(d, nt, nchunks, &pnt1, &pnt2, &post_lo, &ring, &lds) != ESR_OK) return false;
    if (pnt1 != 3 || pnt2 != 1 || post_lo != (d->storage == ESR_STORE_BF16 ? 1 : 0)) return false;
    const long tx = (d->w + TILE - 1) / TILE;
    return (long)d->n * tx * ((d->h + 15) / 16) >= 256;
}

// conv48rq_kernel's descriptors: a 3x3 over 48 physical input channels with three output tiles whose result is stored AND feeds one post 1x1 of
// two output tiles (ESDB c{j}_r -> c{j+1}_d, team18_bsrn.py:150-163), no residual from HBM, fp16 storage (high-part post images only), from
// 256 tiles of 16 x 16
static bool conv48rq_takes(const esr_conv_desc* d)
{
    const int nt = esr_round_up(d->cout, 16) / 16, nchunks = esr_round_up(d->cin, 16) / 16;
    if (d->storage != ESR_STORE_F16 || d->ksize != 3 || nchunks != 3 || nt != 3 || d->out_layout != ESR_NHWC || d->in_seg_stride != 0 || d->hilo) return false;
    if (!d->post_wpacked || d->post2_wpacked || !d->out0.ptr || (d->split > 0 && d->split < d->cout)) return false;
    if (d->res_mode != ESR_RES_NONE && !s16_res_is_input(d)) return false;
    int pnt1 = 0, pnt2 = 0, post_lo = 0, ring = 0;
    size_t lds = 0;
    if (s16_post_plan(d, nt, nchunks, &pnt1, &pnt2, &post_lo, &ring, &lds) != ESR_OK) return false;
    if (pnt1 != 2 || pnt2 != 0 || post_lo != 0) return false;
    return (long)d->n * ((d->w + TILE - 1) / TILE) * ((d->h + 15) / 16) >= 256;
}

// conv64rq_kernel's descriptors: a 3x3 over 64 physical input channels with four output tiles whose result is stored AND feeds one post 1x1
// of two output tiles (RFDB c{j}_r -> c{j+1}_d, rfdn_baseline/block.py:150-160), no residual from HBM, LeakyReLU / none, from 256 tiles of 16 x 16
static bool conv64rq_takes(const esr_conv_desc* d)
{
    const int nt = esr_round_up(d->cout, 16) / 16, nchunks = esr_round_up(d->cin, 16) / 16;
    if (d->ksize != 3 || nchunks != 4 || nt != 4 || d->out_layout != ESR_NHWC || d->in_seg_stride != 0 || d->hilo || d->border_bias) return false;
    if (!d->post_wpacked || d->post2_wpacked || !d->out0.ptr || (d->split > 0 && d->split < d->cout)) return false;
    if (d->act == ESR_ACT_GELU || d->post_act == ESR_ACT_GELU) return false;
    if (d->res_mode != ESR_RES_NONE && !s16_res_is_input(d)) return false;
    int pnt1 = 0, pnt2 = 0, post_lo = 0, ring = 0;
    size_t lds = 0;
    if (s16_post_plan(d, nt, nchunks, &pnt1, &pnt2, &post_lo, &ring, &lds) != ESR_OK) return false;
    if (pnt1 != 2 || pnt2 != 0 || post_lo != (d->storage == ESR_STORE_BF16 ? 1 : 0)) return false;
    return (long)d->n * ((d->w + TILE - 1) / TILE) * ((d->h + 15) / 16) >= 256;
}

// conv48rp_kernel<bf16, LRS>'s descriptors: the LR conv of a 48-channel network on hi + lo pairs -- 48 -> 48 (3 chunks, 3 tiles), residual pair
// from HBM added before the activation, output pair, no post chain -- from 256 tiles of 16 x 16
static bool conv48rl_takes(const esr_conv_desc* d)
{
    const int nt = esr_round_up(d->cout, 16) / 16, nchunks = esr_round_up(d->cin, 16) / 16;
    if (d->storage != ESR_STORE_BF16 || d->hilo != (ESR_HILO_RES | ESR_HILO_OUT) || d->hilo_stride <= 0) return false;
    if (d->ksize != 3 || nchunks != 3 || nt != 3 || d->out_layout != ESR_NHWC || d->in_seg_stride != 0 || d->post_wpacked || d->border_bias) return false;
    if (d->res_mode != ESR_RES_PRE_ACT || d->act == ESR_ACT_GELU || (d->split > 0 && d->split < d->cout)) return false;
    return (long)d->n * ((d->w + TILE - 1) / TILE) * ((d->h + 15) / 16) >= 256;
}

// 1: conv48r_kernel / conv48rp_kernel (one 4-wave block per CU, weights in registers), 4: conv_s16_kernel's two-blocks-per-CU shape (4
// waves, 16 x 16 tiles), 8: one 8-wave block per CU on 16 x 32 tiles
int s16_block_waves(const esr_conv_desc* d)
{
    if (conv48r_takes(d) || conv48rp_takes(d) || conv64r_takes(d) || conv48rl_takes(d) || conv48rq_takes(d) || conv64rq_takes(d)) return 1;
    const int nt = esr_round_up(d->cout, 16) / 16, nchunks = esr_round_up(d->cin, 16) / 16;
    const bool res_hbm = d->res_mode != ESR_RES_NONE && !s16_res_is_input(d);
    if (d->ksize != 3 || nt != 3 || d->border_bias || d->post_wpacked || res_hbm || d->out_layout != ESR_NHWC || d->in_seg_stride != 0) return 8;
    if (d->hilo && !(d->hilo == ESR_HILO_OUT && nchunks == 1 && (long)d->n * ((d->w + TILE - 1) / TILE) * ((d->h + 15) / 16) < 4096)) return 8;   // hi + lo pairs: only a single image's head takes the 4-wave shape
    if ((long)d->n * ((d->w + TILE - 1) / TILE) * ((d->h + 15) / 16) < 512) return 8;           // fewer tiles than resident blocks
    return s16_lds_bytes(nchunks, nt, 3, 4, RING_MIN, 1024) <= (size_t)LDS_LIMIT / 2 ? 4 : 8;
}

// decides how a descriptor with a post chain runs: fills the tile counts and whether the low-part images are resident;
// returns ESR_OK if a fused variant exists and fits the LDS
int s16_post_plan(const esr_conv_desc* d, int nt, int nchunks, int* pnt1, int* pnt2, int* post_lo, int* ring, size_t* lds)
{
    if (d->out_layout != ESR_NHWC || (d->split > 0 && d->split < d->cout)) return ESR_ERR_UNSUPPORTED;
    if (d->post_cout <= 0 || d->post_cout > 48) return ESR_ERR_UNSUPPORTED;
    *pnt1 = esr_round_up(d->post_cout, 16) / 16;
    *pnt2 = d->post2_wpacked ? 1 : 0;
    if (*pnt2 && (d->post2_cout <= 0 || d->post2_cout > 16)) return ESR_ERR_UNSUPPORTED;
    const bool res_is_in = d->res_mode == ESR_RES_PRE_ACT && esr_round_up(d->cin, 16) == esr_round_up(d->cout, 16) && d->res.ptr == d->in.ptr &&
                           d->res.pitch == d->in.pitch && d->res.coff == d->in.coff;
    const bool gres = d->res_mode != ESR_RES_NONE && !res_is_in;
    if (!post_variant_exists(d->ksize, nt, gres, *pnt1, *pnt2)) return ESR_ERR_UNSUPPORTED;
    // fp16 storage: the post weights' low parts (and the activations' low parts, see hilo) are not needed -- 11 mantissa bits, the
    // network's own storage precision; bf16 keeps hi + lo wherever the images fit
    for (int lo = d->storage == ESR_STORE_F16 ? 0 : 1; lo >= 0; --lo) {
        const size_t pb = (size_t)(lo + 1) * (nt * *pnt1 + *pnt1 * *pnt2) * 1024 + 1024 + (d->border_bias ? (size_t)nt * 1024 : 0);
        int r = RING_MAX;
        while (r > RING_MIN && s16_lds_bytes(nchunks, nt, d->ksize, S16_NW, r, pb) > (size_t)LDS_LIMIT) --r;
        if (s16_lds_bytes(nchunks, nt, d->ksize, S16_NW, r, pb) <= (size_t)LDS_LIMIT) {
            *post_lo = lo; *ring = r; *lds = s16_lds_bytes(nchunks, nt, d->ksize, S16_NW, r, pb);
            return ESR_OK;
        }
    }
    return ESR_ERR_UNSUPPORTED;
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
    // (+ the v_mfma_f32_32x32x16 image of the 64 -> 64 3x3s behind the bias: esr_c64m.hip)
    return nchunks * pairs * nt * 1024 + nt * 16 * sizeof(float) + esr_m32_conv_bytes(cin_phys, cout, ksize);
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
    const size_t m32_off = esr_m32_conv_offset(cin_phys, cout, ksize);
    uint16_t* om = m32_off ? reinterpret_cast<uint16_t*>(static_cast<char*>(out) + m32_off) : nullptr;
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
                    // the same value in v_mfma_f32_32x32x16's fragment order (esr_c64m.hip): fragment (chunk, tap, half), lane 32 h + i, slot j
                    if (om) om[(((((size_t)(s / 16) * 9 + tap) * 2 + oc / 32) * 64 + ((s % 16) / 8) * 32 + oc % 32) * 8) + s % 8] = q;
                }
            }
        }
    }
    float* bo = reinterpret_cast<float*>(static_cast<char*>(out) + (size_t)nchunks * pairs * nt * 1024);
    if (bias)
        for (int oc = 0; oc < cout; ++oc) bo[oc] = bias[oc];
    return ESR_OK;
}

size_t esr_packed_post_s16_bytes(int cin, int cout)
{
    if (cin <= 0 || cout <= 0) return 0;
    const size_t kt = (size_t)esr_round_up(cin, 16) / 16, ot = (size_t)esr_round_up(cout, 16) / 16;
    return 2 * kt * ot * 1024 + ot * 16 * sizeof(float) + esr_m32_post_bytes(cin, cout);
}

int esr_pack_post_s16(const float* w, const float* bias, int cin, int cout, int compute, void* out, size_t out_bytes)
{
    if (!w || !out || cin <= 0 || cout <= 0) return ESR_ERR_BAD_ARG;
    if (compute != ESR_COMPUTE_BF16 && compute != ESR_COMPUTE_F16) return ESR_ERR_BAD_ARG;
    const size_t need = esr_packed_post_s16_bytes(cin, cout);
    if (out_bytes < need) return ESR_ERR_BAD_ARG;
    const int kt = esr_round_up(cin, 16) / 16, ot = esr_round_up(cout, 16) / 16;
    memset(out, 0, need);
    uint16_t* hi = static_cast<uint16_t*>(out);
    uint16_t* lo = hi + (size_t)kt * ot * 512;
    const size_t pm_off = esr_m32_post_offset(cin, cout);
    uint16_t* pm = pm_off ? reinterpret_cast<uint16_t*>(static_cast<char*>(out) + pm_off) : nullptr;
    // image [k tile][out tile][lane = kq * 16 + i][j]: input channel 16 kt + 4 kq + (j & 3) for output channel 16 ot + i; the
    // B operand carries the high parts of the four fp32 inputs in slots 0..3 and their low parts in 4..7, so the hi image has
    // the weight's high part in all eight slots, the lo image its low part in slots 0..3 only (lo x lo is dropped)
    for (int o = 0; o < cout; ++o)
        for (int c = 0; c < cin; ++c) {
            const float wv = w[(size_t)o * cin + c];
            const uint16_t h = to16(wv, compute);
            const uint16_t l = to16((double)wv - from16(h, compute), compute);
            const size_t base = ((((size_t)(c / 16) * ot + o / 16) * 64 + ((c % 16) / 4) * 16 + o % 16) * 8) + (c % 4);
            hi[base] = h;
            hi[base + 4] = h;
            lo[base] = l;
            if (pm) {
                // v_mfma_f32_32x32x16 order (esr_c64m.hip): step c / 8, images (hi, lo), lane 32 ((c % 8) / 4) + o, slots c % 4 and c % 4 + 4
                const size_t mb = ((((size_t)(c / 8) * 2) * 64 + ((c % 8) / 4) * 32 + o) * 8) + (c % 4);
                pm[mb] = h;
                pm[mb + 4] = h;
                pm[mb + 512] = l;
            }
        }
    float* bo = reinterpret_cast<float*>(static_cast<char*>(out) + 2 * (size_t)kt * ot * 1024);
    if (bias)
        for (int o = 0; o < cout; ++o) bo[o] = bias[o];
    return ESR_OK;
}

int esr_conv_post_supported(const esr_conv_desc* d)
{
    if (!d || !d->post_wpacked || (d->storage != ESR_STORE_BF16 && d->storage != ESR_STORE_F16)) return 0;
    int a, b, c, r;
    size_t l;
    return s16_post_plan(d, esr_round_up(d->cout, 16) / 16, esr_round_up(d->cin, 16) / 16, &a, &b, &c, &r, &l) == ESR_OK;
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

// ---- network input for the 16-bit plans (esr_pack_input_s16) ----------------------------------------------------------------
namespace {
template <bool BF16>
__global__ __launch_bounds__(256) void pack_input_kernel(const float* __restrict__ x, char* __restrict__ y, int C, long long hw, long long npix, int pitch, int coff)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const long long n = i / hw, s = i - n * hw;
        unsigned short v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = 0;
        float fin[4];               // (every plane's load in flight before the first use: a load inside `if (c < C)` is a branch with its own wait)
#pragma unroll
        for (int c = 0; c < 4; ++c) fin[c] = x[(n * C + (c < C ? c : C - 1)) * hw + s];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < C) {
                const float f = fin[c];
                const unsigned h = pack2<BF16>(f, 0.f) & 0xffffu;
                float fh, dummy;
                unpack2<BF16>(h, fh, dummy);
                const unsigned l = pack2<BF16>(f - fh, 0.f) & 0xffffu;
                v[c] = (unsigned short)h; v[C + c] = (unsigned short)l; v[2 * C + c] = (unsigned short)h;
            }
        uint4* o = reinterpret_cast<uint4*>(y + ((size_t)i * pitch + coff) * 2);
        o[0] = uint4{(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16), (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16)};
        o[1] = uint4{(unsigned)v[8] | ((unsigned)v[9] << 16), (unsigned)v[10] | ((unsigned)v[11] << 16), (unsigned)v[12] | ((unsigned)v[13] << 16), (unsigned)v[14] | ((unsigned)v[15] << 16)};
    }
}
}  // namespace

extern "C" int esr_pack_input_s16(const esr_conv_desc* d, void* hip_stream)
{
    if (!d || !d->in.ptr || !d->out0.ptr || d->n <= 0 || d->h <= 0 || d->w <= 0) return ESR_ERR_BAD_ARG;
    if (d->cin <= 0 || d->cin > 4) return ESR_ERR_UNSUPPORTED;
    if (d->storage != ESR_STORE_BF16 && d->storage != ESR_STORE_F16) return ESR_ERR_BAD_ARG;
    if ((d->out0.pitch & 7) || (d->out0.coff & 7) || d->out0.coff + 16 > d->out0.pitch) return ESR_ERR_BAD_ARG;
    const long long hw = (long long)d->h * d->w, npix = hw * d->n;
    const long long want = (npix + 255) / 256;
    const int grid = (int)(want < 8192 ? want : 8192);
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    esr_note_kernel("pack_input_kernel<%s>", esr_tf(d->storage == ESR_STORE_BF16));
    if (d->storage == ESR_STORE_BF16)
        hipLaunchKernelGGL(pack_input_kernel<true>, dim3(grid), dim3(256), 0, st, static_cast<const float*>(d->in.ptr), static_cast<char*>(d->out0.ptr), d->cin, hw, npix, d->out0.pitch, d->out0.coff);
    else
        hipLaunchKernelGGL(pack_input_kernel<false>, dim3(grid), dim3(256), 0, st, static_cast<const float*>(d->in.ptr), static_cast<char*>(d->out0.ptr), d->cin, hw, npix, d->out0.pitch, d->out0.coff);
    esr_graph_note_io(st, d->in.ptr, 0, nullptr, 0);
    return esr_check_launch("pack_input_kernel launch");
}

// called by esr_conv2d_f32 (esr_hip.hip) for descriptors with 16-bit storage
int esr_s16_block_waves(const esr_conv_desc* d) { return s16_block_waves(d); }

int esr_conv2d_s16(const esr_conv_desc* d, void* hip_stream)
{
    const bool bf16 = d->storage == ESR_STORE_BF16;
    if (d->storage != ESR_STORE_BF16 && d->storage != ESR_STORE_F16) return ESR_ERR_BAD_ARG;
    if (d->compute != (bf16 ? ESR_COMPUTE_BF16 : ESR_COMPUTE_F16)) return ESR_ERR_BAD_ARG;   // operand type = storage type
    if (d->in_layout != ESR_NHWC) return ESR_ERR_UNSUPPORTED;                                  // the NCHW head runs on conv_f32_kernel
    if (d->tail_wpacked || d->blocked8) return ESR_ERR_UNSUPPORTED;                            // fp32 features
    const bool post = d->post_wpacked != nullptr;
    if (d->border_bias && d->out_layout != ESR_NHWC) return ESR_ERR_UNSUPPORTED;
    if (d->border_bias && ((uintptr_t)d->border_bias & 15)) return ESR_ERR_BAD_ARG;       // (staged by 16-byte LDS-DMA pieces, as the packed weights)
    if (!post && d->post2_wpacked) return ESR_ERR_BAD_ARG;
    if ((d->in.pitch & 7) || (d->in.coff & 7)) return ESR_ERR_BAD_ARG;                         // 16-byte granules
    const int cin_phys = esr_round_up(d->cin, 16);
    const bool segmented = d->in_seg_stride != 0;
    if (segmented) {
        if (d->in_seg_chunks <= 0 || (cin_phys / 16) % d->in_seg_chunks || d->in_seg_stride < 0 || (d->in_seg_stride & 15)) return ESR_ERR_BAD_ARG;
        if (d->in.coff + 16 * d->in_seg_chunks > d->in.pitch) return ESR_ERR_BAD_ARG;
        if (d->ksize != 1) return ESR_ERR_UNSUPPORTED;             // (a 3x3 over a concat does not occur on the path)
    } else if (d->in.coff + cin_phys > d->in.pitch) {
        return ESR_ERR_BAD_ARG;                                  // chunk reads stay inside the pixel
    }
    const int nt = esr_round_up(d->cout, 16) / 16;
    const bool shuffle = d->out_layout == ESR_NCHW_SHUFFLE4;
    const int cout8 = esr_round_up(d->cout, 8);
    // hi + lo tensors (ABI v10): two dense tensors of the same shape, the low parts d->hilo_stride bytes behind the high parts
    const int hilo = d->hilo;
    if (hilo & ~(ESR_HILO_IN | ESR_HILO_RES | ESR_HILO_OUT)) return ESR_ERR_BAD_ARG;
    if (hilo) {
        if (!bf16 || d->ksize != 3 || (post && (hilo != ESR_HILO_OUT || d->post2_wpacked)) || segmented || (d->border_bias && (hilo & ESR_HILO_IN)) || (nt != 3 && nt != 4) || (d->split > 0 && d->split < d->cout)) return ESR_ERR_UNSUPPORTED;
        if (d->hilo_stride <= 0 || (d->hilo_stride & 15)) return ESR_ERR_BAD_ARG;
        if ((hilo & ESR_HILO_RES) && d->res_mode == ESR_RES_NONE) return ESR_ERR_BAD_ARG;
        if ((hilo & ESR_HILO_RES) && s16_res_is_input(d)) return ESR_ERR_UNSUPPORTED;      // residual == input is added from the staged tile: the low tensor would be dropped
        if ((hilo & ESR_HILO_OUT) && (shuffle || !d->out0.ptr)) return ESR_ERR_BAD_ARG;
    }
    int split = d->split <= 0 ? cout8 : d->split;
    if (split >= d->cout) split = cout8;
    if (split & 7) return ESR_ERR_BAD_ARG;
    if (shuffle) {
        if (d->cout % 16 || d->res_mode != ESR_RES_NONE) return ESR_ERR_UNSUPPORTED;
    } else if (d->out_layout == ESR_NHWC && post && !d->out0.ptr) {
        // the conv's own result feeds the post chain only
    } else if (d->out_layout == ESR_NHWC) {
        if (!d->out0.ptr) return ESR_ERR_BAD_ARG;
        if ((d->out0.pitch & 7) || (d->out0.coff & 7) || d->out0.coff + split > d->out0.pitch) return ESR_ERR_BAD_ARG;
        if (split < cout8 && (!d->out1.ptr || (d->out1.pitch & 7) || (d->out1.coff & 7) || d->out1.coff + (cout8 - split) > d->out1.pitch))
            return ESR_ERR_BAD_ARG;
    } else {
        return ESR_ERR_BAD_ARG;
    }
    if (d->res_mode != ESR_RES_NONE && (!d->res.ptr || (d->res.pitch & 7) || (d->res.coff & 7) || d->res.coff + cout8 > d->res.pitch))
        return ESR_ERR_BAD_ARG;
    if ((double)d->h * d->w * d->in.pitch * 2.0 >= 2147483647.0) return ESR_ERR_UNSUPPORTED;   // per-image raw buffer < 2 GiB
    const int wchunks = cin_phys / 16;                       // resident weight chunks
    const int nchunks = (hilo & ESR_HILO_IN) ? 2 * wchunks : wchunks;     // input stages per tile
    int ring = RING_MAX;                                     // as many input stages as fit next to the resident weights
    size_t lds = 0;
    int pnt1 = 0, pnt2 = 0, post_lo = 0;
    if (post) {
        const int rc = s16_post_plan(d, nt, nchunks, &pnt1, &pnt2, &post_lo, &ring, &lds);
        if (rc != ESR_OK) return rc;
        const int p1c8 = esr_round_up(d->post_cout, 8);
        if (!d->post_out.ptr || (d->post_out.pitch & 7) || (d->post_out.coff & 7) || d->post_out.coff + p1c8 > d->post_out.pitch) return ESR_ERR_BAD_ARG;
        if ((double)d->h * d->w * d->post_out.pitch * 2.0 >= 2147483647.0) return ESR_ERR_UNSUPPORTED;
        if (pnt2) {
            const int p2c8 = esr_round_up(d->post2_cout, 8);
            if (!d->post2_out.ptr || (d->post2_out.pitch & 7) || (d->post2_out.coff & 7) || d->post2_out.coff + p2c8 > d->post2_out.pitch) return ESR_ERR_BAD_ARG;
            if ((double)d->h * d->w * d->post2_out.pitch * 2.0 >= 2147483647.0) return ESR_ERR_UNSUPPORTED;
        }
        if (d->post_act != ESR_ACT_NONE && d->post_act != ESR_ACT_LRELU && d->post_act != ESR_ACT_RELU && d->post_act != ESR_ACT_GELU) return ESR_ERR_UNSUPPORTED;
    } else {
        const size_t extra = (d->border_bias ? (size_t)nt * 1024 : 0) + 1024;      // border table, the bias KB
        while (ring > RING_MIN && s16_lds_bytes(wchunks, nt, d->ksize, S16_NW, ring, extra) > (size_t)LDS_LIMIT) --ring;
        lds = s16_lds_bytes(wchunks, nt, d->ksize, S16_NW, ring, extra);
    }
    if (lds > (size_t)LDS_LIMIT) return ESR_ERR_UNSUPPORTED;                                     // weight set too large to stay resident
    if (!shuffle && d->out0.ptr) {
        if ((double)d->h * d->w * d->out0.pitch * 2.0 >= 2147483647.0) return ESR_ERR_UNSUPPORTED;
        if (split < cout8 && (double)d->h * d->w * d->out1.pitch * 2.0 >= 2147483647.0) return ESR_ERR_UNSUPPORTED;
    } else if (shuffle && (double)d->cout * d->h * d->w * 4.0 >= 2147483647.0) {
        return ESR_ERR_UNSUPPORTED;                          // per-image raw buffers < 2 GiB (out-of-range offset 0x80000000)
    }
    if (d->res_mode != ESR_RES_NONE && (double)d->h * d->w * d->res.pitch * 2.0 >= 2147483647.0) return ESR_ERR_UNSUPPORTED;
    const int pairs = (d->ksize * d->ksize + 1) / 2;

    S16K k;
    k.x = static_cast<const char*>(d->in.ptr);
    k.wp = static_cast<const char*>(d->wpacked);
    k.bias = reinterpret_cast<const float*>(k.wp + (size_t)wchunks * pairs * nt * 1024);
    k.res = static_cast<const char*>(d->res.ptr);
    k.y0 = static_cast<char*>(d->out0.ptr);
    k.y1 = static_cast<char*>(d->out1.ptr);
    k.N = d->n; k.H = d->h; k.W = d->w;
    k.nchunks = nchunks;
    k.ring = ring;
    k.in_pitch = d->in.pitch; k.in_coff = d->in.coff;
    k.res_pitch = d->res.pitch; k.res_coff = d->res.coff;
    k.y0_pitch = d->out0.pitch; k.y0_coff = d->out0.coff;
    k.y1_pitch = d->out1.pitch; k.y1_coff = d->out1.coff;
    k.cout_store = shuffle ? d->cout : cout8;
    k.split = split;
    k.act = d->act;
    k.slope = d->act == ESR_ACT_LRELU ? d->slope : (d->act == ESR_ACT_RELU ? 0.f : 1.f);
    k.res_mode = d->res_mode;
    k.res_in = 0;
    k.nres = 0;
    if (s16_res_is_input(d)) {
        k.res_in = 1;                               // residual == input: added from the staged tile, no residual loads
        k.res_mode = ESR_RES_NONE;
    }
    if (k.res_mode != ESR_RES_NONE) k.nres = (hilo & ESR_HILO_RES) ? 2 * nt : nt;      // residual from HBM: staged as extra chunks per tile
    k.w_chunks = wchunks;
    k.hilo_out = (hilo & ESR_HILO_OUT) ? 1 : 0;
    k.res_lo_stride = (hilo & ESR_HILO_RES) ? d->hilo_stride : 0;
    k.wm32 = nullptr; k.pm32 = nullptr; k.pbias1 = nullptr;
    if (k.hilo_out) {                                              // the low parts leave through the y1 stores
        k.y1 = k.y0 + d->hilo_stride;
        k.y1_pitch = k.y0_pitch;
        k.y1_coff = k.y0_coff + split;                             // (the kernel subtracts `split` from y1's channel offsets)
    }
    k.out_layout = d->out_layout;
    k.tiles_x = (d->w + TILE - 1) / TILE;
    k.tiles_y = (d->h + 31) / 32;
    k.magic_x = k.tiles_x > 1 ? (unsigned)((0x100000000ull + k.tiles_x - 1) / k.tiles_x) : 0u;
    k.magic_y = k.tiles_y > 1 ? (unsigned)((0x100000000ull + k.tiles_y - 1) / k.tiles_y) : 0u;
    {
        const double nt_all = (double)d->n * k.tiles_x * k.tiles_y;
        if (nt_all * (k.tiles_x > k.tiles_y ? k.tiles_x : k.tiles_y) >= 4294967296.0) return ESR_ERR_UNSUPPORTED;   // magic division range
    }
    k.pw1 = static_cast<const char*>(d->post_wpacked); k.pw2 = static_cast<const char*>(d->post2_wpacked);
    k.py1 = static_cast<char*>(d->post_out.ptr); k.py2 = static_cast<char*>(d->post2_out.ptr);
    k.py1_pitch = d->post_out.pitch; k.py1_coff = d->post_out.coff; k.py2_pitch = d->post2_out.pitch; k.py2_coff = d->post2_out.coff;
    k.p1_cout8 = esr_round_up(d->post_cout > 0 ? d->post_cout : 1, 8); k.p2_cout8 = esr_round_up(d->post2_cout > 0 ? d->post2_cout : 1, 8);
    k.p1_slope = d->post_act == ESR_ACT_LRELU ? d->slope : (d->post_act == ESR_ACT_RELU ? 0.f : 1.f);
    k.p1_gelu = d->post_act == ESR_ACT_GELU;
    k.post_lo = post_lo;
    k.store_main = d->out0.ptr ? 1 : 0;
    k.border = d->border_bias;
    k.seg_chunks = segmented ? d->in_seg_chunks : nchunks;
    k.seg_stride = segmented ? d->in_seg_stride : 0;
    if (hilo & ESR_HILO_IN) {                                      // [hi tensor, lo tensor]: a two-segment concat that meets the same weights twice
        k.seg_chunks = wchunks;
        k.seg_stride = d->hilo_stride;
    }
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if (hilo && conv48rl_takes(d)) {
        S16K kp = k;
        kp.tiles_y = (d->h + 15) / 16;
        kp.magic_y = kp.tiles_y > 1 ? (unsigned)((0x100000000ull + kp.tiles_y - 1) / kp.tiles_y) : 0u;
        const double nt_all = (double)d->n * kp.tiles_x * kp.tiles_y;
        if (nt_all * (kp.tiles_x > kp.tiles_y ? kp.tiles_x : kp.tiles_y) < 4294967296.0) return launch_conv48rp<true, true>(kp, st);
    }
    if (hilo && post) {
        // the head with block 1's first distillation 1x1 in its epilogue (RFDN: 4 main tiles, BSRN: 3; 2 post tiles) + the hi + lo store
        if (pnt2 != 0 || pnt1 != 2 || (nt != 3 && nt != 4)) return ESR_ERR_UNSUPPORTED;
        if (nt == 3) return launch_s16<3, 3, S16_NW, true, false, 2, 0, true>(k, lds, st);
        return launch_s16<4, 3, S16_NW, true, false, 2, 0, true>(k, lds, st);
    }
    if (hilo) {
        const long t16 = (long)d->n * k.tiles_x * ((d->h + 15) / 16);
        if (nt == 3 && hilo == ESR_HILO_OUT && wchunks == 1 && t16 >= 512 && t16 < 4096) {
            // the head of a 48-channel network (16 input slots, hi + lo store) on single images: the two-blocks-per-CU shape on 16 x 16 tiles
            // (one 339 x 510 image: 17.5 against 19.5 us; a batch of 32 is 12 % faster on the 8-wave shape)
            S16K k4 = k;
            k4.ring = RING_MIN;
            k4.tiles_y = (d->h + 15) / 16;
            k4.magic_y = k4.tiles_y > 1 ? (unsigned)((0x100000000ull + k4.tiles_y - 1) / k4.tiles_y) : 0u;
            const double nt_all = (double)d->n * k4.tiles_x * k4.tiles_y;
            if (nt_all * (k4.tiles_x > k4.tiles_y ? k4.tiles_x : k4.tiles_y) < 4294967296.0)
                return launch_s16<3, 3, 4, true, false, 0, 0, true>(k4, s16_lds_bytes(wchunks, nt, 3, 4, RING_MIN, 1024), st);
        }
        if (nt == 3) return launch_s16<3, 3, S16_NW, true, false, 0, 0, true>(k, lds, st);
        return launch_s16<4, 3, S16_NW, true, false, 0, 0, true>(k, lds, st);
    }
    if (conv48rp_takes(d)) {
        // RLFB c3_r (+ block input) -> c5 -> esa.conv1 on 16 x 16 tiles
        S16K kp = k;
        kp.tiles_y = (d->h + 15) / 16;
        kp.magic_y = kp.tiles_y > 1 ? (unsigned)((0x100000000ull + kp.tiles_y - 1) / kp.tiles_y) : 0u;
        const double nt_all = (double)d->n * kp.tiles_x * kp.tiles_y;
        if (nt_all * (kp.tiles_x > kp.tiles_y ? kp.tiles_x : kp.tiles_y) < 4294967296.0)
            return bf16 ? launch_conv48rp<true>(kp, st) : launch_conv48rp<false>(kp, st);
    }
    if (conv64rq_takes(d) || (conv64r_takes(d) && nt == 4)) {
        // round 6: the 64 -> 64 3x3s (RFDB c1_r / c2_r with the next distillation 1x1, c3_r) on v_mfma_f32_32x32x16 (esr_c64m.hip)
        S16K k4 = k;
        k4.tiles_y = (d->h + 15) / 16;
        k4.magic_y = k4.tiles_y > 1 ? (unsigned)((0x100000000ull + k4.tiles_y - 1) / k4.tiles_y) : 0u;
        k4.wm32 = k.wp + esr_m32_conv_offset(cin_phys, d->cout, 3);
        if (post) {
            k4.pm32 = k.pw1 + esr_m32_post_offset(d->cout, d->post_cout);
            k4.pbias1 = reinterpret_cast<const float*>(k.pw1 + (size_t)2 * 4 * 2 * 1024);
        }
        const double nt_all = (double)d->n * k4.tiles_x * k4.tiles_y;
        if (nt_all * (k4.tiles_x > k4.tiles_y ? k4.tiles_x : k4.tiles_y) < 4294967296.0) return esr_launch_conv64m(k4, bf16, post, st);
    }
    if (conv48rq_takes(d)) {
        S16K k4 = k;
        k4.tiles_y = (d->h + 15) / 16;
        k4.magic_y = k4.tiles_y > 1 ? (unsigned)((0x100000000ull + k4.tiles_y - 1) / k4.tiles_y) : 0u;
        const double nt_all = (double)d->n * k4.tiles_x * k4.tiles_y;
        if (nt_all * (k4.tiles_x > k4.tiles_y ? k4.tiles_x : k4.tiles_y) < 4294967296.0) return launch_conv48rq<false>(k4, st);
    }
    if (conv64r_takes(d)) {
        S16K k4 = k;
        k4.tiles_y = (d->h + 15) / 16;
        k4.magic_y = k4.tiles_y > 1 ? (unsigned)((0x100000000ull + k4.tiles_y - 1) / k4.tiles_y) : 0u;
        const double nt_all = (double)d->n * k4.tiles_x * k4.tiles_y;
        if (nt_all * (k4.tiles_x > k4.tiles_y ? k4.tiles_x : k4.tiles_y) < 4294967296.0) {
            // (two output tiles -- RFDB's c4; four take conv64m_kernel above)
            return bf16 ? launch_conv64r<true, 2, true>(k4, st) : launch_conv64r<false, 2, true>(k4, st);
        }
    }
    if (conv48r_takes(d)) {
        // (a post chain stays on conv_s16_kernel: the PNT1 = 2 instantiation -- ESDB c{j}_r + the next distillation 1x1, two GELUs per
        // pixel -- measured 0.396 against 0.368 ms at 32 x 270 x 480: with ONE wave per SIMD the ~380 VALU instructions of a row pair's
        // epilogue have to fit the shadow of its 102 MFMAs exactly, conv_s16_kernel's second wave absorbs them)
        const bool ext = k.border != nullptr || k.res_in || d->act == ESR_ACT_GELU;
        if ((long)d->n * k.tiles_x * k.tiles_y < 1024) {
            // small launches (single images): 16 x 16 tiles
            S16K k4 = k;
            k4.tiles_y = (d->h + 15) / 16;
            k4.magic_y = k4.tiles_y > 1 ? (unsigned)((0x100000000ull + k4.tiles_y - 1) / k4.tiles_y) : 0u;
            if (nt == 2) return bf16 ? launch_conv48r<true, 2, true, 4>(k4, st) : launch_conv48r<false, 2, true, 4>(k4, st);
            if (ext) return bf16 ? launch_conv48r<true, 3, true, 4>(k4, st) : launch_conv48r<false, 3, true, 4>(k4, st);
            return bf16 ? launch_conv48r<true, 3, false, 4>(k4, st) : launch_conv48r<false, 3, false, 4>(k4, st);
        }
        if (nt == 2) return bf16 ? launch_conv48r<true, 2, true>(k, st) : launch_conv48r<false, 2, true>(k, st);
        if (ext) return bf16 ? launch_conv48r<true, 3, true>(k, st) : launch_conv48r<false, 3, true>(k, st);
        return bf16 ? launch_conv48r<true, 3, false>(k, st) : launch_conv48r<false, 3, false>(k, st);
    }
    if (post) {
        const bool gres = k.res_mode != ESR_RES_NONE;
        return bf16 ? launch_s16_post<true>(d->ksize, nt, gres, pnt1, pnt2, k, lds, st)
                    : launch_s16_post<false>(d->ksize, nt, gres, pnt1, pnt2, k, lds, st);
    }
    // the plain 48-channel 3x3 (RLFB c1_r / c2_r): 46 KB of weights + a ring of three 11 KB stages fit 80 KB, so TWO 4-wave blocks
    // share a CU -- their stage barriers are independent and one block's memory phase runs under the other's MFMAs (-2.5 % on the
    // kernel, +1 % RLFN, A/B; 16 x 16 tiles carry more halo and the ring is the shortest, which is why it is not more)
    if (s16_block_waves(d) == 4) {
        const size_t lds4 = s16_lds_bytes(nchunks, nt, 3, 4, RING_MIN, 1024);
        {
            S16K k4 = k;
            k4.ring = RING_MIN;
            k4.tiles_y = (d->h + 15) / 16;
            k4.magic_y = k4.tiles_y > 1 ? (unsigned)((0x100000000ull + k4.tiles_y - 1) / k4.tiles_y) : 0u;
            const double nt_all = (double)d->n * k4.tiles_x * k4.tiles_y;
            if (nt_all * (k4.tiles_x > k4.tiles_y ? k4.tiles_x : k4.tiles_y) < 4294967296.0)
                return bf16 ? launch_s16<3, 3, 4, true, false>(k4, lds4, st) : launch_s16<3, 3, 4, false, false>(k4, lds4, st);
        }
    }
    if (d->ksize == 3) return bf16 ? launch_s16_res<3, true>(nt, k, lds, st) : launch_s16_res<3, false>(nt, k, lds, st);
    return bf16 ? launch_s16_res<1, true>(nt, k, lds, st) : launch_s16_res<1, false>(nt, k, lds, st);
}
