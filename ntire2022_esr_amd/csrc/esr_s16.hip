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
//   esr_r16.hip (split out in round 6): conv48r / conv48rp / conv48rq / conv64r_kernel -- the 48-channel 3x3s and RFDB's c4 with their weights
//       in registers, one wave per SIMD, whole-pixel stages, row pairs
//   esr_c64m.hip (round 6): conv64m_kernel -- the 64 -> 64 shapes, with and without RFDB's post 1x1, on v_mfma_f32_32x32x16
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

// (conv48r / conv48rq / conv48rp / conv64r_kernel and their launchers: esr_r16.hip)

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
static bool conv48r_shape(const esr_conv_desc* d)
{
    const int nt = esr_round_up(d->cout, 16) / 16, nchunks = esr_round_up(d->cin, 16) / 16;
    if (d->ksize != 3 || nchunks != 3 || (nt != 2 && nt != 3) || d->out_layout != ESR_NHWC || d->in_seg_stride != 0 || d->post_wpacked || d->hilo) return false;
    if (d->res_mode != ESR_RES_NONE && !s16_res_is_input(d)) return false;
    if (d->split > 0 && d->split < d->cout) return false;
    return true;
}

static bool conv48rq_takes(const esr_conv_desc* d);

// conv64m_kernel<.., 3, true>'s descriptors (round 6): ESDB's c{j}_r -- the merged BSConvU + input + border table + GELU over 48 channels, plain
// (conv48r_kernel's shape) or fp16 with the next distillation Linear + GELU (conv48rq_kernel's shape) -- from 256 tiles of 16 x 16: the kernel's
// OWN tile, so that a 256 x 256 image alone and the same image inside a batch take the same kernel (its accumulation order differs from
// conv48r_kernel's / conv_s16_kernel's; tests/test_gpu_big.py::test_16bit_batch_equals_per_image)
static bool esdb_r_takes(const esr_conv_desc* d)
{
    const int nt = esr_round_up(d->cout, 16) / 16;
    if (!d->border_bias || d->act != ESR_ACT_GELU || !s16_res_is_input(d) || nt != 3) return false;
    if ((long)d->n * ((d->w + TILE - 1) / TILE) * ((d->h + 15) / 16) < 256) return false;
    return d->post_wpacked ? (conv48rq_takes(d) && d->post_act == ESR_ACT_GELU) : conv48r_shape(d);
}

static bool conv48r_takes(const esr_conv_desc* d)
{
    return conv48r_shape(d) && (long)d->n * ((d->w + TILE - 1) / TILE) * ((d->h + 31) / 32) >= 256;
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

// conv64m_kernel<bf16, plain, HL>'s descriptors: the LR conv of a 64-channel network on hi + lo pairs (RFDN: LR_conv(out_B) + out_fea,
// rfdn_baseline/RFDN.py:50-52) -- 64 -> 64 (4 chunks, 4 tiles), residual pair from HBM added before the activation, output pair, no post
// chain -- from 256 tiles of 16 x 16
static bool conv64ml_takes(const esr_conv_desc* d)
{
    const int nt = esr_round_up(d->cout, 16) / 16, nchunks = esr_round_up(d->cin, 16) / 16;
    if (d->storage != ESR_STORE_BF16 || d->hilo != (ESR_HILO_RES | ESR_HILO_OUT) || d->hilo_stride <= 0) return false;
    if (d->ksize != 3 || nchunks != 4 || nt != 4 || d->out_layout != ESR_NHWC || d->in_seg_stride != 0 || d->post_wpacked || d->border_bias) return false;
    if (d->res_mode != ESR_RES_PRE_ACT || !d->res.ptr || d->act == ESR_ACT_GELU || (d->split > 0 && d->split < d->cout)) return false;
    return (long)d->n * ((d->w + TILE - 1) / TILE) * ((d->h + 15) / 16) >= 256;
}

// rfdb_tail_kernel's descriptors (ABI v12, esr_c64m.hip): a 3x3 over 64 physical input channels with <= 32 outputs whose rounded result is the
// last 32 slots of a 1x1 over three more dense 32-slot tensors, <= 64 outputs stored and fed (unrounded) to a post 1x1 of <= 16 outputs; no
// residual, no activation on the 1x1; from 256 tiles of 16 x 16
static bool rfdb_tail_takes(const esr_conv_desc* d)
{
    if (d->storage != ESR_STORE_BF16 && d->storage != ESR_STORE_F16) return false;
    if (!d->tail_wpacked || !d->wpacked || d->ksize != 3 || d->in_layout != ESR_NHWC || d->out_layout != ESR_NHWC) return false;
    const int nch = esr_round_up(d->cin, 16) / 16;
    if ((nch != 3 && nch != 4) || d->cout < 1 || d->cout > 32 || d->tail_cat_c != 96) return false;
    if (d->res_mode != ESR_RES_NONE || d->hilo || d->in_seg_stride != 0 || d->act != ESR_ACT_NONE || d->blocked8) return false;
    if (nch == 4) {                                 // RFDB: LeakyReLU / ReLU / none on r4, 49 .. 64 outputs
        if (d->border_bias || d->tail_cout < 49 || d->tail_cout > 64) return false;
        if (d->tail_mid_act != ESR_ACT_NONE && d->tail_mid_act != ESR_ACT_LRELU && d->tail_mid_act != ESR_ACT_RELU) return false;
    } else {                                        // ESDB: the merged BSConvU's border table + GELU on r4, 33 .. 48 outputs
        if (!d->border_bias || d->tail_mid_act != ESR_ACT_GELU || d->tail_cout < 33 || d->tail_cout > 48) return false;
    }
    if (d->split > 0 && d->split < d->tail_cout) return false;
    if (!d->post_wpacked || d->post2_wpacked || d->post_cout < 1 || d->post_cout > 16) return false;
    if (d->post_act != ESR_ACT_NONE && d->post_act != ESR_ACT_LRELU && d->post_act != ESR_ACT_RELU) return false;
    if (!d->in.ptr || (d->in.pitch & 7) || (d->in.coff & 7) || d->in.coff + esr_round_up(d->cin, 8) > d->in.pitch) return false;     // (tight pitch: esr_conv2d_s16)
    if (!d->tail_cat.ptr || (d->tail_cat.pitch & 7) || (d->tail_cat.coff & 7) || d->tail_cat.coff + 32 > d->tail_cat.pitch || d->tail_seg_stride16 <= 0) return false;
    if (!d->out0.ptr || (d->out0.pitch & 7) || (d->out0.coff & 7) || d->out0.coff + esr_round_up(d->tail_cout, 8) > d->out0.pitch) return false;
    if (!d->post_out.ptr || (d->post_out.pitch & 7) || (d->post_out.coff & 7) || d->post_out.coff + esr_round_up(d->post_cout, 8) > d->post_out.pitch) return false;
    const double px = (double)d->h * d->w * 2.0, lim = 2147483647.0 - 1048576.0;
    if (px * d->in.pitch >= lim || px * d->tail_cat.pitch >= lim || px * d->out0.pitch >= lim || px * d->post_out.pitch >= lim) return false;
    const long tx = (d->w + TILE - 1) / TILE, ty = (d->h + 15) / 16;
    if ((long)d->n * tx * ty < 256) return false;
    return (double)d->n * tx * ty * (tx > ty ? tx : ty) < 4294967296.0;
}

static int run_rfdb_tail(const esr_conv_desc* d, bool bf16, hipStream_t st)
{
    S16K k;
    memset(&k, 0, sizeof(k));
    const int nt = esr_round_up(d->cout, 16) / 16, ot = esr_round_up(d->post_cout, 16) / 16, nch = esr_round_up(d->cin, 16) / 16;
    const int kt = esr_round_up(d->tail_cout, 16) / 16;
    k.x = static_cast<const char*>(d->in.ptr);
    k.wp = static_cast<const char*>(d->wpacked);
    k.bias = reinterpret_cast<const float*>(k.wp + (size_t)nch * 5 * nt * 1024);
    k.wm32 = k.wp + esr_m32_conv_offset(16 * nch, d->cout, 3);
    k.N = d->n; k.H = d->h; k.W = d->w;
    k.nchunks = nch;
    k.act = d->tail_mid_act;
    k.border = d->border_bias;
    k.in_pitch = d->in.pitch; k.in_coff = d->in.coff;
    k.slope = d->tail_mid_act == ESR_ACT_LRELU ? d->slope : (d->tail_mid_act == ESR_ACT_RELU ? 0.f : 1.f);
    k.tw = static_cast<const char*>(d->tail_wpacked);
    k.cat = static_cast<const char*>(d->tail_cat.ptr);
    k.cat_pitch = d->tail_cat.pitch; k.cat_coff = d->tail_cat.coff;
    k.cat_seg_stride = (long long)d->tail_seg_stride16 * 16;
    k.y0 = static_cast<char*>(d->out0.ptr);
    k.y0_pitch = d->out0.pitch; k.y0_coff = d->out0.coff;
    k.cout_store = esr_round_up(d->tail_cout, 8);
    k.pw1 = static_cast<const char*>(d->post_wpacked);
    k.pm32 = k.pw1 + esr_m32_post_offset(d->tail_cout, d->post_cout);
    k.pbias1 = reinterpret_cast<const float*>(k.pw1 + (size_t)2 * kt * ot * 1024);
    k.py1 = static_cast<char*>(d->post_out.ptr);
    k.py1_pitch = d->post_out.pitch; k.py1_coff = d->post_out.coff;
    k.p1_cout8 = esr_round_up(d->post_cout, 8);
    k.p1_slope = d->post_act == ESR_ACT_LRELU ? d->slope : (d->post_act == ESR_ACT_RELU ? 0.f : 1.f);
    k.out_layout = ESR_NHWC;
    k.tiles_x = (d->w + TILE - 1) / TILE;
    k.tiles_y = (d->h + 15) / 16;
    k.magic_x = k.tiles_x > 1 ? (unsigned)((0x100000000ull + k.tiles_x - 1) / k.tiles_x) : 0u;
    k.magic_y = k.tiles_y > 1 ? (unsigned)((0x100000000ull + k.tiles_y - 1) / k.tiles_y) : 0u;
    return esr_launch_rfdb_tail(k, bf16, st);
}

// 1: conv48r_kernel / conv48rp_kernel (one 4-wave block per CU, weights in registers), 4: conv_s16_kernel's two-blocks-per-CU shape (4
// waves, 16 x 16 tiles), 8: one 8-wave block per CU on 16 x 32 tiles
int s16_block_waves(const esr_conv_desc* d)
{
    if (esdb_r_takes(d) || conv48r_takes(d) || conv48rp_takes(d) || conv64r_takes(d) || conv48rl_takes(d) || conv48rq_takes(d) || conv64rq_takes(d) || conv64ml_takes(d)) return 1;
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
                    if (om) om[(((((size_t)(s / 16) * 9 + tap) * (nt >= 3 ? 2 : 1) + oc / 32) * 64 + ((s % 16) / 8) * 32 + oc % 32) * 8) + s % 8] = q;
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

int esr_conv_tail_supported(const esr_conv_desc* d) { return d && rfdb_tail_takes(d) ? 1 : 0; }

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
    if (d->blocked8) return ESR_ERR_UNSUPPORTED;                                               // an fp32 feature
    if (d->tail_wpacked) return rfdb_tail_takes(d) ? run_rfdb_tail(d, bf16, static_cast<hipStream_t>(hip_stream)) : ESR_ERR_UNSUPPORTED;
    const bool post = d->post_wpacked != nullptr;
    if (d->border_bias && d->out_layout != ESR_NHWC) return ESR_ERR_UNSUPPORTED;
    if (d->border_bias && ((uintptr_t)d->border_bias & 15)) return ESR_ERR_BAD_ARG;       // (staged by 16-byte LDS-DMA pieces, as the packed weights)
    if (!post && d->post2_wpacked) return ESR_ERR_BAD_ARG;
    if ((d->in.pitch & 7) || (d->in.coff & 7)) return ESR_ERR_BAD_ARG;                         // 16-byte granules
    const int cin_phys = esr_round_up(d->cin, 16);
    const bool segmented = d->in_seg_stride != 0;
    if (segmented) {
        if (d->in_seg_chunks <= 0 || (cin_phys / 16) % d->in_seg_chunks || d->in_seg_stride < 0 || (d->in_seg_stride & 15)) return ESR_ERR_BAD_ARG;
        // (tight pitch, round 6: a segment's last chunk may run 8 channels into the next pixel, see below)
        if (d->in.coff + 16 * d->in_seg_chunks - 8 > d->in.pitch) return ESR_ERR_BAD_ARG;
        if (d->ksize != 1) return ESR_ERR_UNSUPPORTED;             // (a 3x3 over a concat does not occur on the path)
    } else if (d->in.coff + esr_round_up(d->cin, 8) > d->in.pitch) {
        // TIGHT PITCH (round 6): the pixel holds round_up(cin, 8) channels -- whole 16-byte granules --, not necessarily whole 16-channel K chunks:
        // the last chunk's second half is then the first 16 bytes of the NEXT pixel (zeros behind the image's last one: the buffer range), and
        // meets weight rows that the packer left zero (slots >= cin).  nf = 50 at pitch 56 instead of 64: 12.5 % fewer bytes in every
        // launch of an HBM-bound model (RFDN).  Values must be finite (0 x Inf), as everywhere
        return ESR_ERR_BAD_ARG;
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
        if (nt_all * (kp.tiles_x > kp.tiles_y ? kp.tiles_x : kp.tiles_y) < 4294967296.0) return esr_launch_conv48rp(kp, true, true, st);
    }
    if (hilo && conv64ml_takes(d)) {
        S16K k4 = k;
        k4.tiles_y = (d->h + 15) / 16;
        k4.magic_y = k4.tiles_y > 1 ? (unsigned)((0x100000000ull + k4.tiles_y - 1) / k4.tiles_y) : 0u;
        k4.wm32 = k.wp + esr_m32_conv_offset(cin_phys, d->cout, 3);
        const double nt_all = (double)d->n * k4.tiles_x * k4.tiles_y;
        if (nt_all * (k4.tiles_x > k4.tiles_y ? k4.tiles_x : k4.tiles_y) < 4294967296.0) return esr_launch_conv64m(k4, true, false, true, st);
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
            return esr_launch_conv48rp(kp, bf16, false, st);
    }
    // round 6 (last): ESDB's c{j}_r -- the merged BSConvU + input + GELU over 48 channels, plain or (fp16) with the next distillation Linear + GELU --
    // on conv64m_kernel's three-chunk form
    const bool esdb_r = esdb_r_takes(d);
    if (esdb_r || conv64rq_takes(d) || (conv64r_takes(d) && nt == 4)) {
        // round 6: the 64 -> 64 3x3s (RFDB c1_r / c2_r with the next distillation 1x1, c3_r) on v_mfma_f32_32x32x16 (esr_c64m.hip)
        S16K k4 = k;
        k4.tiles_y = (d->h + 15) / 16;
        k4.magic_y = k4.tiles_y > 1 ? (unsigned)((0x100000000ull + k4.tiles_y - 1) / k4.tiles_y) : 0u;
        k4.wm32 = k.wp + esr_m32_conv_offset(cin_phys, d->cout, 3);
        if (post) {
            k4.pm32 = k.pw1 + esr_m32_post_offset(d->cout, d->post_cout);
            k4.pbias1 = reinterpret_cast<const float*>(k.pw1 + (size_t)2 * (esr_round_up(d->cout, 16) / 16) * (esr_round_up(d->post_cout, 16) / 16) * 1024);
        }
        const double nt_all = (double)d->n * k4.tiles_x * k4.tiles_y;
        if (nt_all * (k4.tiles_x > k4.tiles_y ? k4.tiles_x : k4.tiles_y) < 4294967296.0) return esr_launch_conv64m(k4, bf16, post, false, st);
    }
    if (conv48rq_takes(d)) {
        S16K k4 = k;
        k4.tiles_y = (d->h + 15) / 16;
        k4.magic_y = k4.tiles_y > 1 ? (unsigned)((0x100000000ull + k4.tiles_y - 1) / k4.tiles_y) : 0u;
        const double nt_all = (double)d->n * k4.tiles_x * k4.tiles_y;
        if (nt_all * (k4.tiles_x > k4.tiles_y ? k4.tiles_x : k4.tiles_y) < 4294967296.0) return esr_launch_conv48rq(k4, st);
    }
    if (conv64r_takes(d)) {
        S16K k4 = k;
        k4.tiles_y = (d->h + 15) / 16;
        k4.magic_y = k4.tiles_y > 1 ? (unsigned)((0x100000000ull + k4.tiles_y - 1) / k4.tiles_y) : 0u;
        const double nt_all = (double)d->n * k4.tiles_x * k4.tiles_y;
        if (nt_all * (k4.tiles_x > k4.tiles_y ? k4.tiles_x : k4.tiles_y) < 4294967296.0) {
            // (two output tiles -- RFDB's c4; four take conv64m_kernel above)
            return esr_launch_conv64r(k4, bf16, st);
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
            return esr_launch_conv48r(k4, bf16, nt, ext, 4, st);
        }
        return esr_launch_conv48r(k, bf16, nt, ext, 8, st);
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
