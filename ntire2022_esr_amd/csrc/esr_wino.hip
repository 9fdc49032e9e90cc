// esr_wino.hip -- Winograd F(2x2, 3x3) fp32 convolution on v_mfma_f32_16x16x4_f32 (gfx950), NHWC in / NHWC out.
// Interface: esr_conv_desc.wino_wpacked (include/esr_hip.h, ABI v7); design notes: DESIGN.md section 4.1w.
//
// Same operation as conv_f32_kernel<.,3,..> (nn.Conv2d(k=3, s=1, p=1, bias) + fused epilogue, models/basicblock.py:61-98),
// other arithmetic: 16 transformed-domain products per 2x2 output pixels and (cin, cout) instead of 36,
//     Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A           (Lavin & Gray; B^T, G, A^T below)
// i.e. 2.25x fewer MFMAs than the direct form, fp32 throughout (rounding ~1e-6 relative: tools/wino/wino_numerics.py).
//
//   work item   one 16x16-pixel tile (8x8 Winograd tiles) x 32 output channels (a "cout half")
//   block       4 waves, TWO blocks per CU (81 376 B of LDS each, <= 256 registers per wave): the blocks drift apart, one block's
//               epilogue / barrier stalls run under the other's MFMAs
//   wave w      Winograd tile rows 2w, 2w+1 (16 tiles = the N side of the MFMA) x 2 cout tiles x ALL 16 positions:
//               128 accumulator registers; the output transform is register-local
//   MFMA        A = U = G g G^T (lane (i, k): cout i of the tile, cin 2k+s of the chunk), B = V = B^T d B (lane (j, k): Winograd
//               tile j, cin 2k+s), D[cout][tile]: lane (j, g) holds couts 4g..4g+3 of tile j
//   K loop      cin in chunks of 8.  Per chunk the block stages, global -> LDS by DMA (buffer_load ... lds), rings of 3:
//                 U chunk    16 KB  [pos][lane][ct0 s0, ct0 s1, ct1 s0, ct1 s1]   (always an L2 hit)
//                 raw halo   18x18 pixels x 32 B = 10.4 KB, [half][row pair][37 slots of 16 B]
//               V NEVER touches LDS: lane (j, k) reads the 4x4 patch of ITS tile and ITS channel pair from the raw stage
//               (16 ds_read_b64), transforms it (32 v_pk_add_f32) and holds the 16 positions in registers as the B operands of
//               the NEXT chunk while the current chunk's 64 MFMAs run.
//   stage       16 positions x 4 MFMAs; everything else rides between them: A fragments two positions ahead, the transform in
//               positions 0..8, the raw DMA (chunk +3) at positions 2 / 4 / 6, the ONE barrier of the stage at position 12
//               (hand-over of U +1 and raw +2), behind it the U DMA (chunk +2) and the first A fragments of the next stage --
//               no bubble at the stage boundary.  One M0 write serves all pieces of a DMA group (the instruction's immediate
//               offset moves the LDS destination AND the global address: tools/wino/m0_offset_probe.hip).
//   LDS reads   per wave and chunk: 16 ds_read_b128 (A) + 16 ds_read_b64 (raw) for 64 MFMAs
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>

#include "esr_hip.h"
#include "esr_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int wn_i32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int WN_THREADS = 256;
constexpr int WN_TILE = 16;                 // output pixels per tile edge
constexpr int WN_HALO = WN_TILE + 2;
constexpr int WN_PAIR = 37;                 // 16-byte slots per pair of halo rows: 18 + 18 + 1 pad (see wn_slot)
constexpr int WN_PLANE = 9 * WN_PAIR;       // 333 slots per channel-half plane
constexpr int WN_RAW_SLOTS = 2 * WN_PLANE;  // 666 = 10 full DMA pieces + one of 26 lanes
constexpr int WN_LAST_LANES = WN_RAW_SLOTS - 640;
constexpr int WN_RAW_BYTES = WN_RAW_SLOTS * 16;
constexpr int WN_U_BYTES = 16 * 1024;
constexpr int WN_RING = 3;                  // both rings
constexpr int WN_RAW_OFF = WN_RING * WN_U_BYTES;
constexpr int WN_BIAS_OFF = WN_RAW_OFF + WN_RING * WN_RAW_BYTES;
constexpr int WN_LDS = WN_BIAS_OFF + 256;                            // 81 376 B: two blocks per CU (163 840 B)
static_assert(2 * WN_LDS <= 160 * 1024, "two blocks per CU");
constexpr int WN_MAX_BLOCKS = 512;          // 256 CUs x 2
constexpr unsigned WN_OOB = 0x80000000u;
constexpr int WN_BARRIER_POS = 12;
constexpr int WN_FIRST_SHARE = 9;           // 9/16 of the items to the first-dispatched half of the grid

struct WinoK {
    const float* x;
    const float* up;          // packed U blob
    const float* bias;        // nhalves * 32 floats inside the blob
    const float* res;
    float* y0;
    float* y1;
    int N, H, W;
    int nchunks, nhalves;
    unsigned up_bytes;
    int in_pitch, in_coff;
    int res_pitch, res_coff;
    int y0_pitch, y0_coff, y1_pitch, y1_coff;
    int cout_store, split;
    int act;
    float slope;
    int res_mode;
    int tiles_x, tiles_y;
    unsigned magic_x, magic_y;   // floor(2^32 / d) + 1 (0 for d == 1): n / d == umulhi(n, magic) for n * d < 2^32 (host-checked)
    int y1_blk;
    int first_share;             // 1/16ths of the items the first-dispatched half of the blocks takes (8 = even split)
    // wino8_f32_kernel: input addressing for NHWC and for channel-blocked [n][C/8][h][w][8] inputs (esr_conv_desc.blocked8 & ESR_BLOCKED_IN)
    unsigned pix_floats;         // floats from one pixel to the next inside a chunk: in_pitch (NHWC) | 8 (blocked)
    unsigned in_base;            // byte offset of the first input channel inside an image: in_coff * 4 | (in_coff / 8) * H * W * 32
    unsigned chunk_stride;       // bytes from one 8-channel chunk to the next: 32 | H * W * 32
};

__device__ __forceinline__ unsigned wn_div(unsigned n, unsigned d, unsigned magic) { return d == 1 ? n : __umulhi(n, magic); }

__device__ __forceinline__ wn_i32x4 wn_rsrc(const void* base, size_t bytes)
{
    wn_i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(size_t)base);
    r.y = __builtin_amdgcn_readfirstlane((int)(((size_t)base >> 32) & 0xffff));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

// 64 lanes x 16 bytes, global -> LDS: lane l's 16 bytes land at m0 + OFF + 16 l, read from rsrc + soff + voff + OFF (the immediate
// offset moves BOTH addresses: tools/wino/m0_offset_probe.hip); an out-of-range voff writes zeros.  Inline asm: hipcc cannot see
// which LDS bytes a DMA touches and would put vmcnt(0) in front of later ds_reads; the stage loop counts instead.  M0 is bound
// through the "{m0}" constraint: hipcc writes it once per run of pieces that share a base.  s_nop: the wait state between an M0
// write and its LDS-DMA use.
template <int OFF>
__device__ __forceinline__ void wn_dma16(unsigned m0_base, unsigned voff, wn_i32x4 rsrc, unsigned soff)
{
    asm volatile("s_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds"
                 :: "v"(voff), "s"(rsrc), "s"(soff), "n"(OFF), "{m0}"(m0_base) : "memory");
}

__device__ __forceinline__ float wn_act1(float v, int act, float slope)
{
    switch (act) {
        case ESR_ACT_LRELU: return fmaxf(v, slope * v);
        case ESR_ACT_RELU: return fmaxf(v, 0.f);
        case ESR_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        default: return v;
    }
}

// ACT: compile-time activation of the hot instantiations (ESR_ACT_LRELU / ESR_ACT_NONE); -1 = read p.act at run time.
template <int ACT>
__device__ __forceinline__ f32x4 wn_act4(f32x4 v, int act, float slope)
{
    if (ACT == ESR_ACT_LRELU) {
        const f32x4 m = v * slope;
        v.x = fmaxf(v.x, m.x); v.y = fmaxf(v.y, m.y); v.z = fmaxf(v.z, m.z); v.w = fmaxf(v.w, m.w);
    } else if (ACT < 0) {
        v.x = wn_act1(v.x, act, slope); v.y = wn_act1(v.y, act, slope);
        v.z = wn_act1(v.z, act, slope); v.w = wn_act1(v.w, act, slope);
    }
    return v;
}

// slot (16-byte granule) of halo pixel (ly, lx) inside a channel-half plane.  Rows come in pairs of 37 slots so that two rows
// down is an ODD number of slots: lane (tile (tx, ty), k) reads pixel (2 tx + dx, 4 w + 2 ty + r), and for the 32 lanes
// (16 tiles x k in {0, 1}) one ds_read_b64 services together the dword address is 8 tx + 148 ty + 2 k + {0, 1} (+ const):
// all 64 banks once.
__device__ __forceinline__ int wn_slot(int ly, int lx) { return (ly >> 1) * WN_PAIR + (ly & 1) * WN_HALO + lx; }

// one ds_read_b64 from an LDS byte address.  Volatile (on an explicit LDS pointer: a volatile generic pointer becomes a flat load)
// so that hipcc does not fuse two of them into ds_read2_b64 -- half the rate, and 32 banks instead of 64 (MI355X_MICROARCH.md, LDS)
__device__ __forceinline__ f32x2 wn_lds_b64(unsigned addr)
{
    typedef const volatile __attribute__((address_space(3))) f32x2* lds_ptr;
    return *(lds_ptr)addr;
}

// a - b on a channel pair as ONE packed instruction (v_pk_add_f32 with negated source): left to itself hipcc splits 19 of the 24
// subtractions of a transform into two scalar v_add_f32 each (SQ_INSTS_VALU: 1.7 non-MFMA VALU instructions per MFMA, r04a).
// No op_sel: the gfx950 erratum of esr_internal.h does not apply.
__device__ __forceinline__ f32x2 wn_sub2(f32x2 a, f32x2 b)
{
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Channel-blocked stores as whole lines.  A lane (tile j, cout group g) holds 4 channels of the pixels b = 0 / 1 of its tile; stored as they
// are, an instruction writes 32 bytes (lanes g, g + 1: one 8-channel plane) of every other pixel.  v_permlane32_swap on the b = 0 / b = 1
// registers moves the upper half-wave's b = 0 data down and the lower half-wave's b = 1 data up: afterwards `lo` holds, for plane A
// (channels 0..7 of the cout tile), rows g = 0, 1: pixel b = 0 and rows g = 2, 3: pixel b = 1 -- 64 contiguous bytes per tile, 512 per
// row of 8 tiles -- and `hi` the same for plane B (channels 8..15).  Pixel of a lane: b = g >> 1, 16-byte slot g & 1.
__device__ __forceinline__ void wn_pair_planes(f32x4& v0, f32x4& v1)
{
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a = v0[e], b = v1[e];          // (hipcc: __builtin_bit_cast on an ext-vector ELEMENT reads element 0 -- copy to a scalar first)
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
        const unsigned r0 = r[0], r1 = r[1];
        v0[e] = __builtin_bit_cast(float, r0);
        v1[e] = __builtin_bit_cast(float, r1);
    }
}

__device__ __forceinline__ f32x2 wn_add2(f32x2 a, f32x2 b)          // (hipcc makes two scalar v_add_f32 of a plain f32x2 sum here)
{
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// acc += a * b on v_mfma_f32_16x16x4_f32, IN PLACE.  With the builtin hipcc unties vdst from srcC, rotates every accumulator through
// temporaries, reuses the freed registers for the next A fragment and pads the resulting WAR hazard (an LDS load into a register an
// MFMA still reads as srcC) with four to six s_nop 7 per stage.  An asm keeps vdst == srcC.  What hipcc's hazard recognizer then no
// longer sees: the VALU read of an MFMA result (epilogue) -- wn_mfma_drain() in front of the first one.
__device__ __forceinline__ void wn_mfma(f32x4& acc, float a, float b)
{
    asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void wn_mfma_drain(f32x4& acc) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc)); }   // >= 18 wait states: an 8-pass XDL write -> VALU read needs 11 (+1 on gfx950)

// keeps a value's computation where it is written: without it hipcc sinks the whole input transform (it is only consumed by the
// NEXT stage's MFMAs) out of this stage's MFMA stream into the top of the next stage
#define WN_PIN(x) asm volatile("" : "+v"(x))

// OUT: 0 = NHWC views (split store allowed), 1 = out1 channel-blocked [n][C/8][h][w][8] (esr_conv_desc.blocked8), 2 = the network
// output, conv + nn.PixelShuffle(4) fused (ESR_NCHW_SHUFFLE4: out[n, c, 4y+i, 4x+j] = conv[n, 16c+4i+j, y, x], basicblock.py:84-85,446-449)
template <int ACT, int RES, int OUT>
__global__ __launch_bounds__(WN_THREADS, 2) void wino_f32_kernel(const WinoK p)
{
    __shared__ __attribute__((aligned(16))) char smem[WN_LDS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;          // MFMA column (Winograd tile) / K slot (B side), cout group (D side)
    const int tx = j & 7, ty = j >> 3;
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- work walk: item = (tile, cout half); within a pass XCD x (blocks b % 8 == x) takes a contiguous run and the two halves of a
    // tile go to neighbouring blocks.  The two blocks of a CU do not share the MFMA pipe evenly: the older wave of a SIMD is served
    // first and gets through an item ~25 % sooner (tools/wino/abl.py probe: 4.6 k against 5.8 k cycles per stage), and the hardware
    // places blocks 0 .. G/2-1 first, so they are the older ones.  With an even split the younger block of every CU runs alone at the
    // end; the first half of the grid therefore takes first_share/16 of the items.  Speed only: any map is correct.
    const int nwork = p.N * p.tiles_y * p.tiles_x * p.nhalves;
    const int G = gridDim.x;
    const bool classes = p.first_share != 8 && (G & 15) == 0;
    const int GC = classes ? G >> 1 : G;                                   // blocks per class
    const bool second = classes && (int)blockIdx.x >= GC;
    const int n_first = classes ? (int)(((long)nwork * p.first_share / 16) & ~1L) : nwork;
    const int cbase = second ? n_first : 0, cend = second || !classes ? nwork : n_first;
    const int bl = second ? (int)blockIdx.x - GC : (int)blockIdx.x;
    auto work_index = [&](int k) -> int {
        const int base = cbase + k * GC;
        if (base >= cend) return -1;
        int off = bl;
        if ((GC & 7) == 0 && base + GC <= cend) off = (bl & 7) * (GC >> 3) + (bl >> 3);
        const int t = base + off;
        return t < cend ? t : -1;
    };

    // A wave issues 3 raw pieces per stage: slots (3 wv + i) * 64 + lane of the stage image.  Pieces 0 and 1 share one M0 (piece 1
    // through the immediate offset 1024, so its per-lane offset is stored 1024 lower); piece 2 has its own M0.  Wave 3's piece 1 is
    // the partial one (26 lanes: the others stay off, they would write into the neighbouring stage), its piece 2 does not exist:
    // it repeats piece 0 (same bytes to the same place) so that every wave issues the same number of instructions.
    struct Ctx { int n, x0, y0, half; unsigned voff[3]; };
    auto setup = [&](int work, Ctx& c) {
        if (work < 0) {                              // behind the last item: the DMAs still issue (uniform counts), all lanes out of range
            c.n = 0; c.x0 = 0; c.y0 = 0; c.half = 0;
            c.voff[0] = c.voff[1] = c.voff[2] = WN_OOB;
            return;
        }
        const unsigned t = p.nhalves == 2 ? (unsigned)work >> 1 : (unsigned)work;
        c.half = p.nhalves == 2 ? work & 1 : 0;
        const unsigned tq = wn_div(t, (unsigned)p.tiles_x, p.magic_x);
        const unsigned txi = t - tq * p.tiles_x;
        c.n = (int)wn_div(tq, (unsigned)p.tiles_y, p.magic_y);
        const unsigned tyi = tq - (unsigned)c.n * p.tiles_y;
        c.x0 = (int)txi * WN_TILE;
        c.y0 = (int)tyi * WN_TILE;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int item = (wv * 3 + (wv == 3 && i == 2 ? 0 : i)) * 64 + lane;
            const int half = item >= WN_PLANE ? 1 : 0;
            const int slot = item - half * WN_PLANE;
            const int pr = slot / WN_PAIR, rem = slot - pr * WN_PAIR;
            const int odd = rem >= WN_HALO ? 1 : 0;
            const int ly = 2 * pr + odd, lx = rem - odd * WN_HALO;
            const int gy = c.y0 - 1 + ly, gx = c.x0 - 1 + lx;
            const bool ok = item < WN_RAW_SLOTS && rem < 2 * WN_HALO && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            c.voff[i] = (ok ? ((unsigned)(gy * p.W + gx) * p.pix_floats + 4u * half) * 4u + p.in_base : WN_OOB) - (i == 1 ? 1024u : 0u);
        }
    };
    const size_t img_bytes = (size_t)p.H * p.W * p.in_pitch * 4;
    const wn_i32x4 ursrc = wn_rsrc(p.up, p.up_bytes);
    const bool live1 = wv != 3 || lane < WN_LAST_LANES;
    const unsigned raw_m0 = smem_lds + (unsigned)(WN_RAW_OFF + wv * 3 * 1024);            // + ring slot * WN_RAW_BYTES
    const unsigned raw_m0_2 = raw_m0 + (wv == 3 ? 0u : 2048u);
    const unsigned u_m0 = smem_lds + (unsigned)(wv * 4 * 1024);                             // + ring slot * WN_U_BYTES

    // piece i of the raw halo of chunk `chunk` of an item (image rsrc xr, per-lane offset v) -> raw ring slot `rs`
    auto issue_raw = [&](int i, int rs, wn_i32x4 xr, int chunk, unsigned v) __attribute__((always_inline)) {
        const unsigned soff = (unsigned)chunk * p.chunk_stride;
        const unsigned ro = (unsigned)(rs * WN_RAW_BYTES);
        if (i == 0) wn_dma16<0>(raw_m0 + ro, v, xr, soff);
        else if (i == 1) { if (live1) wn_dma16<1024>(raw_m0 + ro, v, xr, soff); }
        else wn_dma16<0>(raw_m0_2 + ro, v, xr, soff);
    };
    auto image_rsrc = [&](int n) { return wn_rsrc(p.x + (size_t)n * (img_bytes / 4), img_bytes); };
    // piece i of the U chunk of cout half `half` -> U ring slot `us`; live = false: out of range (zeros into a slot nobody reads)
    auto issue_u = [&](int i, int us, int half, int chunk, bool live) __attribute__((always_inline)) {
        const unsigned soff = (unsigned)((chunk * p.nhalves + half) * WN_U_BYTES + wv * 4 * 1024);
        const unsigned vo = live ? (unsigned)lane * 16u : WN_OOB;
        const unsigned m = u_m0 + (unsigned)(us * WN_U_BYTES);
        if (i == 0) wn_dma16<0>(m, vo, ursrc, soff);
        else if (i == 1) wn_dma16<1024>(m, vo, ursrc, soff);
        else if (i == 2) wn_dma16<2048>(m, vo, ursrc, soff);
        else wn_dma16<3072>(m, vo, ursrc, soff);
    };

    // lane-constant LDS byte offsets
    const int raw_lane = (g >> 1) * (WN_PLANE * 16) + wn_slot(4 * wv + 2 * ty, 2 * tx) * 16 + (g & 1) * 8;
    const int u_lane = lane * 16;

    // ---- input transform of one (tile, channel pair): V = B^T d B,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
    // row pass (needs one patch row): w[r][.] = d[r][.] B;   column pass: V[.][c] = B^T w[.][c];   position = 4 * row + column.
    // The patch reads stay single ds_read_b64 (wn_lds_b64: two 32-lane groups, 64 banks -- conflict-free for this layout); fused
    // into ds_read2_b64 they ran at half the rate on 32 banks (PMC: a third of all LDS cycles were bank conflicts).
    // rb = ring slot base + this lane's patch origin, ONE register per stage (made opaque: otherwise hipcc keeps the 16 loop-invariant
    // sums raw_lane + constant in 16 registers and adds the slot base to each of them per read: 16 v_add_u32 per stage, no offsets)
    auto raw_row = [&](unsigned rb, int r, f32x2 (&d)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int dx = 0; dx < 4; ++dx)
            d[dx] = wn_lds_b64(rb + ((r >> 1) * WN_PAIR + (r & 1) * WN_HALO + dx) * 16);
    };
    auto row_pass = [&](f32x2 (&V)[16], int r, const f32x2 (&d)[4]) __attribute__((always_inline)) {
        V[4 * r + 0] = wn_sub2(d[0], d[2]);
        V[4 * r + 1] = wn_add2(d[1], d[2]);
        V[4 * r + 2] = wn_sub2(d[2], d[1]);
        V[4 * r + 3] = wn_sub2(d[1], d[3]);
    };
    auto col_pass = [&](f32x2 (&V)[16], int c) __attribute__((always_inline)) {
        const f32x2 w0 = V[c], w1 = V[4 + c], w2 = V[8 + c], w3 = V[12 + c];
        V[c] = wn_sub2(w0, w2);
        V[4 + c] = wn_add2(w1, w2);
        V[8 + c] = wn_sub2(w2, w1);
        V[12 + c] = wn_sub2(w1, w3);
    };

    int k = 0;
    int work = work_index(0);
    if (work < 0) return;
    Ctx cur, nxt;
    setup(work, cur);
    int wn = work_index(1);
    setup(wn, nxt);

    // ---- prologue: raw chunks 0..2 and U chunk 0 of the first item; V of chunk 0 is computed alone
    {
        const wn_i32x4 xr = image_rsrc(cur.n);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 3; ++i) issue_raw(i, c, xr, c, cur.voff[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_u(i, 0, cur.half, 0, true);
    }
    // bias of both cout halves -> LDS (a per-item global load would make hipcc wait vmcnt(0), i.e. for the DMAs in flight)
    if (tid < p.nhalves * 8) *reinterpret_cast<f32x4*>(smem + WN_BIAS_OFF + tid * 16) = *reinterpret_cast<const f32x4*>(p.bias + tid * 4);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x2 V0[16], V1[16];
    {
        unsigned rs = smem_lds + WN_RAW_OFF + raw_lane;
        asm volatile("" : "+v"(rs));
        f32x2 d[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { raw_row(rs, r, d); row_pass(V0, r, d); }
#pragma unroll
        for (int c = 0; c < 4; ++c) col_pass(V0, c);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // raw slot 0 is overwritten by the first stage
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_u(i, 1, cur.half, 1, true);
    f32x4 a[4];                                   // A fragments two positions ahead of their MFMAs, across stage boundaries
    a[0] = *reinterpret_cast<const f32x4*>(smem + u_lane);
    a[1] = *reinterpret_cast<const f32x4*>(smem + u_lane + 1024);

    int us = 0;                                   // U ring slot of the current stage
    int rsn = 1;                                  // raw ring slot of the NEXT stage (read by this stage's transform)

    f32x4 acc[16][2];
    for (;;) {
        const bool has_next = wn >= 0;
        f32x4 biasv[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) biasv[ct] = *reinterpret_cast<const f32x4*>(smem + WN_BIAS_OFF + (cur.half * 32 + ct * 16 + g * 4) * 4);

        // one K stage g (chunk c of the item): 64 MFMAs from (U ring slot us, Vc) while Vn = transform(raw ring slot rsn) is built.
        // Ring contents: behind the barrier of stage h the block issues U chunk h + 2 and (in the first positions of stage h + 1) raw
        // chunk h + 4; the barrier of stage g hands over U g + 1 and raw g + 2.
        // FIRST: the item's first stage -- the accumulators start from constants (A^T M A of a bias b placed at position (1, 1)
        // is b in all four outputs: no bias add in the epilogue, no accumulator clearing)
        auto stage = [&](auto first_tag, int c, f32x2 (&Vc)[16], f32x2 (&Vn)[16]) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const bool rc = c + 3 < p.nchunks;
            const wn_i32x4 xr = image_rsrc(rc ? cur.n : nxt.n);
            const int rchunk = rc ? c + 3 : c + 3 - p.nchunks;
            int rs3 = rsn + 2;
            rs3 = rs3 >= 3 ? rs3 - 3 : rs3;
            const bool uc = c + 2 < p.nchunks;
            const int uhalf = uc ? cur.half : nxt.half, uchunk = uc ? c + 2 : c + 2 - p.nchunks;
            const bool ulive = uc || has_next;
            int us1 = us + 1, us2 = us + 2;
            us1 = us1 >= 3 ? us1 - 3 : us1;
            us2 = us2 >= 3 ? us2 - 3 : us2;
            const char* ust = smem + us * WN_U_BYTES + u_lane;
            const char* ust1 = smem + us1 * WN_U_BYTES + u_lane;
            unsigned rs = smem_lds + (unsigned)(WN_RAW_OFF + rsn * WN_RAW_BYTES) + raw_lane;
            asm volatile("" : "+v"(rs));
            f32x2 d[2][4];
#pragma unroll
            for (int pos = 0; pos < 16; ++pos) {
                if (pos == WN_BARRIER_POS) {
                    // hand-over: U chunk +1 (older than the 3 raw pieces of this stage) and raw chunk +2 (issued a stage ago) have landed
                    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                if (pos + 2 < 16) a[(pos + 2) & 3] = *reinterpret_cast<const f32x4*>(ust + (pos + 2) * 1024);
                else a[(pos + 2) & 3] = *reinterpret_cast<const f32x4*>(ust1 + (pos + 2 - 16) * 1024);
                if (pos < 4) raw_row(rs, pos, d[pos & 1]);
                if (pos >= 1 && pos < 5) {
                    row_pass(Vn, pos - 1, d[(pos - 1) & 1]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) WN_PIN(Vn[4 * (pos - 1) + q]);
                }
                if (pos >= 5 && pos < 9) {
                    col_pass(Vn, pos - 5);
#pragma unroll
                    for (int q = 0; q < 4; ++q) WN_PIN(Vn[4 * q + pos - 5]);
                }
                if (pos == 2 || pos == 4 || pos == 6) {
                    const int i = pos / 2 - 1;
                    issue_raw(i, rs3, xr, rchunk, rc ? cur.voff[i] : nxt.voff[i]);
                }
                if (pos >= WN_BARRIER_POS) issue_u(pos - WN_BARRIER_POS, us2, uhalf, uchunk, ulive);
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                const f32x4 c0 = FIRST ? (pos == 5 ? biasv[0] : zero) : acc[pos][0];
                const f32x4 c1 = FIRST ? (pos == 5 ? biasv[1] : zero) : acc[pos][1];
                const f32x4 af = a[pos & 3];
                if (FIRST) {
                    acc[pos][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, Vc[pos].x, c0, 0, 0, 0);
                    acc[pos][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, Vc[pos].x, c1, 0, 0, 0);
                } else {                                       // in place (wn_mfma): no accumulator rotation, no WAR padding
                    wn_mfma(acc[pos][0], af.x, Vc[pos].x);
                    wn_mfma(acc[pos][1], af.z, Vc[pos].x);
                }
                wn_mfma(acc[pos][0], af.y, Vc[pos].y);
                wn_mfma(acc[pos][1], af.w, Vc[pos].y);
                __builtin_amdgcn_sched_barrier(0);
            }
            us = us1;
            rsn = rsn == 2 ? 0 : rsn + 1;
        };
        stage(std::true_type{}, 0, V0, V1);
        stage(std::false_type{}, 1, V1, V0);
        for (int c = 2; c < p.nchunks; c += 2) {
            stage(std::false_type{}, c, V0, V1);
            stage(std::false_type{}, c + 1, V1, V0);
        }

        // ---- output transform  Y = A^T M A,  A^T = [1 1 1 0; 0 1 -1 -1], then residual / activation / store
        {
            wn_mfma_drain(acc[15][1]);
            const int ybase = cur.y0 + 4 * wv + 2 * ty, xbase = cur.x0 + 2 * tx;
            unsigned pix[2][2];
            bool pok[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    pok[a][b] = ybase + a < p.H && xbase + b < p.W;
                    pix[a][b] = (unsigned)((ybase + a) * p.W + xbase + b);
                }
            const size_t hw = (size_t)p.H * p.W;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int cg = cur.half * 32 + ct * 16 + g * 4;            // this lane's first output channel
                const bool to0 = cur.half * 32 + ct * 16 < p.split;        // uniform: host guarantees split % 16 == 0 or no split
                const bool cok = cg < p.cout_store;
                const int ch = to0 ? p.y0_coff + cg : p.y1_coff + cg - p.split;
                const bool blk = OUT == 1 && !to0;
                char* const dbase = reinterpret_cast<char*>(to0 ? p.y0 : p.y1) + (size_t)cur.n * hw * (size_t)((to0 ? p.y0_pitch : p.y1_pitch) * 4);
                const unsigned dps = blk ? 32u : (unsigned)((to0 ? p.y0_pitch : p.y1_pitch) * 4);
                const size_t dlane = blk ? (size_t)(ch >> 3) * hw * 32 + (size_t)(ch & 7) * 4 : (size_t)ch * 4;
                f32x4 rv[2][2];
                if (RES != ESR_RES_NONE) {
                    const char* rbase = reinterpret_cast<const char*>(p.res) + (size_t)cur.n * hw * (size_t)(p.res_pitch * 4) + (size_t)(p.res_coff + cg) * 4;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            rv[a][b] = (pok[a][b] && cok) ? *reinterpret_cast<const f32x4*>(rbase + (size_t)pix[a][b] * (unsigned)(p.res_pitch * 4))
                                                          : f32x4{0.f, 0.f, 0.f, 0.f};
                }
                f32x4 Y[2][2];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const f32x4 m0 = acc[jj][ct], m1 = acc[4 + jj][ct], m2 = acc[8 + jj][ct], m3 = acc[12 + jj][ct];
                    const f32x4 t0 = m0 + m1 + m2, t1 = m1 - m2 - m3;
                    if (jj == 0) { Y[0][0] = t0; Y[1][0] = t1; }
                    else if (jj == 1) { Y[0][0] += t0; Y[1][0] += t1; Y[0][1] = t0; Y[1][1] = t1; }
                    else if (jj == 2) { Y[0][0] += t0; Y[1][0] += t1; Y[0][1] -= t0; Y[1][1] -= t1; }
                    else { Y[0][1] -= t0; Y[1][1] -= t1; }
                }
                if (blk) {
                    // RES_NONE by construction (esr_wino_supported); whole-line stores of the tile's two 8-channel planes (wn_pair_planes)
                    char* const pa = dbase + (size_t)((p.y1_coff + cur.half * 32 + ct * 16 - p.split) >> 3) * hw * 32 + (size_t)(g & 1) * 16;
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        f32x4 v0 = wn_act4<ACT>(Y[a][0], p.act, p.slope), v1 = wn_act4<ACT>(Y[a][1], p.act, p.slope);
                        wn_pair_planes(v0, v1);
                        const bool ok = (g >> 1) ? pok[a][1] : pok[a][0];
                        char* const q = pa + (size_t)((g >> 1) ? pix[a][1] : pix[a][0]) * 32;
                        if (ok) {
                            *reinterpret_cast<f32x4*>(q) = v0;
                            *reinterpret_cast<f32x4*>(q + hw * 32) = v1;
                        }
                    }
                } else
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        f32x4 v = Y[a][b];
                        if (RES == ESR_RES_PRE_ACT) v = wn_act4<ACT>(v + rv[a][b], p.act, p.slope);
                        else if (RES == ESR_RES_POST_ACT) v = wn_act4<ACT>(v, p.act, p.slope) + rv[a][b];
                        else v = wn_act4<ACT>(v, p.act, p.slope);
                        if (OUT == 2) {
                            // lane (tile, g) holds channels 16c + 4g + {0..3} of pixel (y, x): the four HR pixels (4y + g, 4x .. 4x+3) of plane c
                            const int cidx = cur.half * 2 + ct, nco = p.cout_store >> 4;
                            float* const o = p.y0 + ((size_t)(cur.n * nco + cidx) * (4 * p.H) + 4 * (ybase + a) + g) * (size_t)(4 * p.W) + 4 * (xbase + b);
                            if (pok[a][b] && cidx < nco) *reinterpret_cast<f32x4*>(o) = v;
                        } else if (pok[a][b] && cok) *reinterpret_cast<f32x4*>(dbase + dlane + (size_t)pix[a][b] * dps) = v;
                    }
            }
        }
        if (!has_next) break;
        cur = nxt;
        ++k;
        wn = work_index(k + 1);
        setup(wn, nxt);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the out-of-range DMAs issued behind the last item
}


// ==================================================================================================================================
// wino8_f32_kernel<ACT, OUT, NCH> -- the same arithmetic for layers of NCH <= 6 input chunks (cin <= 48: IMDBlock conv2 / conv3), organised
// around what round 4's counters said bounds wino_f32_kernel (profiles/r04_c1_sq_counters.md, LAB_NOTES.md 9.2): the length of a wave's
// own instruction stream per MFMA -- 7 LDS-DMA pieces (~100 cycles of issue each) and a block barrier per 64 MFMAs.
//
//   block     8 waves, ONE block per CU.  The U of the block's cout half for ALL input chunks is RESIDENT in LDS (NCH x 16 KB, loaded once
//             per block): no U DMA in the loop.
//   wave      autonomous: it walks its own list of STRIPS (4 output rows x 16 columns = 2 x 8 Winograd tiles = the N side of the MFMA,
//             all 16 positions, 2 cout tiles -- the accumulator shape of wino_f32_kernel) and stages its own 6 x 18-pixel halo, one 8-channel
//             chunk at a time, into a PRIVATE ring of two slots (4 DMA pieces per chunk instead of 7).  No wave ever reads what another wave
//             staged, so the loop has NO barrier: the two waves of a SIMD drift apart, one's epilogue and DMA issue run under the other's MFMAs.
//   stage     chunk c of a strip: 64 MFMAs (A fragments two positions ahead from the resident U); between them, from position W8_TP on: s_waitcnt
//             vmcnt(4 (+ 8 stores of an epilogue in between)), the transform of chunk c + 1 from slot (c + 1) & 1, and -- as soon as its four
//             rows have been read -- the DMA of chunk c + 3 into the same slot: 1.75 stages ahead of the transform that reads it.
//             Chunks run on across strip boundaries (the last stages of a strip stage and transform the next strip's first chunks).
//   epilogue  stores as unconditional buffer stores (invalid lanes: out-of-range offsets), so the vmcnt arithmetic is exact.
// Shapes: RES_NONE only (conv2 / conv3 have no residual), NHWC split or channel-blocked second output.
constexpr int W8_THREADS = 512;
constexpr int W8_PLANE = 3 * WN_PAIR;                  // 6 halo rows = 3 row pairs of 37 slots, per channel-half plane
constexpr int W8_SLOTS = 2 * W8_PLANE;                 // 222 slots of 16 B = 3 full DMA pieces + one of 30 lanes
constexpr int W8_SLOT_BYTES = W8_SLOTS * 16;           // 3552
constexpr int W8_LAST_LANES = W8_SLOTS - 192;
constexpr int W8_EPI_STORES = 8;                       // buffer stores per wave and strip (2 cout tiles x 2 x 2 pixels)
constexpr int W8_MAX_BLOCKS = 256;
#ifndef W8_TP
#define W8_TP 2
#endif
#ifndef W8_DMA_TOP
#define W8_DMA_TOP 0
#endif
// ablation switches of tools/wino/w8_variants.sh (research builds only; results are WRONG with any of them set)
#ifndef W8_STAGGER
#define W8_STAGGER 0          // s_sleep units (64 cycles) the second wave of every SIMD waits before its first stage
#endif
#ifndef W8_ABL_NODMA
#define W8_ABL_NODMA 0
#endif
#ifndef W8_ABL_NOXF
#define W8_ABL_NOXF 0
#endif
#ifndef W8_ABL_NOA
#define W8_ABL_NOA 0
#endif
#ifndef W8_ABL_NOEPI
#define W8_ABL_NOEPI 0
#endif
#ifndef W8_ABL_NOMFMA
#define W8_ABL_NOMFMA 0
#endif
#ifndef W8_ABL_LAXWAIT
#define W8_ABL_LAXWAIT 0        // timing probe (wrong results possible): stages 2 and 3 also leave the epilogue's stores outstanding
#endif
#ifndef W8_ABL_OOBST
#define W8_ABL_OOBST 0          // every store out of range (issued, dropped: no write traffic)
#endif
#ifndef W8_ABL_OOBDMA
#define W8_ABL_OOBDMA 0         // every halo piece out of range (issued, zero fill: no read traffic)
#endif

template <int NCH> struct W8L {
    static constexpr int U_BYTES = NCH * WN_U_BYTES;
    static constexpr int RAW_OFF = U_BYTES;
    static constexpr int BIAS_OFF = RAW_OFF + 8 * 2 * W8_SLOT_BYTES;
    static constexpr int TOTAL = BIAS_OFF + 256;
    static_assert(TOTAL <= 160 * 1024, "one block per CU");
};

// s_nop 1: a store of more than 64 bits reads its data registers AFTER it has issued, and a VALU write to them needs 2 wait states behind
// it on gfx940+ (LLVM: checkVALUHazardsHelper / VmemStoreHazard).  hipcc inserts them for its own stores but cannot see into an asm:
// without the nop the next output's arithmetic landed in v.x of lanes 12..15 of every row before the store had read it (r04f).
__device__ __forceinline__ void wn_store16(f32x4 v, unsigned voff, wn_i32x4 rsrc)
{
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" :: "v"(v), "v"(voff), "s"(rsrc) : "memory");
}

template <int ACT, int OUT, int NCH>
__global__ __launch_bounds__(W8_THREADS, 1) void wino8_f32_kernel(const WinoK p)
{
    typedef W8L<NCH> LY;
    __shared__ __attribute__((aligned(16))) char smem[LY::TOTAL];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int tx = j & 7, ty = j >> 3;
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- which strips: blocks b and b + 8 (same XCD) take the two cout halves of the same strips; a block owns a contiguous run of
    // strips and its 8 waves take 8 consecutive ones (horizontal neighbours) per step
    const int nh = p.nhalves;
    const int half = nh == 2 ? ((int)blockIdx.x >> 3) & 1 : 0;
    const int pidx = nh == 2 ? (((int)blockIdx.x >> 4) << 3) + ((int)blockIdx.x & 7) : (int)blockIdx.x;
    const int npair = nh == 2 ? (int)gridDim.x >> 1 : (int)gridDim.x;
    const int nstrips = p.N * p.tiles_y * p.tiles_x;                       // tiles_x = strips per row (16 px), tiles_y = strips per column (4 px)
    const int per_block = (nstrips + npair - 1) / npair;
    const int s_end = min((pidx + 1) * per_block, nstrips);
    auto strip_index = [&](int k) -> int {
        const int s = pidx * per_block + k * 8 + wv;
        return s < s_end ? s : -1;
    };

    // Two cursors: `cur` (image, origin of the strip whose MFMAs run: wave-uniform, SGPRs) and the DMA cursor `dvoff` / `dn` (per-lane
    // offsets of the 4 halo pieces and the image of the strip whose chunks are being STAGED) -- it runs two chunks ahead and moves to
    // the next strip in stage NCH - 2, so only one set of offsets is live
    struct Ctx { int n, x0, y0; };
    unsigned dvoff[4];
    int dn = 0;
    auto locate = [&](int strip, Ctx& c) {
        if (strip < 0) { c.n = 0; c.x0 = 0; c.y0 = 0; return; }
        const unsigned sq = wn_div((unsigned)strip, (unsigned)p.tiles_x, p.magic_x);
        const unsigned sxi = (unsigned)strip - sq * p.tiles_x;
        c.n = (int)wn_div(sq, (unsigned)p.tiles_y, p.magic_y);
        const unsigned syi = sq - (unsigned)c.n * p.tiles_y;
        c.x0 = (int)sxi * 16;
        c.y0 = (int)syi * 4;
    };
    auto dma_to = [&](int strip, const Ctx& c) {
        dn = c.n;
        if (strip < 0) {                             // behind the last strip: the pieces still issue (uniform counts), all lanes out of range
            dvoff[0] = dvoff[1] = dvoff[2] = dvoff[3] = WN_OOB;
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int item = i * 64 + lane;
            const int plane = item >= W8_PLANE ? 1 : 0;
            const int slot = item - plane * W8_PLANE;
            const int pr = slot / WN_PAIR, rem = slot - pr * WN_PAIR;
            const int odd = rem >= WN_HALO ? 1 : 0;
            const int ly = 2 * pr + odd, lx = rem - odd * WN_HALO;
            const int gy = c.y0 - 1 + ly, gx = c.x0 - 1 + lx;
            const bool ok = item < W8_SLOTS && rem < 2 * WN_HALO && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            dvoff[i] = ((ok && !W8_ABL_OOBDMA) ? ((unsigned)(gy * p.W + gx) * p.pix_floats + 4u * plane) * 4u + p.in_base : WN_OOB) - (unsigned)(i * 1024);
        }
    };
    const size_t img_bytes = (size_t)p.H * p.W * p.in_pitch * 4;
    auto image_rsrc = [&](int n) { return wn_rsrc(p.x + (size_t)n * (img_bytes / 4), img_bytes); };
    const unsigned ring_m0 = smem_lds + (unsigned)(LY::RAW_OFF + wv * 2 * W8_SLOT_BYTES);
    const bool live3 = lane < W8_LAST_LANES;
    // the 4 pieces of chunk `chunk` of an image (rsrc xr, per-lane offsets v[]) -> this wave's ring slot `par`
    auto issue_raw = [&](int par, int chunk) __attribute__((always_inline)) {
        const wn_i32x4 xr = image_rsrc(dn);
        const unsigned soff = (unsigned)chunk * p.chunk_stride;
        const unsigned m = ring_m0 + (unsigned)(par * W8_SLOT_BYTES);
        wn_dma16<0>(m, dvoff[0], xr, soff);
        wn_dma16<1024>(m, dvoff[1], xr, soff);
        wn_dma16<2048>(m, dvoff[2], xr, soff);
        if (live3) wn_dma16<3072>(m, dvoff[3], xr, soff);      // 30 lanes: the others would write into the next slot
    };

    int work = strip_index(0);
    // ---- resident U of this block's cout half + bias (every wave takes part, also one without strips: the barrier below is the only one)
    {
        const wn_i32x4 ursrc = wn_rsrc(p.up, p.up_bytes);
#pragma unroll 1
        for (int pc = wv; pc < NCH * 16; pc += 8) {
            const int chunk = pc >> 4, piece = pc & 15;
            wn_dma16<0>(smem_lds + (unsigned)(chunk * WN_U_BYTES + piece * 1024), (unsigned)lane * 16u, ursrc,
                        (unsigned)((chunk * nh + half) * WN_U_BYTES + piece * 1024));
        }
        if (tid < nh * 8) *reinterpret_cast<f32x4*>(smem + LY::BIAS_OFF + tid * 16) = *reinterpret_cast<const f32x4*>(p.bias + tid * 4);
    }
    Ctx cur, nxt;
    locate(work, cur);
    int wn = strip_index(1);
    locate(wn, nxt);
    if (work >= 0) {
        dma_to(work, cur);
        issue_raw(0, 0);
        issue_raw(1, 1);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (work < 0) return;

    const int raw_lane = (g >> 1) * (W8_PLANE * 16) + wn_slot(2 * ty, 2 * tx) * 16 + (g & 1) * 8;
    const int u_lane = lane * 16;
    auto raw_row = [&](unsigned rb, int r, f32x2 (&d)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int dx = 0; dx < 4; ++dx)
            d[dx] = wn_lds_b64(rb + ((r >> 1) * WN_PAIR + (r & 1) * WN_HALO + dx) * 16);
    };
    auto row_pass = [&](f32x2 (&V)[16], int r, const f32x2 (&d)[4]) __attribute__((always_inline)) {
        V[4 * r + 0] = wn_sub2(d[0], d[2]);
        V[4 * r + 1] = wn_add2(d[1], d[2]);
        V[4 * r + 2] = wn_sub2(d[2], d[1]);
        V[4 * r + 3] = wn_sub2(d[1], d[3]);
    };
    auto col_pass = [&](f32x2 (&V)[16], int c) __attribute__((always_inline)) {
        const f32x2 w0 = V[c], w1 = V[4 + c], w2 = V[8 + c], w3 = V[12 + c];
        V[c] = wn_sub2(w0, w2);
        V[4 + c] = wn_add2(w1, w2);
        V[8 + c] = wn_sub2(w2, w1);
        V[12 + c] = wn_sub2(w1, w3);
    };

    // V of chunk 0 of the first strip, alone
    f32x2 V0[16], V1[16];
    {
        unsigned rs = ring_m0 + raw_lane;
        asm volatile("" : "+v"(rs));
        f32x2 d[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { raw_row(rs, r, d); row_pass(V0, r, d); }
#pragma unroll
        for (int c = 0; c < 4; ++c) col_pass(V0, c);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // slot 0 is free: chunk 2 goes there
    if (!W8_DMA_TOP) issue_raw(0, 2);
    f32x4 a[4];
    a[0] = *reinterpret_cast<const f32x4*>(smem + u_lane);
    a[1] = *reinterpret_cast<const f32x4*>(smem + u_lane + 1024);

    // The two waves of a SIMD (w and w + 4) run the same instruction stream on equal work: left alone they stay in step, reach their
    // epilogues (~350 instructions without an MFMA) and their DMA issue together, and the matrix pipe idles -- the ablations of r04 put
    // 14 % + 15 % of the launch there.  Half a strip of head start for one of them keeps one wave's epilogue under the other's MFMAs.
    if (W8_STAGGER > 0 && wv >= 4) {
#pragma unroll
        for (int i = 0; i < (W8_STAGGER + 126) / 127; ++i) __builtin_amdgcn_s_sleep(W8_STAGGER < 127 ? W8_STAGGER : 127);
    }
    int k = 0;
    f32x4 acc[16][2];
    if (W8_ABL_NOXF) {
#pragma unroll
        for (int q = 0; q < 16; ++q) V1[q] = V0[q];
    }
    if (W8_ABL_NOA) { a[2] = a[0]; a[3] = a[1]; }
    const char* const bias_lds = smem + LY::BIAS_OFF + (half * 32 + g * 4) * 4;
    for (;;) {
        const bool has_next = wn >= 0;
        // stage of chunk c: DMA of chunk c + 2 into slot c & 1 (the slot the previous stage's transform has finished reading), transform of
        // chunk c + 1 (slot (c + 1) & 1), 64 MFMAs.  AFTER_EPI: the first stage of a strip other than the wave's first -- the previous
        // strip's W8_EPI_STORES stores were issued between the DMA this stage waits for and the DMA it issues.
        auto stage = [&](auto first_tag, int c, f32x2 (&Vc)[16], f32x2 (&Vn)[16]) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            if (W8_DMA_TOP) {                                               // research variant: chunk c + 2 at the top of stage c (1 stage + W8_TP positions of lead)
                if (c + 2 == NCH) dma_to(wn, nxt);
                issue_raw(c & 1, c + 2 < NCH ? c + 2 : c + 2 - NCH);
            }
            f32x4 biasv0, biasv1;
            const int cn = c + 1 < NCH ? c + 1 : 0;                       // U chunk of the next stage (position look-ahead)
            const char* ust = smem + c * WN_U_BYTES + u_lane;
            const char* ust1 = smem + cn * WN_U_BYTES + u_lane;
            unsigned rs = ring_m0 + (unsigned)(((c + 1) & 1) * W8_SLOT_BYTES) + raw_lane;
            asm volatile("" : "+v"(rs));
            // chunk c + 1 has landed when only this stage's 3 or 4 pieces (+ the epilogue's stores) are still in flight.  A wave with lanes
            // >= 30 ... every wave issues the 4th piece (lanes < 30 exist in every wave), so the count is 4
            f32x2 d[2][4];
#pragma unroll
            for (int pos = 0; pos < 16; ++pos) {
                if (!W8_ABL_NOA) {
                if (pos + 2 < 16) a[(pos + 2) & 3] = *reinterpret_cast<const f32x4*>(ust + (pos + 2) * 1024);
                else a[(pos + 2) & 3] = *reinterpret_cast<const f32x4*>(ust1 + (pos + 2 - 16) * 1024);
                }
                // the transform of the NEXT chunk sits in positions W8_TP .. W8_TP + 8: with a ring of two slots its DMA was issued only
                // one stage ago, so the wait comes as late as the stage allows
                if (pos == W8_TP) {
                    // chunk c + 1 has landed when only the 4 pieces of chunk c + 2 are younger -- and, in the first two stages of a strip
                    // that is not the wave's first, the 8 stores of the epilogue in between (every wave issues all 4 pieces and all 8 stores)
                    if (c < (W8_DMA_TOP ? 1 : 2) + 2 * W8_ABL_LAXWAIT && k > 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 + W8_EPI_STORES) : "memory");
                    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                }
                if (!W8_DMA_TOP && pos == W8_TP + 5) {
                    // the four rows of chunk c + 1 have been read (row_pass has consumed them): its slot takes chunk c + 3, two stages ahead
                    // of the transform that will read it
                    if (c + 3 == NCH) dma_to(wn, nxt);                        // the DMA cursor moves on to the next strip
                    if (!W8_ABL_NODMA) issue_raw((c + 1) & 1, c + 3 < NCH ? c + 3 : c + 3 - NCH);
                }
                if (!W8_ABL_NOXF && pos >= W8_TP && pos < W8_TP + 4) raw_row(rs, pos - W8_TP, d[(pos - W8_TP) & 1]);
                if (!W8_ABL_NOXF && pos >= W8_TP + 1 && pos < W8_TP + 5) {
                    row_pass(Vn, pos - W8_TP - 1, d[(pos - W8_TP - 1) & 1]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) WN_PIN(Vn[4 * (pos - W8_TP - 1) + q]);
                }
                if (!W8_ABL_NOXF && pos >= W8_TP + 5 && pos < W8_TP + 9) {
                    col_pass(Vn, pos - W8_TP - 5);
#pragma unroll
                    for (int q = 0; q < 4; ++q) WN_PIN(Vn[4 * q + pos - W8_TP - 5]);
                }
                if (FIRST && pos == 3) {                                     // the bias enters at position 5 (see wino_f32_kernel)
                    biasv0 = *reinterpret_cast<const f32x4*>(bias_lds);
                    biasv1 = *reinterpret_cast<const f32x4*>(bias_lds + 64);
                }
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                const f32x4 c0 = FIRST ? (pos == 5 ? biasv0 : zero) : acc[pos][0];
                const f32x4 c1 = FIRST ? (pos == 5 ? biasv1 : zero) : acc[pos][1];
                const f32x4 af = a[pos & 3];
                if (W8_ABL_NOMFMA) {
                    acc[pos][0] = c0; acc[pos][1] = c1;
                    acc[pos][0].x += af.x * Vc[pos].x + af.y * Vc[pos].y; acc[pos][1].x += af.z * Vc[pos].x + af.w * Vc[pos].y;
                } else if (FIRST) {
                acc[pos][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, Vc[pos].x, c0, 0, 0, 0);
                acc[pos][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, Vc[pos].x, c1, 0, 0, 0);
                wn_mfma(acc[pos][0], af.y, Vc[pos].y);
                wn_mfma(acc[pos][1], af.w, Vc[pos].y);
                } else {
                wn_mfma(acc[pos][0], af.x, Vc[pos].x);
                wn_mfma(acc[pos][1], af.z, Vc[pos].x);
                wn_mfma(acc[pos][0], af.y, Vc[pos].y);
                wn_mfma(acc[pos][1], af.w, Vc[pos].y);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        stage(std::true_type{}, 0, V0, V1);
        stage(std::false_type{}, 1, V1, V0);
#pragma unroll
        for (int c = 2; c < NCH; c += 2) {
            stage(std::false_type{}, c, V0, V1);
            stage(std::false_type{}, c + 1, V1, V0);
        }

        if (W8_ABL_NOEPI) {
            if (p.N < 0) {                        // never true: keeps the accumulators alive
                f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 16; ++q) sum += acc[q][0] + acc[q][1];
                *reinterpret_cast<f32x4*>(p.y0 + lane * 4) = sum;
            }
        } else
        // ---- output transform Y = A^T M A, activation, 8 unconditional buffer stores (out-of-range offset = dropped)
        {
            wn_mfma_drain(acc[15][1]);
            const int ybase = cur.y0 + 2 * ty, xbase = cur.x0 + 2 * tx;
            const size_t hw = (size_t)p.H * p.W;
            unsigned pix[2][2];
            bool pok[2][2];
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    pok[aa][b] = ybase + aa < p.H && xbase + b < p.W;
                    pix[aa][b] = (unsigned)((ybase + aa) * p.W + xbase + b);
                }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int cg = half * 32 + ct * 16 + g * 4;
                const bool to0 = half * 32 + ct * 16 < p.split;
                const bool cok = cg < p.cout_store;
                const int ch = to0 ? p.y0_coff + cg : p.y1_coff + cg - p.split;
                const bool blk = OUT == 1 && !to0;
                const unsigned pitch_b = (unsigned)((to0 ? p.y0_pitch : p.y1_pitch) * 4);
                const size_t ibytes = hw * pitch_b;
                const wn_i32x4 yr = wn_rsrc(reinterpret_cast<const char*>(to0 ? p.y0 : p.y1) + (size_t)cur.n * ibytes, ibytes);
                const unsigned dps = blk ? 32u : pitch_b;
                const unsigned dlane = blk ? (unsigned)(ch >> 3) * (unsigned)(hw * 32) + (unsigned)(ch & 7) * 4u : (unsigned)ch * 4u;
                f32x4 Y[2][2];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const f32x4 m0 = acc[jj][ct], m1 = acc[4 + jj][ct], m2 = acc[8 + jj][ct], m3 = acc[12 + jj][ct];
                    const f32x4 t0 = m0 + m1 + m2, t1 = m1 - m2 - m3;
                    if (jj == 0) { Y[0][0] = t0; Y[1][0] = t1; }
                    else if (jj == 1) { Y[0][0] += t0; Y[1][0] += t1; Y[0][1] = t0; Y[1][1] = t1; }
                    else if (jj == 2) { Y[0][0] += t0; Y[1][0] += t1; Y[0][1] -= t0; Y[1][1] -= t1; }
                    else { Y[0][1] -= t0; Y[1][1] -= t1; }
                }
                if (blk) {
                    // whole-line stores of the two 8-channel planes of this cout tile (wn_pair_planes)
                    const unsigned plane_a = (unsigned)((p.y1_coff + half * 32 + ct * 16 - p.split) >> 3) * (unsigned)(hw * 32) + (unsigned)(g & 1) * 16u;
#pragma unroll
                    for (int aa = 0; aa < 2; ++aa) {
                        f32x4 v0 = wn_act4<ACT>(Y[aa][0], p.act, p.slope), v1 = wn_act4<ACT>(Y[aa][1], p.act, p.slope);
                        wn_pair_planes(v0, v1);
                        const bool ok = (g >> 1) ? pok[aa][1] : pok[aa][0];
                        const unsigned po = plane_a + ((g >> 1) ? pix[aa][1] : pix[aa][0]) * 32u;
                        wn_store16(v0, (ok && !W8_ABL_OOBST) ? po : WN_OOB, yr);
                        wn_store16(v1, (ok && !W8_ABL_OOBST) ? po + (unsigned)(hw * 32) : WN_OOB, yr);
                    }
                } else {
#pragma unroll
                for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const f32x4 v = wn_act4<ACT>(Y[aa][b], p.act, p.slope);
                        wn_store16(v, (pok[aa][b] && cok && !W8_ABL_OOBST) ? dlane + pix[aa][b] * dps : WN_OOB, yr);
                    }
                }
            }
        }
        if (!has_next) break;
        cur = nxt;
        ++k;
        wn = strip_index(k + 1);
        locate(wn, nxt);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the out-of-range DMAs issued behind the last strip
}

template <int ACT, int OUT, int NCH>
int w8_launch(const WinoK& k, int grid, hipStream_t st)
{
    esr_note_kernel("wino8_f32_kernel<%d, %d, %d>", ACT, OUT, NCH);
    hipLaunchKernelGGL((wino8_f32_kernel<ACT, OUT, NCH>), dim3(grid), dim3(W8_THREADS), 0, st, k);
    esr_graph_note_io(st, k.x, offsetof(WinoK, x), k.y0, offsetof(WinoK, y0));
    return esr_check_launch("wino8_f32_kernel launch");
}

template <int ACT, int RES, int OUT>
int wn_launch(const WinoK& k, int grid, hipStream_t st)
{
    esr_note_kernel("wino_f32_kernel<%d, %d, %d>", ACT, RES, OUT);
    hipLaunchKernelGGL((wino_f32_kernel<ACT, RES, OUT>), dim3(grid), dim3(WN_THREADS), 0, st, k);
    esr_graph_note_io(st, k.x, offsetof(WinoK, x), k.y0, offsetof(WinoK, y0));
    return esr_check_launch("wino_f32_kernel launch");
}

bool g_wino8_enabled = true;          // research switch (esr_dbg_wino8): A/B of the two Winograd kernels inside one process

// G of F(2x2, 3x3)
const double WN_G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};

inline size_t wn_index(int nhalves, int slot, int pos, int o)
{
    const int chunk = slot / 8, within = slot % 8;
    const int kq = within / 2, s = within % 2;
    const int half = o / 32, ctl = (o % 32) / 16, i = o % 16;
    return ((((size_t)chunk * nhalves + half) * 16 + pos) * 64 + kq * 16 + i) * 4 + ctl * 2 + s;
}

}  // namespace

// Shapes the Winograd kernel takes (everything else stays on conv_f32_kernel): see esr_conv_desc.wino_wpacked.
int esr_wino_supported(const esr_conv_desc* d)
{
    if (!d || d->ksize != 3 || d->in_layout != ESR_NHWC) return 0;
    if (d->out_layout == ESR_NCHW_SHUFFLE4) {                      // the network's last convolution: no residual, no split
        if ((d->cout & 15) || d->res_mode != ESR_RES_NONE || (d->split > 0 && d->split < d->cout) || (d->blocked8 & ESR_BLOCKED_OUT1)) return 0;
    } else if (d->out_layout != ESR_NHWC) return 0;
    if (d->storage != ESR_STORE_F32 || d->compute != ESR_COMPUTE_F32) return 0;
    if (d->tail_wpacked || d->post_wpacked || d->border_bias || d->in_seg_stride) return 0;
    if ((d->blocked8 & ESR_BLOCKED_OUT1) && d->res_mode != ESR_RES_NONE) return 0;
    if (d->cin <= 0 || d->cout <= 0 || d->cout > 64) return 0;
    const int cin_phys = esr_round_up(d->cin, 8);
    const int nchunks = cin_phys / 8;
    if (nchunks < 4 || (nchunks & 1)) return 0;                  // raw ring looks 3 chunks ahead; stages are unrolled in pairs
    // a channel-blocked INPUT [n][C/8][h][w][8] (both Winograd kernels read it: a chunk is a plane): whole planes only
    if ((d->blocked8 & ESR_BLOCKED_IN) && ((d->in.coff & 7) || (d->in.pitch & 7))) return 0;
    if (d->blocked8 & ~(ESR_BLOCKED_IN | ESR_BLOCKED_OUT1)) return 0;   // blocked out0 / res: the fused IMDB tail only
    if (d->hilo) return 0;
    const int cout4 = esr_round_up(d->cout, 4);
    int split = d->split <= 0 ? cout4 : d->split;
    if (split >= d->cout) split = cout4;
    if (split < cout4 && (split & 15)) return 0;                 // a cout tile goes to ONE destination
    // tile decode by multiplication (wn_div): tiles * max(tiles_x, tiles_y) < 2^32; per-image raw buffer < 2 GiB - 1 KB (OOB offsets)
    const double tx = (d->w + 15) / 16, ty = (d->h + 15) / 16;
    if ((double)d->n * tx * ty * (tx > ty ? tx : ty) >= 4294967296.0) return 0;
    if ((double)d->h * d->w * d->in.pitch * 4.0 >= 2147482624.0) return 0;
    return 1;
}

extern "C" {

size_t esr_packed_wino_bytes(int cin_phys, int cout)
{
    if (cin_phys <= 0 || cout <= 0) return 0;
    const size_t nchunks = (size_t)esr_round_up(cin_phys, 8) / 8;
    const size_t nhalves = (size_t)esr_round_up(cout, 32) / 32;
    return (nchunks * nhalves * 16 * 256 + nhalves * 32) * sizeof(float);
}

int esr_pack_wino_f32(const float* w, const float* bias, int cin, int cout, const int32_t* cin_map, int cin_phys, void* out, size_t out_bytes)
{
    if (!w || !out || cin <= 0 || cout <= 0) return ESR_ERR_BAD_ARG;
    if (!cin_map && cin_phys < cin) return ESR_ERR_BAD_ARG;
    const size_t need = esr_packed_wino_bytes(cin_phys, cout);
    if (need == 0 || out_bytes < need) return ESR_ERR_BAD_ARG;
    const int nhalves = esr_round_up(cout, 32) / 32;
    float* o = static_cast<float*>(out);
    memset(o, 0, need);
    for (int s = 0; s < cin_phys; ++s) {
        const int c = cin_map ? cin_map[s] : (s < cin ? s : -1);
        if (c < 0) continue;
        if (c >= cin) return ESR_ERR_BAD_ARG;
        for (int oc = 0; oc < cout; ++oc) {
            const float* g = w + ((size_t)oc * cin + c) * 9;
            double t[4][3];                                        // G g
            for (int i = 0; i < 4; ++i)
                for (int b = 0; b < 3; ++b) t[i][b] = WN_G[i][0] * g[b] + WN_G[i][1] * g[3 + b] + WN_G[i][2] * g[6 + b];
            for (int i = 0; i < 4; ++i)
                for (int jj = 0; jj < 4; ++jj) {
                    const double u = t[i][0] * WN_G[jj][0] + t[i][1] * WN_G[jj][1] + t[i][2] * WN_G[jj][2];
                    o[wn_index(nhalves, s, 4 * i + jj, oc)] = (float)u;          // ONE rounding of the fp64 value
                }
        }
    }
    float* bo = o + (size_t)(esr_round_up(cin_phys, 8) / 8) * nhalves * 16 * 256;
    if (bias)
        for (int oc = 0; oc < cout; ++oc) bo[oc] = bias[oc];
    return ESR_OK;
}

int esr_unpack_wino_f32(const void* packed, size_t bytes, int cin, int cout, const int32_t* cin_map, int cin_phys, float* u, float* bias)
{
    if (!packed || !u || cin <= 0 || cout <= 0) return ESR_ERR_BAD_ARG;
    if (bytes < esr_packed_wino_bytes(cin_phys, cout)) return ESR_ERR_BAD_ARG;
    const int nhalves = esr_round_up(cout, 32) / 32;
    const float* o = static_cast<const float*>(packed);
    memset(u, 0, sizeof(float) * (size_t)cout * cin * 16);
    for (int s = 0; s < cin_phys; ++s) {
        const int c = cin_map ? cin_map[s] : (s < cin ? s : -1);
        if (c < 0) continue;
        for (int oc = 0; oc < cout; ++oc)
            for (int pos = 0; pos < 16; ++pos) u[((size_t)oc * cin + c) * 16 + pos] = o[wn_index(nhalves, s, pos, oc)];
    }
    if (bias) {
        const float* bo = o + (size_t)(esr_round_up(cin_phys, 8) / 8) * nhalves * 16 * 256;
        for (int oc = 0; oc < cout; ++oc) bias[oc] = bo[oc];
    }
    return ESR_OK;
}

/* research switch, not part of the ABI header: 0 = every Winograd launch on wino_f32_kernel */
void esr_dbg_wino8(int on) { g_wino8_enabled = on != 0; }

}  // extern "C"

// Called by esr_conv2d_f32 after its argument checks when d->wino_wpacked is set and esr_wino_supported(d).
int esr_conv2d_wino(const esr_conv_desc* d, void* hip_stream)
{
    const int cin_phys = esr_round_up(d->cin, 8);
    const int cout4 = esr_round_up(d->cout, 4);
    int split = d->split <= 0 ? cout4 : d->split;
    if (split >= d->cout) split = cout4;
    WinoK k;
    k.x = static_cast<const float*>(d->in.ptr);
    k.up = static_cast<const float*>(d->wino_wpacked);
    k.nchunks = cin_phys / 8;
    k.nhalves = esr_round_up(d->cout, 32) / 32;
    k.up_bytes = (unsigned)esr_packed_wino_bytes(cin_phys, d->cout);
    k.bias = k.up + (size_t)k.nchunks * k.nhalves * 16 * 256;
    k.res = static_cast<const float*>(d->res.ptr);
    k.y0 = static_cast<float*>(d->out0.ptr);
    k.y1 = static_cast<float*>(d->out1.ptr);
    k.N = d->n; k.H = d->h; k.W = d->w;
    k.in_pitch = d->in.pitch; k.in_coff = d->in.coff;
    k.res_pitch = d->res.pitch; k.res_coff = d->res.coff;
    k.y0_pitch = d->out0.pitch; k.y0_coff = d->out0.coff;
    k.y1_pitch = d->out1.pitch; k.y1_coff = d->out1.coff;
    k.cout_store = cout4;
    k.split = split;
    k.act = d->act; k.slope = d->slope; k.res_mode = d->res_mode;
    k.tiles_x = (d->w + WN_TILE - 1) / WN_TILE;
    k.tiles_y = (d->h + WN_TILE - 1) / WN_TILE;
    k.y1_blk = (d->blocked8 & ESR_BLOCKED_OUT1) ? 1 : 0;
    k.magic_x = k.tiles_x == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)k.tiles_x) + 1u;
    k.magic_y = k.tiles_y == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)k.tiles_y) + 1u;
    const long nwork = (long)k.N * k.tiles_x * k.tiles_y * k.nhalves;
    const int grid = nwork < WN_MAX_BLOCKS ? (int)nwork : WN_MAX_BLOCKS;
    // uneven split between the first- and the second-dispatched half of the grid (see the kernel's work walk): only when both
    // blocks of every CU are there and walk several items
    k.first_share = (grid == WN_MAX_BLOCKS && nwork >= 8L * grid) ? WN_FIRST_SHARE : 8;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    // instantiations: the activation is a template constant for LeakyReLU / none (the networks' cases), read at run time otherwise
    const int a = d->act;
    // wino8_f32_kernel (resident U, wave-private halo rings, no barrier): 4 or 6 input chunks, no residual, NHWC / blocked outputs, and enough
    // strips (4 rows x 16 columns) that each of the 2048 waves walks several; both per-image byte ranges behind 31-bit buffer offsets
    const bool blocked_in = (d->blocked8 & ESR_BLOCKED_IN) != 0;
    k.pix_floats = blocked_in ? 8u : (unsigned)d->in.pitch;
    k.in_base = blocked_in ? (unsigned)(d->in.coff / 8) * (unsigned)(d->h * d->w * 32) : (unsigned)d->in.coff * 4u;
    k.chunk_stride = blocked_in ? (unsigned)(d->h * d->w * 32) : 32u;
    if ((k.nchunks == 4 || k.nchunks == 6) && d->res_mode == ESR_RES_NONE && d->out_layout == ESR_NHWC && g_wino8_enabled) {
        const long sx = (d->w + 15) / 16, sy = (d->h + 3) / 4;
        const long nstrips = (long)k.N * sx * sy;
        const double out_bytes = (double)d->h * d->w * 4.0 * (d->out0.pitch > d->out1.pitch ? d->out0.pitch : d->out1.pitch);
        const bool fits = (double)k.N * sx * sy * (sx > sy ? sx : sy) < 4294967296.0 && out_bytes < 2147482624.0;
        if (fits && nstrips >= 4L * 8 * W8_MAX_BLOCKS / k.nhalves) {
            WinoK w = k;
            w.tiles_x = (int)sx; w.tiles_y = (int)sy;
            w.magic_x = sx == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)sx) + 1u;
            w.magic_y = sy == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)sy) + 1u;
            const int grid8 = W8_MAX_BLOCKS;
            const bool blk = k.y1_blk != 0;
            const bool lr = a == ESR_ACT_LRELU;
            if (k.nchunks == 6) {
                if (blk) return lr ? w8_launch<ESR_ACT_LRELU, 1, 6>(w, grid8, st) : w8_launch<-1, 1, 6>(w, grid8, st);
                return lr ? w8_launch<ESR_ACT_LRELU, 0, 6>(w, grid8, st) : w8_launch<-1, 0, 6>(w, grid8, st);
            }
            if (blk) return lr ? w8_launch<ESR_ACT_LRELU, 1, 4>(w, grid8, st) : w8_launch<-1, 1, 4>(w, grid8, st);
            return lr ? w8_launch<ESR_ACT_LRELU, 0, 4>(w, grid8, st) : w8_launch<-1, 0, 4>(w, grid8, st);
        }
    }
    if (d->out_layout == ESR_NCHW_SHUFFLE4)
        return a == ESR_ACT_NONE ? wn_launch<ESR_ACT_NONE, ESR_RES_NONE, 2>(k, grid, st) : wn_launch<-1, ESR_RES_NONE, 2>(k, grid, st);
    if (k.y1_blk)
        return a == ESR_ACT_LRELU ? wn_launch<ESR_ACT_LRELU, ESR_RES_NONE, 1>(k, grid, st) : wn_launch<-1, ESR_RES_NONE, 1>(k, grid, st);
    if (d->res_mode == ESR_RES_NONE)
        return a == ESR_ACT_LRELU ? wn_launch<ESR_ACT_LRELU, ESR_RES_NONE, 0>(k, grid, st)
             : a == ESR_ACT_NONE  ? wn_launch<ESR_ACT_NONE, ESR_RES_NONE, 0>(k, grid, st) : wn_launch<-1, ESR_RES_NONE, 0>(k, grid, st);
    if (d->res_mode == ESR_RES_PRE_ACT)
        return a == ESR_ACT_LRELU ? wn_launch<ESR_ACT_LRELU, ESR_RES_PRE_ACT, 0>(k, grid, st)
             : a == ESR_ACT_NONE  ? wn_launch<ESR_ACT_NONE, ESR_RES_PRE_ACT, 0>(k, grid, st) : wn_launch<-1, ESR_RES_PRE_ACT, 0>(k, grid, st);
    return a == ESR_ACT_LRELU ? wn_launch<ESR_ACT_LRELU, ESR_RES_POST_ACT, 0>(k, grid, st) : wn_launch<-1, ESR_RES_POST_ACT, 0>(k, grid, st);
}
