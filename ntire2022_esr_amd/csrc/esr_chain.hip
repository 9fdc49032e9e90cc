// esr_chain.hip -- a residual block's 3x3 CHAIN as one launch (round 5; esr_chain_desc, ABI v11; 16-bit storage).
//
// RLFB (team04_rlfn.py:109-122): t1 = lrelu(c1_r(x)), t2 = lrelu(c2_r(t1)), u = lrelu(c3_r(t2)) + x, v = c5(u), c1 = esa.conv1(v).
// As separate launches (conv48r_kernel x 2 + conv48rp_kernel) one 339 x 510 image spends 49 us in them, each launch at 0.27-0.33 of the HBM
// rate with 3-5 us of prologue, and a batch moves t1 / t2 through HBM twice.  Here the chain is a LAYER-PER-SIMD PIPELINE inside one block:
//
//   wave 0 = c1_r    wave 1 = c2_r    wave 2 = c3_r (+ x)    wave 3 = c5, esa.conv1, the stores and the input DMA
//
// Every wave keeps ITS layer's weights in registers for the life of the block (3x3: 45 A fragments = 180 accumulation registers, as
// conv48r_kernel; c5 / conv1: 24 hi + lo fragments), so nothing is re-staged per tile and per layer.  A block walks JOBS: a column strip of
// 16 G - 4 = 28 output pixels x RS output rows, one image ROW per step.  In step S wave l computes row S - 3 l of its layer (G = 2 MFMA
// pixel groups x 45 MFMAs) from three rows of its producer's ring in LDS and writes its own row into the next ring -- bf16 / fp16 pixels,
// exactly the values the separate launches would have stored (t1, t2) or kept in fp32 (u, as hi + lo B operands) -- with the finished
// row's epilogue running as micro-steps behind the next row's MFMAs (LAB_NOTES 9.5).  One s_barrier per step is the only synchronisation;
// a layer lags its producer by three steps (one row of halo, one step of MFMAs, one step of deferred epilogue).  Halo: a layer computes
// all 32 columns, each one pixel further right than its producer's, so the strip's 28 + 6 input columns shrink to 28 valid output columns
// (columns / rows outside the IMAGE are written as zeros: every layer's own zero padding); rows: RS + 6 slots per job.  Jobs of a block
// follow each other without draining the pipeline.  Rings: input 16 rows x 4 KB (DMA, 8 rows ahead), t1 / t2 4 rows, u 2 rows: 102 KB.
//
// Same packed weights, fragment maps, per-accumulator operation order and roundings as conv48r_kernel / conv48rp_kernel: the results are
// bit-identical to the three separate launches (tests/test_gpu_chain.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <atomic>

#include "esr_s16_dev.h"

namespace {

struct ChainK {
    const char* x;            // NHWC 16-bit block input: 48 physical channels at in_coff of in_pitch
    const char* w0;           // esr_pack_conv_s16 blobs of the three 3x3 layers (45 KB image + 48 fp32 biases each)
    const char* w1;
    const char* w2;
    const char* pw1;          // esr_pack_post_s16 blob of post 1 (c5: 48 -> 48): hi images, lo images, fp32 bias
    const char* pw2;          // ... of post 2 (esa.conv1: 48 -> 16)
    char* y1;                 // v
    char* y2;                 // c1
    int N, H, W;
    int in_pitch, in_coff;
    int y1_pitch, y1_coff, y2_pitch, y2_coff;
    int p1_cout8, p2_cout8;   // channels stored
    float slope;              // activation of the 3x3 layers as max(v, slope v)
    float p1_slope;
    int SX, SY;               // strips per image row, row segments per image
    int WS, RS;               // output columns per strip (<= 28), output rows per segment
    int njobs;                // N * SY * SX
};

constexpr int CH_G = 2;                       // MFMA pixel groups per row
constexpr int CH_COLS = 16 * CH_G + 2;        // pixels of a ring row
constexpr int CH_WS = 16 * CH_G - 4;          // valid output columns of a strip
constexpr int CH_IN_PITCH = 4096;             // input ring row: 34 pixels x 96 B = 3264 B, staged as four 1 KB DMA pieces
constexpr int CH_NR_IN = 16;
constexpr int CH_T_PITCH = 3328;              // t1 / t2 ring row (>= 34 x 96)
constexpr int CH_NR_T = 4;
constexpr int CH_U_ROW = 3 * CH_G * 1024;     // u ring row: [tile][group] 1 KB B-operand fragments
constexpr int CH_OFF_IN = 0;
constexpr int CH_OFF_T1 = CH_NR_IN * CH_IN_PITCH;
constexpr int CH_OFF_T2 = CH_OFF_T1 + CH_NR_T * CH_T_PITCH;
constexpr int CH_OFF_U = CH_OFF_T2 + CH_NR_T * CH_T_PITCH;
constexpr int CH_LDS = CH_OFF_U + 2 * CH_U_ROW;
constexpr int CH_D = 8;                       // the input DMA runs this many rows ahead of layer 1
constexpr int CH_LAGP = 8;                    // wave 3 (post chain) lags layer 1 by this many steps; layer l by 3 l
constexpr int CH_HALO = 3;                    // row slot r of a job is image row Y0 - 3 + r; RS + 6 slots per job
constexpr int CH_WIMG = 45 * 1024;            // 3 chunks x 5 tap pairs x 3 tiles

// a wave's cursor over (job, row slot); every member is wave-uniform
struct ChainCur {
    int V;        // virtual row index: ring slots are V & (rows - 1)
    int r;        // row slot of the job
    int k;        // job ordinal of this block (-1: before the first)
    int ok;       // the job exists
    int n, X0, Y0;
};

__device__ __forceinline__ int chain_job_index(const ChainK& p, int k)
{
    const int G = gridDim.x;
    const int base = k * G;
    if (k < 0 || base >= p.njobs) return -1;
    int off = blockIdx.x;
    if ((G & 7) == 0 && base + G <= p.njobs) off = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);      // neighbouring strips on one XCD
    const int t = base + off;
    return t < p.njobs ? t : -1;
}

__device__ __forceinline__ void chain_decode(const ChainK& p, ChainCur& c)
{
    const int j = chain_job_index(p, c.k);
    c.ok = j >= 0;
    if (j < 0) { c.n = 0; c.X0 = 0; c.Y0 = 0; return; }
    const int per = p.SX * p.SY;
    c.n = j / per;
    const int rem = j - c.n * per;
    const int sy = rem / p.SX, sx = rem - sy * p.SX;
    c.X0 = sx * p.WS;
    c.Y0 = sy * p.RS;
}

// one step forward; true when a new job starts
__device__ __forceinline__ bool chain_advance(const ChainK& p, ChainCur& c, int RJ)
{
    ++c.V;
    if (++c.r < RJ) return false;
    c.r = 0;
    ++c.k;
    chain_decode(p, c);
    return true;
}

__device__ __forceinline__ ChainCur chain_cursor(int lag, int RJ)
{
    // after lag + 1 advances the cursor stands at (job 0, slot 0), V = 0
    ChainCur c;
    c.V = -1 - lag; c.r = RJ - 1 - lag; c.k = -1; c.ok = 0; c.n = 0; c.X0 = 0; c.Y0 = 0;
    return c;
}

__device__ __forceinline__ void chain_barrier()
{
    // every LDS access of the step has completed (writes visible, ring slots released); the "memory" clobber keeps hipcc from moving
    // LDS accesses across it
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- waves 0 .. 2: one 3x3 layer each -------------------------------------------------------------------------------------------------
// LAST = false: act(conv) rounded into the next layer's ring (zeros outside the image).  LAST = true (c3_r): act(conv) + x (the block input
// from the input ring), kept in fp32 and handed to wave 3 as the post chain's B operands (16-bit high parts | low parts).
template <bool BF16, bool LAST>
__device__ __forceinline__ void chain_layer_wave(const ChainK& p, char* const smem, const int layer, const int nsteps, const int RJ)
{
    constexpr int G = CH_G, NCH = 3, PAIRS = 5, NT = 3, NG = NCH * PAIRS;
    const int lane = threadIdx.x & 63, px = lane & 15, kq = lane >> 4;
    const int src_base = layer == 0 ? CH_OFF_IN : (layer == 1 ? CH_OFF_T1 : CH_OFF_T2);
    const int src_pitch = layer == 0 ? CH_IN_PITCH : CH_T_PITCH;
    const int src_mask = layer == 0 ? CH_NR_IN - 1 : CH_NR_T - 1;
    const int dst_base = layer == 0 ? CH_OFF_T1 : CH_OFF_T2;
    const char* const wp = layer == 0 ? p.w0 : (layer == 1 ? p.w1 : p.w2);

    // ---- the layer's weights: registers for the life of the block ----------------------------------------------------------------------
    i32x4 wr[NCH][PAIRS][NT];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int q = 0; q < PAIRS; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                wr[c][q][t] = *reinterpret_cast<const i32x4*>(wp + (size_t)(((c * PAIRS + q) * NT + t) * 1024 + lane * 16));
    f32x4 bia[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bia[t] = *reinterpret_cast<const f32x4*>(wp + CH_WIMG + (t * 16 + kq * 4) * 4);
    const float slope = p.slope;

    // lane-constant parts of the B fragment addresses: pair q reads tap min(2 q + (kq >> 1), 8), channel half kq & 1 of the chunk; only pair 1
    // (taps 2 | 3) straddles two rows
    int laneoff[PAIRS];
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
        const int tap = min(2 * q + (kq >> 1), 8);
        laneoff[q] = (px + tap % 3) * 96 + (kq & 1) * 16;
    }
    const bool hi_tap = (kq >> 1) != 0;
    const int lane_w = px * 96 + kq * 8;                 // this lane's 4 channels of tile 0, group 0 in a ring row
    const int lane_r = (px + CH_HALO) * 96 + kq * 8;     // ... of the block input (residual): three columns to the right in the input ring

    auto rowaddr = [&](int v) __attribute__((always_inline)) -> int { return src_base + (v & src_mask) * src_pitch; };
    auto set_baddr = [&](int (&ba)[PAIRS], int V) __attribute__((always_inline)) {       // B addresses of the step that computes row V
        const int ra0 = __builtin_amdgcn_readfirstlane(rowaddr(V - 1)), ra1 = __builtin_amdgcn_readfirstlane(rowaddr(V)),
                  ra2 = __builtin_amdgcn_readfirstlane(rowaddr(V + 1));
        ba[0] = ra0 + laneoff[0];
        ba[1] = (hi_tap ? ra1 : ra0) + laneoff[1];
        ba[2] = ra1 + laneoff[2];
        ba[3] = ra2 + laneoff[3];
        ba[4] = ra2 + laneoff[4];
    };

    ChainCur cur = chain_cursor(3 * layer, RJ);
    unsigned cm0 = 0u, cm1 = 0u;                // this job's column masks (the layer's output column 16 e + px lies inside the image)
    int baddr[2][PAIRS];
    set_baddr(baddr[0], cur.V + 1);

    // epilogue parameters of a row, stashed by the step that computes it for the step that finishes it
    struct Epi { int wa; unsigned m0, m1; int ra; };
    Epi ep[2] = {{dst_base + lane_w, 0u, 0u, CH_OFF_IN + lane_r}, {dst_base + lane_w, 0u, 0u, CH_OFF_IN + lane_r}};
    if (LAST) { ep[0].wa = CH_OFF_U + lane * 16; ep[1].wa = CH_OFF_U + lane * 16; }

    f32x4 acc[2][NT][G];
    i32x4 b[5][G];                              // B fragments: ring of five, read three groups ahead
    i32x4 pb[3][G];                             // the next step's first three groups (rows that are already complete), read mid-step
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the weights are in their registers
    chain_barrier();                            // wave 3 has staged the first CH_D input rows
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int e = 0; e < G; ++e) pb[g][e] = *reinterpret_cast<const i32x4*>(smem + baddr[0][g] + e * (16 * 96));
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < G; ++e) acc[par][t][e] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 ev = {0.f, 0.f, 0.f, 0.f};
    uint2 pk = {0u, 0u};
    uint2 rraw[2] = {{0u, 0u}, {0u, 0u}};
    i32x4 ob = {0, 0, 0, 0};

    // the finished row's epilogue, micro-step s (behind MFMA s of the next row); pp = the finished row's accumulators
    auto micro = [&](auto pp_, auto s_) __attribute__((always_inline)) {
        constexpr int pp = decltype(pp_)::value, s = decltype(s_)::value;
        const Epi& E = ep[pp];
        if constexpr (!LAST) {
            // fragment f = 2 t + e in slots 6 + 4 f ..: activation halves, rounding + mask, store
            if constexpr (s >= 6 && s < 6 + 4 * NT * G) {
                constexpr int f = (s - 6) >> 2, m = (s - 6) & 3, t = f >> 1, e = f & 1;
                if constexpr (m == 0) {
                    ev = acc[pp][t][e];
                    ev.x = act1(ev.x, slope); ev.y = act1(ev.y, slope);
                } else if constexpr (m == 1) {
                    ev.z = act1(ev.z, slope); ev.w = act1(ev.w, slope);
                } else if constexpr (m == 2) {
                    const unsigned msk = e ? E.m1 : E.m0;
                    pk.x = pack2<BF16>(ev.x, ev.y) & msk;
                    pk.y = pack2<BF16>(ev.z, ev.w) & msk;
                } else {
                    *reinterpret_cast<uint2*>(smem + E.wa + e * (16 * 96) + t * 32) = pk;
                }
            }
        } else {
            // fragment f: residual read in slot 6 f, act + residual in 6 f + 6 / + 7, high parts 6 f + 8, low parts 6 f + 9, store 6 f + 10
            static_for<NT * G>([&](auto f_) __attribute__((always_inline)) {
                constexpr int f = decltype(f_)::value, t = f >> 1, e = f & 1;
                if constexpr (s == 6 * f) rraw[f & 1] = *reinterpret_cast<const uint2*>(smem + E.ra + e * (16 * 96) + t * 32);
                if constexpr (s == 6 * f + 6 || s == 6 * f + 7) {
                    constexpr int h = s - (6 * f + 6);
                    float ra_, rb_;
                    unpack2<BF16>(h ? rraw[f & 1].y : rraw[f & 1].x, ra_, rb_);
                    float va = h ? acc[pp][t][e].z : acc[pp][t][e].x, vb = h ? acc[pp][t][e].w : acc[pp][t][e].y;
                    va = act1(va, slope) + ra_; vb = act1(vb, slope) + rb_;              // lrelu(conv) + x (team04_rlfn.py:117-119)
                    if (h) { acc[pp][t][e].z = va; acc[pp][t][e].w = vb; } else { acc[pp][t][e].x = va; acc[pp][t][e].y = vb; }
                }
                if constexpr (s == 6 * f + 8) {
                    const f32x4 v = acc[pp][t][e];
                    ob.x = (int)pack2<BF16>(v.x, v.y); ob.y = (int)pack2<BF16>(v.z, v.w);
                    if (!BF16) { ob.z = 0; ob.w = 0; }
                }
                if constexpr (s == 6 * f + 9 && BF16) {
                    const f32x4 v = acc[pp][t][e];
                    float a, bq, c, d;
                    unpack2<BF16>((unsigned)ob.x, a, bq);
                    unpack2<BF16>((unsigned)ob.y, c, d);
                    ob.z = (int)pack2<BF16>(v.x - a, v.y - bq); ob.w = (int)pack2<BF16>(v.z - c, v.w - d);
                }
                if constexpr (s == 6 * f + 10) *reinterpret_cast<i32x4*>(smem + E.wa + (t * G + e) * 1024) = ob;
            });
        }
    };

    auto run_step = [&](auto par_) __attribute__((always_inline)) {
        constexpr int par = decltype(par_)::value;
        if (chain_advance(p, cur, RJ) && !LAST) {
            const int gx = cur.X0 - 2 + layer + px;
            cm0 = (cur.ok && (unsigned)gx < (unsigned)p.W) ? 0xffffffffu : 0u;
            cm1 = (cur.ok && (unsigned)(gx + 16) < (unsigned)p.W) ? 0xffffffffu : 0u;
        }
        {
            // this row's epilogue parameters (used one step later)
            if (!LAST) {
                const bool rowok = (unsigned)(cur.Y0 - CH_HALO + cur.r) < (unsigned)p.H;
                ep[par].wa = dst_base + (cur.V & (CH_NR_T - 1)) * CH_T_PITCH + lane_w;
                ep[par].m0 = rowok ? cm0 : 0u;
                ep[par].m1 = rowok ? cm1 : 0u;
            } else {
                ep[par].wa = CH_OFF_U + (cur.V & 1) * CH_U_ROW + lane * 16;
                ep[par].ra = CH_OFF_IN + (cur.V & (CH_NR_IN - 1)) * CH_IN_PITCH + lane_r;
            }
        }
        const int* const ba = baddr[par];
        auto read_b = [&](int g) __attribute__((always_inline)) {
            const int c_ = g / PAIRS, q_ = g % PAIRS;
#pragma unroll
            for (int e = 0; e < G; ++e) b[g % 5][e] = *reinterpret_cast<const i32x4*>(smem + ba[q_] + c_ * 32 + e * (16 * 96));
        };
        static_for<NG>([&](auto g_) __attribute__((always_inline)) {
            constexpr int g = decltype(g_)::value;
            constexpr int c = g / PAIRS, q = g % PAIRS;
            static_for<NT * G>([&](auto m_) __attribute__((always_inline)) {
                constexpr int mi = decltype(m_)::value, t = mi >> 1, e = mi & 1;
                // groups 0 .. 2 take the fragments read during the previous step; they are consumed before group 8 overwrites them
                const i32x4 bf = g < 3 ? pb[g < 3 ? g : 0][e] : b[g % 5][e];
                if (g == 0) {
                    // (early clobber: the first write of an accumulator must not land on the registers of a fragment that dies here)
                    if (BF16) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(bf), "v"(bia[t]));
                    else asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(bf), "v"(bia[t]));
                } else {
                    if (BF16) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(bf));
                    else asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(bf));
                }
                // LDS reads are issued BEHIND the group's first MFMA: hipcc closes the step's barrier (an asm it cannot see into) with
                // lgkmcnt(0) in front of the first use of a fragment read before it -- nothing is outstanding there yet
                if constexpr (mi == 0 && g + 3 < NG) read_b(g + 3);
                if constexpr (mi == 0 && g == 7) set_baddr(baddr[par ^ 1], cur.V + 1);
                if constexpr (mi == 0 && g >= 8 && g < 11) {
                    // the next step's groups 0 .. 2 (taps 0 .. 5: the two older rows of its window, complete since the last barrier)
#pragma unroll
                    for (int e2 = 0; e2 < G; ++e2) pb[g - 8][e2] = *reinterpret_cast<const i32x4*>(smem + baddr[par ^ 1][g - 8] + e2 * (16 * 96));
                }
                micro(std::integral_constant<int, par ^ 1>{}, std::integral_constant<int, 6 * g + 2 * t + e>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        chain_barrier();
    };
    for (int it = 0; it < nsteps; it += 2) {
        run_step(std::integral_constant<int, 0>{});
        run_step(std::integral_constant<int, 1>{});
    }
}

// ---- wave 3: the input DMA, c5 and esa.conv1 on the fp32 rows of u, the stores ---------------------------------------------------------------
template <bool BF16>
__device__ __forceinline__ void chain_post_wave(const ChainK& p, char* const smem, const int nsteps, const int RJ)
{
    constexpr int G = CH_G, NT = 3, PNT1 = 3;
    constexpr int P1_IMG = NT * PNT1 * 1024, P2_IMG = PNT1 * 1024;
    constexpr bool plo = BF16;
    const int lane = threadIdx.x & 63, px = lane & 15, kq = lane >> 4;
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- input DMA: row V of the input ring = 34 pixels x 6 sixteen-byte parts, four pieces of 64 lanes ---------------------------------
    const size_t img_bytes = (size_t)p.H * p.W * p.in_pitch * 2;
    const unsigned rowb_in = (unsigned)p.W * (unsigned)p.in_pitch * 2u;
    ChainCur dc = chain_cursor(0, RJ);          // runs CH_D rows ahead of layer 1 after the prologue
    unsigned dbase[4] = {OOB, OOB, OOB, OOB};
    auto dma_lanes = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned sl = (unsigned)(i * 64 + lane);
            const unsigned pixel = sl / 6u, part = sl - pixel * 6u;
            const int gx = dc.X0 - CH_HALO + (int)pixel;
            const bool ok = dc.ok && pixel < (unsigned)CH_COLS && (unsigned)gx < (unsigned)p.W;
            dbase[i] = ok ? (unsigned)(gx * p.in_pitch + p.in_coff) * 2u + part * 16u : OOB;
        }
    };
    auto dma_row = [&]() __attribute__((always_inline)) {
        const int gy = dc.Y0 - CH_HALO + dc.r;
        const bool rowok = dc.ok && (unsigned)gy < (unsigned)p.H;
        const i32x4 rs = make_rsrc(p.x + (size_t)dc.n * img_bytes, img_bytes);
        const unsigned dst = smem_lds + (unsigned)(CH_OFF_IN + (dc.V & (CH_NR_IN - 1)) * CH_IN_PITCH);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_buf16(dst + (unsigned)i * 1024u, rowok ? dbase[i] + (unsigned)gy * rowb_in : OOB, rs, 0u);
    };
    for (int i = 0; i < CH_D; ++i) {
        if (chain_advance(p, dc, RJ)) dma_lanes();
        dma_row();
    }

    // ---- post weights: registers ---------------------------------------------------------------------------------------------------------
    i32x4 a1h[PNT1][PNT1], a1l[PNT1][PNT1], a2h[PNT1], a2l[PNT1];       // [k tile][out tile]
#pragma unroll
    for (int kt = 0; kt < PNT1; ++kt) {
#pragma unroll
        for (int ot = 0; ot < PNT1; ++ot) {
            a1h[kt][ot] = *reinterpret_cast<const i32x4*>(p.pw1 + (size_t)((kt * PNT1 + ot) * 1024 + lane * 16));
            if (plo) a1l[kt][ot] = *reinterpret_cast<const i32x4*>(p.pw1 + (size_t)(P1_IMG + (kt * PNT1 + ot) * 1024 + lane * 16));
        }
        a2h[kt] = *reinterpret_cast<const i32x4*>(p.pw2 + (size_t)(kt * 1024 + lane * 16));
        if (plo) a2l[kt] = *reinterpret_cast<const i32x4*>(p.pw2 + (size_t)(P2_IMG + kt * 1024 + lane * 16));
    }
    f32x4 pb1[PNT1], pb2;
#pragma unroll
    for (int t = 0; t < PNT1; ++t) pb1[t] = *reinterpret_cast<const f32x4*>(p.pw1 + (size_t)2 * P1_IMG + (t * 16 + kq * 4) * 4);
    pb2 = *reinterpret_cast<const f32x4*>(p.pw2 + (size_t)2 * P2_IMG + (kq * 4) * 4);
    const float p1s = p.p1_slope;

    // ---- stores: per job the lane's byte offsets in row 0 of the image (OOB: column outside the strip / image, channel not stored) --------
    const size_t y1_img = (size_t)p.H * p.W * p.y1_pitch * 2, y2_img = (size_t)p.H * p.W * p.y2_pitch * 2;
    const unsigned rowb1 = (unsigned)p.W * (unsigned)p.y1_pitch * 2u, rowb2 = (unsigned)p.W * (unsigned)p.y2_pitch * 2u;
    ChainCur cur = chain_cursor(CH_LAGP, RJ);
    unsigned sA0 = OOB, sA1 = OOB, sB = OOB, sC = OOB;
    auto store_lanes = [&]() __attribute__((always_inline)) {
        const int wsj = min(p.WS, p.W - cur.X0);                       // valid output columns of this strip
        const int chA = (kq & 1) * 16 + (kq >> 1) * 8, chB = 32 + (kq >> 1) * 8, ch2 = (kq >> 1) * 8;
        const int colB = (kq & 1) * 16 + px;
        auto off1 = [&](int col, int ch) -> unsigned {
            return (cur.ok && col < wsj && ch < p.p1_cout8) ? (unsigned)((cur.X0 + col) * p.y1_pitch + p.y1_coff + ch) * 2u : OOB;
        };
        sA0 = off1(px, chA);
        sA1 = off1(16 + px, chA);
        sB = off1(colB, chB);
        sC = (cur.ok && colB < wsj && ch2 < p.p2_cout8) ? (unsigned)((cur.X0 + colB) * p.y2_pitch + p.y2_coff + ch2) * 2u : OOB;
    };
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto swap16 = [&](uint2 X, uint2 Y) __attribute__((always_inline)) -> i32x4 {
        const u32x2 a = __builtin_amdgcn_permlane16_swap(X.x, Y.x, false, false);
        const u32x2 bq = __builtin_amdgcn_permlane16_swap(X.y, Y.y, false, false);
        return i32x4{(int)a.x, (int)bq.x, (int)a.y, (int)bq.y};
    };
    auto hl = [&](f32x4 v) __attribute__((always_inline)) -> i32x4 {      // an fp32 fragment as a B operand: 16-bit high parts | low parts
        i32x4 o;
        o.x = (int)pack2<BF16>(v.x, v.y); o.y = (int)pack2<BF16>(v.z, v.w);
        if (BF16) {
            float a, bq, c, d;
            unpack2<BF16>((unsigned)o.x, a, bq);
            unpack2<BF16>((unsigned)o.y, c, d);
            o.z = (int)pack2<BF16>(v.x - a, v.y - bq); o.w = (int)pack2<BF16>(v.z - c, v.w - d);
        } else {
            o.z = 0; o.w = 0;
        }
        return o;
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the first CH_D input rows have landed (and the weights)
    chain_barrier();                                        // (pairs with the barrier in the kernel body in front of the layer waves' loops)

    for (int S = 0; S < nsteps; ++S) {
        // the input row CH_D ahead of layer 1, into the slot layer 3's epilogue released a step ago
        if (chain_advance(p, dc, RJ)) dma_lanes();
        dma_row();
        if (chain_advance(p, cur, RJ)) store_lanes();
        const char* const ub = smem + CH_OFF_U + (cur.V & 1) * CH_U_ROW + lane * 16;
        i32x4 bs[NT][G];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < G; ++e) bs[t][e] = *reinterpret_cast<const i32x4*>(ub + (t * G + e) * 1024);
        // c5: per accumulator k tiles ascending, high image then low image (conv48rp_kernel's order)
        f32x4 d1[PNT1][G];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int lo = 0; lo < (plo ? 2 : 1); ++lo)
#pragma unroll
                for (int ot = 0; ot < PNT1; ++ot)
#pragma unroll
                    for (int e = 0; e < G; ++e)
                        d1[ot][e] = mfma32<BF16>(lo ? a1l[kt][ot] : a1h[kt][ot], bs[kt][e], (kt == 0 && lo == 0) ? pb1[ot] : d1[ot][e]);
        uint2 pk1[PNT1][G];
        i32x4 bs2[PNT1][G];
#pragma unroll
        for (int ot = 0; ot < PNT1; ++ot)
#pragma unroll
            for (int e = 0; e < G; ++e) {
                f32x4 v = d1[ot][e];
                v.x = act1(v.x, p1s); v.y = act1(v.y, p1s); v.z = act1(v.z, p1s); v.w = act1(v.w, p1s);
                pk1[ot][e].x = pack2<BF16>(v.x, v.y);
                pk1[ot][e].y = pack2<BF16>(v.z, v.w);
                bs2[ot][e] = hl(v);
            }
        // esa.conv1 on c5's fp32 result
        f32x4 d2[G];
#pragma unroll
        for (int kt = 0; kt < PNT1; ++kt)
#pragma unroll
            for (int e = 0; e < G; ++e)
#pragma unroll
                for (int lo = 0; lo < (plo ? 2 : 1); ++lo)
                    d2[e] = mfma32<BF16>(lo ? a2l[kt] : a2h[kt], bs2[kt][e], (kt == 0 && lo == 0) ? pb2 : d2[e]);
        uint2 pk2[G];
#pragma unroll
        for (int e = 0; e < G; ++e) {
            pk2[e].x = pack2<BF16>(d2[e].x, d2[e].y);
            pk2[e].y = pack2<BF16>(d2[e].z, d2[e].w);
        }
        // four stores per row: tiles 0 | 1 of each pixel group (64 B per pixel), tile 2 of both groups, conv1's tile of both groups
        {
            const int gy = cur.Y0 - CH_HALO + cur.r;
            const bool rowok = cur.ok && cur.r >= CH_HALO && cur.r < RJ - CH_HALO && gy < p.H;
            const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(p.y1 + (size_t)cur.n * y1_img, 0, (int)y1_img, 0x00020000);
            const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(p.y2 + (size_t)cur.n * y2_img, 0, (int)y2_img, 0x00020000);
            const unsigned ro1 = (unsigned)gy * rowb1, ro2 = (unsigned)gy * rowb2;
            __builtin_amdgcn_raw_buffer_store_b128(swap16(pk1[0][0], pk1[1][0]), r1, rowok ? sA0 + ro1 : OOB, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(swap16(pk1[0][1], pk1[1][1]), r1, rowok ? sA1 + ro1 : OOB, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(swap16(pk1[2][0], pk1[2][1]), r1, rowok ? sB + ro1 : OOB, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(swap16(pk2[0], pk2[1]), r2, rowok ? sC + ro2 : OOB, 0, 0);
        }
        // Layer 1 reads input row S + 2 in the next step: it was requested in step S - 6 (or by the prologue, which waited for everything);
        // younger than its four pieces are the four stores of step S - 6, and four pieces + four stores of each of the steps S - 5 .. S
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 * CH_D - 12) : "memory");
        chain_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the trailing zero-fill DMA must not outlive the block)
}

template <bool BF16>
__global__ __launch_bounds__(256, 1) void rlfb_chain_kernel(const ChainK p)
{
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int RJ = p.RS + 2 * CH_HALO;
    // jobs of this block -> steps: the post chain finishes the last job's last slot CH_LAGP steps behind layer 1 (even count: two steps per loop pass)
    int nj = 0;
    while (chain_job_index(p, nj) >= 0) ++nj;
    const int nsteps = (nj * RJ + CH_LAGP + 1) & ~1;
    if (wv == 3) chain_post_wave<BF16>(p, smem, nsteps, RJ);
    else if (wv == 2) chain_layer_wave<BF16, true>(p, smem, 2, nsteps, RJ);
    else chain_layer_wave<BF16, false>(p, smem, wv, nsteps, RJ);
}

}  // namespace

namespace {

template <bool BF16>
int launch_rlfb_chain(const ChainK& k, hipStream_t st)
{
    static std::atomic<unsigned> attr_set[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return ESR_ERR_LAUNCH;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rlfb_chain_kernel<BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS);
        if (e != hipSuccess) {
            esr_set_err("hipFuncSetAttribute(rlfb_chain_kernel, MaxDynamicSharedMemorySize)", e);
            return ESR_ERR_LAUNCH;
        }
        attr_set[dev].store(1u, std::memory_order_relaxed);
    }
    const int grid = k.njobs < 256 ? k.njobs : 256;
    esr_note_kernel("rlfb_chain_kernel<%s>", esr_tf(BF16));
    hipLaunchKernelGGL((rlfb_chain_kernel<BF16>), dim3(grid), dim3(256), CH_LDS, st, k);
    return esr_check_launch("rlfb_chain_kernel launch");
}

// Row segments per image: a block's cost is (jobs per block) x (RS + 6) + 9 steps, a step is the same work whatever the job -- take the
// segment count with the cheapest slowest block on 256 CUs (ties: fewer, longer segments = less halo traffic).
void chain_geometry(int n, int h, int w, int* sx, int* sy, int* rs)
{
    const int SX = (w + CH_WS - 1) / CH_WS;
    long best = -1;
    int best_sy = 1;
    const int sy_max = h / 4 > 0 ? h / 4 : 1;                // RS >= 4
    for (int SY = 1; SY <= sy_max && SY <= 4096; ++SY) {
        const int RS = (h + SY - 1) / SY;
        const int sy_eff = (h + RS - 1) / RS;
        if (sy_eff != SY) continue;                          // the same RS with fewer segments was already priced
        const long jobs = (long)n * SX * SY;
        const long per_block = (jobs + 255) / 256;
        const long cost = per_block * (RS + 2 * CH_HALO) + CH_LAGP + 1;
        if (best < 0 || cost < best) { best = cost; best_sy = SY; }
    }
    *sx = SX;
    *sy = best_sy;
    *rs = (h + best_sy - 1) / best_sy;
}

}  // namespace

extern "C" {

int esr_conv_chain_supported(const esr_chain_desc* d)
{
    if (!d || d->n <= 0 || d->h <= 0 || d->w <= 0) return 0;
    if (d->storage != ESR_STORE_BF16 && d->storage != ESR_STORE_F16) return 0;
    if (d->compute != (d->storage == ESR_STORE_BF16 ? ESR_COMPUTE_BF16 : ESR_COMPUTE_F16)) return 0;
    // RLFB: three 3x3 layers over 48 physical channels, LeakyReLU / none, the third one + the chain's input after its activation, then a
    // 1x1 of three output tiles and a 1x1 of one (no activation on the second)
    if (d->n_layers != 3 || d->res_mode != ESR_RES_POST_ACT) return 0;
    if (d->act != ESR_ACT_LRELU && d->act != ESR_ACT_RELU && d->act != ESR_ACT_NONE) return 0;
    if (esr_round_up(d->cin, 16) != 48 || esr_round_up(d->cmid, 16) != 48 || esr_round_up(d->cout, 16) != 48) return 0;
    if (!d->post_wpacked || !d->post2_wpacked || esr_round_up(d->post_cout, 16) != 48 || d->post2_cout <= 0 || d->post2_cout > 16) return 0;
    if (d->post_act != ESR_ACT_NONE && d->post_act != ESR_ACT_LRELU && d->post_act != ESR_ACT_RELU) return 0;
    if (d->h < 4) return 0;
    if ((double)d->h * d->w * d->in.pitch * 2.0 >= 2147483647.0 || (double)d->h * d->w * d->post_out.pitch * 2.0 >= 2147483647.0 ||
        (double)d->h * d->w * d->post2_out.pitch * 2.0 >= 2147483647.0)
        return 0;                                            // per-image raw buffers < 2 GiB (out-of-range offset 0x80000000)
    return 1;
}

int esr_conv_chain_s16(const esr_chain_desc* d, void* hip_stream)
{
    if (!d || !d->in.ptr || !d->post_out.ptr || !d->post2_out.ptr) return ESR_ERR_BAD_ARG;
    for (int i = 0; i < 3; ++i)
        if (!d->wpacked[i]) return ESR_ERR_BAD_ARG;
    if (!esr_conv_chain_supported(d)) return ESR_ERR_UNSUPPORTED;
    if ((d->in.pitch & 7) || (d->in.coff & 7) || d->in.coff + 48 > d->in.pitch) return ESR_ERR_BAD_ARG;
    const int p1c8 = esr_round_up(d->post_cout, 8), p2c8 = esr_round_up(d->post2_cout, 8);
    if ((d->post_out.pitch & 7) || (d->post_out.coff & 7) || d->post_out.coff + p1c8 > d->post_out.pitch) return ESR_ERR_BAD_ARG;
    if ((d->post2_out.pitch & 7) || (d->post2_out.coff & 7) || d->post2_out.coff + p2c8 > d->post2_out.pitch) return ESR_ERR_BAD_ARG;
    ChainK k;
    memset(&k, 0, sizeof(k));
    k.x = static_cast<const char*>(d->in.ptr);
    k.w0 = static_cast<const char*>(d->wpacked[0]); k.w1 = static_cast<const char*>(d->wpacked[1]); k.w2 = static_cast<const char*>(d->wpacked[2]);
    k.pw1 = static_cast<const char*>(d->post_wpacked); k.pw2 = static_cast<const char*>(d->post2_wpacked);
    k.y1 = static_cast<char*>(d->post_out.ptr); k.y2 = static_cast<char*>(d->post2_out.ptr);
    k.N = d->n; k.H = d->h; k.W = d->w;
    k.in_pitch = d->in.pitch; k.in_coff = d->in.coff;
    k.y1_pitch = d->post_out.pitch; k.y1_coff = d->post_out.coff; k.y2_pitch = d->post2_out.pitch; k.y2_coff = d->post2_out.coff;
    k.p1_cout8 = p1c8; k.p2_cout8 = p2c8;
    k.slope = d->act == ESR_ACT_LRELU ? d->slope : (d->act == ESR_ACT_RELU ? 0.f : 1.f);
    k.p1_slope = d->post_act == ESR_ACT_LRELU ? d->slope : (d->post_act == ESR_ACT_RELU ? 0.f : 1.f);
    k.WS = CH_WS;
    chain_geometry(d->n, d->h, d->w, &k.SX, &k.SY, &k.RS);
    const double jobs = (double)d->n * k.SX * k.SY;
    if (jobs >= 2147483647.0) return ESR_ERR_UNSUPPORTED;
    k.njobs = (int)jobs;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    return d->storage == ESR_STORE_BF16 ? launch_rlfb_chain<true>(k, st) : launch_rlfb_chain<false>(k, st);
}

}  // extern "C"
