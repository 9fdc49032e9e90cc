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
// exactly the values the separate launches would have stored (t1, t2) or kept in fp32 (u, as hi + lo B operands).  One s_barrier per step
// is the only synchronisation; a layer lags its producer by three steps (one row of halo, one step of MFMAs, one step of deferred epilogue).
//
// What a SIMD can do (tools/r05/mfma_pair_probe.hip, mfma_valu_probe.hip; LAB_NOTES round 5): one v_mfma_f32_16x16x32 per 16 cycles, from
// one wave or two -- two waves of a SIMD SERIALISE on the matrix pipe (the older one first) and a partner's VALU work gets about one issue slot
// per MFMA, so a second wave per SIMD buys nothing (an 8-wave version of this kernel, tools/r05/variants/, ran the same step in the same
// time).  A wave issues one instruction per 4 cycles: behind each MFMA TWO plain VALU instructions are free, every further instruction
// (VALU, SALU, s_waitcnt alike) costs 4 cycles, a ds_read_b128 ~6.  So the finished row's epilogue is cut into BUNDLES of two VALU
// instructions, one bundle behind each MFMA of the next row, LDS reads one per MFMA, and the step's scalar bookkeeping is kept to a few
// dozen instructions.
// Halo: a layer computes all 32 columns, each one pixel further right than its producer's, so the strip's 28 + 6 input columns shrink to 28
// valid output columns (columns / rows outside the IMAGE are written as zeros: every layer's own zero padding); rows: RS + 6 slots per job.
// Jobs of a block follow each other without draining the pipeline.  Rings: input 16 rows x 4 KB (DMA, 8 rows ahead), t1 / t2 4 rows, u 2 rows.
//
// Same packed weights, fragment maps, per-accumulator operation order and roundings as conv48r_kernel / conv48rp_kernel: the results are
// bit-identical to the three separate launches (tests/test_gpu_chain.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
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
#ifdef ESR_CHAIN_TRACE
    unsigned long long* trace;   // research builds (tools/r05/chain_trace.py): [block < 4][wave][step < CH_TRACE_STEPS][CH_TRACE_W] s_memtime stamps
#endif
};

constexpr int CH_NW = 4;                      // waves per block: one per SIMD
// Geometry of a block that computes G MFMA pixel groups (16 G columns) per row: strips of 16 G - 4 valid output columns.  G = 2 for single
// images (more, smaller jobs: 339 x 510 = 247 jobs on 256 CUs), G = 3 for batches (44 of 48 columns valid instead of 28 of 32, and the
// per-step overhead -- barrier, bookkeeping -- spread over 135 instead of 90 MFMAs per wave).
template <int G>
struct ChGeo {
    static constexpr int COLS = 16 * G + 2;                     // pixels of a ring row
    static constexpr int WS = 16 * G - 4;                       // valid output columns of a strip
    static constexpr int PIECES = (COLS * 6 + 63) / 64;         // 1 KB DMA pieces of an input row (COLS pixels x 96 B)
    static constexpr int IN_PITCH = PIECES * 1024;
    static constexpr int NR_IN = 16;
    static constexpr int T_PITCH = (COLS * 96 + 255) / 256 * 256;   // t1 / t2 ring row
    static constexpr int NR_T = 4;
    static constexpr int U_ROW = 3 * G * 1024;                  // u ring row: [tile][group] 1 KB B-operand fragments
    static constexpr int NR_U = 2;
    static constexpr int OFF_IN = 0;
    static constexpr int OFF_T1 = NR_IN * IN_PITCH;
    static constexpr int OFF_T2 = OFF_T1 + NR_T * T_PITCH;
    static constexpr int OFF_U = OFF_T2 + NR_T * T_PITCH;
    static constexpr int LDS = OFF_U + NR_U * U_ROW;
    static constexpr int STORES = G + 2 * ((G + 1) / 2);        // stores per row of the post wave: tiles 0 | 1 per group, tile 2 and conv1's tile per group PAIR
    // the input DMA runs D rows ahead of layer 1; the post wave's counted wait (PIECES + STORES) (D - 2) + STORES must fit vmcnt's 6 bits
    static constexpr int D = (PIECES + STORES) * 6 + STORES <= 63 ? 8 : 6;
    static_assert((PIECES + STORES) * (D - 2) + STORES <= 63 && D + 8 <= NR_IN, "vmcnt range / input ring depth");
};
constexpr int CH_LAG = 3;                     // a layer lags its producer by this many steps
constexpr int CH_LAGP = 2 * CH_LAG + 2;       // wave 3 (post chain) lags layer 1 by this many steps
constexpr int CH_HALO = 3;                    // row slot r of a job is image row Y0 - 3 + r; RS + 6 slots per job
constexpr int CH_WIMG = 45 * 1024;            // 3 chunks x 5 tap pairs x 3 tiles
// Out-of-range markers of the per-lane / per-row byte offsets that are ADDED to form a buffer offset: a lane that must not touch memory
// carries CH_LOOB, a row that must not the (wave-uniform) CH_ROOB; images are < 1 GiB (checked by the host), so any sum with a marker in it
// is >= the buffer's num_records (hardware: loads return zero -- the zero padding --, stores are dropped) and no sum wraps to a valid offset
constexpr unsigned CH_LOOB = 0x40000000u, CH_ROOB = 0x80000000u;

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
    return k < 0 ? -1 : s16_tile_index(k, p.njobs);          // (the kernels' common walk: neighbouring strips on one XCD)
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

#ifdef ESR_CHAIN_TRACE
constexpr int CH_TRACE_STEPS = 40;
constexpr int CH_TRACE_W = 6;             // stamps per step: before / after the barrier, four inside the step
#define CH_STAMP(v) asm volatile("s_memtime %0" : "=s"(v) :: "memory")
#else
#define CH_STAMP(v)
#endif
#ifdef CH_ABL_VOL
#define CH_MFMA_ASM asm volatile       // ablation builds (tools/r05/chain_trace.py): the MFMAs stay although nothing reads their results
#else
#define CH_MFMA_ASM asm
#endif

__device__ __forceinline__ void chain_barrier(char* const smem = nullptr, int wv = 0, int step = -1, unsigned long long m0 = 0, unsigned long long m1 = 0,
                                              unsigned long long m2 = 0, unsigned long long m3 = 0, int lds_end = 0)
{
    // every LDS access of the step has completed (writes visible, ring slots released); the "memory" clobber keeps hipcc from moving
    // LDS accesses across it
#ifdef ESR_CHAIN_TRACE
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1) :: "memory");
    if (smem && step >= 0 && step < CH_TRACE_STEPS && (threadIdx.x & 63) == 0) {
        unsigned long long* tr = reinterpret_cast<unsigned long long*>(smem + lds_end) + (wv * CH_TRACE_STEPS + step) * CH_TRACE_W;
        tr[0] = t0; tr[1] = t1; tr[2] = m0; tr[3] = m1; tr[4] = m2; tr[5] = m3;
    }
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// two independent plain VALU instructions = what fits behind an MFMA for free; written as asm so that hipcc neither fuses, packs (v_pk_*
// beside MFMAs costs 4x a plain instruction) nor moves them
__device__ __forceinline__ void mul2(float& a, float& b, float x, float y, float s)
{
    asm("v_mul_f32 %0, %2, %3\n\tv_mul_f32 %1, %2, %4" : "=&v"(a), "=&v"(b) : "v"(s), "v"(x), "v"(y));
}
#define CH_MAX2(X, Y, a, b) do { float x_ = (X), y_ = (Y); asm("v_max_f32 %0, %0, %2\n\tv_max_f32 %1, %1, %3" : "+v"(x_), "+v"(y_) : "v"(a), "v"(b)); (X) = x_; (Y) = y_; } while (0)

// ---- waves 0 .. 2: one 3x3 layer each -------------------------------------------------------------------------------------------------
// LAST = false: act(conv) rounded into the next layer's ring (zeros outside the image).  LAST = true (c3_r): act(conv) + x (the block input
// from the input ring), kept in fp32 and handed to wave 3 as the post chain's B operands (16-bit high parts | low parts).
template <bool BF16, bool LAST, int G>
__device__ __forceinline__ void chain_layer_wave(const ChainK& p, char* const smem, const int layer, const int nsteps, const int RJ)
{
    typedef ChGeo<G> Geo;
    constexpr int NCH = 3, PAIRS = 5, NT = 3, NG = NCH * PAIRS, NF = NT * G;      // NF: fragments (MFMAs of a group) per row
    const int lane = threadIdx.x & 63, px = lane & 15, kq = lane >> 4;
    const int src_base = layer == 0 ? Geo::OFF_IN : (layer == 1 ? Geo::OFF_T1 : Geo::OFF_T2);
    const int src_pitch = layer == 0 ? Geo::IN_PITCH : Geo::T_PITCH;
    const int src_mask = layer == 0 ? Geo::NR_IN - 1 : Geo::NR_T - 1;
    const int dst_base = layer == 0 ? Geo::OFF_T1 : Geo::OFF_T2;
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

    ChainCur cur = chain_cursor(CH_LAG * layer, RJ);
    unsigned cm[G];                             // this job's column masks (the layer's output column 16 e + px lies inside the image)
#pragma unroll
    for (int e = 0; e < G; ++e) cm[e] = 0u;
    int baddr[2][PAIRS];
    {
        const int ra0 = rowaddr(cur.V), ra1 = rowaddr(cur.V + 1), ra2 = rowaddr(cur.V + 2);
        baddr[0][0] = ra0 + laneoff[0]; baddr[0][1] = (hi_tap ? ra1 : ra0) + laneoff[1]; baddr[0][2] = ra1 + laneoff[2];
        baddr[0][3] = ra2 + laneoff[3]; baddr[0][4] = ra2 + laneoff[4];
    }

    // epilogue parameters of a row, stashed by the step that computes it for the step that finishes it
    struct Epi { int wa; unsigned m[G]; int ra; };
    Epi ep[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ep[i].wa = LAST ? Geo::OFF_U + lane * 16 : dst_base + lane_w;
        ep[i].ra = Geo::OFF_IN + lane_r;
#pragma unroll
        for (int e = 0; e < G; ++e) ep[i].m[e] = 0u;
    }

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

    float ta = 0.f, tb = 0.f, tc = 0.f, td = 0.f;
    uint2 pk = {0u, 0u};
    uint2 rraw[2] = {{0u, 0u}, {0u, 0u}};
    i32x4 ob = {0, 0, 0, 0};

    // The finished row's epilogue as BUNDLES of (at most) two VALU instructions or one LDS access: bundle s runs behind MFMA s of the next
    // row.  pp = the finished row's accumulators.
    auto micro = [&](auto pp_, auto s_) __attribute__((always_inline)) {
        constexpr int pp = decltype(pp_)::value, s = decltype(s_)::value;
#ifdef CH_ABL_NOEPI
        return;
#endif
        const Epi& E = ep[pp];
        if constexpr (!LAST) {
            // fragment f = 2 t + e in bundles 2 + 7 f ..: slope products, maxima (x, y | z, w), rounding, mask, store
            constexpr int S0 = 2, PER = 7;
            if constexpr (s >= S0 && s < S0 + PER * NF) {
                constexpr int f = (s - S0) / PER, m = (s - S0) % PER, t = f / G, e = f % G;
                f32x4& A = acc[pp][t][e];
                if constexpr (m == 0) mul2(ta, tb, A.x, A.y, slope);
                else if constexpr (m == 1) CH_MAX2(A.x, A.y, ta, tb);
                else if constexpr (m == 2) mul2(ta, tb, A.z, A.w, slope);
                else if constexpr (m == 3) CH_MAX2(A.z, A.w, ta, tb);
                else if constexpr (m == 4) { pk.x = pack2<BF16>(A.x, A.y); pk.y = pack2<BF16>(A.z, A.w); }
                else if constexpr (m == 5) { const unsigned msk = E.m[e]; pk.x &= msk; pk.y &= msk; }
                else *reinterpret_cast<uint2*>(smem + E.wa + e * (16 * 96) + t * 32) = pk;
            }
        } else {
            // fragment f in bundles 15 f .. 15 f + 14 (all 90): lrelu(conv) + x (team04_rlfn.py:117-119), the fp32 value as high parts | low
            // parts; its residual is read five bundles ahead (fragment 0's: behind the previous row's last bundles is too early -- bundle 0)
            constexpr int PER = 15;
            constexpr int f = s / PER, m = s % PER, t = f / G, e = f % G;
            f32x4& A = acc[pp][t][e];
            // residual of fragment f + 1 (of fragment 0 in bundle 1 of fragment 0: its first use is bundle 4)
            if constexpr (m == 9 && f + 1 < NF) {
                constexpr int f1 = f + 1;
                rraw[f1 & 1] = *reinterpret_cast<const uint2*>(smem + E.ra + (f1 % G) * (16 * 96) + (f1 / G) * 32);
            }
            if constexpr (s == 0) rraw[0] = *reinterpret_cast<const uint2*>(smem + E.ra);
            if constexpr (m == 0) mul2(ta, tb, A.x, A.y, slope);
            else if constexpr (m == 1) CH_MAX2(A.x, A.y, ta, tb);
            else if constexpr (m == 2) mul2(ta, tb, A.z, A.w, slope);
            else if constexpr (m == 3) CH_MAX2(A.z, A.w, ta, tb);
            else if constexpr (m == 4) unpack2<BF16>(rraw[f & 1].x, ta, tb);
            else if constexpr (m == 5) { A.x += ta; A.y += tb; }
            else if constexpr (m == 6) unpack2<BF16>(rraw[f & 1].y, ta, tb);
            else if constexpr (m == 7) { A.z += ta; A.w += tb; }
            else if constexpr (m == 8) {
                ob.x = (int)pack2<BF16>(A.x, A.y); ob.y = (int)pack2<BF16>(A.z, A.w);
                if (!BF16) { ob.z = 0; ob.w = 0; }
            } else if constexpr (m == 9) { if (BF16) unpack2<BF16>((unsigned)ob.x, ta, tb); }
            else if constexpr (m == 10) { if (BF16) unpack2<BF16>((unsigned)ob.y, tc, td); }
            else if constexpr (m == 11) { if (BF16) { ta = A.x - ta; tb = A.y - tb; } }
            else if constexpr (m == 12) { if (BF16) { tc = A.z - tc; td = A.w - td; } }
            else if constexpr (m == 13) { if (BF16) { ob.z = (int)pack2<BF16>(ta, tb); ob.w = (int)pack2<BF16>(tc, td); } }
            else *reinterpret_cast<i32x4*>(smem + E.wa + (t * G + e) * 1024) = ob;
        }
    };

    int step_no = 0;
    auto run_step = [&](auto par_) __attribute__((always_inline)) {
        constexpr int par = decltype(par_)::value;
        unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0;
        if (chain_advance(p, cur, RJ) && !LAST) {
            const int gx = cur.X0 - 2 + layer + px;
#pragma unroll
            for (int e = 0; e < G; ++e) cm[e] = (cur.ok && (unsigned)(gx + 16 * e) < (unsigned)p.W) ? 0xffffffffu : 0u;
        }
        {
            // this row's epilogue parameters (used one step later)
            if (!LAST) {
                const bool rowok = (unsigned)(cur.Y0 - CH_HALO + cur.r) < (unsigned)p.H;
                ep[par].wa = dst_base + (cur.V & (Geo::NR_T - 1)) * Geo::T_PITCH + lane_w;
#pragma unroll
                for (int e = 0; e < G; ++e) ep[par].m[e] = rowok ? cm[e] : 0u;
            } else {
                ep[par].wa = Geo::OFF_U + (cur.V & (Geo::NR_U - 1)) * Geo::U_ROW + lane * 16;
                ep[par].ra = Geo::OFF_IN + (cur.V & (Geo::NR_IN - 1)) * Geo::IN_PITCH + lane_r;
            }
        }
        // the next step's row addresses (its window = rows V, V + 1, V + 2 of the producer)
        const int na0 = __builtin_amdgcn_readfirstlane(rowaddr(cur.V)), na1 = __builtin_amdgcn_readfirstlane(rowaddr(cur.V + 1)),
                  na2 = __builtin_amdgcn_readfirstlane(rowaddr(cur.V + 2));
        const int* const ba = baddr[par];
        int* const nb = baddr[par ^ 1];
        CH_STAMP(m0);
        static_for<NG>([&](auto g_) __attribute__((always_inline)) {
            constexpr int g = decltype(g_)::value;
            constexpr int c = g / PAIRS, q = g % PAIRS;
            static_for<NF>([&](auto m_) __attribute__((always_inline)) {
                constexpr int mi = decltype(m_)::value, t = mi / G, e = mi % G;
                // groups 0 .. 2 take the fragments read during the previous step; they are consumed before group 8 overwrites them
                const i32x4 bf = g < 3 ? pb[g < 3 ? g : 0][e] : b[g % 5][e];
#ifndef CH_ABL_NOMFMA
                if (g == 0) {
                    // (early clobber: the first write of an accumulator must not land on the registers of a fragment that dies here)
                    if (BF16) CH_MFMA_ASM("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(bf), "v"(bia[t]));
                    else CH_MFMA_ASM("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(bf), "v"(bia[t]));
                } else {
                    if (BF16) CH_MFMA_ASM("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(bf));
                    else CH_MFMA_ASM("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[par][t][e]) : "a"(wr[c][q][t]), "v"(bf));
                }
#endif
                // LDS reads ONE per MFMA, behind the group's first two (hipcc closes the step's barrier, an asm it cannot see into, with
                // lgkmcnt(0) in front of the first use of a fragment read before it: nothing is outstanding there yet)
#ifndef CH_ABL_NOBREAD
                if constexpr (mi < G && g + 3 < NG) {
                    // (pixel group G - 1 first: the group's first MFMA uses group 0's fragment, the YOUNGEST read -- LDS returns in order, so the one
                    // s_waitcnt in front of it covers the others and hipcc emits no second one: 11 instructions less per step)
                    constexpr int g3 = g + 3, e3 = G - 1 - mi;
                    b[g3 % 5][e3] = *reinterpret_cast<const i32x4*>(smem + ba[g3 % PAIRS] + (g3 / PAIRS) * 32 + e3 * (16 * 96));
                }
                // the next step's groups 0 .. 2 (taps 0 .. 5: the two older rows of its window, complete since the last barrier)
                if constexpr (mi >= G && mi < 2 * G && g >= 8 && g < 11)
                    pb[g - 8][2 * G - 1 - mi] = *reinterpret_cast<const i32x4*>(smem + nb[g - 8] + (2 * G - 1 - mi) * (16 * 96));
#endif
                // the next step's B addresses, one VALU instruction at a time (groups 5 .. 7: ahead of their first use in group 8)
                if constexpr (g == 5 && mi == NF - 2) nb[0] = na0 + laneoff[0];
                if constexpr (g == 5 && mi == NF - 1) nb[1] = (hi_tap ? na1 : na0) + laneoff[1];
                if constexpr (g == 6 && mi == NF - 2) nb[2] = na1 + laneoff[2];
                if constexpr (g == 6 && mi == NF - 1) nb[3] = na2 + laneoff[3];
                if constexpr (g == 7 && mi == NF - 2) nb[4] = na2 + laneoff[4];
                micro(std::integral_constant<int, par ^ 1>{}, std::integral_constant<int, NF * g + mi>{});
                if constexpr (mi == NF - 1 && g == 4) CH_STAMP(m1);
                if constexpr (mi == NF - 1 && g == 9) CH_STAMP(m2);
                if constexpr (mi == NF - 1 && g == 14) CH_STAMP(m3);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        chain_barrier(smem, layer, step_no++, m0, m1, m2, m3, Geo::LDS);
    };
    for (int it = 0; it < nsteps; it += 2) {
        run_step(std::integral_constant<int, 0>{});
        run_step(std::integral_constant<int, 1>{});
    }
}

// ---- wave 3: the input DMA, c5 and esa.conv1 on the fp32 rows of u, the stores ---------------------------------------------------------------
template <bool BF16, int G>
__device__ __forceinline__ void chain_post_wave(const ChainK& p, char* const smem, const int nsteps, const int RJ)
{
    typedef ChGeo<G> Geo;
    constexpr int NT = 3, PNT1 = 3, NPAIR = G / 2, ODD = G & 1;
    constexpr int P1_IMG = NT * PNT1 * 1024, P2_IMG = PNT1 * 1024;
    constexpr bool plo = BF16;
    const int lane = threadIdx.x & 63, px = lane & 15, kq = lane >> 4;
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- input DMA: row V of the input ring = COLS pixels x 6 sixteen-byte parts, PIECES pieces of 64 lanes.  Per job: the buffer resource
    // of the image and every lane's byte offset in image row 0 (CH_LOOB: a pixel outside the strip's window or the image: zero fill);
    // per row: ONE v_add per piece with the row's (wave-uniform) offset or CH_ROOB
    const size_t img_bytes = (size_t)p.H * p.W * p.in_pitch * 2;
    const unsigned rowb_in = (unsigned)p.W * (unsigned)p.in_pitch * 2u;
    ChainCur dc = chain_cursor(0, RJ);          // runs Geo::D rows ahead of layer 1 after the prologue
    unsigned dbase[Geo::PIECES];
#pragma unroll
    for (int i = 0; i < Geo::PIECES; ++i) dbase[i] = CH_LOOB;
    i32x4 drs = make_rsrc(p.x, img_bytes);
    auto dma_job = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < Geo::PIECES; ++i) {
            const unsigned sl = (unsigned)(i * 64 + lane);
            const unsigned pixel = sl / 6u, part = sl - pixel * 6u;
            const int gx = dc.X0 - CH_HALO + (int)pixel;
            const bool ok = dc.ok && pixel < (unsigned)Geo::COLS && (unsigned)gx < (unsigned)p.W;
            dbase[i] = ok ? (unsigned)(gx * p.in_pitch + p.in_coff) * 2u + part * 16u : CH_LOOB;
        }
        drs = make_rsrc(p.x + (size_t)dc.n * img_bytes, img_bytes);
        drs.x = __builtin_amdgcn_readfirstlane(drs.x); drs.y = __builtin_amdgcn_readfirstlane(drs.y);
        drs.z = __builtin_amdgcn_readfirstlane(drs.z); drs.w = __builtin_amdgcn_readfirstlane(drs.w);
    };
    auto dma_row = [&]() __attribute__((always_inline)) {
        const int gy = dc.Y0 - CH_HALO + dc.r;
        const unsigned roff = (dc.ok && (unsigned)gy < (unsigned)p.H) ? (unsigned)gy * rowb_in : CH_ROOB;
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)(Geo::OFF_IN + (dc.V & (Geo::NR_IN - 1)) * Geo::IN_PITCH));
        unsigned v[Geo::PIECES];
#pragma unroll
        for (int i = 0; i < Geo::PIECES; ++i) v[i] = dbase[i] + roff;
        unsigned keep;
        // 1 KB pieces, M0 (the LDS destination) stepped between them
        if constexpr (Geo::PIECES == 4)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %6, 0 offen lds\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %6, 0 offen lds\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %6, 0 offen lds\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %6, 0 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(dst), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "s"(drs) : "memory", "scc");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %7, 0 offen lds\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %7, 0 offen lds\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %7, 0 offen lds\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %7, 0 offen lds\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %7, 0 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(dst), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[Geo::PIECES - 1]), "s"(drs) : "memory", "scc");
    };
    static_assert(Geo::PIECES == 4 || Geo::PIECES == 5, "DMA asm blocks");
    for (int i = 0; i < Geo::D; ++i) {
        if (chain_advance(p, dc, RJ)) dma_job();
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
    const bool p1act = p1s != 1.f;              // (wave-uniform: RLFB's c5 has no activation)

    // ---- stores: per job the images' buffer resources and the lane's byte offsets in row 0 (CH_LOOB: column outside the strip / image,
    // channel not stored); per row one v_add per store.  Tiles 0 | 1 of a pixel group leave as one 16-byte store per lane
    // (v_permlane16_swap pairs the tiles: 64 contiguous bytes per pixel); tile 2 and conv1's tile of a group PAIR likewise (the swap pairs
    // the groups); an odd last group stores them as 8 bytes per lane
    const size_t y1_img = (size_t)p.H * p.W * p.y1_pitch * 2, y2_img = (size_t)p.H * p.W * p.y2_pitch * 2;
    const unsigned rowb1 = (unsigned)p.W * (unsigned)p.y1_pitch * 2u, rowb2 = (unsigned)p.W * (unsigned)p.y2_pitch * 2u;
    ChainCur cur = chain_cursor(CH_LAGP, RJ);
    unsigned sA[G], sB[NPAIR + ODD], sC[NPAIR + ODD];
#pragma unroll
    for (int e = 0; e < G; ++e) sA[e] = CH_LOOB;
#pragma unroll
    for (int i = 0; i < NPAIR + ODD; ++i) { sB[i] = CH_LOOB; sC[i] = CH_LOOB; }
    __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(p.y1, 0, (int)y1_img, 0x00020000);
    __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(p.y2, 0, (int)y2_img, 0x00020000);
    auto store_job = [&]() __attribute__((always_inline)) {
        const int wsj = min(p.WS, p.W - cur.X0);                       // valid output columns of this strip
        const int chA = (kq & 1) * 16 + (kq >> 1) * 8, chB = 32 + (kq >> 1) * 8, ch2 = (kq >> 1) * 8;
        auto off1 = [&](int col, int ch) -> unsigned {
            return (cur.ok && col < wsj && ch < p.p1_cout8) ? (unsigned)((cur.X0 + col) * p.y1_pitch + p.y1_coff + ch) * 2u : CH_LOOB;
        };
        auto off2 = [&](int col, int ch) -> unsigned {
            return (cur.ok && col < wsj && ch < p.p2_cout8) ? (unsigned)((cur.X0 + col) * p.y2_pitch + p.y2_coff + ch) * 2u : CH_LOOB;
        };
#pragma unroll
        for (int e = 0; e < G; ++e) sA[e] = off1(16 * e + px, chA);
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) {
            const int colB = 32 * i + (kq & 1) * 16 + px;               // the swap gives lanes kq & 1 the pixel of group 2 i + (kq & 1)
            sB[i] = off1(colB, chB);
            sC[i] = off2(colB, ch2);
        }
        if (ODD) {
            sB[NPAIR] = off1(16 * (G - 1) + px, 32 + kq * 4);           // 8 bytes per lane: channels 4 kq .. of the tile
            sC[NPAIR] = off2(16 * (G - 1) + px, kq * 4);
        }
        r1 = __builtin_amdgcn_make_buffer_rsrc(p.y1 + (size_t)cur.n * y1_img, 0, (int)y1_img, 0x00020000);
        r2 = __builtin_amdgcn_make_buffer_rsrc(p.y2 + (size_t)cur.n * y2_img, 0, (int)y2_img, 0x00020000);
    };
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    auto swap16 = [&](uint2 X, uint2 Y) __attribute__((always_inline)) -> i32x4 { return s16_swap16(X, Y); };
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

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the first Geo::D input rows have landed (and the weights)
    chain_barrier();                                        // (pairs with the barrier in front of the layer waves' loops)

    for (int S = 0; S < nsteps; ++S) {
        // the input row Geo::D ahead of layer 1, into the slot layer 3's epilogue released a step ago
        if (chain_advance(p, dc, RJ)) dma_job();
        dma_row();
        if (chain_advance(p, cur, RJ)) store_job();
        const char* const ub = smem + Geo::OFF_U + (cur.V & (Geo::NR_U - 1)) * Geo::U_ROW + lane * 16;
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
        if (p1act) {
#pragma unroll
            for (int ot = 0; ot < PNT1; ++ot)
#pragma unroll
                for (int e = 0; e < G; ++e) {
                    f32x4& v = d1[ot][e];
                    v.x = act1(v.x, p1s); v.y = act1(v.y, p1s); v.z = act1(v.z, p1s); v.w = act1(v.w, p1s);
                }
        }
        uint2 pk1[PNT1][G];
        i32x4 bs2[PNT1][G];
#pragma unroll
        for (int ot = 0; ot < PNT1; ++ot)
#pragma unroll
            for (int e = 0; e < G; ++e) {
                const f32x4 v = d1[ot][e];
                bs2[ot][e] = hl(v);
                pk1[ot][e].x = (unsigned)bs2[ot][e].x;               // (the rounded value = the high parts)
                pk1[ot][e].y = (unsigned)bs2[ot][e].y;
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
        {
            const int gy = cur.Y0 - CH_HALO + cur.r;
            const bool rowok = cur.ok && cur.r >= CH_HALO && cur.r < RJ - CH_HALO && gy < p.H;
            const unsigned ro1 = rowok ? (unsigned)gy * rowb1 : CH_ROOB, ro2 = rowok ? (unsigned)gy * rowb2 : CH_ROOB;
#pragma unroll
            for (int e = 0; e < G; ++e) __builtin_amdgcn_raw_buffer_store_b128(swap16(pk1[0][e], pk1[1][e]), r1, sA[e] + ro1, 0, 0);
#pragma unroll
            for (int i = 0; i < NPAIR; ++i) {
                __builtin_amdgcn_raw_buffer_store_b128(swap16(pk1[2][2 * i], pk1[2][2 * i + 1]), r1, sB[i] + ro1, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(swap16(pk2[2 * i], pk2[2 * i + 1]), r2, sC[i] + ro2, 0, 0);
            }
            if (ODD) {
                __builtin_amdgcn_raw_buffer_store_b64(i32x2{(int)pk1[2][G - 1].x, (int)pk1[2][G - 1].y}, r1, sB[NPAIR] + ro1, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(i32x2{(int)pk2[G - 1].x, (int)pk2[G - 1].y}, r2, sC[NPAIR] + ro2, 0, 0);
            }
        }
        // Layer 1 reads input row S + 2 in the next step: it was requested in step S + 2 - D (or by the prologue, which waited for
        // everything); younger than its pieces are the stores of that step, and the pieces + stores of each of the D - 2 steps since
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((Geo::PIECES + Geo::STORES) * (Geo::D - 2) + Geo::STORES) : "memory");
        chain_barrier(smem, 3, S, 0, 0, 0, 0, Geo::LDS);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the trailing zero-fill DMA must not outlive the block)
}

template <bool BF16, int G>
__global__ __launch_bounds__(64 * CH_NW, 1) void rlfb_chain_kernel(const ChainK p)
{
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int RJ = p.RS + 2 * CH_HALO;
    // jobs of this block -> steps: the post chain finishes the last job's last slot CH_LAGP steps behind layer 1 (even count: two steps per loop pass)
    int nj = 0;
    while (chain_job_index(p, nj) >= 0) ++nj;
    const int nsteps = (nj * RJ + CH_LAGP + 1) & ~1;
    if (wv == 3) chain_post_wave<BF16, G>(p, smem, nsteps, RJ);
    else if (wv == 2) chain_layer_wave<BF16, true, G>(p, smem, 2, nsteps, RJ);
    else chain_layer_wave<BF16, false, G>(p, smem, wv, nsteps, RJ);
#ifdef ESR_CHAIN_TRACE
    __syncthreads();
    if (p.trace && blockIdx.x < 4)
        for (int i = threadIdx.x; i < CH_NW * CH_TRACE_STEPS * CH_TRACE_W; i += 64 * CH_NW)
            p.trace[(size_t)blockIdx.x * CH_NW * CH_TRACE_STEPS * CH_TRACE_W + i] = reinterpret_cast<const unsigned long long*>(smem + ChGeo<G>::LDS)[i];
#endif
}

}  // namespace

namespace {

#ifdef ESR_CHAIN_TRACE
unsigned long long* g_chain_trace = nullptr;
constexpr int CH_TRACE_LDS = CH_NW * CH_TRACE_STEPS * CH_TRACE_W * 8;
#else
constexpr int CH_TRACE_LDS = 0;
#endif

template <bool BF16, int G>
int launch_rlfb_chain(const ChainK& k, hipStream_t st)
{
    constexpr int LDS = ChGeo<G>::LDS + CH_TRACE_LDS;
    static_assert(LDS <= LDS_LIMIT, "rings fit the LDS");
    static std::atomic<unsigned> attr_set[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return ESR_ERR_LAUNCH;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rlfb_chain_kernel<BF16, G>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            esr_set_err("hipFuncSetAttribute(rlfb_chain_kernel, MaxDynamicSharedMemorySize)", e);
            return ESR_ERR_LAUNCH;
        }
        attr_set[dev].store(1u, std::memory_order_relaxed);
    }
    const int grid = k.njobs < 256 ? k.njobs : 256;
    esr_note_kernel("rlfb_chain_kernel<%s, %d>", esr_tf(BF16), G);
    hipLaunchKernelGGL((rlfb_chain_kernel<BF16, G>), dim3(grid), dim3(64 * CH_NW), LDS, st, k);
    return esr_check_launch("rlfb_chain_kernel launch");
}

// Strip width and row segments: a block's cost is (jobs per block) x (RS + 6) + 9 steps, a step costs about the same whatever the job --
// for each strip width take the segment count with the cheapest slowest block on 256 CUs (ties: fewer, longer segments = less halo traffic),
// then the width whose steps x (cycles per step) is smaller.  Relative step cost measured at 32 x 256 x 256 (0.404 ms / 359 steps against 0.364 ms / 219 steps): G = 3 : G = 2 = 1.48.
long chain_geometry_g(int n, int h, int w, int ws, int* sx, int* sy, int* rs)
{
    const int SX = (w + ws - 1) / ws;
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
    return best;
}

constexpr long CH_STEP_CYCLES_G2 = 2200, CH_STEP_CYCLES_G3 = 3250;

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
    if ((double)d->h * d->w * d->in.pitch * 2.0 >= 1073741824.0 || (double)d->h * d->w * d->post_out.pitch * 2.0 >= 1073741824.0 ||
        (double)d->h * d->w * d->post2_out.pitch * 2.0 >= 1073741824.0)
        return 0;                                            // per-image raw buffers < 1 GiB (the out-of-range markers CH_LOOB / CH_ROOB)
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
    int g = 2;
    {
        int sx2, sy2, rs2, sx3, sy3, rs3;
        const long c2 = chain_geometry_g(d->n, d->h, d->w, ChGeo<2>::WS, &sx2, &sy2, &rs2) * CH_STEP_CYCLES_G2;
        const long c3 = chain_geometry_g(d->n, d->h, d->w, ChGeo<3>::WS, &sx3, &sy3, &rs3) * CH_STEP_CYCLES_G3;
        const char* force = getenv("ESR_CHAIN_G");          // (research: force a strip width)
        if (force ? force[0] == '3' : c3 < c2) { g = 3; k.SX = sx3; k.SY = sy3; k.RS = rs3; }
        else { k.SX = sx2; k.SY = sy2; k.RS = rs2; }
    }
    k.WS = g == 3 ? ChGeo<3>::WS : ChGeo<2>::WS;
    const double jobs = (double)d->n * k.SX * k.SY;
    if (jobs >= 2147483647.0) return ESR_ERR_UNSUPPORTED;
    k.njobs = (int)jobs;
#ifdef ESR_CHAIN_TRACE
    k.trace = g_chain_trace;
#endif
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if (g == 3) return d->storage == ESR_STORE_BF16 ? launch_rlfb_chain<true, 3>(k, st) : launch_rlfb_chain<false, 3>(k, st);
    return d->storage == ESR_STORE_BF16 ? launch_rlfb_chain<true, 2>(k, st) : launch_rlfb_chain<false, 2>(k, st);
}

#ifdef ESR_CHAIN_TRACE
void esr_chain_set_trace(void* p) { g_chain_trace = static_cast<unsigned long long*>(p); }
// (a research build is a library of its own: the three helpers of esr_hip.hip it needs)
#endif

}  // extern "C"
