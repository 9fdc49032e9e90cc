// Round 6: does a v_mfma_f32_32x32x16 with C = 0 that WRITES v[0:15] clobber the sources of a VALU instruction issued just in front of it?
// (rfdb_tail_kernel<fp16>'s flush: `v_cvt_pk_f16_f32 v39, v2, v3` directly in front of `v_mfma_f32_32x32x16_f16 v[0:15], .., .., 0` gave
// pack(0, 0) -- the MFMA's result -- for every lane; the bf16 kernel packs with other instructions and was right.)
// Variants: the VALU instruction (cvt_pk_f16_f32 | add_f32 | pack_b32_f16 | perm_b32), 0 .. 3 wait states in between, with and without
// another MFMA in flight.  Result (MI355X): every variant is right, three repetitions each -- the sequence alone is no hazard; what went
// wrong in the kernel is not reproduced here (it disappeared with `s_nop 1` in front of the MFMA and scheduling barriers between the flush's steps).
//      hipcc --offload-arch=gfx950 -O3 -o mfma_war_probe mfma_war_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v40","v41"
#define SET "v_mov_b32 v0, %2\n\tv_mov_b32 v1, %3\n\tv_mov_b32 v2, %2\n\tv_mov_b32 v3, %3\n\ts_nop 7\n\t"
#define MF "v_mfma_f32_32x32x16_f16 v[0:15], %4, %5, 0\n\t"
#define BUSYM "v_mfma_f32_32x32x16_f16 v[16:31], %4, %5, 0\n\t"
#define TAILS "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\tv_mov_b32 %0, v40\n\tv_mov_b32 %1, v41"
#define OP0 "v_cvt_pk_f16_f32 v40, v0, v1\n\tv_cvt_pk_f16_f32 v41, v2, v3\n\t"
#define OP1 "v_add_f32 v40, v0, v1\n\tv_add_f32 v41, v2, v3\n\t"
#define OP2 "v_pack_b32_f16 v40, v0, v1\n\tv_pack_b32_f16 v41, v2, v3\n\t"
#define OP3 "v_perm_b32 v40, v0, v1, v1\n\tv_perm_b32 v41, v2, v3, v3\n\t"
#define N0 ""
#define N1 "s_nop 0\n\t"
#define N2 "s_nop 1\n\t"
#define N4 "s_nop 3\n\t"

#define VARIANT(name, BUSY, OP, NOP)                                                                      \
    __global__ void name(unsigned* out)                                                                   \
    {                                                                                                     \
        const int lane = threadIdx.x;                                                                     \
        float x = 1.0f + lane, y = 100.0f + lane;                                                         \
        i32x4 A = {0x3c003c00, 0x3c003c00, 0x3c003c00, 0x3c003c00}, B = A;                                \
        unsigned r0, r1;                                                                                  \
        asm volatile(SET BUSY OP NOP MF TAILS : "=v"(r0), "=v"(r1) : "v"(x), "v"(y), "v"(A), "v"(B) : CLOB); \
        out[lane] = r0; out[64 + lane] = r1;                                                              \
    }
VARIANT(k_cvt_0, "", OP0, N0) VARIANT(k_cvt_1, "", OP0, N1) VARIANT(k_cvt_2, "", OP0, N2) VARIANT(k_cvt_4, "", OP0, N4)
VARIANT(k_cvt_b0, BUSYM, OP0, N0) VARIANT(k_cvt_b1, BUSYM, OP0, N1) VARIANT(k_cvt_b2, BUSYM, OP0, N2) VARIANT(k_cvt_b4, BUSYM, OP0, N4)
VARIANT(k_add_0, "", OP1, N0) VARIANT(k_add_b0, BUSYM, OP1, N0)
VARIANT(k_pack_0, "", OP2, N0) VARIANT(k_pack_b0, BUSYM, OP2, N0)
VARIANT(k_perm_0, "", OP3, N0) VARIANT(k_perm_b0, BUSYM, OP3, N0)
// reference: the VALU instructions alone
VARIANT(k_cvt_ref, "", OP0, "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t")
VARIANT(k_add_ref, "", OP1, "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t")
VARIANT(k_pack_ref, "", OP2, "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t")
VARIANT(k_perm_ref, "", OP3, "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t")

int main()
{
    unsigned *d, h[128], ref[4][128];
    (void)hipMalloc(&d, sizeof(h));
    typedef void (*K)(unsigned*);
    struct { const char* name; K k; int ref; } v[] = {
        {"cvt ref", k_cvt_ref, -1}, {"add ref", k_add_ref, -2}, {"pack ref", k_pack_ref, -3}, {"perm ref", k_perm_ref, -4},
        {"cvt_pk_f16_f32, 0 wait states", k_cvt_0, 0}, {"cvt_pk_f16_f32, 1", k_cvt_1, 0}, {"cvt_pk_f16_f32, 2", k_cvt_2, 0}, {"cvt_pk_f16_f32, 4", k_cvt_4, 0},
        {"cvt_pk_f16_f32 behind an MFMA, 0", k_cvt_b0, 0}, {"cvt_pk_f16_f32 behind an MFMA, 1", k_cvt_b1, 0}, {"cvt_pk_f16_f32 behind an MFMA, 2", k_cvt_b2, 0},
        {"cvt_pk_f16_f32 behind an MFMA, 4", k_cvt_b4, 0},
        {"add_f32, 0", k_add_0, 1}, {"add_f32 behind an MFMA, 0", k_add_b0, 1}, {"pack_b32_f16, 0", k_pack_0, 2}, {"pack_b32_f16 behind an MFMA, 0", k_pack_b0, 2},
        {"perm_b32, 0", k_perm_0, 3}, {"perm_b32 behind an MFMA, 0", k_perm_b0, 3}};
    for (auto& e : v) {
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipMemset(d, 0xff, sizeof(h));
            hipLaunchKernelGGL(e.k, dim3(1), dim3(64), 0, 0, d);
            (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            if (e.ref < 0) { memcpy(ref[-e.ref - 1], h, sizeof(h)); break; }
            int bad0 = 0, bad1 = 0;
            for (int i = 0; i < 64; ++i) { bad0 += h[i] != ref[e.ref][i]; bad1 += h[64 + i] != ref[e.ref][64 + i]; }
            printf("%-40s rep %d: first result %2d lanes wrong, second (directly in front of the MFMA) %2d lanes wrong   lane 5: %08x / %08x (ref %08x)\n",
                   e.name, rep, bad0, bad1, h[5], h[64 + 5], ref[e.ref][64 + 5]);
        }
    }
    return 0;
}
