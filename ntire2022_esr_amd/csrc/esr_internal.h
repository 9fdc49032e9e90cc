// shared by the translation units of libesr_hip.so (not part of the public ABI)
#pragma once
#include <hip/hip_runtime.h>
#include "esr_hip.h"

void esr_set_err(const char* what, hipError_t e);
int esr_check_launch(const char* what);
// Records the device symbol of the launch that follows, as rocprofv3 prints it -- only while esr_run_ops_profiled runs (a bench /
// profile leg names its kernels by the symbol that really ran: esr_prof_kernel_symbol).  printf-style.
void esr_note_kernel(const char* fmt, ...);
static inline const char* esr_tf(bool b) { return b ? "true" : "false"; }
static inline int esr_round_up(int v, int m) { return (v + m - 1) / m * m; }

// esr_s16.hip: NHWC convolution on 16-bit storage (called by esr_conv2d_f32 when d->storage != ESR_STORE_F32)
int esr_conv2d_s16(const esr_conv_desc* d, void* hip_stream);
int esr_s16_block_waves(const esr_conv_desc* d);      // 4: two 4-wave blocks per CU (16 x 16 tiles), 8: one 8-wave block (16 x 32)

// esr_wino.hip: Winograd F(2x2, 3x3) fp32 convolution (called by esr_conv2d_f32 when d->wino_wpacked is set and the shape qualifies)
int esr_conv2d_wino(const esr_conv_desc* d, void* hip_stream);

// esr_graph.hip: while esr_graph_create captures an op list, a launcher whose kernel can read the network input or write the network
// output reports the launch it has JUST enqueued on `st`: the pointer values it passed and their byte offsets inside the kernel's first
// argument (offsetof in the parameter struct; 0 for a leading scalar pointer argument).  No-op outside a capture.
void esr_graph_note_io(hipStream_t st, const void* in_ptr, size_t in_off, const void* out_ptr, size_t out_off);

#ifdef __HIPCC__
// gfx950 erratum found in round 4 (LAB_NOTES.md "packed fp32 op_sel"; tools/dbg/pk_opsel_probe.hip reproduces it in isolation): a
// packed fp32 VALU instruction whose op_sel makes a result half read the HIGH dword of a 64-bit source pair (v_pk_mul_f32 ...
// op_sel:[0,1]) returns 0 in lanes 48..63 when a wave of ANOTHER kernel issues MFMAs on the same SIMD -- forwards overlapping on
// several HIP streams differed from serial ones.  hipcc picks that encoding when a scalar factor happens to live in the odd
// register of a pair (an .y / .w element of a loaded vector, or lx next to ly in a struct).  esr_lone() gives the factor a register
// of its own, so the broadcast is encoded with op_sel_hi (both halves read the LOW dword: the form that never failed);
// tools/lint_isa.py (a CPU test) fails the build if the bad encoding shows up in any kernel of the library.
__device__ __forceinline__ float esr_lone(float v)
{
    asm("" : "+v"(v));
    return v;
}

// GELU of the 16-bit storage modes (the scalar definition conv_s16_kernel's packed version follows bit for bit; accuracy and
// derivation: esr_s16.hip, tools/fit_gelu.py)
__device__ __forceinline__ float esr_gelu16(float x)
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
#endif
