/*
 * esr_hip.h -- C ABI of libesr_hip.so, the MI355X (gfx950) implementation of the
 * NTIRE2022_ESR test_demo.py forward path.
 *
 * The reference has no FFI: its boundary is the torch.nn.Module protocol
 * (select_model test_demo.py:13-341, forward test_demo.py:364-391).  This header
 * is what a from-scratch binding of that path calls underneath `model(x)`:
 * plain pointers and sizes, an explicit HIP stream, status codes, no allocation,
 * no host synchronisation, no torch types.  The Python host side
 * (ntire2022_esr_amd/) binds it with ctypes; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Activation tensors between ops are NHWC fp32 with an explicit channel pitch
 * (floats per pixel) and a first-channel offset, so torch.split / torch.cat on
 * the channel axis (models/basicblock.py:260-264, models/rfdn_baseline/block.py:163,
 * RFDN.py:36) are free: producers store straight into a slice of the consumer's
 * buffer.  The network input/output stay NCHW fp32 exactly as `model(x)` sees them.
 *
 * All entry points return ESR_OK (0) or a negative esr_status; none throws.
 */
#ifndef ESR_HIP_H
#define ESR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESR_ABI_VERSION 12

typedef enum esr_status {
    ESR_OK = 0,
    ESR_ERR_BAD_ARG = -1,      /* NULL pointer, non-positive size, unsupported combination */
    ESR_ERR_UNSUPPORTED = -2,  /* shape outside what the kernels implement (e.g. cout > 64) */
    ESR_ERR_LAUNCH = -3,       /* hipLaunchKernel reported an error (see esr_last_hip_error) */
    ESR_ERR_TOO_SMALL = -4     /* ESA needs H,W >= 15 (models/rfdn_baseline/block.py:119-120) */
} esr_status;

/* nn.LeakyReLU(0.05) models/basicblock.py:77 | nn.ReLU rfdn_baseline/block.py:115 |
 * nn.GELU() (erf form) models/team18_bsrn.py:107 */
typedef enum esr_act { ESR_ACT_NONE = 0, ESR_ACT_LRELU = 1, ESR_ACT_RELU = 2, ESR_ACT_GELU = 3 } esr_act;

/* where the residual enters relative to the activation:
 *   PRE : act(conv(x) + r)   RFDB  rfdn_baseline/block.py:151 ; ShortcutBlock basicblock.py:197-199 (act none)
 *   POST: act(conv(x)) + r   RLFB  team04_rlfn.py:117-119 */
typedef enum esr_res { ESR_RES_NONE = 0, ESR_RES_PRE_ACT = 1, ESR_RES_POST_ACT = 2 } esr_res;

/* arithmetic of the matrix products (accumulation is always fp32):
 *   F32          exact fp32 MFMA (v_mfma_f32_16x16x4_f32); storage must be ESR_STORE_F32 (or the NCHW head of a 16-bit network)
 *   BF16 / F16   v_mfma_f32_16x16x32_{bf16,f16} on 16-bit operands; goes with the SAME 16-bit storage type (esr_storage):
 *                the stored activation IS the MFMA operand, weights come from esr_pack_conv_s16 */
typedef enum esr_compute { ESR_COMPUTE_F32 = 0, ESR_COMPUTE_BF16 = 1, ESR_COMPUTE_F16 = 2 } esr_compute;

/* ABI v4 -- element type of the FULL-RESOLUTION NHWC activation views of an op (in, res, out0, out1, ESA x / y / c1,
 * BSConvU in / res / out / d_out).  pitch / coff of those views count ELEMENTS.  16-bit storage halves the HBM traffic of
 * the memory-bound layers (BASELINE.json configs [2]-[4]); arithmetic stays fp32-accumulate, values are rounded (RNE) once,
 * when stored.  The network input / output (NCHW) and the ESA low-resolution maps (pitch ESR_ESA_FP) are always fp32.
 * For esr_conv2d_f32 on NHWC input, storage BF16 / F16 requires compute BF16 / F16 and weights from esr_pack_conv_s16;
 * views then need pitch and coff in multiples of 8 elements (16 bytes) and in.coff + round_up(cin, 8) <= in.pitch.
 * TIGHT PITCH (round 6): the K loop walks 16-channel chunks; where the pixel holds round_up(cin, 8) channels but not round_up(cin, 16) -- nf = 50
 * at pitch 56 -- the last chunk's second half is the first 16 bytes of the NEXT pixel (zeros behind an image's last pixel) and meets the weight rows
 * esr_pack_conv_s16 leaves zero.  Values must be finite.  Segmented inputs (in_seg_*): in.coff + 16 in_seg_chunks - 8 <= in.pitch likewise. */
typedef enum esr_storage { ESR_STORE_F32 = 0, ESR_STORE_BF16 = 1, ESR_STORE_F16 = 2 } esr_storage;

typedef enum esr_layout {
    ESR_NHWC = 0,            /* [n][h][w][pitch] fp32, channel slice [coff, coff+c) */
    ESR_NCHW_IN = 1,         /* network input  N x cin x H x W contiguous (uint2tensor4 output, utils_image.py:190-193) */
    ESR_NCHW_SHUFFLE4 = 2    /* network output N x cout/16 x 4H x 4W: conv + nn.PixelShuffle(4) fused
                                (basicblock.py:446-449,84-85): out[n,c,4h+i,4w+j] = conv[n,16c+4i+j,h,w] */
} esr_layout;

/* One NHWC view: device pointer + channel pitch + first channel. */
typedef struct esr_view {
    void*   ptr;
    int32_t pitch;   /* floats per pixel (>= coff + channels, multiple of 4) */
    int32_t coff;    /* first channel of the slice (multiple of 4) */
} esr_view;

/*
 * esr_conv2d_f32 -- nn.Conv2d(k in {1,3}, stride 1, padding (k-1)/2, bias) + fused epilogue.
 * Replaces every full-resolution conv call on the path: basicblock.conv 'C'/'CL'
 * (models/basicblock.py:61-98), rfdn conv_layer (models/rfdn_baseline/block.py:7-10),
 * rlfn conv_layer (models/team04_rlfn.py:7-10) and the 1x1 "fusion" convs that follow a
 * torch.cat.  fp32 in, fp32 accumulate on v_mfma_f32_16x16x4_f32 (exact fmaf chain), fp32 out.
 *
 *   y = epilogue(bias + sum_{c,i,j} W[o,c,i,j] * x[n, h+i-p, w+j-p, c])
 *   channels [0, split) of the result go to out0, channels [split, cout) to out1
 *   (torch.split(t,(d,r),dim=1) basicblock.py:260-262); split == cout -> single output.
 */
typedef struct esr_conv_desc {
    int32_t n, h, w;            /* batch, input height, width (output spatial = same; x4 for SHUFFLE4) */
    int32_t cin, cout;          /* logical channels; cin <= 512, cout <= 64 */
    int32_t ksize;              /* 1 or 3 */
    int32_t in_layout;          /* ESR_NHWC | ESR_NCHW_IN (cin <= 4 only) */
    int32_t out_layout;         /* ESR_NHWC | ESR_NCHW_SHUFFLE4 (cout == 48 -> 3 x 4H x 4W) */
    int32_t act;                /* esr_act */
    float   slope;              /* LeakyReLU negative slope */
    int32_t res_mode;           /* esr_res */
    int32_t split;              /* multiple of 4, 0 < split <= cout */
    esr_view in;                /* pitch/coff ignored for ESR_NCHW_IN */
    esr_view res;               /* residual, NHWC, `cout` channels from coff; unused if ESR_RES_NONE */
    esr_view out0, out1;        /* out0.ptr is the NCHW tensor for ESR_NCHW_SHUFFLE4 */
    const void* wpacked;        /* device pointer to esr_pack_conv_f32 (16-bit storage: esr_pack_conv_s16) output */
    int32_t compute;            /* esr_compute; 0 = fp32 */
    int32_t storage;            /* esr_storage of the NHWC views (ABI v4; 0 = fp32) */
    /* ABI v3 -- optional fused 1x1 tail: the 3x3 result (cout <= 16, after `tail_mid_act`) never goes to memory; it is
     * the LAST 16 input channels of a 1x1 convolution whose first `tail_cat_c` input channels are read from `tail_cat`:
     *     y = W1x1 . concat(tail_cat[0:tail_cat_c), mid_act(conv3x3(in))) + b1x1
     * and act / res / res_mode / split / out0 / out1 above then describe the epilogue of THAT 1x1 (`tail_cout` output
     * channels).  This is IMDBlock's conv4 -> cat -> conv1x1 -> + x (models/basicblock.py:263-265) in one kernel.
     * tail_wpacked = esr_pack_conv_f32 of the 1x1 (cin = tail_cat_c + 16, ksize 1); NULL = no tail.
     * Supported: ksize 3, NHWC in/out, cout <= 16, tail_cat_c a multiple of 16, 48 < tail_cout <= 64, fp32. */
    const void* tail_wpacked;
    esr_view tail_cat;
    int32_t tail_cat_c;
    int32_t tail_cout;
    int32_t tail_mid_act;       /* esr_act applied to the 3x3 result before the 1x1 (slope = `slope`) */
    /* ABI v12 -- the tail in 16-bit storage (csrc/esr_c64m.hip: rfdb_tail_kernel): RFDB's c4 -> cat(d1, d2, d3, r4) -> c5 -> esa.conv1
     * (models/rfdn_baseline/block.py:161-164, :117) in ONE launch.  The 3x3 (64 physical input channels, cout <= 32) is activated by
     * tail_mid_act and ROUNDED to the storage type (exactly the tensor the separate launches store) but never stored; tail_cat are
     * three dense tensors of pitch 32 (tail_cat_c = 96 physical slots) that lie tail_seg_stride16 * 16 bytes apart; the 1x1 (tail_cout
     * <= 64, no activation, weights from esr_pack_tail_s16) is stored to out0 AND, unrounded, feeds the post 1x1 (post_wpacked from
     * esr_pack_post_s16, post_cout <= 16, post_out).  The same with 48 physical input channels, border_bias (the merged BSConvU's table,
     * added to the 3x3 result in front of the activation) and tail_mid_act = ESR_ACT_GELU is ESDB's tail (models/team18_bsrn.py:165-171,
     * :109; 33 <= tail_cout <= 48).  esr_conv_tail_supported(d) tells whether a descriptor runs this way. */
    int32_t tail_seg_stride16;
    /* ABI v3 -- optional "post" 1x1: besides its own output, the conv's ACTIVATED result x' feeds a 1x1 whose output goes to
     * `post_out`:  post_out = post_act(W_p . x' + b_p).  RFDB: r_j = act(c{j}_r(..)), d_{j+1} = lrelu(c{j+1}_d(r_j))
     * (models/rfdn_baseline/block.py:150-160) -- the distillation conv is evaluated by the kernel that produces its input.
     * post_wpacked = esr_pack_conv_f32 of the 1x1 (cin = this conv's cout, ksize 1); NULL = none.  Supported: ksize 3, NHWC
     * in/out, 48 < cout <= 64, post_cout <= 32, fp32, residual none or identical to the input (pre-activation). */
    const void* post_wpacked;
    esr_view post_out;
    int32_t post_cout;
    int32_t post_act;
    /* ABI v4 -- 16-bit storage: the post 1x1 is applied to the conv's FINISHED result (activation and residual included, fp32,
     * not rounded), post_wpacked comes from esr_pack_post_s16, out0.ptr == NULL means the conv's own result is not stored (only
     * the chain consumes it), and a second 1x1 can be chained on the first one's result (no activation):
     *     x' = epilogue(conv(in));  post_out = post_act(W_p . x' + b_p);  post2_out = W_q . post_out_fp32 + b_q
     * RLFB: c3_r -> c5 -> esa.conv1 (models/team04_rlfn.py:117-121, 76).  esr_conv_post_supported() tells whether a descriptor's
     * chain runs fused (shape variants exist for the in-scope networks); otherwise build separate ops. */
    const void* post2_wpacked;
    esr_view post2_out;
    int32_t post2_cout;
    int32_t reserved3;
    /* ABI v5 -- position-dependent bias at the image border (16-bit storage, NHWC, no post chain; NULL = none): fp32 table
     * [16][round_up(cout, 16)], row m is ADDED (before residual / activation) to the pixels whose outside-mask is m:
     * bit 0: x == 0, bit 1: x == w-1, bit 2: y == 0, bit 3: y == h-1 (row 0 is never read).  This is what lets BSConvU
     * (models/team18_bsrn.py:82-88: depthwise 3x3 over the ZERO-PADDED pointwise output pw(x) + b) run as ONE dense 3x3 with
     * the merged weights dw[c,tap] * pw[c,k]: in the interior the pointwise bias contributes b[c] * sum_tap dw[c,tap] (folded
     * into the bias), at the border the taps that fall outside the image contribute nothing -- row m holds
     * -b[c] * sum over the taps outside for mask m of dw[c,tap].  The table must be 16-byte aligned (it is staged by 16-byte
     * LDS-DMA pieces like the packed weights; ESR_ERR_BAD_ARG otherwise). */
    const float* border_bias;
    /* ABI v5 -- segmented (planar) input for the 1x1 over a channel concat (16-bit storage): the input channels come from
     * `cin_phys / (16 * in_seg_chunks)` tensors of identical geometry (pitch / coff as given by `in`) that lie in_seg_stride bytes
     * apart; 16-channel chunk c is chunk c % in_seg_chunks of segment c / in_seg_chunks.  in_seg_stride == 0: one tensor.
     * Why: a distilled slice written into a shared [.., 4 x dc] concat buffer is a partial-line store (48 of 192 bytes per
     * pixel for BSRN) and costs 2.3x the dense store (tools/dbg/s16_1x1_probe.py); with one dense tensor per slice the
     * producers write whole lines and torch.cat still never happens -- the consumer walks the segments. */
    int64_t in_seg_stride;
    int32_t in_seg_chunks;
    /* ABI v6 -- channel-blocked fp32 tensors ("NC/8HW8"): bit 0 (ESR_BLOCKED_IN): `in`, bit 1 (ESR_BLOCKED_OUT1): `out1` is stored as
     * [n][pitch / 8][h][w][8] instead of [n][h][w][pitch] (pitch and coff multiples of 8): element (n, y, x, c) of the view lies at
     * ptr[((n * pitch / 8 + (coff + c) / 8) * h * w + y * w + x) * 8 + (coff + c) % 8].
     * Why: conv_f32_kernel and imdb_tail_kernel stage the input one 8-channel K chunk at a time, so a 128-byte line of an
     * [.., 48]-channel NHWC pixel is touched by four stages microseconds apart and is re-fetched when the L2 has dropped it in
     * between (PMC: 1.33x the algorithmic reads for the memory-bound IMDB tail); in the blocked layout a stage reads whole lines
     * and every line exactly once.  Supported: ESR_BLOCKED_OUT1 for fp32 NHWC convolutions with a split store (the "remaining"
     * channels of IMDBlock's conv3), ESR_BLOCKED_IN for the fused IMDB tail at the network's shape (imdb_tail_kernel) and -- ABI v9 --
     * for every Winograd descriptor (both kernels read a chunk as a plane); ABI v9 also: bit 2 (ESR_BLOCKED_OUT0) `out0` and bit 3
     * (ESR_BLOCKED_RES) `res` of the fused IMDB tail (IMDBlock's input / output x: the 64-input-channel Winograd layers then read
     * whole lines too); any other use returns ESR_ERR_UNSUPPORTED. */
    int32_t blocked8;
    /* ABI v10 -- hi + lo tensors (bf16 storage, plain 3x3 with 33..64 output channels): a value is kept as the SUM of two bf16 numbers in two
     * dense NHWC tensors of the same shape -- the high parts bf16(v) where the view points, the low parts bf16(v - hi) `hilo_stride` bytes
     * behind them (the field at the end of this struct; a multiple of 16) -- 16 significant bits instead of 8.  Bits: ESR_HILO_IN -- `in` is
     * such a pair (the K loop runs over both tensors against the same weights: w (hi + lo) = w hi + w lo); ESR_HILO_RES -- `res` is one (both
     * are added in fp32); ESR_HILO_OUT -- `out0` becomes one (NHWC only; `out1` unused).  For the long skip of the x4 networks,
     * `upsampler(LR_conv(body) + fea)` (team04_rlfn.py:149-150; rfdn_baseline/RFDN.py:44-47): the image itself travels through `fea` and
     * `out_lr`, and two bf16 roundings of it cost 0.03-0.09 dB on near-detail-free content (LAB_NOTES.md 9.4).  ESR_HILO_OUT alone may carry one
     * post 1x1 of two output tiles (`post_*`) and a border table -- the head of RFDN / BSRN with block 1's first distillation conv in its
     * epilogue.  Anything else returns ESR_ERR_UNSUPPORTED. */
    int32_t hilo;
    /* ABI v7 -- Winograd F(2x2, 3x3) weights (esr_pack_wino_f32) of the SAME convolution; NULL = none.  When set and
     * esr_wino_supported(d) (fp32 storage and compute, ksize 3, NHWC in / NHWC out, round_up(cin, 8) / 8 even and >= 4, no tail /
     * post / border table / segmented or blocked input, a split store only at a multiple of 16 channels), esr_conv2d_f32 runs
     * wino_f32_kernel: 16 transformed-domain products per 2x2 output pixels and (cin, cout) instead of 36 -- 2.25x fewer MFMAs,
     * fp32 throughout (its rounding differs from the direct sum's by ~1e-6 relative, inside the 2e-5 budget of SURVEY 8c).
     * Otherwise the descriptor takes conv_f32_kernel with `wpacked`, which must always be valid. */
    const void* wino_wpacked;
    int64_t hilo_stride;        /* ABI v10: bytes from the high-part tensor to the low-part tensor of every hi + lo pair of this descriptor (`hilo`) */
} esr_conv_desc;
#define ESR_BLOCKED_IN   1
#define ESR_BLOCKED_OUT1 2
#define ESR_BLOCKED_OUT0 4
#define ESR_BLOCKED_RES  8
#define ESR_HILO_IN 1
#define ESR_HILO_RES 2
#define ESR_HILO_OUT 4

/* Host-side weight packer (the K10 "weight packer" of SURVEY 7.2): OIHW fp32 (the layout of every
 * Conv2d in the reference state_dicts; nn.Linear [out,in] is the k=1 case) + bias -> the MFMA-tiled,
 * zero-padded blob esr_conv2d_f32 consumes.  `cin_map` (may be NULL = identity) gives, for each of
 * `cin_phys` physical input-channel slots, the logical input channel it carries or -1 for a padding
 * slot: that is how padded concat buffers are described.  Pure CPU code: callable without a GPU. */
size_t esr_packed_conv_bytes(int cin_phys, int cout, int ksize);
/* ABI v12 -- weights of the 1x1 of a 16-bit-storage tail (esr_conv_desc.tail_*): w [cout][nseg * seg_c + mid_c] fp32 (the reference's
 * c5.weight: inputs in concat order, the 3x3's mid_c channels last) -> fragment images for v_mfma_f32_32x32x16 (high and low 16-bit parts)
 * + fp32 bias.  Shapes: nseg == 3, seg_c <= 32, mid_c <= 32, cout <= 64.  Host-side, no GPU needed. */
size_t esr_packed_tail_s16_bytes(int nseg, int seg_c, int mid_c, int cout);
int    esr_pack_tail_s16(const float* w, const float* bias, int nseg, int seg_c, int mid_c, int cout, int compute, void* out, size_t out_bytes);
int    esr_conv_tail_supported(const esr_conv_desc* d);       /* 1: esr_conv2d_f32 runs this 16-bit descriptor's tail fused */
int    esr_pack_conv_f32(const float* w_oihw, const float* bias, int cin, int cout, int ksize,
                         const int32_t* cin_map, int cin_phys, void* out, size_t out_bytes);
/* inverse, for tests: recovers OIHW + bias from a packed blob */
int    esr_unpack_conv_f32(const void* packed, size_t bytes, int cin, int cout, int ksize,
                           const int32_t* cin_map, int cin_phys, float* w_oihw, float* bias);

/* 16-bit-STORAGE weights (ESR_STORE_BF16 / ESR_STORE_F16 descriptors, ksize 1 or 3): K chunks of 16 input channels,
 * [chunk][tap pair][16-cout tile][lane][8] 16-bit values for v_mfma_f32_16x16x32_{bf16,f16}, followed by the fp32 bias.
 * 3x3: the 9 taps of every (cout, cin) filter are rounded with error diffusion (tap k absorbs the rounding error of tap
 * k-1), so the filter's DC gain keeps fp32 accuracy; 1x1: the second tap slot carries the rounding residual (w = hi + lo).
 * esr_unpack_conv_s16 returns the EFFECTIVE fp32 weights the kernel multiplies by (tests). */
size_t esr_packed_conv_s16_bytes(int cin_phys, int cout, int ksize);
int    esr_pack_conv_s16(const float* w_oihw, const float* bias, int cin, int cout, int ksize, const int32_t* cin_map,
                         int cin_phys, int compute, void* out, size_t out_bytes);
int    esr_unpack_conv_s16(const void* packed, size_t bytes, int cin, int cout, int ksize, const int32_t* cin_map,
                           int cin_phys, int compute, float* w_oihw, float* bias);

/* post-chain 1x1 weights for 16-bit storage (esr_conv_desc.post_wpacked / post2_wpacked): [cout][cin] fp32 -> MFMA images of the
 * high and low 16-bit parts + fp32 bias */
size_t esr_packed_post_s16_bytes(int cin, int cout);
int    esr_pack_post_s16(const float* w_oi, const float* bias, int cin, int cout, int compute, void* out, size_t out_bytes);
int    esr_conv_post_supported(const esr_conv_desc* d);   /* 1: the descriptor's post chain has a fused kernel that fits */

/* Winograd F(2x2, 3x3) weight packer (host C++, no GPU needed): OIHW fp32 3x3 weights -> U = G g G^T (computed in fp64, rounded
 * once) in the MFMA A-operand order wino_f32_kernel stages: [cin chunk of 8][cout half of 32][position 16][lane 64][ct0 s0, ct0 s1,
 * ct1 s0, ct1 s1] floats, followed by round_up(cout, 32) bias floats.  cin_map / cin_phys as for esr_pack_conv_f32.
 * esr_unpack_wino_f32 (tests) returns U as [cout][cin][16] and the bias. */
size_t esr_packed_wino_bytes(int cin_phys, int cout);
int    esr_pack_wino_f32(const float* w_oihw, const float* bias, int cin, int cout, const int32_t* cin_map, int cin_phys,
                         void* out, size_t out_bytes);
int    esr_unpack_wino_f32(const void* packed, size_t bytes, int cin, int cout, const int32_t* cin_map, int cin_phys,
                           float* u_oc16, float* bias);
int    esr_wino_supported(const esr_conv_desc* d);   /* 1: a descriptor of this shape runs on wino_f32_kernel when wino_wpacked is set */

int esr_conv2d_f32(const esr_conv_desc* d, void* hip_stream);
/* Diagnostics: waves per block of the conv_f32_kernel variant esr_conv2d_f32 launches for `d` (4 = 16x16-pixel tiles,
 * two blocks per CU; 8 = 16x32-pixel tiles, one block per CU, used for large 3x3 launches; likewise for conv_s16_kernel;
 * 1 = conv48r_kernel: 16-bit storage, a 3x3 over 48 physical input channels with 2 or 3 output tiles and at least 256 tiles of
 * 16x32 -- one 4-wave block per CU, one wave per SIMD, weights in registers), 0 for a NULL / empty descriptor.  Lets a profiler
 * name the device symbol that ran (the reference has torch.profiler for that). */
int esr_conv_block_waves(const esr_conv_desc* d);

/*
 * ESA (enhanced spatial attention) -- models/rfdn_baseline/block.py:103-129, models/team04_rlfn.py:62-89,
 * models/team18_bsrn.py:91-122.  The f = n_feats/4 (12) or esa_channels (16) wide maps live in NHWC buffers of
 * pitch ESR_ESA_FP = 16 whose pad channels are zero.  Small dense weights use the plain layout written by
 * esr_pack_dense_f32: [tap][cin_p][cout_p] floats followed by bias[cout_p].
 *
 *   esr_conv3x3s2_f32   conv2: nn.Conv2d(f, f, 3, stride 2, padding 0)   H2 = (H-3)/2+1  (block.py:110,119)
 *   esr_maxpool7s3_f32  F.max_pool2d(kernel_size=7, stride=3)             H3 = (H2-7)/3+1 (block.py:120)
 *   esr_esa_apply_f32   y = x * sigmoid(conv4(bilinear(c3 -> HxW, align_corners=False) + conv_f(c1_)))
 *                       (block.py:124-129): one full-resolution pass, x read once, y written once.
 */
#define ESR_ESA_FP 16

size_t esr_packed_dense_bytes(int cin_p, int cout_p, int ksize);
int    esr_pack_dense_f32(const float* w_oihw, const float* bias, int cin, int cout, int ksize,
                          int cin_p, int cout_p, void* out, size_t out_bytes);

/*
 * A 1x1 convolution riding in esr_esa_apply_f32's launch (ABI v8, 16-bit storage only): post[0] is evaluated on the apply result
 * y AS STORED (the 16-bit rounded values: exactly what a separate 1x1 launch would read back), post[1] on post[0]'s fp32 result
 * (16-bit high + low parts, as esr_conv_desc.post_* does).  v = W.in + b (+ res, ESR_RES_PRE_ACT) -> act -> 16-bit store to `out`.
 *   RFDN  : y = the RFDB's output, post[0] = the NEXT block's c1_d + LeakyReLU (rfdn_baseline/block.py:150)
 *   BSRN  : y = the ESA output (store_y = 0: nothing else reads it), post[0] = conv_out . cw + block input (team18_bsrn.py:169-172),
 *           post[1] = the next block's c1_d + GELU (:150)
 * Weights: esr_esa_desc.post_w, ONE esr_pack_apply_post blob for the chain (MFMA images of the weights' 16-bit high and low parts
 * + fp32 biases).
 */
typedef struct esr_esa_post {
    int32_t cout;               /* 0: unused */
    int32_t act;                /* esr_act */
    float   slope;
    int32_t res_mode;           /* ESR_RES_NONE | ESR_RES_PRE_ACT (post[0] only) */
    esr_view res;
    esr_view out;
} esr_esa_post;
int    esr_esa_apply_post_supported(int c, int cout0, int cout1);   /* 1: esr_esa_apply_f32 has a kernel for this chain (cout1 = 0: post[0] only) */
size_t esr_packed_apply_post_bytes(int cin, int cout0, int cout1, int storage);
/* w0: [cout0][cin] fp32 (+ b0[cout0] or NULL), w1: [cout1][cout0] (+ b1) or NULL with cout1 = 0 */
int    esr_pack_apply_post(const float* w0, const float* b0, const float* w1, const float* b1, int cin, int cout0, int cout1,
                           int storage, void* out, size_t out_bytes);

typedef struct esr_esa_desc {
    int32_t n, h, w;            /* full-resolution dims */
    int32_t c;                  /* n_feats (logical channels of x / y) */
    int32_t f;                  /* ESA width (<= 16) */
    int32_t h_lo, w_lo;         /* dims of the low-resolution map (source of the op) */
    int32_t storage;            /* esr_storage of the full-resolution views x / y / c1 (ABI v4; 0 = fp32) */
    esr_view x;                 /* apply: block input x ; conv3x3s2 / maxpool: source map (pitch 16) */
    esr_view y;                 /* destination */
    const void* c1;             /* apply: c1_ = conv1(x), [n*h*w][16] */
    const void* c3;             /* apply: low-res map [n][h_lo][w_lo][16] */
    const void* w0;             /* conv3x3s2: packed dense k=3 ; apply: conv_f packed dense k=1 (16 x 16) */
    const void* w1;             /* apply: conv4 packed dense k=1 (16 x round_up(c,4)) */
    const void* post_w;         /* apply, ABI v8: esr_pack_apply_post blob of post[0] (+ post[1]), or NULL */
    esr_esa_post post[2];
    int32_t skip_y;             /* apply: 1 = y is not stored (only the post chain consumes it) */
    int32_t reserved;
} esr_esa_desc;

/*
 * esr_esa_lowres_f32 -- ESA's whole low-resolution branch in two launches (ABI v7):
 *     c3 = layers(max_pool2d(conv2(c1_), 7, 3))
 * conv2 = nn.Conv2d(f, f, 3, stride 2, padding 0) (rfdn_baseline/block.py:110,119), the pooling (:120), then `n_layers` (1..3)
 * 3x3 / padding 1 layers on the pooled map: kind 0 = nn.Conv2d(f, f, 3, padding 1) (block.py:121-123 conv_max, conv3, conv3_;
 * team04_rlfn.py:81 conv3), kind 1 = BSConvU (pointwise nn.Linear + depthwise 3x3 over the zero-padded pointwise output,
 * team18_bsrn.py:113-116), each followed by `act` (esr_act).  Halo recompute instead of intermediate tensors: conv2's output and
 * the layers' intermediates never reach memory; only the pooled map does (`pooled`, caller-provided scratch [n][h3][w3][16] fp32).
 * x: the conv1 map [n][h][w][16] in `storage`; y: [n][h3][w3][16] fp32 with h2 = (h-3)/2+1, h3 = (h2-7)/3+1 (likewise w).
 * Weights: w_s2 and kind-0 layers esr_pack_dense_f32(k = 3, 16, 16); kind-1 layers w = esr_pack_dense_f32(k = 1, 16, 16) (pointwise),
 * w_dw = esr_pack_dw_f32.  Replaces esr_conv3x3s2_f32 + esr_maxpool7s3_f32 + 1..6 small esr_conv2d_f32 / esr_dwconv3x3_f32 launches
 * (which stay available): on one DIV2K image they were a third of a forward's launches for < 1 % of its arithmetic.
 */
#define ESR_ESA_MAX_LAYERS 3
typedef struct esr_esa_layer {
    int32_t kind;               /* 0: dense 3x3, 1: BSConvU */
    int32_t act;                /* esr_act behind the layer */
    const void* w;
    const void* w_dw;           /* kind 1 only */
} esr_esa_layer;
typedef struct esr_esa_lowres_desc {
    int32_t n, h, w;            /* full-resolution dims of the conv1 map */
    int32_t f;                  /* ESA width (<= 16) */
    int32_t storage;            /* esr_storage of x */
    int32_t n_layers;
    esr_view x;                 /* pitch 16, coff 0 */
    const void* w_s2;
    void* pooled;               /* scratch, n * h3 * w3 * 16 floats */
    void* y;                    /* result, n * h3 * w3 * 16 floats */
    esr_esa_layer layer[ESR_ESA_MAX_LAYERS];
} esr_esa_lowres_desc;
int esr_esa_lowres_f32(const esr_esa_lowres_desc* d, void* hip_stream);

int esr_conv3x3s2_f32(const esr_esa_desc* d, void* hip_stream);   /* x: [n][h][w][16] -> y: [n][h_lo][w_lo][16] */
int esr_maxpool7s3_f32(const esr_esa_desc* d, void* hip_stream);  /* x: [n][h][w][16] -> y: [n][h_lo][w_lo][16] */
int esr_esa_apply_f32(const esr_esa_desc* d, void* hip_stream);

/*
 * esr_dwconv3x3_f32 -- depthwise nn.Conv2d(C, C, 3, 1, 1, groups=C, padding_mode="zeros") + the same fused
 * epilogue as esr_conv2d_f32 (residual PRE/POST, activation).  The second half of BSConvU
 * (models/team18_bsrn.py:70-88).  Uses esr_conv_desc with cin == cout == C, ksize == 3, NHWC in/out views
 * (split/out1 unused) and `wpacked` = esr_pack_dw_f32 output: [tap][c_p] floats + bias[c_p], c_p = round_up(C,4).
 */
size_t esr_packed_dw_bytes(int c);
int    esr_pack_dw_f32(const float* w_c133, const float* bias, int c, void* out, size_t out_bytes);
int    esr_dwconv3x3_f32(const esr_conv_desc* d, void* hip_stream);

/*
 * Post-processing of run() on the device, so that only the uint8 image (for imsave) and one
 * scalar leave the GPU:
 *   esr_tensor2uint_u8   utils_image.tensor2uint (utils/utils_image.py:204-208): NCHW fp32 [C][H][W] (one image)
 *                        -> clamp(0, data_range) -> * 255/data_range (fp32) -> round half to even -> HWC uint8
 *   esr_sqerr_u8         sum over the border-cropped region of (a - b)^2 for two HWC uint8 images, as exact
 *                        uint64 (calculate_psnr :490-503 then is 20*log10(255/sqrt(sum/count)) on the host)
 */
int esr_tensor2uint_u8(const float* x_chw, uint8_t* y_hwc, int c, int h, int w, float data_range, void* hip_stream);
int esr_sqerr_u8(const uint8_t* a_hwc, const uint8_t* b_hwc, int h, int w, int c, int border,
                 unsigned long long* sum_out /* device, zeroed by the call */, void* hip_stream);
/*
 * ABI v9 (csrc/esr_metrics.hip):
 *   esr_tensor2uint_u8_chk  esr_tensor2uint_u8 that also ORs 1 into *nonfinite (device int, caller-zeroed) when the fp32 image holds an
 *                           Inf or NaN: the harness learns about overflowed 16-bit activations without a full-size isfinite pass.
 *   esr_ssim_u8             calculate_ssim (utils/utils_image.py:509-554) for two HWC uint8 images (c = 1 or 3) already on the device:
 *                           crop `border`, 11x11 Gaussian window (sigma 1.5) as two separable float64 passes, 'valid' region, the SSIM
 *                           map summed per block into partials[esr_ssim_partials(h, w, c, border)] (device doubles, every entry
 *                           written); the caller adds them up in order and divides by (h - 2 border - 10)(w - 2 border - 10) c.  For c = 3
 *                           that is the reference's result: its loop evaluates ssim() on the whole HxWx3 array three times
 *                           (:521-527).  ESR_ERR_BAD_ARG when the cropped image is smaller than the 11x11 window.
 */
int    esr_tensor2uint_u8_chk(const float* x_chw, uint8_t* y_hwc, int c, int h, int w, float data_range, int* nonfinite, void* hip_stream);
size_t esr_ssim_partials(int h, int w, int c, int border);
int    esr_ssim_u8(const uint8_t* a_hwc, const uint8_t* b_hwc, int h, int w, int c, int border, double* partials, size_t n_partials,
                   void* hip_stream);

/*
 * A forward pass is a flat list of ops executed in order on one stream: the native
 * replacement for test_demo.py's `model(img_lq)` (forward(), test_demo.py:364-367).
 * The Python host builds the list once per (model, N, H, W) and replays it.
 */
typedef enum esr_op_kind {
    ESR_OP_CONV = 0, ESR_OP_CONV3X3S2 = 1, ESR_OP_MAXPOOL7S3 = 2, ESR_OP_ESA_APPLY = 3, ESR_OP_DWCONV = 4,
    ESR_OP_BSCONV = 5,
    ESR_OP_PACK_INPUT = 6,      /* esr_pack_input_s16 on esr_op.conv (ABI v5) */
    ESR_OP_ESA_LOWRES = 7,      /* esr_esa_lowres_f32 on esr_op.lo (ABI v7) */
    ESR_OP_CONV_CHAIN = 8       /* esr_conv_chain_s16 on esr_op.chain (ABI v11) */
} esr_op_kind;

/*
 * esr_bsconv_f32 -- BSConvU (models/team18_bsrn.py:44-88) in one launch: y = act(dw3x3(pw1x1(x)) [+ res]), the
 * pointwise result never reaches memory (the depthwise conv zero-pads it, so out-of-image halo positions are 0, not
 * the pointwise bias).  Optionally the distillation 1x1 of the enclosing block (team18_bsrn.py:135-148: c{j}_d reads
 * the same input as c{j}_r) is evaluated on the same input tile: d_out = d_act(W_d . x + b_d).
 * pw_packed / d_packed = esr_pack_conv_f32(..., ksize = 1) blobs, dw_packed = esr_pack_dw_f32.  cin, c <= 64, d_cout <= 32.
 */
typedef struct esr_bsconv_desc {
    int32_t n, h, w;
    int32_t cin;                /* input channels of the pointwise conv */
    int32_t c;                  /* pointwise outputs = depthwise channels */
    int32_t act;                /* esr_act of the depthwise output */
    float   slope;
    int32_t res_mode;           /* esr_res, applied to the depthwise output */
    esr_view in, res, out;
    const void* pw_packed;
    const void* dw_packed;
    const void* d_packed;       /* NULL = no distillation conv */
    int32_t d_cout;
    int32_t d_act;
    esr_view d_out;
    int32_t storage;            /* esr_storage of in / res / out / d_out (ABI v4; 0 = fp32) */
    int32_t reserved;
} esr_bsconv_desc;

int esr_bsconv_f32(const esr_bsconv_desc* d, void* hip_stream);

/*
 * esr_channel_attention_f32 -- CALayer (models/basicblock.py:333-348) and the contrast-aware CCALayer
 * (models/team05_efdn/plainblock.py:106-122):
 *     y = x * sigmoid(W2 . relu(W1 . s + b1) + b2),   s[c] = mean_hw(x[c])            (contrast = 0, CA)
 *                                                     s[c] = std_hw(x[c]) + mean_hw(x[c])  (contrast = 1, CCA; population std)
 * One reduction pass (fp64 sums of x and x^2 per image and channel) and one scaling pass.  `layout` ESR_NHWC: x / y are NHWC
 * views (any esr_storage); ESR_NCHW_IN: x / y are plain NCHW fp32 tensors (pitch / coff ignored).  w1 / w2 are
 * esr_pack_dense_f32 blobs of the two 1x1 convolutions: w1 (cin = c, cout = cr, cin_p = c, cout_p = cr), w2 (cin = cr,
 * cout = c, cin_p = cr, cout_p = round_up(c, 4)).  `stats` is caller-provided scratch of n * 2 * round_up(c, 4) doubles
 * (the call zeroes it on the stream).  c <= 64, cr <= 16.
 */
typedef struct esr_ca_desc {
    int32_t n, h, w, c;
    int32_t cr;                 /* channel / reduction */
    int32_t contrast;           /* 0 = CALayer, 1 = CCALayer */
    int32_t layout;             /* ESR_NHWC | ESR_NCHW_IN */
    int32_t storage;            /* esr_storage of the NHWC views */
    esr_view x, y;
    const void* w1;
    const void* w2;
    void* stats;
} esr_ca_desc;

int esr_channel_attention_f32(const esr_ca_desc* d, void* hip_stream);

/*
 * ABI v11 -- esr_conv_chain_s16: a residual block's chain of 3x3 convolutions as ONE launch (16-bit storage):
 *     t_0 = in;  t_i = act(conv3x3_i(t_{i-1}))  (i = 1 .. n_layers - 1);
 *     u   = act(conv3x3_n(t_{n-1})) + in                        (res_mode ESR_RES_POST_ACT: the chain's input, after the activation)
 *     post_out = post_act(W_p . u + b_p);  post2_out = W_q . post_out_fp32 + b_q
 * RLFB.forward (models/team04_rlfn.py:109-122, 76): c1_r -> c2_r -> c3_r (+ x) -> c5 -> esa.conv1.  The intermediate tensors never reach
 * memory: they travel as 16-bit pixels (t_i, rounded exactly as esr_conv2d_f32 would store them) or fp32 (u, the post chain's input as in
 * esr_conv_desc.post_*) through LDS, one image row at a time, between the waves of a block that each own one layer and keep its weights in
 * registers (rlfb_chain_kernel, csrc/esr_chain.hip).  Results are bit-identical to the three esr_conv2d_f32 launches it replaces.
 * wpacked[i] = esr_pack_conv_s16 (ksize 3) of layer i, post_wpacked / post2_wpacked = esr_pack_post_s16.  esr_conv_chain_supported() tells
 * whether a descriptor's shape has a kernel (today: three layers over 33..48 channels, a first 1x1 of 33..48 and a second of <= 16 outputs).
 */
#define ESR_CHAIN_MAX_LAYERS 4
typedef struct esr_chain_desc {
    int32_t n, h, w;
    int32_t n_layers;
    int32_t cin, cmid, cout;    /* logical channels: in -> cmid -> ... -> cmid -> cout */
    int32_t act;                /* esr_act of every 3x3 layer */
    float   slope;
    int32_t res_mode;           /* esr_res of the LAST layer (residual = `in`) */
    int32_t storage;            /* esr_storage of in / post_out / post2_out */
    int32_t compute;            /* esr_compute: the storage's type */
    esr_view in;
    const void* wpacked[ESR_CHAIN_MAX_LAYERS];
    const void* post_wpacked;
    esr_view post_out;
    int32_t post_cout;
    int32_t post_act;
    const void* post2_wpacked;
    esr_view post2_out;
    int32_t post2_cout;
    int32_t reserved;
} esr_chain_desc;
int esr_conv_chain_supported(const esr_chain_desc* d);   /* 1: esr_conv_chain_s16 has a kernel for this shape */
int esr_conv_chain_s16(const esr_chain_desc* d, void* hip_stream);

typedef struct esr_op {
    int32_t kind;               /* esr_op_kind */
    int32_t reserved;
    esr_conv_desc conv;         /* ESR_OP_CONV, ESR_OP_DWCONV */
    esr_esa_desc esa;           /* the three ESA kinds */
    esr_bsconv_desc bs;         /* ESR_OP_BSCONV (ABI v3) */
    esr_esa_lowres_desc lo;     /* ESR_OP_ESA_LOWRES (ABI v7) */
    esr_chain_desc chain;       /* ESR_OP_CONV_CHAIN (ABI v11) */
} esr_op;

/* ABI v5 -- the network input for the 16-bit plans: NCHW fp32 [n, cin <= 4, h, w] (d->in.ptr) -> NHWC 16-bit (d->out0, pitch >= 16,
 * storage d->storage), 16 channels per pixel: [x_hi (cin) | x_lo (cin) | x_hi (cin) | 0 ..] with x = x_hi + x_lo in the storage type.
 * A 3x3 conv_s16 over these slots with the weights [w_hi | w_hi | w_lo] (w = w_hi + w_lo, dropping lo x lo) reproduces the fp32 head
 * convolution to ~2^-16 (bf16) / 2^-22 (fp16) relative -- at the 16-bit matrix rate instead of the fp32 one (conv_f32_kernel's NCHW
 * head: 0.21 ms at batch 32 for 226 MB).  Only n, h, w, cin, storage, in.ptr, out0 of the descriptor are read. */
int esr_pack_input_s16(const esr_conv_desc* d, void* hip_stream);

int esr_run_ops(const esr_op* ops, int n_ops, void* hip_stream);

/*
 * ABI v11 -- an op list as ONE HIP graph launch (csrc/esr_graph.hip).  test_demo.py runs one image per forward (:416-433); esr_run_ops then
 * costs the host 20-35 kernel launches (5-8 us each).  esr_graph_create captures the launches of `ops` once (nothing executes; `x` / `y` are
 * the network input / output pointers the ops currently hold: the ops that read ESR_NCHW_IN / esr_pack_input_s16's source and the op that
 * writes ESR_NCHW_SHUFFLE4); esr_graph_launch patches the two pointers in the captured kernel arguments when they changed and enqueues the
 * whole forward with one hipGraphLaunch on the caller's stream.  Same kernels, same order, same results as esr_run_ops; the graph must be
 * rebuilt when anything else in the op list changes (workspace address, weights, shape).  Every launch of a graph must be enqueued on a
 * stream that orders it behind the previous launch of the SAME graph (its ops share one workspace).
 */
typedef struct esr_graph esr_graph;
int  esr_graph_create(const esr_op* ops, int n_ops, const void* x, void* y, esr_graph** out);
int  esr_graph_launch(esr_graph* g, const void* x, void* y, void* hip_stream);
int  esr_graph_nodes(const esr_graph* g);      /* diagnostics: nodes of the captured graph */
void esr_graph_destroy(esr_graph* g);

/*
 * In-stream per-op timing (what test_demo.py:413-433 does per image with a CUDA event pair, done
 * per kernel): esr_run_ops_profiled records a HIP event pair around every op ON THE LAUNCH STREAM
 * and never synchronises; esr_prof_collect (call after the stream is idle) adds each op's elapsed
 * milliseconds over all recorded passes into ms_sum[n_ops] and returns the pass count.
 */
typedef struct esr_profiler esr_profiler;
int  esr_prof_create(int n_ops, int max_passes, esr_profiler** out);
int  esr_run_ops_profiled(const esr_op* ops, int n_ops, void* hip_stream, esr_profiler* prof);
int  esr_prof_collect(esr_profiler* prof, double* ms_sum, int n_ops, int* passes);
void esr_prof_destroy(esr_profiler* prof);
/* Device symbol(s) of op `op` as launched in the last profiled pass, in rocprofv3's spelling ("wino_f32_kernel<1, 0, 0>",
 * "conv_s16_kernel<4, 3, 8, true, false, 2, 0>", ...; "a + b" when an op was lowered to two launches): what a profiler matches its
 * kernel trace against (the reference has torch.profiler for that). */
int  esr_prof_kernel_symbol(esr_profiler* prof, int op, char* buf, size_t n);

/* ABI v11 -- measurement helpers of bench.py (csrc/esr_metrics.hip; they allocate and synchronise: NOT for a timed region).
 * esr_event_pair_ms = what per-launch hipEvent brackets add to a launch: n launches of a probe kernel timed by one pair around all of them
 * and by a pair around each, (sum - whole) / n -- the inflation of esr_run_ops_profiled's per-op numbers; esr_bw_probe = read + write
 * GB/s a plain copy kernel reaches inside `buf` (2 x bytes, device memory) when it repeats the pass `reps` times in one launch: the ceiling of a
 * streaming kernel at that working set (a tensor the previous launch wrote is Infinity-Cache resident at single-image sizes). */
int esr_event_pair_ms(void* hip_stream, int n, double* ms_out);
int esr_bw_probe(void* buf, size_t bytes, int reps, void* hip_stream, double* gbs_out);

/* diagnostics */
int         esr_abi_version(void);
const char* esr_last_hip_error(void);     /* thread-local, "" if none */
/* sizeof of the ABI structs as this library was compiled -- 0: esr_view, 1: esr_conv_desc, 2: esr_esa_desc, 3: esr_bsconv_desc,
 * 4: esr_ca_desc, 5: esr_op, 6: esr_esa_lowres_desc, 7: esr_chain_desc; anything else: 0.  A binding checks its own struct definitions against these once at load time (the
 * reference has no counterpart: its boundary is Python objects). */
size_t      esr_sizeof(int which);
const char* esr_build_info(void);         /* e.g. "gfx950 f32-mfma16x16x4 tile16x16 chunk8" */
/* ABI v9: SHA-256 (hex) over the library's sources (every .hip / .inc / .h under csrc + this header) as __graft_entry__.build() saw them, or
 * "unknown" for a library compiled by hand.  Measurements stored next to the code (profiles/pmc_traffic.json) carry it, and bench.py
 * refuses to replay PMC traffic recorded for another build. */
const char* esr_source_hash(void);

#ifdef __cplusplus
}
#endif
#endif /* ESR_HIP_H */
