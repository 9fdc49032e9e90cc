"""ctypes binding of libesr_hip.so (C ABI: include/esr_hip.h).

There is deliberately NO fallback: if the HIP library cannot be loaded, every
op raises.  A product path that silently ran on the CPU or on eager PyTorch
would void the parity claims.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (ESR_HIP_LIB: research builds of the SAME library -- compile-time A/B switches, tools/r05 -- loaded in place of the product's; never set in production)
SO_PATH = os.environ.get("ESR_HIP_LIB") or os.path.join(_HERE, "libesr_hip.so")

ESR_OK = 0
STATUS = {0: "ESR_OK", -1: "ESR_ERR_BAD_ARG", -2: "ESR_ERR_UNSUPPORTED", -3: "ESR_ERR_LAUNCH", -4: "ESR_ERR_TOO_SMALL"}

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_GELU = 0, 1, 2, 3
RES_NONE, RES_PRE_ACT, RES_POST_ACT = 0, 1, 2
NHWC, NCHW_IN, NCHW_SHUFFLE4 = 0, 1, 2
BLOCKED_IN, BLOCKED_OUT1, BLOCKED_OUT0, BLOCKED_RES = 1, 2, 4, 8          # esr_conv_desc.blocked8 bits (ABI v6 / v9)
HILO_IN, HILO_RES, HILO_OUT = 1, 2, 4                                     # esr_conv_desc.hilo bits (ABI v10)
OP_CONV, OP_CONV3X3S2, OP_MAXPOOL7S3, OP_ESA_APPLY, OP_DWCONV, OP_BSCONV, OP_PACK_INPUT, OP_ESA_LOWRES, OP_CONV_CHAIN = 0, 1, 2, 3, 4, 5, 6, 7, 8
CHAIN_MAX_LAYERS = 4
ESA_MAX_LAYERS = 3
ESA_FP = 16
COMPUTE_F32, COMPUTE_BF16, COMPUTE_F16 = 0, 1, 2
COMPUTE = {"f32": 0, "bf16": 1, "f16": 2}
STORE = {"f32": 0, "bf16": 1, "f16": 2}     # esr_storage: element type of the full-resolution NHWC views


class View(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("pitch", ctypes.c_int32), ("coff", ctypes.c_int32)]


class ConvDesc(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_int32), ("h", ctypes.c_int32), ("w", ctypes.c_int32),
        ("cin", ctypes.c_int32), ("cout", ctypes.c_int32), ("ksize", ctypes.c_int32),
        ("in_layout", ctypes.c_int32), ("out_layout", ctypes.c_int32),
        ("act", ctypes.c_int32), ("slope", ctypes.c_float),
        ("res_mode", ctypes.c_int32), ("split", ctypes.c_int32),
        ("inp", View), ("res", View), ("out0", View), ("out1", View),
        ("wpacked", ctypes.c_void_p),
        ("compute", ctypes.c_int32), ("storage", ctypes.c_int32),
        ("tail_wpacked", ctypes.c_void_p), ("tail_cat", View),
        ("tail_cat_c", ctypes.c_int32), ("tail_cout", ctypes.c_int32),
        ("tail_mid_act", ctypes.c_int32), ("tail_seg_stride16", ctypes.c_int32),
        ("post_wpacked", ctypes.c_void_p), ("post_out", View),
        ("post_cout", ctypes.c_int32), ("post_act", ctypes.c_int32),
        ("post2_wpacked", ctypes.c_void_p), ("post2_out", View),
        ("post2_cout", ctypes.c_int32), ("reserved3", ctypes.c_int32), ("border_bias", ctypes.c_void_p), ("in_seg_stride", ctypes.c_int64), ("in_seg_chunks", ctypes.c_int32), ("blocked8", ctypes.c_int32),
        ("hilo", ctypes.c_int32), ("wino_wpacked", ctypes.c_void_p),           # ABI v10 / v7
        ("hilo_stride", ctypes.c_int64),                                       # ABI v10
    ]


class EsaPost(ctypes.Structure):
    _fields_ = [
        ("cout", ctypes.c_int32), ("act", ctypes.c_int32), ("slope", ctypes.c_float), ("res_mode", ctypes.c_int32),
        ("res", View), ("out", View),
    ]


class EsaDesc(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_int32), ("h", ctypes.c_int32), ("w", ctypes.c_int32),
        ("c", ctypes.c_int32), ("f", ctypes.c_int32), ("h_lo", ctypes.c_int32), ("w_lo", ctypes.c_int32),
        ("storage", ctypes.c_int32),
        ("x", View), ("y", View),
        ("c1", ctypes.c_void_p), ("c3", ctypes.c_void_p), ("w0", ctypes.c_void_p), ("w1", ctypes.c_void_p),
        ("post_w", ctypes.c_void_p), ("post", EsaPost * 2), ("skip_y", ctypes.c_int32), ("reserved", ctypes.c_int32),
    ]


class BsDesc(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_int32), ("h", ctypes.c_int32), ("w", ctypes.c_int32),
        ("cin", ctypes.c_int32), ("c", ctypes.c_int32), ("act", ctypes.c_int32), ("slope", ctypes.c_float),
        ("res_mode", ctypes.c_int32),
        ("inp", View), ("res", View), ("out", View),
        ("pw_packed", ctypes.c_void_p), ("dw_packed", ctypes.c_void_p), ("d_packed", ctypes.c_void_p),
        ("d_cout", ctypes.c_int32), ("d_act", ctypes.c_int32),
        ("d_out", View),
        ("storage", ctypes.c_int32), ("reserved", ctypes.c_int32),
    ]


class CaDesc(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_int32), ("h", ctypes.c_int32), ("w", ctypes.c_int32), ("c", ctypes.c_int32),
        ("cr", ctypes.c_int32), ("contrast", ctypes.c_int32), ("layout", ctypes.c_int32), ("storage", ctypes.c_int32),
        ("x", View), ("y", View),
        ("w1", ctypes.c_void_p), ("w2", ctypes.c_void_p), ("stats", ctypes.c_void_p),
    ]


class EsaLayer(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("act", ctypes.c_int32), ("w", ctypes.c_void_p), ("w_dw", ctypes.c_void_p)]


class EsaLowresDesc(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_int32), ("h", ctypes.c_int32), ("w", ctypes.c_int32), ("f", ctypes.c_int32),
        ("storage", ctypes.c_int32), ("n_layers", ctypes.c_int32),
        ("x", View), ("w_s2", ctypes.c_void_p), ("pooled", ctypes.c_void_p), ("y", ctypes.c_void_p),
        ("layer", EsaLayer * ESA_MAX_LAYERS),
    ]


class ChainDesc(ctypes.Structure):          # esr_chain_desc (ABI v11)
    _fields_ = [
        ("n", ctypes.c_int32), ("h", ctypes.c_int32), ("w", ctypes.c_int32), ("n_layers", ctypes.c_int32),
        ("cin", ctypes.c_int32), ("cmid", ctypes.c_int32), ("cout", ctypes.c_int32),
        ("act", ctypes.c_int32), ("slope", ctypes.c_float), ("res_mode", ctypes.c_int32),
        ("storage", ctypes.c_int32), ("compute", ctypes.c_int32),
        ("inp", View), ("wpacked", ctypes.c_void_p * CHAIN_MAX_LAYERS),
        ("post_wpacked", ctypes.c_void_p), ("post_out", View), ("post_cout", ctypes.c_int32), ("post_act", ctypes.c_int32),
        ("post2_wpacked", ctypes.c_void_p), ("post2_out", View), ("post2_cout", ctypes.c_int32), ("reserved", ctypes.c_int32),
    ]


class Op(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("reserved", ctypes.c_int32), ("conv", ConvDesc), ("esa", EsaDesc),
                ("bs", BsDesc), ("lo", EsaLowresDesc), ("chain", ChainDesc)]


# every symbol include/esr_hip.h declares (tests check the .so exports all of them)
EXPORTS = [
    "esr_abi_version", "esr_last_hip_error", "esr_build_info", "esr_source_hash", "esr_sizeof",
    "esr_packed_conv_bytes", "esr_pack_conv_f32", "esr_unpack_conv_f32",
    "esr_packed_conv_s16_bytes", "esr_pack_conv_s16", "esr_unpack_conv_s16",
    "esr_packed_post_s16_bytes", "esr_pack_post_s16", "esr_conv_post_supported",
    "esr_packed_tail_s16_bytes", "esr_pack_tail_s16", "esr_conv_tail_supported",
    "esr_packed_wino_bytes", "esr_pack_wino_f32", "esr_unpack_wino_f32", "esr_wino_supported",
    "esr_conv2d_f32", "esr_conv_block_waves", "esr_run_ops", "esr_pack_input_s16",
    "esr_prof_create", "esr_run_ops_profiled", "esr_prof_collect", "esr_prof_destroy", "esr_prof_kernel_symbol",
    "esr_packed_dense_bytes", "esr_pack_dense_f32",
    "esr_conv3x3s2_f32", "esr_maxpool7s3_f32", "esr_esa_apply_f32", "esr_esa_lowres_f32",
    "esr_esa_apply_post_supported", "esr_packed_apply_post_bytes", "esr_pack_apply_post",
    "esr_packed_dw_bytes", "esr_pack_dw_f32", "esr_dwconv3x3_f32", "esr_bsconv_f32",
    "esr_tensor2uint_u8", "esr_sqerr_u8", "esr_channel_attention_f32",
    "esr_tensor2uint_u8_chk", "esr_ssim_partials", "esr_ssim_u8",
    "esr_conv_chain_supported", "esr_conv_chain_s16",
    "esr_graph_create", "esr_graph_launch", "esr_graph_nodes", "esr_graph_destroy",
    "esr_event_pair_ms", "esr_bw_probe",
]

_lib = None


class EsrError(RuntimeError):
    pass


def lib():
    """Load libesr_hip.so or raise.  Never returns None."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise EsrError(f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = ctypes.CDLL(SO_PATH)
    vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    L.esr_abi_version.restype = ci
    L.esr_last_hip_error.restype = ctypes.c_char_p
    L.esr_build_info.restype = ctypes.c_char_p
    L.esr_packed_conv_bytes.argtypes = [ci, ci, ci]
    L.esr_packed_conv_bytes.restype = sz
    L.esr_pack_conv_f32.argtypes = [vp, vp, ci, ci, ci, vp, ci, vp, sz]
    L.esr_pack_conv_f32.restype = ci
    L.esr_unpack_conv_f32.argtypes = [vp, sz, ci, ci, ci, vp, ci, vp, vp]
    L.esr_unpack_conv_f32.restype = ci
    L.esr_packed_conv_s16_bytes.argtypes = [ci, ci, ci]
    L.esr_packed_conv_s16_bytes.restype = sz
    L.esr_pack_conv_s16.argtypes = [vp, vp, ci, ci, ci, vp, ci, ci, vp, sz]
    L.esr_pack_conv_s16.restype = ci
    L.esr_unpack_conv_s16.argtypes = [vp, sz, ci, ci, ci, vp, ci, ci, vp, vp]
    L.esr_unpack_conv_s16.restype = ci
    L.esr_packed_post_s16_bytes.argtypes = [ci, ci]
    L.esr_packed_post_s16_bytes.restype = sz
    L.esr_pack_post_s16.argtypes = [vp, vp, ci, ci, ci, vp, sz]
    L.esr_pack_post_s16.restype = ci
    L.esr_packed_tail_s16_bytes.argtypes = [ci, ci, ci, ci]
    L.esr_packed_tail_s16_bytes.restype = sz
    L.esr_pack_tail_s16.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, sz]
    L.esr_pack_tail_s16.restype = ci
    L.esr_conv_tail_supported.argtypes = [ctypes.POINTER(ConvDesc)]
    L.esr_conv_tail_supported.restype = ci
    L.esr_conv_post_supported.argtypes = [ctypes.POINTER(ConvDesc)]
    L.esr_conv_post_supported.restype = ci
    L.esr_packed_wino_bytes.argtypes = [ci, ci]
    L.esr_packed_wino_bytes.restype = sz
    L.esr_pack_wino_f32.argtypes = [vp, vp, ci, ci, vp, ci, vp, sz]
    L.esr_pack_wino_f32.restype = ci
    L.esr_unpack_wino_f32.argtypes = [vp, sz, ci, ci, vp, ci, vp, vp]
    L.esr_unpack_wino_f32.restype = ci
    L.esr_wino_supported.argtypes = [ctypes.POINTER(ConvDesc)]
    L.esr_wino_supported.restype = ci
    L.esr_conv2d_f32.argtypes = [ctypes.POINTER(ConvDesc), vp]
    L.esr_conv2d_f32.restype = ci
    L.esr_conv_block_waves.argtypes = [ctypes.POINTER(ConvDesc)]
    L.esr_conv_block_waves.restype = ci
    L.esr_run_ops.argtypes = [ctypes.POINTER(Op), ci, vp]
    L.esr_run_ops.restype = ci
    L.esr_packed_dense_bytes.argtypes = [ci, ci, ci]
    L.esr_packed_dense_bytes.restype = sz
    L.esr_pack_dense_f32.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, sz]
    L.esr_pack_dense_f32.restype = ci
    for fn in (L.esr_conv3x3s2_f32, L.esr_maxpool7s3_f32, L.esr_esa_apply_f32):
        fn.argtypes = [ctypes.POINTER(EsaDesc), vp]
        fn.restype = ci
    L.esr_esa_lowres_f32.argtypes = [ctypes.POINTER(EsaLowresDesc), vp]
    L.esr_esa_lowres_f32.restype = ci
    L.esr_packed_dw_bytes.argtypes = [ci]
    L.esr_packed_dw_bytes.restype = sz
    L.esr_pack_dw_f32.argtypes = [vp, vp, ci, vp, sz]
    L.esr_pack_dw_f32.restype = ci
    L.esr_dwconv3x3_f32.argtypes = [ctypes.POINTER(ConvDesc), vp]
    L.esr_dwconv3x3_f32.restype = ci
    L.esr_bsconv_f32.argtypes = [ctypes.POINTER(BsDesc), vp]
    L.esr_bsconv_f32.restype = ci
    L.esr_channel_attention_f32.argtypes = [ctypes.POINTER(CaDesc), vp]
    L.esr_channel_attention_f32.restype = ci
    L.esr_tensor2uint_u8.argtypes = [vp, vp, ci, ci, ci, ctypes.c_float, vp]
    L.esr_tensor2uint_u8.restype = ci
    L.esr_sqerr_u8.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp]
    L.esr_sqerr_u8.restype = ci
    L.esr_tensor2uint_u8_chk.argtypes = [vp, vp, ci, ci, ci, ctypes.c_float, vp, vp]
    L.esr_tensor2uint_u8_chk.restype = ci
    L.esr_ssim_partials.argtypes = [ci, ci, ci, ci]
    L.esr_ssim_partials.restype = sz
    L.esr_ssim_u8.argtypes = [vp, vp, ci, ci, ci, ci, vp, sz, vp]
    L.esr_ssim_u8.restype = ci
    L.esr_source_hash.restype = ctypes.c_char_p
    L.esr_prof_create.argtypes = [ci, ci, ctypes.POINTER(vp)]
    L.esr_prof_create.restype = ci
    L.esr_run_ops_profiled.argtypes = [ctypes.POINTER(Op), ci, vp, vp]
    L.esr_run_ops_profiled.restype = ci
    L.esr_prof_collect.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ci, ctypes.POINTER(ci)]
    L.esr_prof_collect.restype = ci
    L.esr_prof_kernel_symbol.argtypes = [vp, ci, ctypes.c_char_p, sz]
    L.esr_prof_kernel_symbol.restype = ci
    L.esr_prof_destroy.argtypes = [vp]
    L.esr_prof_destroy.restype = None
    L.esr_esa_apply_post_supported.argtypes = [ci, ci, ci]
    L.esr_esa_apply_post_supported.restype = ci
    L.esr_packed_apply_post_bytes.argtypes = [ci, ci, ci, ci]
    L.esr_packed_apply_post_bytes.restype = sz
    L.esr_pack_apply_post.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp, sz]
    L.esr_pack_apply_post.restype = ci
    L.esr_conv_chain_supported.argtypes = [ctypes.POINTER(ChainDesc)]
    L.esr_conv_chain_supported.restype = ci
    L.esr_conv_chain_s16.argtypes = [ctypes.POINTER(ChainDesc), vp]
    L.esr_conv_chain_s16.restype = ci
    L.esr_graph_create.argtypes = [ctypes.POINTER(Op), ci, vp, vp, ctypes.POINTER(vp)]
    L.esr_graph_create.restype = ci
    L.esr_graph_launch.argtypes = [vp, vp, vp, vp]
    L.esr_graph_launch.restype = ci
    L.esr_graph_nodes.argtypes = [vp]
    L.esr_graph_nodes.restype = ci
    L.esr_graph_destroy.argtypes = [vp]
    L.esr_graph_destroy.restype = None
    L.esr_event_pair_ms.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_double)]
    L.esr_event_pair_ms.restype = ci
    L.esr_bw_probe.argtypes = [vp, sz, ci, vp, ctypes.POINTER(ctypes.c_double)]
    L.esr_bw_probe.restype = ci
    if L.esr_abi_version() != 12:
        raise EsrError("libesr_hip.so ABI version mismatch")
    L.esr_sizeof.argtypes = [ci]
    L.esr_sizeof.restype = ctypes.c_size_t
    for which, st in enumerate((View, ConvDesc, EsaDesc, BsDesc, CaDesc, Op, EsaLowresDesc, ChainDesc)):
        if L.esr_sizeof(which) != ctypes.sizeof(st):
            raise EsrError(f"libesr_hip.so: sizeof({st.__name__}) is {L.esr_sizeof(which)} in the library, {ctypes.sizeof(st)} in the binding")
    _lib = L
    return L


def check(rc, what):
    if rc != ESR_OK:
        msg = lib().esr_last_hip_error().decode()
        raise EsrError(f"{what} failed: {STATUS.get(rc, rc)} {msg}")
