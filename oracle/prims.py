"""ctypes wrappers over oracle/libesr_oracle.so (NCHW contiguous fp32 numpy arrays).

Test infrastructure only -- see oracle/__init__.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libesr_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "esr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        fp, ip, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
        L.orc_conv2d.argtypes = [fp, fp, fp, fp] + [ip] * 9
        L.orc_conv2d.restype = ip
        L.orc_leaky_relu.argtypes = [fp, sz, ctypes.c_float]
        L.orc_relu.argtypes = [fp, sz]
        L.orc_gelu.argtypes = [fp, sz]
        L.orc_sigmoid_mul.argtypes = [fp, fp, fp, sz]
        L.orc_add.argtypes = [fp, fp, fp, sz]
        L.orc_channel_scale.argtypes = [fp, fp, ip, ip, ip]
        L.orc_max_pool2d.argtypes = [fp, fp] + [ip] * 5
        L.orc_max_pool2d.restype = ip
        L.orc_bilinear.argtypes = [fp, fp] + [ip] * 5
        L.orc_pixel_shuffle.argtypes = [fp, fp] + [ip] * 5
        L.orc_tensor2uint.argtypes = [fp, fp, ip, ip, ip, ctypes.c_float]
        L.orc_psnr_u8.argtypes = [fp, fp, ip, ip, ip, ip]
        L.orc_psnr_u8.restype = ctypes.c_double
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def conv2d(x, w, b=None, stride=1, pad=0, groups=1):
    x, w = _f32(x), _f32(w)
    N, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = np.empty((N, Cout, Ho, Wo), np.float32)
    bb = _f32(b) if b is not None else None
    rc = lib().orc_conv2d(_p(x), _p(w), _p(bb) if bb is not None else None, _p(y),
                          N, Cin, H, W, Cout, k, stride, pad, groups)
    if rc != 0:
        raise ValueError(f"orc_conv2d failed rc={rc}")
    return y


def linear_nchw(x, w, b=None):
    """nn.Linear applied on the channel axis of an NCHW tensor (BSRN permutes to NHWC
    around it, team18_bsrn.py:83-84,110-119,150-170): identical to a 1x1 conv."""
    w = _f32(w)
    return conv2d(x, w.reshape(w.shape[0], w.shape[1], 1, 1), b)


def leaky_relu(x, slope=0.05):
    y = _f32(x).copy()
    lib().orc_leaky_relu(_p(y), y.size, slope)
    return y


def relu(x):
    y = _f32(x).copy()
    lib().orc_relu(_p(y), y.size)
    return y


def gelu(x):
    y = _f32(x).copy()
    lib().orc_gelu(_p(y), y.size)
    return y


def sigmoid_mul(x, m):
    x, m = _f32(x), _f32(m)
    y = np.empty_like(x)
    lib().orc_sigmoid_mul(_p(x), _p(m), _p(y), x.size)
    return y


def add(a, b):
    a, b = _f32(a), _f32(b)
    y = np.empty_like(a)
    lib().orc_add(_p(a), _p(b), _p(y), a.size)
    return y


def channel_scale(x, s):
    y = _f32(x).copy()
    s = _f32(s).reshape(-1)
    N, C, H, W = y.shape
    lib().orc_channel_scale(_p(y), _p(s), N, C, H * W)
    return y


def max_pool2d(x, k, s):
    x = _f32(x)
    N, C, H, W = x.shape
    Ho, Wo = (H - k) // s + 1, (W - k) // s + 1
    y = np.empty((N, C, Ho, Wo), np.float32)
    if lib().orc_max_pool2d(_p(x), _p(y), N * C, H, W, k, s) != 0:
        raise ValueError("orc_max_pool2d: input too small")
    return y


def bilinear(x, Ho, Wo):
    x = _f32(x)
    N, C, H, W = x.shape
    y = np.empty((N, C, Ho, Wo), np.float32)
    lib().orc_bilinear(_p(x), _p(y), N * C, H, W, Ho, Wo)
    return y


def pixel_shuffle(x, r):
    x = _f32(x)
    N, C, H, W = x.shape
    y = np.empty((N, C // (r * r), H * r, W * r), np.float32)
    lib().orc_pixel_shuffle(_p(x), _p(y), N, C, H, W, r)
    return y


def tensor2uint(x_chw, data_range):
    x = _f32(x_chw)
    C, H, W = x.shape
    y = np.empty((H, W, C), np.uint8)
    lib().orc_tensor2uint(_p(x), _p(y), C, H, W, float(data_range))
    return y


def psnr_u8(a, b, border=0):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    if a.shape != b.shape:
        raise ValueError("Input images must have the same dimensions.")
    H, W = a.shape[:2]
    C = a.shape[2] if a.ndim == 3 else 1
    return lib().orc_psnr_u8(_p(a), _p(b), H, W, C, border)
