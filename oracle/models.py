"""numpy/C graphs of the four hot-path networks (the oracle proper).

Each function consumes the reference state_dict (name -> array, exactly the
keys of weights/<ckpt>.safetensors) and an NCHW fp32 array, and follows the
reference forward line by line (citations relative to /root/reference).
Test infrastructure only -- see oracle/__init__.py.
"""
import numpy as np

from . import prims as P


def _conv(sd, name, x, stride=1, pad=None):
    w = sd[name + ".weight"]
    if pad is None:
        pad = (w.shape[2] - 1) // 2
    return P.conv2d(x, w, sd[name + ".bias"], stride=stride, pad=pad)


# ----------------------------------------------------------------------------
# IMDN  (models/imdn_baseline.py:33-65; blocks models/basicblock.py:230-265)
# ----------------------------------------------------------------------------
def imdn_block(sd, pre, x, d_nc=16):
    """IMDBlock.forward, basicblock.py:259-265 (conv+LeakyReLU(0.05) x3, bare conv4, 1x1, residual)."""
    t1 = P.leaky_relu(_conv(sd, pre + "conv1.0", x))
    d1, r1 = t1[:, :d_nc], t1[:, d_nc:]
    t2 = P.leaky_relu(_conv(sd, pre + "conv2.0", r1))
    d2, r2 = t2[:, :d_nc], t2[:, d_nc:]
    t3 = P.leaky_relu(_conv(sd, pre + "conv3.0", r2))
    d3, r3 = t3[:, :d_nc], t3[:, d_nc:]
    d4 = _conv(sd, pre + "conv4", r3)
    res = _conv(sd, pre + "conv1x1", np.concatenate([d1, d2, d3, d4], axis=1))
    return P.add(x, res)


def imdn(sd, x, nb=8, upscale=4):
    head = _conv(sd, "model.0", x)                          # imdn_baseline.py:46
    t = head
    for i in range(nb):                                     # :47
        t = imdn_block(sd, f"model.1.sub.{i}.", t)
    t = _conv(sd, f"model.1.sub.{nb}", t)                   # :48
    t = P.add(head, t)                                      # ShortcutBlock basicblock.py:197-199
    t = _conv(sd, "model.2", t)                             # upsample_pixelshuffle basicblock.py:446-449
    return P.pixel_shuffle(t, upscale)


# ----------------------------------------------------------------------------
# RFDN  (models/rfdn_baseline/RFDN.py:11-41, block.py:103-166)
# ----------------------------------------------------------------------------
def rfdn_esa(sd, pre, x):
    """ESA.forward, rfdn_baseline/block.py:117-129."""
    c1_ = _conv(sd, pre + "conv1", x)
    c1 = _conv(sd, pre + "conv2", c1_, stride=2, pad=0)
    v_max = P.max_pool2d(c1, 7, 3)
    v_range = P.relu(_conv(sd, pre + "conv_max", v_max))
    c3 = P.relu(_conv(sd, pre + "conv3", v_range))
    c3 = _conv(sd, pre + "conv3_", c3)
    c3 = P.bilinear(c3, x.shape[2], x.shape[3])
    cf = _conv(sd, pre + "conv_f", c1_)
    c4 = _conv(sd, pre + "conv4", P.add(c3, cf))
    return P.sigmoid_mul(x, c4)


def rfdb(sd, pre, x):
    """RFDB.forward, rfdn_baseline/block.py:148-166."""
    d1 = P.leaky_relu(_conv(sd, pre + "c1_d", x))
    r1 = P.leaky_relu(P.add(_conv(sd, pre + "c1_r", x), x))
    d2 = P.leaky_relu(_conv(sd, pre + "c2_d", r1))
    r2 = P.leaky_relu(P.add(_conv(sd, pre + "c2_r", r1), r1))
    d3 = P.leaky_relu(_conv(sd, pre + "c3_d", r2))
    r3 = P.leaky_relu(P.add(_conv(sd, pre + "c3_r", r2), r2))
    r4 = P.leaky_relu(_conv(sd, pre + "c4", r3))
    out = np.concatenate([d1, d2, d3, r4], axis=1)
    return rfdn_esa(sd, pre + "esa.", _conv(sd, pre + "c5", out))


def rfdn(sd, x, num_modules=4, upscale=4):
    fea = _conv(sd, "fea_conv", x)                          # RFDN.py:30
    outs, t = [], fea
    for i in range(1, num_modules + 1):                     # :31-34
        t = rfdb(sd, f"B{i}.", t)
        outs.append(t)
    out_b = P.leaky_relu(_conv(sd, "c.0", np.concatenate(outs, axis=1)))   # :36
    out_lr = P.add(_conv(sd, "LR_conv", out_b), fea)        # :37
    return P.pixel_shuffle(_conv(sd, "upsampler.0", out_lr), upscale)      # :39


# ----------------------------------------------------------------------------
# RLFN_cut  (models/team04_rlfn.py:62-152)
# ----------------------------------------------------------------------------
def rlfn_esa(sd, pre, x):
    """slim ESA.forward, team04_rlfn.py:76-89 (single conv3, no ReLU stack)."""
    c1_ = _conv(sd, pre + "conv1", x)
    c1 = _conv(sd, pre + "conv2", c1_, stride=2, pad=0)
    v_max = P.max_pool2d(c1, 7, 3)
    c3 = _conv(sd, pre + "conv3", v_max)
    c3 = P.bilinear(c3, x.shape[2], x.shape[3])
    cf = _conv(sd, pre + "conv_f", c1_)
    c4 = _conv(sd, pre + "conv4", P.add(c3, cf))
    return P.sigmoid_mul(x, c4)


def rlfb(sd, pre, x):
    """RLFB.forward, team04_rlfn.py:109-122."""
    t = P.leaky_relu(_conv(sd, pre + "c1_r", x))
    t = P.leaky_relu(_conv(sd, pre + "c2_r", t))
    t = P.leaky_relu(_conv(sd, pre + "c3_r", t))
    t = P.add(t, x)
    return rlfn_esa(sd, pre + "esa.", _conv(sd, pre + "c5", t))


def rlfn(sd, x, upscale=4):
    fea = _conv(sd, "fea_conv", x)                          # team04_rlfn.py:142
    t = fea
    for i in range(1, 5):                                   # :144-147
        t = rlfb(sd, f"B{i}.", t)
    out_lr = P.add(_conv(sd, "LR_conv", t), fea)            # :149
    return P.pixel_shuffle(_conv(sd, "upsampler.0", out_lr), upscale)      # :150


# ----------------------------------------------------------------------------
# BSRN  (models/team18_bsrn.py:44-236)
# ----------------------------------------------------------------------------
def _lin(sd, name, x):
    return P.linear_nchw(x, sd[name + ".weight"], sd[name + ".bias"])


def bsconv_u(sd, pre, x):
    """BSConvU.forward, team18_bsrn.py:82-88: pointwise Linear then depthwise 3x3 (zeros padding)."""
    t = _lin(sd, pre + "pw", x)
    w = sd[pre + "dw.weight"]
    return P.conv2d(t, w, sd[pre + "dw.bias"], stride=1, pad=1, groups=w.shape[0])


def bsrn_esa(sd, pre, x):
    """ESA.forward, team18_bsrn.py:109-122."""
    c1_ = _lin(sd, pre + "conv1", x)
    c1 = _conv(sd, pre + "conv2", c1_, stride=2, pad=0)
    v_max = P.max_pool2d(c1, 7, 3)
    v_range = P.gelu(bsconv_u(sd, pre + "conv_max.", v_max))
    c3 = P.gelu(bsconv_u(sd, pre + "conv3.", v_range))
    c3 = bsconv_u(sd, pre + "conv3_.", c3)
    c3 = P.bilinear(c3, x.shape[2], x.shape[3])
    cf = _lin(sd, pre + "conv_f", c1_)
    c4 = _lin(sd, pre + "conv4", P.add(c3, cf))
    return P.sigmoid_mul(x, c4)


def bsrn_rfdb(sd, pre, x):
    """RFDB.forward, team18_bsrn.py:150-172."""
    d1 = P.gelu(_lin(sd, pre + "c1_d", x))
    r1 = P.gelu(P.add(bsconv_u(sd, pre + "c1_r.", x), x))
    d2 = P.gelu(_lin(sd, pre + "c2_d", r1))
    r2 = P.gelu(P.add(bsconv_u(sd, pre + "c2_r.", r1), r1))
    d3 = P.gelu(_lin(sd, pre + "c3_d", r2))
    r3 = P.gelu(P.add(bsconv_u(sd, pre + "c3_r.", r2), r2))
    r4 = P.gelu(bsconv_u(sd, pre + "c4.", r3))
    out = _lin(sd, pre + "c5", np.concatenate([d1, d2, d3, r4], axis=1))
    fused = bsrn_esa(sd, pre + "esa.", out)
    fused = P.channel_scale(fused, sd[pre + "cw"])          # :169
    return P.add(_lin(sd, pre + "conv_out", fused), x)      # :170-172


def bsrn(sd, x, num_block=5, upscale=4):
    x4 = np.concatenate([x, x, x, x], axis=1)               # team18_bsrn.py:218
    fea = bsconv_u(sd, "fea_conv.", x4)
    outs, t = [], fea
    for i in range(1, num_block + 1):
        t = bsrn_rfdb(sd, f"B{i}.", t)
        outs.append(t)
    out_b = P.gelu(_lin(sd, "c1", np.concatenate(outs, axis=1)))           # :227-229
    out_lr = P.add(bsconv_u(sd, "c2.", out_b), fea)         # :231
    t = _conv(sd, "upsampler.upsampleOneStep.0", out_lr)    # :24,234
    return P.pixel_shuffle(t, upscale)


FORWARD = {
    "imdn_baseline": imdn,
    "rfdn_baseline": rfdn,
    "team04_rlfn": rlfn,
    "team18_bsrn": bsrn,
}
DATA_RANGE = {"imdn_baseline": 1.0, "rfdn_baseline": 255.0, "team04_rlfn": 255.0, "team18_bsrn": 1.0}
