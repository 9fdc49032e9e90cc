"""The four hot-path graphs as the ATen op sequence the reference issues.

Why this exists next to oracle/models.py: the reference's CPU path IS PyTorch
(ATen -> oneDNN); BASELINE.md section 3 names "the same ATen op sequence" as the
CPU baseline of record.  This module restates the graphs functionally (state_dict
in, tensor out: conv2d / linear / leaky_relu_ / gelu / split / cat / add / mul /
max_pool2d / interpolate / sigmoid / pixel_shuffle in the reference's order) so
that bench.py can time it on the GPU box's host cores (cpu_baseline.kind =
"port") and tests can cross-check it against the C oracle and the goldens.
Test infrastructure only -- see oracle/__init__.py.  Works on any torch device,
so tests may also run it on the GPU as a second opinion.
"""
import torch
import torch.nn.functional as F


def _c(sd, name, x, stride=1, padding=None, groups=1):
    w = sd[name + ".weight"]
    if padding is None:
        padding = (w.shape[2] - 1) // 2
    return F.conv2d(x, w, sd[name + ".bias"], stride=stride, padding=padding, groups=groups)


def _act(x):
    return F.leaky_relu_(x, 0.05)


# IMDN -- models/imdn_baseline.py:63-65, models/basicblock.py:259-265
def imdn(sd, x, nb=8):
    head = _c(sd, "model.0", x)
    t = head
    for i in range(nb):
        p = f"model.1.sub.{i}."
        d1, r1 = torch.split(_act(_c(sd, p + "conv1.0", t)), (16, 48), dim=1)
        d2, r2 = torch.split(_act(_c(sd, p + "conv2.0", r1)), (16, 48), dim=1)
        d3, r3 = torch.split(_act(_c(sd, p + "conv3.0", r2)), (16, 48), dim=1)
        d4 = _c(sd, p + "conv4", r3)
        t = t + _c(sd, p + "conv1x1", torch.cat((d1, d2, d3, d4), dim=1))
    t = head + _c(sd, f"model.1.sub.{nb}", t)
    return F.pixel_shuffle(_c(sd, "model.2", t), 4)


def _esa_tail(sd, p, x, c1_, c3):
    c3 = F.interpolate(c3, (x.size(2), x.size(3)), mode="bilinear", align_corners=False)
    cf = _c(sd, p + "conv_f", c1_)
    c4 = _c(sd, p + "conv4", c3 + cf)
    return x * torch.sigmoid(c4)


# RFDN -- models/rfdn_baseline/RFDN.py:29-41, block.py:117-129,148-166
def _rfdn_esa(sd, p, x):
    c1_ = _c(sd, p + "conv1", x)
    c1 = _c(sd, p + "conv2", c1_, stride=2, padding=0)
    v = F.max_pool2d(c1, kernel_size=7, stride=3)
    v = F.relu_(_c(sd, p + "conv_max", v))
    c3 = F.relu_(_c(sd, p + "conv3", v))
    c3 = _c(sd, p + "conv3_", c3)
    return _esa_tail(sd, p, x, c1_, c3)


def _rfdb(sd, p, x):
    d1 = _act(_c(sd, p + "c1_d", x))
    r1 = _act(_c(sd, p + "c1_r", x) + x)
    d2 = _act(_c(sd, p + "c2_d", r1))
    r2 = _act(_c(sd, p + "c2_r", r1) + r1)
    d3 = _act(_c(sd, p + "c3_d", r2))
    r3 = _act(_c(sd, p + "c3_r", r2) + r2)
    r4 = _act(_c(sd, p + "c4", r3))
    out = torch.cat([d1, d2, d3, r4], dim=1)
    return _rfdn_esa(sd, p + "esa.", _c(sd, p + "c5", out))


def rfdn(sd, x):
    fea = _c(sd, "fea_conv", x)
    outs, t = [], fea
    for i in range(1, 5):
        t = _rfdb(sd, f"B{i}.", t)
        outs.append(t)
    out_b = _act(_c(sd, "c.0", torch.cat(outs, dim=1)))
    out_lr = _c(sd, "LR_conv", out_b) + fea
    return F.pixel_shuffle(_c(sd, "upsampler.0", out_lr), 4)


# RLFN_cut -- models/team04_rlfn.py:76-89,109-122,141-152
def _rlfn_esa(sd, p, x):
    c1_ = _c(sd, p + "conv1", x)
    c1 = _c(sd, p + "conv2", c1_, stride=2, padding=0)
    v = F.max_pool2d(c1, kernel_size=7, stride=3)
    c3 = _c(sd, p + "conv3", v)
    return _esa_tail(sd, p, x, c1_, c3)


def rlfn(sd, x):
    fea = _c(sd, "fea_conv", x)
    t = fea
    for i in range(1, 5):
        p = f"B{i}."
        o = _act(_c(sd, p + "c1_r", t))
        o = _act(_c(sd, p + "c2_r", o))
        o = _act(_c(sd, p + "c3_r", o))
        o = o + t
        t = _rlfn_esa(sd, p + "esa.", _c(sd, p + "c5", o))
    out_lr = _c(sd, "LR_conv", t) + fea
    return F.pixel_shuffle(_c(sd, "upsampler.0", out_lr), 4)


# BSRN -- models/team18_bsrn.py:82-88,109-122,150-172,217-236 (permutes kept: they are
# part of the reference's CPU cost)
def _lin(sd, name, x_nhwc):
    return F.linear(x_nhwc, sd[name + ".weight"], sd[name + ".bias"])


def _bsconv(sd, p, x):
    t = _lin(sd, p + "pw", x.permute(0, 2, 3, 1))
    w = sd[p + "dw.weight"]
    return F.conv2d(t.permute(0, 3, 1, 2), w, sd[p + "dw.bias"], stride=1, padding=1, groups=w.shape[0])


def _bsrn_esa(sd, p, inp):
    x = inp.permute(0, 2, 3, 1)
    c1_ = _lin(sd, p + "conv1", x)
    c1 = _c(sd, p + "conv2", c1_.permute(0, 3, 1, 2), stride=2, padding=0)
    v = F.max_pool2d(c1, kernel_size=7, stride=3)
    v = F.gelu(_bsconv(sd, p + "conv_max.", v))
    c3 = F.gelu(_bsconv(sd, p + "conv3.", v))
    c3 = _bsconv(sd, p + "conv3_.", c3)
    c3 = F.interpolate(c3, (inp.size(2), inp.size(3)), mode="bilinear", align_corners=False)
    cf = _lin(sd, p + "conv_f", c1_)
    c4 = _lin(sd, p + "conv4", c3.permute(0, 2, 3, 1) + cf)
    return inp * torch.sigmoid(c4.permute(0, 3, 1, 2))


def _bsrn_rfdb(sd, p, x):
    d1 = F.gelu(_lin(sd, p + "c1_d", x.permute(0, 2, 3, 1)))
    r1 = F.gelu(_bsconv(sd, p + "c1_r.", x) + x)
    d2 = F.gelu(_lin(sd, p + "c2_d", r1.permute(0, 2, 3, 1)))
    r2 = F.gelu(_bsconv(sd, p + "c2_r.", r1) + r1)
    d3 = F.gelu(_lin(sd, p + "c3_d", r2.permute(0, 2, 3, 1)))
    r3 = F.gelu(_bsconv(sd, p + "c3_r.", r2) + r2)
    r4 = F.gelu(_bsconv(sd, p + "c4.", r3))
    out = torch.cat([d1, d2, d3, r4.permute(0, 2, 3, 1)], dim=3)
    out = _lin(sd, p + "c5", out).permute(0, 3, 1, 2)
    f = _bsrn_esa(sd, p + "esa.", out)
    f = f.permute(0, 2, 3, 1) * sd[p + "cw"]
    f = _lin(sd, p + "conv_out", f)
    return f.permute(0, 3, 1, 2) + x


def bsrn(sd, x, num_block=5):
    x = torch.cat([x, x, x, x], dim=1)
    fea = _bsconv(sd, "fea_conv.", x)
    outs, t = [], fea
    for i in range(1, num_block + 1):
        t = _bsrn_rfdb(sd, f"B{i}.", t)
        outs.append(t)
    trunk = torch.cat(outs, dim=1)
    out_b = F.gelu(_lin(sd, "c1", trunk.permute(0, 2, 3, 1)).permute(0, 3, 1, 2))
    out_lr = _bsconv(sd, "c2.", out_b) + fea
    return F.pixel_shuffle(_c(sd, "upsampler.upsampleOneStep.0", out_lr), 4)


FORWARD = {
    "imdn_baseline": imdn,
    "rfdn_baseline": rfdn,
    "team04_rlfn": rlfn,
    "team18_bsrn": bsrn,
}
