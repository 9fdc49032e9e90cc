"""Channel-attention drop-ins: `CALayer` (models/basicblock.py:333-348) and the contrast-aware `CCALayer`
(models/team05_efdn/plainblock.py:106-122) on the HIP engine.  Same constructor arguments and the same state_dict keys as the
reference classes (`conv_fc.{0,2}.{weight,bias}` / `conv_du.{0,2}.{weight,bias}`); forward takes and returns NCHW fp32 CUDA
tensors like the reference modules and is one esr_channel_attention_f32 call (reduce + scale).  None of the four in-scope
networks executes them (both are defined-but-unused for ids -1 / 0 / 4 / 18); they serve the wider IMDN / RFDN family."""
import torch.nn as nn

from . import _lib as L
from . import ops


class _ChannelAttention(nn.Module):
    _seq, _contrast = "conv_fc", False

    def __init__(self, channel=64, reduction=16):
        super().__init__()
        if channel > 64 or channel // reduction > 16 or channel // reduction < 1:
            raise NotImplementedError("HIP channel attention supports channel <= 64 and channel // reduction in [1, 16]")
        # parameter holders with the reference's key names: index 0 and 2 of the nn.Sequential (1 = ReLU, 3 = Sigmoid)
        seq = nn.Module()
        seq.add_module("0", nn.Conv2d(channel, channel // reduction, 1, padding=0, bias=True))
        seq.add_module("2", nn.Conv2d(channel // reduction, channel, 1, padding=0, bias=True))
        self.add_module(self._seq, seq)

    def forward(self, x):
        if not x.is_cuda:
            raise L.EsrError(f"{type(self).__name__}: input is on {x.device}; this engine only runs on an MI355X "
                             "(no CPU fallback)")
        seq = self._modules[self._seq]
        a, b = seq._modules["0"], seq._modules["2"]
        return ops.channel_attention(x.contiguous().float(), a.weight, a.bias, b.weight, b.bias,
                                     contrast=self._contrast, nchw=True)


class CALayer(_ChannelAttention):
    """models/basicblock.py:333-348: x * sigmoid(conv_fc(avg_pool(x)))"""
    _seq, _contrast = "conv_fc", False


class CCALayer(_ChannelAttention):
    """models/team05_efdn/plainblock.py:106-122: x * sigmoid(conv_du(std(x) + avg_pool(x)))"""
    _seq, _contrast = "conv_du", True

    def __init__(self, channel, reduction=4):
        super().__init__(channel, reduction)
