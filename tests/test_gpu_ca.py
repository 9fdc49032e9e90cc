"""-m gpu: CALayer / CCALayer kernels against vectors produced by the reference classes (tools/gen_golden_r2.py ->
tests/golden/ca_cca.npz: models/basicblock.py:333-348, models/team05_efdn/plainblock.py:106-122)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("tag,cls,kw", [("ca", "CALayer", dict(channel=64, reduction=16)), ("cca", "CCALayer", dict(channel=64, reduction=4)),
                                        ("cca48", "CCALayer", dict(channel=48, reduction=4))])
def test_drop_in_modules_match_reference(tag, cls, kw):
    from ntire2022_esr_amd import attention
    g = np.load(os.path.join(GOLD, "ca_cca.npz"))
    m = getattr(attention, cls)(**kw)
    sd = {k[len(tag) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "_w_")}
    m.load_state_dict(sd, strict=True)                            # the reference's key names
    y = m.to(DEV)(torch.from_numpy(g[tag + "_x"]).to(DEV))
    ref = g[tag + "_y"]
    assert float(np.abs(y.cpu().numpy() - ref).max()) < 2e-5 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2.0 ** -7), (torch.float16, 2.0 ** -10)])
def test_nhwc_views_and_storage_types(dt, tol):
    """the engine-native form: NHWC slices of wider buffers, fp32 and 16-bit storage, against fp64 on the same inputs"""
    from ntire2022_esr_amd import ops
    g = torch.Generator().manual_seed(3)
    c, cr = 48, 12
    x = (torch.randn(2, 19, 23, c, generator=g) * 2 + 1.5).to(dt)
    w1, b1 = torch.randn(cr, c, generator=g) * 0.3, torch.randn(cr, generator=g) * 0.1
    w2, b2 = torch.randn(c, cr, generator=g) * 0.3, torch.randn(c, generator=g) * 0.1
    xd = x.double()
    s = xd.std(dim=(1, 2), unbiased=False) + xd.mean(dim=(1, 2))
    gate = torch.sigmoid(torch.relu(s @ w1.double().T + b1.double()) @ w2.double().T + b2.double())
    ref = xd * gate[:, None, None, :]
    y = ops.channel_attention(x.to(DEV), w1, b1, w2, b2, contrast=True)
    assert y.dtype == dt
    err = (y.double().cpu() - ref).abs() / ref.abs().clamp_min(1.0)
    assert float(err.max()) < tol
