"""CPU (-m "not gpu"): the shim packages resolve the import paths an unmodified test_demo.py uses (test_demo.py:8-10, 20,
26, 54, 152) to this repo's own classes, with the reference's ctor / load_state_dict / eval / to lifecycle and the
reference's complexity numbers; and `model(x)` is ONE registered torch.library operator with a fake (shape) implementation."""
import contextlib
import importlib
import io
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import GOLD, REPO, load_sd_torch

SHIM = os.path.join(REPO, "shim")


def _run_in_shim(code):
    env = dict(os.environ, PYTHONPATH=SHIM + os.pathsep + REPO)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=SHIM, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stdout


def test_shim_import_paths_lifecycle_and_counters():
    code = r'''
import contextlib, io, json, os, sys, torch
from safetensors.torch import load_file
from models.imdn_baseline import IMDN
from models.rfdn_baseline.RFDN import RFDN
from models.team04_rlfn import RLFN_cut
from models.team18_bsrn import BSRN
from utils.model_summary import get_model_activation, get_model_flops
from utils import utils_logger
from utils import utils_image as util
import ntire2022_esr_amd as E
assert IMDN is E.IMDN and RFDN is E.RFDN and RLFN_cut is E.RLFN_cut and BSRN is E.BSRN
W = os.environ["ESR_WEIGHTS"]
out = {}
with contextlib.redirect_stdout(io.StringIO()):
    nets = {"imdn_baseline": IMDN(in_nc=3, out_nc=3, nc=64, nb=8, upscale=4), "rfdn_baseline": RFDN(),
            "team04_rlfn": RLFN_cut(in_nc=3, out_nc=3),
            "team18_bsrn": BSRN(num_in_ch=3, num_feat=48, num_block=5, num_out_ch=3, upscale=4, conv='BSConvU', upsampler='pixelshuffledirect')}
for name, model in nets.items():
    model.load_state_dict(load_file(os.path.join(W, name + ".safetensors")), strict=True)      # test_demo.py:23
    model.eval()                                                                                  # :336
    for k, v in model.named_parameters():
        v.requires_grad = False                                                                   # :338-339
    model = model.to(torch.device("cpu"))                                                         # :340
    acts, nconv = get_model_activation(model, (3, 256, 256))                                      # :525
    flops = get_model_flops(model, (3, 256, 256), False)                                          # :530
    out[name] = {"activations": acts, "num_conv": nconv, "flops": flops,
                 "num_parameters": sum(map(lambda x: x.numel(), model.parameters()))}            # :534
utils_logger.logger_info("t", log_path=os.devnull)
assert util.uint2tensor4(__import__("numpy").zeros((4, 5, 3), "uint8"), 255.0).shape == (1, 3, 4, 5)
print(json.dumps(out))
'''
    os.environ["ESR_WEIGHTS"] = os.path.join(REPO, "weights")
    got = json.loads(_run_in_shim(code).strip().splitlines()[-1])
    want = json.load(open(os.path.join(GOLD, "summary.json")))
    assert got == want


def test_shim_rider_modules():
    out = _run_in_shim("from models.team06_v1 import v1; from models.team22_rep_rfdn import RFDN40; "
                       "m = RFDN40(); print(len(m.state_dict()), len(v1(in_nc=3, nf=50, num_modules=4, out_nc=3, upscale=4).state_dict()))")
    assert out.split() == ["128", "128"]


def test_forward_is_a_registered_custom_op_with_fake_impl():
    from ntire2022_esr_amd import IMDN, RFDN, _lib as L, engine
    assert hasattr(torch.ops.esr, "sr_forward")
    m = IMDN()
    with pytest.raises(L.EsrError, match="no CPU fallback"):
        m(torch.rand(1, 3, 16, 16))                                  # the real kernel refuses CPU tensors
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        y = m(torch.empty(2, 3, 40, 56))                              # shape function only, nothing runs
        assert tuple(y.shape) == (2, 3, 160, 224) and y.dtype == torch.float32
    y = torch.ops.esr.sr_forward(torch.empty(1, 3, 17, 15, device="meta"), id(m))
    assert tuple(y.shape) == (1, 3, 68, 60)
    with pytest.raises(Exception):
        torch.ops.esr.sr_forward(torch.empty(1, 3, 17, 15, device="meta"), 12345)       # no live model with that handle
    r = RFDN()
    with FakeTensorMode():
        assert tuple(r(torch.empty(1, 3, 33, 21)).shape) == (1, 3, 132, 84)
