"""`from utils.model_summary import get_model_activation, get_model_flops` (test_demo.py:8).

The reference counts with forward hooks on nn.Conv2d / nn.Linear / activation modules (utils/model_summary.py:230-245,
398-405); inside the fused HIP modules no such submodule is ever called, so the same numbers are derived analytically
from the op list (ntire2022_esr_amd/summary.py, pinned to the reference's own output in tests/golden/summary.json)."""
from ntire2022_esr_amd.summary import model_complexity


def get_model_flops(model, input_res, print_per_layer_stat=True, input_constructor=None):
    assert type(input_res) is tuple and len(input_res) >= 2
    return model_complexity(model, tuple(input_res))["flops"]


def get_model_activation(model, input_res, input_constructor=None):
    assert type(input_res) is tuple and len(input_res) >= 2
    c = model_complexity(model, tuple(input_res))
    return c["activations"], c["num_conv"]
