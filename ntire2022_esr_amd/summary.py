"""Analytic complexity counters.

`utils/model_summary.py` counts with forward hooks on nn.Conv2d / activation modules
(:230-245, :274-300, :398-440); those hooks never fire through the fused HIP ops, so the same numbers
are derived from the op list: FLOPs = conv MACs (k*k*cin*cout per output pixel, no bias term, :274-294)
+ one per element of every ReLU-family activation output (:298-300); #Acts = numel of every Conv2d
output (:430-440); #Conv = number of Conv2d calls.  Pinned to the reference's own output
(tests/golden/summary.json) in tests/test_harness.py.
"""
from . import _lib as L
from .engine import Plan


def model_complexity(model, input_dim=(3, 256, 256)):
    c, h, w = input_dim
    plan = Plan(1, h, w)
    model._build_plan(plan, c)
    flops = 0
    acts = 0
    nconv = 0
    for o in plan.ops:
        f, a, n = model._complexity_terms(plan, o)
        flops, acts, nconv = flops + f, acts + a, nconv + n
    return {"activations": float(acts), "num_conv": int(nconv), "flops": float(flops),
            "num_parameters": int(sum(p.numel() for p in model.parameters()))}
