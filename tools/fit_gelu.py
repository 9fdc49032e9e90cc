#!/usr/bin/env python3
"""Minimax fit behind gelu16() (csrc/esr_s16.hip, esr_bsconv.hip): Phi(x) - 0.5 = 0.5 erf(x / sqrt 2) ~ x P(x^2) on [0, 4],
P of degree 7 (Lawson-weighted least squares on Chebyshev nodes), then the fp32 Horner evaluation is checked against the exact
GELU on [-8, 8].  Prints the coefficients (low -> high) and the error bounds quoted in the kernels."""
import math
import numpy as np

erf = np.vectorize(math.erf)
L, deg = 4.0, 8
xs = np.cos(np.pi * (np.arange(6000) + 0.5) / 6000) * L / 2 + L / 2
f = 0.5 * erf(xs / np.sqrt(2))
A = np.stack([xs ** (2 * k + 1) for k in range(deg)], 1)
w = np.ones_like(xs)
for _ in range(200):
    c, *_ = np.linalg.lstsq(A * w[:, None], f * w, rcond=None)
    e = np.abs(A @ c - f)
    w = w * (e / e.max() + 1e-3) ** 0.5
    w /= w.max()
print("coefficients low->high:", ", ".join("%.9ef" % v for v in c))
xx = np.linspace(0, L, 200001)
print("max |Phi error| on [0, 4]: %.2e" % np.abs(np.stack([xx ** (2 * k + 1) for k in range(deg)], 1) @ c - 0.5 * erf(xx / np.sqrt(2))).max())
x = np.linspace(-8, 8, 400001).astype(np.float32)
xc = np.clip(x, -4, 4)
t = xc * xc
p = np.float32(c[-1]) * np.ones_like(x)
for k in range(deg - 2, -1, -1):
    p = p * t + np.float32(c[k])
g = np.maximum(x, np.float32(-4)) * (np.float32(0.5) + xc * p)
ge = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
print("max |gelu16 - gelu| on [-8, 8] (fp32 Horner): %.2e at x = %.2f" % (np.abs(g - ge).max(), x[np.abs(g - ge).argmax()]))
