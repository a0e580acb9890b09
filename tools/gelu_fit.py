#!/usr/bin/env python3
"""Fits the polynomial of csrc/gelu.h and checks the fp32 evaluation order the kernel uses (runs anywhere, needs scipy + torch).

   GELU(v) = v * (1 - erfc(t)/2) for v >= 0,  v * erfc(t)/2 for v < 0,  t = min(|v|/sqrt 2, 5),  erfc(t) = 2^(-t q(t))

q is a weighted least-squares / Remez-style fit of g(t) = -log2(erfc(t)) / t on [0, 5]; the weight is the absolute error the
approximation leaves in the GELU output, 0.5 sqrt2 t^2 erfc(t) ln2 |dq| (plus a small floor so the far tail keeps ~1e-3
relative accuracy).  The script prints the fp32 coefficients, then evaluates the kernel's exact operation sequence (fp32 FMA
Horner, one exp2, the sign select) on a dense grid and compares with the fp64 function and with torch's fp32 GELU."""
import sys

import numpy as np
import torch
from scipy import special

T = 5.0
f32 = np.float32


def g(t):
    t = np.asarray(t, dtype=np.float64)
    out = np.full_like(t, 2 / np.sqrt(np.pi) / np.log(2))
    nz = t >= 1e-8
    out[nz] = -(special.log_ndtr(-t[nz] * np.sqrt(2)) + np.log(2)) / np.log(2) / t[nz]      # erfc(t) = 2 ndtr(-t sqrt 2)
    return out


def weight(t):
    return 0.5 * np.sqrt(2) * t * t * special.erfc(t) * np.log(2) + 1e-7 * t * np.log(2)


def fit(deg, iters=60):
    t = np.sort((np.cos(np.linspace(0, np.pi, 4000)) + 1) * T / 2)
    w0 = weight(t)
    w = w0.copy()
    V = np.polynomial.chebyshev.chebvander(2 * t / T - 1, deg)
    y = g(t)
    for _ in range(iters):                       # push the weight towards the points with the largest error (equi-ripple)
        c, *_ = np.linalg.lstsq(V * w[:, None], y * w, rcond=None)
        err = np.abs(V @ c - y) * w0
        w = w * (1 + 2.0 * err / err.max())
        w /= w.max() / w0.max()
    mono = np.polynomial.chebyshev.Chebyshev(c, domain=[0, T]).convert(kind=np.polynomial.Polynomial).coef
    return mono, float((np.abs(V @ c - y) * w0).max())


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def gelu_kernel_order(v, coef):
    v = v.astype(f32)
    z = (v * f32(0.70710678118654752440)).astype(f32)
    t = np.minimum(np.abs(z), f32(T)).astype(f32)
    c = coef.astype(f32)
    q = np.full_like(t, c[-1])
    for k in range(len(c) - 2, -1, -1):
        q = fma(q, t, np.full_like(t, c[k]))
    u = (t * q).astype(f32)
    h = (f32(0.5) * np.exp2(-u.astype(np.float64)).astype(f32)).astype(f32)
    return (v * np.where(z >= 0, (f32(1.0) - h).astype(f32), h)).astype(f32)


def main():
    deg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    coef, e = fit(deg)
    print(f"degree {deg}: weighted approximation error {e:.2e}")
    print("coefficients (fp32, c0 first):", ", ".join(repr(float(f32(c))) + "f" for c in coef))
    rng = np.random.default_rng(0)
    v = np.concatenate([np.linspace(-12, 12, 4_000_001), rng.standard_normal(2_000_000) * 2,
                        rng.standard_normal(1_000_000) * 0.1]).astype(f32)
    true = 0.5 * v.astype(np.float64) * special.erfc(-v.astype(np.float64) / np.sqrt(2))
    for name, y in (("gelu.h order", gelu_kernel_order(v, coef)), ("torch fp32", torch.nn.functional.gelu(torch.from_numpy(v)).numpy())):
        ae = np.abs(y.astype(np.float64) - true)
        print(f"{name:13s} against fp64: max abs {ae.max():.3e} at v = {v[ae.argmax()]:.4f}, rms {np.sqrt((ae ** 2).mean()):.3e}")


if __name__ == "__main__":
    main()
