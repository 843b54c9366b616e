"""Coefficients of the polynomial pair that replaces the eleven bandwidths alpha >= 10 of the MMD kernel sum
   f(s) = sum_a exp(-s / (2 alpha_a)),  fp(s) = sum_a exp(-s / (2 alpha_a)) / (2 alpha_a),   s = d^2 in [0, SMAX],
in the variable t = s / (SMAX/2) - 1 (Chebyshev interpolation in fp64, converted to the monomial basis, rounded to fp32).
Prints the C arrays pasted into scarches_amd/csrc/mmd.hip and the errors of an fp32 Horner evaluation against fp64.
usage: python scripts/mmd_poly_fit.py [degree] [smax]"""
import sys
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P

deg = int(sys.argv[1]) if len(sys.argv) > 1 else 11
smax = float(sys.argv[2]) if len(sys.argv) > 2 else 64.0
# 1 / (2 alpha) in float32 from float32 alphas, as the reference (torch.FloatTensor(alphas)) and MMD_BETA build them
alphas = np.array([10, 15, 20, 25, 30, 35, 100, 1e3, 1e4, 1e5, 1e6], dtype=np.float32)
beta = (np.float32(1.0) / (np.float32(2.0) * alphas)).astype(np.float64)


def f(s):
    return np.exp(-np.outer(s, beta)).sum(1)


def fp(s):
    return (np.exp(-np.outer(s, beta)) * beta).sum(1)


def fit(fun):
    k = np.arange(deg + 1)
    t = np.cos(np.pi * (k + 0.5) / (deg + 1))          # Chebyshev nodes
    s = (t + 1.0) * smax / 2.0
    c = C.chebfit(t, fun(s), deg)
    return C.cheb2poly(c)                              # monomial coefficients in t, ascending


def horner32(co, t):
    acc = np.full_like(t, np.float32(co[-1]), dtype=np.float32)
    for c in co[-2::-1]:
        acc = np.float32(acc * t + np.float32(c))       # (fma on the device: one rounding less)
    return acc


s = np.linspace(0.0, smax, 200001)
t32 = (s.astype(np.float32) * np.float32(2.0 / smax) - np.float32(1.0)).astype(np.float32)
for name, fun in (("K", f), ("Kp", fp)):
    co = fit(fun)
    ref = fun(s)
    e64 = np.abs(P.polyval(s * 2.0 / smax - 1.0, co) - ref).max()
    e32 = np.abs(horner32(co.astype(np.float32), t32).astype(np.float64) - ref)
    print(f"// {name}: degree {deg} on [0, {smax:g}]: fit error {e64:.2e}, fp32 Horner max {e32.max():.2e} "
          f"rms {np.sqrt((e32 ** 2).mean()):.2e}; f ranges {ref.min():.4g} .. {ref.max():.4g}")
    print(f"constexpr float MMD_PC_{name}[{deg + 1}] = {{" + ", ".join(f"{np.float32(c):.9e}f" for c in co) + "};")
