"""Reproduces the coefficients of gelu_pk (csrc/common.h): g(v) = 0.5*erf(v/sqrt2) ~ vc * P(vc^2), vc = clamp(v, -L, L).
Weighted least-squares Chebyshev fit of P(u) = g(sqrt u)/sqrt u on u in [0, L^2]; the error reported is that of
gelu = v*(0.5 + g) evaluated with float32 Horner arithmetic over v in [-12, 12]."""
import numpy as np
from numpy.polynomial import chebyshev as Ch, polynomial as Po
from scipy.special import erf

L, N, WEND = 4.5, 10, 3


def fit():
    k = np.arange(4000)
    s = np.cos(np.pi * (k + 0.5) / 4000)
    u = (s + 1) * L * L / 2
    v = np.sqrt(u)
    y = 0.5 * erf(v / np.sqrt(2)) / v
    w = v * np.maximum(1, v) * (1 + WEND * (v / L) ** 8)
    ps = Ch.cheb2poly(Ch.chebfit(s, y, N - 1, w=w))
    pu = np.zeros(1)
    for i, c in enumerate(ps):
        pu = Po.polyadd(pu, c * Po.polypow([-1.0, 2 / (L * L)], i))
    return pu


def gelu_f32(c, v):
    v = v.astype(np.float32)
    vc = np.clip(v, -L, L).astype(np.float32)
    u = (vc * vc).astype(np.float32)
    p = np.full_like(u, np.float32(c[-1]))
    for ci in c[-2::-1]:
        p = (p * u + np.float32(ci)).astype(np.float32)
    return (v * (vc * p).astype(np.float32) + np.float32(0.5) * v).astype(np.float32)


if __name__ == "__main__":
    c = fit()
    vv = np.linspace(-12, 12, 960001)
    err = np.abs(gelu_f32(c, vv) - 0.5 * vv * (1 + erf(vv / np.sqrt(2))))
    print("coefficients (low -> high):", ", ".join("%.9ef" % x for x in c))
    print("max |gelu_pk - gelu| = %.3g at v = %.2f" % (err.max(), vv[err.argmax()]))
