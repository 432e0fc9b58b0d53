"""Coefficients of common.h geglu_scaled: erfc(z) = 2^P(t), t = |gs| = z sqrt(log2 e), P a degree-6 polynomial through the origin
fitted by least squares on Chebyshev nodes of z in [0, 6] with weight erfc(z) (minimises the absolute error of erfc); then an fp32
emulation of x * GELU(g) in this form and in the Abramowitz-Stegun 7.1.26 form against float64.   python tools/fit_erfc_poly.py"""
import numpy as np
from scipy.special import erfc, erf

c = np.sqrt(np.log2(np.e))
deg, zmax = 6, 6.0
k = np.arange(6000)
z = 0.5 * zmax * (1 - np.cos(np.pi * (k + 0.5) / 6000))
t = z * c
y, w = np.log2(erfc(z)), erfc(z)
V = np.vander(t, deg + 1, increasing=True)
coef = np.concatenate([[0.0], np.linalg.lstsq(V[:, 1:] * w[:, None], y * w, rcond=None)[0]])
print("clamp T =", repr(np.float32(zmax * c)))
print("c1..c6 :", [repr(np.float32(v)) for v in coef[1:]])
zz = np.linspace(0, zmax, 200001)
print("max |2^P - erfc| (float64):", np.abs(np.exp2(np.polynomial.polynomial.polyval(zz * c, coef)) - erfc(zz)).max())

rng = np.random.default_rng(0)
g = np.concatenate([rng.normal(0, 2.5, 2_000_000), np.linspace(-12, 12, 200001)]).astype(np.float32)
x = rng.normal(0, 2.0, g.shape).astype(np.float32)
GS = np.float32(0.70710678118654752 * 1.2011224087864498); XS = np.float32(0.5) / GS
xs, gs = x * XS, g * GS
cf = [np.float32(v) for v in coef]
tt = np.minimum(np.abs(gs), np.float32(zmax * c)).astype(np.float32)
P = np.full_like(tt, cf[6])
for i in (5, 4, 3, 2, 1):
    P = (P * tt + cf[i]).astype(np.float32)
P = (P * tt).astype(np.float32)
ha = (xs * np.abs(gs)).astype(np.float32)
hs = (xs * gs + ha).astype(np.float32)
o = (hs - ha * np.exp2(P).astype(np.float32)).astype(np.float32)
ref = x.astype(np.float64) * (0.5 * g.astype(np.float64) * (1 + erf(g.astype(np.float64) / np.sqrt(2))))
tq = (1.0 / (1 + np.float32(0.3275911 / 1.2011224087864498) * np.abs(gs))).astype(np.float32)
poly = np.float32(1.061405429)
for a in (-1.453152027, 1.421413741, -0.284496736, 0.254829592):
    poly = (poly * tq + np.float32(a)).astype(np.float32)
o2 = (hs - ha * ((poly * tq).astype(np.float32) * np.exp2(-(gs * gs)).astype(np.float32))).astype(np.float32)
for name, v in (("2^P form", o), ("A-S 7.1.26 form", o2)):
    e = np.abs(v - ref)
    print(f"{name:16s}: max |err| {e.max():.3e}   max |err| / (|x g| + 1e-3) {(e / (np.abs(x * g) + 1e-3)).max():.3e}")
