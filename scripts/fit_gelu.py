"""Coefficients of wd_gelu (csrc/common.h, round 5):  GELU(x) = max(x, 0) - |x| exp2(P5(|x|)),  P5(a) ~ log2 Phi(-a),  P5(0) = -1.
Weighted least squares with Lawson re-weighting towards the minimax of the ABSOLUTE error of a Phi(-a); prints the fp32
coefficients, the fit error, the error of an fp32 Horner evaluation on 4 M points of [-12, 12] and the extrapolation check
(P5 must keep falling beyond the fit interval: the leading coefficient is negative).      python scripts/fit_gelu.py"""
import numpy as np
from scipy.special import erf, erfc

DEG, AMAX = 5, 6.5
a = np.linspace(0, AMAX, 80001)
phi = 0.5 * erfc(a / np.sqrt(2))
f = np.log2(phi) + 1.0                                   # c0 = -1 fixed
w = a * phi + 1e-9
w /= w.max()
V = np.vander(a, DEG + 1, increasing=True)[:, 1:]
best = (1.0, None)
for _ in range(600):
    c = np.linalg.lstsq(V * w[:, None], f * w, rcond=None)[0]
    err = np.abs(a * np.exp2(V @ c - 1.0) - a * phi)
    if err.max() < best[0]:
        best = (err.max(), c.copy())
    w = w * (1 + (err / err.max()) ** 4)
    w /= w.max()
c = np.concatenate([[-1.0], best[1]])
c32 = c.astype(np.float32)
x = np.linspace(-12, 12, 4_000_001).astype(np.float32)
ax = np.abs(x)
p = np.full_like(ax, c32[DEG])
for k in range(DEG - 1, -1, -1):
    p = (p.astype(np.float64) * ax.astype(np.float64) + np.float64(c32[k])).astype(np.float32)       # fmaf
g = (np.maximum(x, 0).astype(np.float64) - ax.astype(np.float64) * np.exp2(p.astype(np.float64)).astype(np.float32)).astype(np.float32)
ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
e32 = np.abs(g.astype(np.float64) - ref)
far = np.linspace(AMAX, 1e4, 200001)
print("coefficients c0..c5 (fp32):", ", ".join(f"{float(v):.10e}f" for v in c32))
print(f"fit: max |a Phi(-a) - a 2^P5(a)| = {best[0]:.3g};  fp32 Horner on [-12, 12]: max |GELU error| = {e32.max():.3g} at x = {x[e32.argmax()]:.3f}")
print(f"extrapolation: max P5 on [{AMAX}, 1e4] = {np.polyval(c[::-1], far).max():.1f} (at a = {far[np.polyval(c[::-1], far).argmax()]:.1f}); leading coefficient {c[-1]:.3e}")
