"""Coefficients of the branch-free erf used by the fused FFN kernel's GELU (csrc/ffn.hip).

erf(t) = 1 - 2^(-t * Q(t)),  t = min(|x|, TMAX),  Q a degree-DEG polynomial: one v_exp_f32 and DEG FMAs.  The fit
minimises the ABSOLUTE error of erf (weight erfc(t) on the exponent error), which is what 0.5*y*(1 + erf(y/sqrt 2))
needs; the float32 evaluation error is measured with the same operation order the kernel uses.
"""
import numpy as np
from scipy.special import erf, erfc

DEG, TMAX = 7, 4.0
t = (np.cos(np.pi * (np.arange(4000) + 0.5) / 4000) * 0.5 + 0.5) * TMAX
t = t[t > 1e-6]
u = -np.log2(erfc(t)) / t  # target Q(t)
# d erf = erfc(t) * ln2 * t * dQ  -> weight
w = erfc(t) * np.log(2.0) * t
coef = None
for it in range(60):  # iteratively re-weighted least squares -> near-minimax of the weighted error
    V = np.vander(t, DEG + 1, increasing=True)
    c, *_ = np.linalg.lstsq(V * w[:, None], u * w, rcond=None)
    err = (V @ c - u) * erfc(t) * np.log(2.0) * t
    coef = c
    w = w * (1 + 3.0 * np.abs(err) / np.abs(err).max()) ** 0.5
    w *= (erfc(t) * np.log(2.0) * t).max() / w.max()
print("coefficients (Q0..Q%d):" % DEG)
for ci in coef:
    print("  %.9ef," % np.float32(ci))


def erf_f32(x, c):
    x = x.astype(np.float32)
    tt = np.minimum(np.abs(x), np.float32(TMAX))
    q = np.float32(c[-1]) * np.ones_like(tt)
    for ci in c[-2::-1]:
        q = (q * tt + np.float32(ci)).astype(np.float32)
    e = np.exp2((-(tt * q)).astype(np.float32)).astype(np.float32)
    return np.copysign((np.float32(1) - e).astype(np.float32), x)


xs = np.linspace(-6, 6, 2000001)
c32 = [np.float32(ci) for ci in coef]
ee = np.abs(erf_f32(xs, c32).astype(np.float64) - erf(xs))
print("max abs erf error (float32 evaluation): %.3e at x=%.4f" % (ee.max(), xs[ee.argmax()]))
y = xs * np.sqrt(2.0)
g_ref = 0.5 * y * (1 + erf(xs))
y32 = y.astype(np.float32)
hy = (np.float32(0.5) * y32).astype(np.float32)
g = (hy * erf_f32((y32 * np.float32(0.70710678118654752)).astype(np.float32), c32) + hy).astype(np.float32)
ge = np.abs(g.astype(np.float64) - 0.5 * y32.astype(np.float64) * (1 + erf(y32.astype(np.float64) / np.sqrt(2.0))))
print("max abs GELU error: %.3e ; max error relative to max(|gelu|, 1e-3): %.3e" % (ge.max(), (ge / np.maximum(np.abs(g_ref), 1e-3)).max()))
