"""Study (test infrastructure): fit of GeluQ5 (smalltts_amd/csrc/common.hpp) — gelu(x) = max(x, 0) - |x| 2^q(|x|), q of degree 5.

g(a) = Phi(-a) = erfc(a / sqrt 2) / 2; the fit minimises the maximum ABSOLUTE error of a g(a) (that is the gelu error) over a in
[0, 9] by iteratively re-weighted least squares on the coefficients of q, then checks the fp32 evaluation on [-40, 40].
    python -m tests.studies.gelu_q5_fit
"""
import numpy as np
from scipy.optimize import least_squares
from scipy.special import erf, erfc


def fit(deg=5, amax=9.0):
    a = np.linspace(0, amax, 30001)
    g = 0.5 * erfc(a / np.sqrt(2))
    tgt = a * g
    A = np.vander(a, deg + 1, increasing=True)
    w = np.maximum(tgt, 1e-6)
    c = np.linalg.lstsq(A * w[:, None], np.log2(np.maximum(g, 1e-300)) * w, rcond=None)[0]
    res = lambda c: a * np.exp2(A @ c) - tgt
    c = least_squares(res, c, xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=20000).x
    for _ in range(60):
        e = res(c)
        wgt = (np.abs(e) / np.abs(e).max()) ** 2 + 0.05
        c = least_squares(lambda c: res(c) * wgt, c, xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=5000).x
    return c.astype(np.float32)


def fp32_error(c32):
    x = np.linspace(-40, 40, 4000001).astype(np.float32)
    ax = np.abs(x)
    q = np.full_like(x, c32[-1])
    for k in range(len(c32) - 2, -1, -1):
        q = (q * ax + c32[k]).astype(np.float32)
    y = (np.maximum(x, 0) - ax * np.exp2(q).astype(np.float32)).astype(np.float32)
    xd = x.astype(np.float64)
    return float(np.abs(y - 0.5 * xd * (1 + erf(xd / np.sqrt(2)))).max())


if __name__ == "__main__":
    c = fit()
    print("coefficients (fp32):", [repr(float(v)) for v in c])
    print("max |gelu error| in fp32 on [-40, 40]:", fp32_error(c))
