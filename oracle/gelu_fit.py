"""Test infrastructure (not shipped code): the fit behind csrc/common.h's gelu_erf_f (round 5).

    GELU(x) = x * Phi(x)  ~  x * sigmoid(x * P(min(|x|, 6))),   P of degree 4

The reference evaluates the exact-erf GELU (dp/models/activations.py:109, F.gelu with approximate='none'); the HIP
epilogues need it in as few VALU instructions as possible.  This script refits P (iteratively re-weighted least squares
towards the minimax solution, then a simplex polish on the maximum error), prints the coefficients multiplied by -log2(e)
as common.h holds them, and the maximum absolute error of the fp32 evaluation over [-10, 10]:

    python oracle/gelu_fit.py        # -> 5.1e-06 at x = 2.81

tests/test_gelu_form.py pins the constants in common.h to this error bound with numpy (no GPU).
"""
import numpy as np
from scipy.optimize import least_squares, minimize
from scipy.special import erf

CLAMP = 6.0
COEFFS_NEG_LOG2E = (-2.3031814098358154, 0.0026162799913436174, -0.1058432012796402, -0.0018446178874000907,
                    0.0014851129380986094)          # WIW_GELU_C0 .. C4 of csrc/common.h


def gelu_exact(x):
    return x * 0.5 * (1.0 + erf(x / np.sqrt(2.0)))


def gelu_sigmoid_form_f32(x, coeffs=COEFFS_NEG_LOG2E, clamp=CLAMP):
    """The arithmetic of gelu_erf_f in fp32, operation by operation (Horner from C4 down, exp2, 1 +, reciprocal, x *)."""
    x = np.asarray(x, np.float32)
    a = np.minimum(np.abs(x), np.float32(clamp))
    p = np.full_like(x, np.float32(coeffs[4]))
    for c in coeffs[3::-1]:
        p = p * a + np.float32(c)
    with np.errstate(over="ignore"):
        e = np.exp2(x * p)
        return x * (np.float32(1.0) / (np.float32(1.0) + e))


def _model(c, x):
    a = np.minimum(np.abs(x), CLAMP)
    p = np.zeros_like(x)
    for k in c[::-1]:
        p = p * a + k
    return x / (1.0 + np.exp(-x * p))


def fit(deg=4):
    X = np.linspace(-10.0, 10.0, 160001)
    G = gelu_exact(X)
    xs = X[np.abs(X) <= CLAMP + 0.5]
    gs = G[np.abs(X) <= CLAMP + 0.5]
    c = np.zeros(deg + 1)
    c[0], c[2] = 1.5957691216, 0.0713548            # the tanh form's constants as the starting point
    w = np.ones_like(xs)
    best = None
    for _ in range(100):
        c = least_squares(lambda q: (_model(q, xs) - gs) * w, c, xtol=1e-15, ftol=1e-15, gtol=1e-15).x
        err = np.abs(_model(c, X) - G).max()
        e2 = np.abs(_model(c, xs) - gs)
        w = w * (1.0 + 3.0 * e2 / e2.max())
        w /= w.mean()
        if best is None or err < best[0]:
            best = (err, c.copy())
    sc = np.abs(best[1])
    f = lambda q: np.abs(_model(q * sc, X) - G).max()
    r = minimize(f, best[1] / sc, method="Nelder-Mead", options=dict(xatol=1e-13, fatol=1e-15, maxiter=40000, maxfev=40000, adaptive=True))
    c = (r.x if r.fun < best[0] else best[1] / sc) * sc
    return c, X, G


if __name__ == "__main__":
    c, X, G = fit()
    print("P coefficients (natural units):", [float("%.9g" % v) for v in c])
    print("times -log2(e), as float32   :", [float(np.float32(v * -1.4426950408889634)) for v in c])
    e = gelu_sigmoid_form_f32(X).astype(np.float64) - G
    print("committed constants, fp32 evaluation: max |error| %.3g at x = %.2f" % (np.abs(e).max(), X[np.abs(e).argmax()]))
