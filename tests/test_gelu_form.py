"""The fitted GELU form of csrc/common.h (round 5): its constants, parsed from the header, against the exact-erf GELU the
reference evaluates (dp/models/activations.py:109).  CPU only: numpy restatement of the kernel arithmetic (oracle/gelu_fit.py)."""
import os
import re

import numpy as np

from oracle.gelu_fit import CLAMP, COEFFS_NEG_LOG2E, gelu_exact, gelu_sigmoid_form_f32

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "world-in-world_amd", "csrc", "common.h")


def header_constants():
    src = open(HDR).read()
    cs = [float(re.search(r"#define WIW_GELU_C%d \((-?[0-9.eE+-]+)f\)" % i, src).group(1)) for i in range(5)]
    clamp = float(re.search(r"#define WIW_GELU_CLAMP ([0-9.]+)f", src).group(1))
    return cs, clamp


def test_header_holds_the_fitted_constants():
    cs, clamp = header_constants()
    assert clamp == CLAMP
    assert np.allclose(cs, COEFFS_NEG_LOG2E, rtol=0, atol=0)


def test_error_bound_of_the_form_in_fp32():
    cs, clamp = header_constants()
    x = np.linspace(-12.0, 12.0, 480001)
    err = np.abs(gelu_sigmoid_form_f32(x, cs, clamp).astype(np.float64) - gelu_exact(x))
    assert err.max() <= 5.2e-6, err.max()
    # relative to half an fp16 ulp of the result (what the 16-bit hidden tensor resolves): 1/6 of it where |h| >= 0.1,
    # 1/30 where |h| >= 0.3; one half-ulp in the negative tail (h = -0.01 at x = -2.7: 4e-6 absolute on a value that small)
    g = np.abs(gelu_exact(x))
    for floor, frac in ((1e-2, 1.1), (0.1, 0.16), (0.3, 0.03)):
        big = g >= floor
        half_ulp16 = np.exp2(np.floor(np.log2(g[big])) - 11)
        assert (err[big] / half_ulp16).max() < frac, (floor, (err[big] / half_ulp16).max())


def test_saturation_and_specials():
    cs, clamp = header_constants()
    x = np.array([0.0, -0.0, 6.0, -6.0, 30.0, -30.0, 65504.0, -65504.0, 1e30, -1e30], np.float32)
    y = gelu_sigmoid_form_f32(x, cs, clamp)
    assert np.all(np.isfinite(y))
    assert y[0] == 0 and y[1] == 0
    assert np.allclose(y[2::2], x[2::2], rtol=1e-6)          # x >> 0: identity
    assert np.all(np.abs(y[3::2]) <= 4e-8)                     # x << 0: zero (gelu(-6) = -5.9e-9; the form gives -3.3e-8)
    assert np.isnan(gelu_sigmoid_form_f32(np.array([np.nan], np.float32), cs, clamp))[0]
