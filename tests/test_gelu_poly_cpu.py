"""CPU: the polynomial erf-GELU forms of the GEMM epilogues (icka_amd/csrc/common.h, gelu_f / dgelu_f) against the exact
functions of the reference (Cross_Modal_Interaction_Module.py:31-37: x * 0.5 * (1 + erf(x / sqrt(2)))), evaluated with
the header's own coefficients in float32 Horner form.  Bound: far inside the bf16 grid of the outputs (2^-9)."""
import math
import os
import re

import numpy as np

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "icka_amd", "csrc", "common.h")


def _coefficients():
    src = open(HDR).read()
    out = {}
    for name in ("gelu_f", "dgelu_f"):
        body = src[src.index("float %s(float x)" % name):]
        arr = re.search(r"const float c\[8\] = \{([^}]*)\}", body).group(1)
        out[name] = [np.float32(v.strip().rstrip("f")) for v in arr.split(",")]
        assert len(out[name]) == 8
    return out


def _odd_poly8(x, c):
    xc = np.clip(x, np.float32(-4), np.float32(4)).astype(np.float32)
    t = (xc * xc).astype(np.float32)
    p = np.full_like(x, c[7])
    for k in range(6, -1, -1):
        p = (p * t + c[k]).astype(np.float32)
    return (xc * p + np.float32(0.5)).astype(np.float32)


def test_polynomial_gelu_and_derivative_error_bounds():
    c = _coefficients()
    x = np.linspace(-12, 12, 600001).astype(np.float32)
    xd = x.astype(np.float64)
    erf = np.vectorize(math.erf)
    Phi = 0.5 * (1.0 + erf(xd / math.sqrt(2.0)))
    phi = np.exp(-xd * xd / 2.0) / math.sqrt(2.0 * math.pi)
    gelu = np.where(x < -4, np.float32(0), x * _odd_poly8(x, c["gelu_f"])).astype(np.float32)   # as gelu_f
    dgelu = _odd_poly8(x, c["dgelu_f"])
    e_phi = np.abs(_odd_poly8(x, c["gelu_f"]) - Phi)[np.abs(x) <= 4].max()
    ref = xd * Phi
    e_gelu_abs = np.abs(gelu - ref)[x < 1].max()                 # small outputs: absolute
    e_gelu_rel = (np.abs(gelu - ref) / np.abs(ref))[x >= 1].max()  # large outputs: relative
    e_dgelu = np.abs(dgelu - (Phi + xd * phi)).max()
    assert e_phi < 6e-5, e_phi
    assert e_dgelu < 3e-4, e_dgelu
    assert e_gelu_abs < 2.5e-4, e_gelu_abs
    assert e_gelu_rel < 1e-4, e_gelu_rel     # bf16 half-ulp is 2e-3
