"""CPU: closed-form approximations baked into the HIP epilogues, evaluated here in float32 from the constants as they stand in the source."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast_erf_gelu_constants_in_gemm_dev_h():
    """gelu_erf_fast2 (csrc/gemm_dev.h): gelu(v) = max(v, 0) - a 2^q(a), a = min(|v|, 12). The constants are read from the header, the formula is
    evaluated in float32 the way the kernel does (Horner, highest order first) and compared with the exact erf-GELU in float64
    (the reference's nn.GELU(), Whisper/Export_Whisper.py:430-447): absolute error within the f32 rounding of the result (6e-8 from the fit itself), relative
    error below 1e-3 -- a quarter of a bf16 ulp -- wherever |gelu| > 1e-4 and below one bf16 ulp down to |gelu| = 1e-6 (v = -5)."""
    from math import erf
    src = open(os.path.join(ROOT, "automatic-speech-recognition-asr-onnx_amd", "csrc", "gemm_dev.h")).read()
    body = src[src.index("void gelu_erf_fast2"):src.index("float gelu_erf_fast(float v)")]
    lead = re.search(r"q = a \* ([-0-9.e+]+)f \+ ([-0-9.e+]+)f;", body)
    rest = re.findall(r"q = q \* a \+ ([-0-9.e+]+)f;", body)
    clamp = float(re.search(r"fminf\(fabsf\(v0\), ([0-9.]+)f\)", body).group(1))
    coef = [np.float32(lead.group(1)), np.float32(lead.group(2))] + [np.float32(c) for c in rest]
    assert len(coef) == 7 and clamp == 12.0
    v = np.concatenate([np.linspace(-40, 40, 400001), np.linspace(-1, 1, 100001), [0.0, -0.0, 1e-20, -1e-20, 300.0, -300.0]]).astype(np.float32)
    a = np.minimum(np.abs(v), np.float32(clamp))
    q = coef[0] * a + coef[1]
    for c in coef[2:]:
        q = q * a + c
    got = (np.maximum(v, np.float32(0)) - a * np.exp2(q)).astype(np.float64)
    want = np.array([0.5 * x * (1.0 + erf(x / np.sqrt(2.0))) for x in v.astype(np.float64)])
    assert np.isfinite(got).all()
    err = np.abs(got - want)
    assert (err <= 1.2e-7 + 2.0 ** -23 * np.abs(want)).all()
    for floor, tol in ((1e-4, 1e-3), (1e-6, 2.0 ** -8)):
        big = np.abs(want) > floor
        assert (err[big] / np.abs(want)[big]).max() < tol
