"""Fit of the erf-GELU used by the bf16 / e4m3 GEMM epilogues (csrc/gemm_dev.h, gelu_erf_fast2):
    gelu(v) = max(v, 0) - a * 2^q(a),   a = min(|v|, 12),   q = degree-6 polynomial ~ log2(erfc(a / sqrt 2) / 2)
Least squares on [0, 13], re-weighted towards the minimax of the ABSOLUTE error of a * 2^q (the quantity that lands in the output). Prints the
coefficients (lowest order first) and the errors. CPU only (numpy + scipy)."""
import numpy as np
from scipy.special import erfc, erfcx

a = np.linspace(0, 13, 26001)
z = a / np.sqrt(2)
L = np.log2(0.5 * erfcx(z)) - z * z * np.log2(np.e)          # log2(erfc(z) / 2) without underflow
term = a * 0.5 * erfc(z)
wt = np.maximum(term, 1e-12)
V = np.vander(a, 7, increasing=True)
for _ in range(30):
    coef = np.linalg.lstsq(V * wt[:, None], L * wt, rcond=None)[0]
    err = np.abs(a * np.exp2(np.minimum(V @ coef, 0)) - term)
    wt = wt * (1 + 2 * err / err.max())
c32 = coef.astype(np.float32)
q = V @ c32.astype(np.float64)
err = np.abs(a * np.exp2(q) - term)
print("coefficients (a^0 .. a^6):", ", ".join("%.9g" % c for c in c32))
print("max |error| of a * 2^q: %.2e at a = %.2f; relative to the term for a < 4: %.2e; q(12) = %.1f, q(13) = %.1f" % (
    err.max(), a[err.argmax()], np.max((err / np.maximum(term, 1e-30))[a < 4]), q[np.searchsorted(a, 12)], q[-1]))
