"""GPU probe: the FP8 matrix-pipe GEMM (csrc/gemm_fp8.hip) -- correctness against float64 on the decoded e4m3 bytes, then microseconds / TF on the
Whisper-large-v3 encoder FFN shapes next to the bf16 ping-pong kernel."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
eng = importlib.import_module("automatic-speech-recognition-asr-onnx_amd.engine")
probe = importlib.import_module("automatic-speech-recognition-asr-onnx_amd._probe")
tab = probe.e4m3_table()


def rand_e4m3(rng, shape, spread):
    """bytes whose values are roughly N(0, spread), rounded to e4m3 through the table (nearest)."""
    x = rng.standard_normal(shape) * spread
    finite = np.where(np.isnan(tab), np.inf, tab)
    idx = np.abs(x[..., None] - finite[None, :]).argmin(-1) if x.size < 2_000_000 else None
    if idx is None:
        order = np.argsort(finite); vals = finite[order]; vals = vals[np.isfinite(vals)]; order = order[:vals.size]
        pos = np.clip(np.searchsorted(vals, x), 1, vals.size - 1)
        idx = np.where(np.abs(x - vals[pos - 1]) <= np.abs(x - vals[pos]), order[pos - 1], order[pos])
    return idx.astype(np.uint8)


ok = True
rng = np.random.default_rng(0)
for (M, N, K) in [(256, 256, 256), (300, 512, 512), (1000, 1280, 1280), (2048, 1280, 5120)]:
    a8, w8 = rand_e4m3(rng, (M, K), 1.0), rand_e4m3(rng, (N, K), 32.0)
    sc = 2.0 ** rng.integers(-12, -8, N).astype(np.float64)
    bias = rng.standard_normal(N)
    add = rng.standard_normal((M, N)).astype(np.float32)
    ref = (tab[a8] @ tab[w8].T) * sc[None, :] + bias[None, :]
    mag = (np.abs(tab[a8]) @ np.abs(tab[w8]).T) * sc[None, :]
    got, _ = probe.gemm_fp8(a8, w8, sc, bias, add=add)
    err = np.abs(got - (ref + add)) / mag
    out8, _ = probe.gemm_fp8(a8, w8, sc, bias, act=2)
    gelu = 0.5 * ref * (1.0 + np.vectorize(__import__("math").erf)(ref / np.sqrt(2.0))) if M * N < 400_000 else None
    msg = f"M={M} N={N} K={K}: f32 out max |err| / sum|terms| {err.max():.2e}"
    good = err.max() < 2e-5
    if gelu is not None:
        d8 = np.abs(tab[out8] - np.clip(gelu, -448, 448))
        rel = d8 / np.maximum(np.abs(gelu), 2.0 ** -6)
        msg += f"; byte out max rel err {rel.max():.3f} (half an e4m3 ulp = 0.0625)"
        good &= rel.max() < 0.07
    print(msg, "OK" if good else "FAIL", flush=True)
    ok &= good
print("CORRECTNESS", "PASS" if ok else "FAIL", flush=True)
for name, M, N, K in [("fc1 b64", 25600, 5120, 1280), ("fc2 b64", 25600, 1280, 5120), ("fc1 b32x30", 48000, 5120, 1280), ("sq8192", 8192, 8192, 8192)]:
    a8, w8 = rng.integers(0, 120, (M, K), dtype=np.uint8), rng.integers(0, 120, (N, K), dtype=np.uint8)
    a8 |= (rng.integers(0, 2, (M, K), dtype=np.uint8) << 7); w8 |= (rng.integers(0, 2, (N, K), dtype=np.uint8) << 7)
    sc, bias = np.ones(N), np.zeros(N)
    _, us8 = probe.gemm_fp8(a8, w8, sc, bias, act=2, iters=20)
    _, usf = probe.gemm_fp8(a8, w8, sc, bias, add=np.zeros((M, N), np.float32), iters=20)
    bf = min(eng.op_gemm_bench(M, N, K, 8, 0, 20) for _ in range(2)) * 1e3
    print(f"{name:10s} M={M} N={N} K={K} | fp8 -> bytes {us8:7.1f} us {2*M*N*K/us8/1e6:6.0f} TF | fp8 + residual -> f32 {usf:7.1f} us {2*M*N*K/usf/1e6:6.0f} TF | bf16 ping-pong {bf:7.1f} us {2*M*N*K/bf/1e6:6.0f} TF", flush=True)
