"""GPU probe (run via gpurun): the ping-pong 256 x 256 GEMM (variant 8) -- correctness against float64 on bf16-rounded operands, then
microseconds / TF on the Whisper-large-v3 encoder and Qwen3-ASR prefill shapes next to the older tilings and to torch.matmul (yardstick only:
torch is never a product dependency)."""
import importlib, os, sys, time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
eng = importlib.import_module("automatic-speech-recognition-asr-onnx_amd.engine")
probe = importlib.import_module("automatic-speech-recognition-asr-onnx_amd._probe")


def bf16r(x):
    return probe._bf16_to_f32(probe._bf16_bits(x))


def check(M, N, K, variant, act=0, bias=True, add=False, seed=0):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((M, K), dtype=np.float32)
    w = rng.standard_normal((N, K), dtype=np.float32) * 0.05
    b = rng.standard_normal((N,), dtype=np.float32) if bias else None
    r = rng.standard_normal((M, N), dtype=np.float32) if add else None
    out, kern = probe.gemm(a, w, bias=b, add=r, act=act, variant=variant, want=("f32",) if add else ("lo",))
    ref = bf16r(a).astype(np.float64) @ bf16r(w).astype(np.float64).T
    if b is not None: ref += b
    if r is not None: ref += r
    if act == 1: ref = np.maximum(ref, 0)
    got = out["f32"] if add else out["lo"]
    tol = 2e-4 * np.abs(ref).max() + (0 if add else np.abs(ref).max() * 2 ** -8)
    err = np.abs(got - ref).max()
    print(f"check M={M} N={N} K={K} act={act} add={add}: kernel={kern} max err {err:.3e} (tol {tol:.3e}) {'OK' if err <= tol else 'FAIL'}", flush=True)
    return err <= tol


ok = True
for (M, N, K) in [(256, 256, 128), (512, 512, 256), (300, 256, 384), (1000, 768, 1280), (2048, 1280, 5120)]:
    ok &= check(M, N, K, 8)
    ok &= check(M, N, K, 8, add=True, seed=1)
    ok &= check(M, N, K, 12, seed=3)                  # 32 x 32 x 16 MFMA variant
    ok &= check(M, N, K, 16, add=True, seed=4)
    ok &= check(M, N, K, 108, seed=5)                 # persistent form
    ok &= check(M, N, K, 108, add=True, seed=6)
ok &= check(9000, 2560, 1280, 108, seed=7)            # several tiles per workgroup
ok &= check(9000, 2560, 1280, 108, add=True, seed=8)
ok &= check(777, 512, 512, 8, act=1, seed=2)
print("CORRECTNESS", "PASS" if ok else "FAIL", flush=True)

shapes = [("wh8 qk b64", 25600, 2560, 1280), ("wh8 out b64", 25600, 1280, 1280), ("wh8 fc1 b64", 25600, 5120, 1280), ("wh8 fc2 b64", 25600, 1280, 5120),
          ("wh8 fc1 b32", 12800, 5120, 1280), ("wh8 fc2 b32", 12800, 1280, 5120), ("wh8 fc1 26624", 26624, 5120, 1280),
          ("wh30 fc1", 48000, 5120, 1280), ("wh30 fc2", 48000, 1280, 5120), ("wh30 out", 48000, 1280, 1280),
          ("qw qkv", 8192, 4096, 1024), ("qw gate_up", 8192, 6144, 1024), ("qw down", 8192, 1024, 3072), ("sq 8192", 8192, 8192, 8192)]
try:
    import torch
except Exception:
    torch = None
for name, M, N, K in shapes:
    row = []
    for v, ep in ((8, 0), (108, 0), (8, 2), (108, 2)):
        try:
            best = min(eng.op_gemm_bench(M, N, K, v, ep, 20) for _ in range(3))
            row.append(f"v{v}/e{ep}: {best*1e3:7.1f} us {2*M*N*K/best/1e9:6.0f} TF")
        except Exception as e:
            row.append(f"v{v}/e{ep}: n/a ({str(e)[:40]})")
    if torch is not None:
        a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16(); w = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
        for _ in range(3): torch.matmul(a, w.t())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): torch.matmul(a, w.t())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        row.append(f"torch: {dt*1e6:7.1f} us {2*M*N*K/dt/1e12:6.0f} TF")
    print(f"{name:14s} M={M} N={N} K={K} | " + " | ".join(row), flush=True)
