"""Isolated durations of the decode-step GEMM shapes (back-to-back launches, weights hot) vs their in-graph cost."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
eng = importlib.import_module("automatic-speech-recognition-asr-onnx_amd.engine")
for M in (1, 32):
    for name, N, K, epi in (("qkv", 3840, 1280, 0), ("out", 1280, 1280, 2), ("fc1", 5120, 1280, 1), ("fc2", 1280, 5120, 2), ("logits", 51968, 1280, 0)):
        ms = min(eng.op_gemm_bench(M, N, K, -1, epi, 200) for _ in range(3))
        print(f"M={M:3d} {name:6s} N={N:6d} K={K:5d}: {ms*1e3:7.2f} us  weights {N*K*2/1e6:6.1f} MB -> {N*K*2/ms/1e6:8.1f} GB/s", flush=True)
