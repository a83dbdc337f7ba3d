"""Decode GEMM shapes of Whisper-large-v3 / Qwen3-ASR-0.6B as cold-weight launch chains: us per launch vs the HBM streaming time."""
import importlib, sys
sys.path.insert(0, ".")
probe = importlib.import_module("automatic-speech-recognition-asr-onnx_amd._probe")
shapes = [("whisper qkv (LN)", 3840, 1280, 3), ("whisper out / cross-out", 1280, 1280, 2), ("whisper cross-q (LN)", 1280, 1280, 3),
          ("whisper fc1", 5120, 1280, 1), ("whisper fc2", 1280, 5120, 2), ("qwen qkv", 4096, 1024, 0), ("qwen down", 1024, 3072, 2)]
for M in (32, 64, 16, 1):
    for name, N, K, epi in shapes:
        if epi == 3 and M > 32:
            continue
        for cold in (768, 0):
            us, kern = probe.gemm_chain(M, N, K, epi, cold, 5)
            mb = N * K * 2 / 1e6
            print(f"M={M:3d} {name:26s} N={N:5d} K={K:5d} {'cold' if cold else 'hot '} {us:7.2f} us  {kern:12s} weights {mb:5.1f} MB = {mb / 5.0:5.2f} us at 5 TB/s", flush=True)
