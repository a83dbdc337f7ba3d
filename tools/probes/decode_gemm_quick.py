"""Whisper-large-v3 decode GEMM shapes as cold-weight launch chains: generic weight-streaming kernel vs the decode GEMM."""
import importlib, sys
sys.path.insert(0, ".")
probe = importlib.import_module("automatic-speech-recognition-asr-onnx_amd._probe")
for M in (32, 64, 1):
    for name, N, K, old, new in [("qkv (LN)", 3840, 1280, 3, 13), ("out / cross-out", 1280, 1280, 2, 12), ("cross-q (LN)", 1280, 1280, 3, 13),
                                 ("fc1 (LN, gelu)", 5120, 1280, 1, 11), ("fc2", 1280, 5120, 2, 12)]:
        if M > 32 and old == 3:
            old = 0
        a, ka = probe.gemm_chain(M, N, K, old, 768, 5)
        b, kb = probe.gemm_chain(M, N, K, new, 768, 5)
        print(f"M={M:3d} {name:16s} generic {a:6.2f} us ({ka:11s})   decode gemm {b:6.2f} us ({kb})", flush=True)
