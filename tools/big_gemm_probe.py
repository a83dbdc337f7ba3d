"""GPU probe: large-M encoder / prefill GEMM shapes (Whisper-large-v3 8 s x 32, Qwen3-ASR prefill) across tilings (run via gpurun)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
eng = importlib.import_module("automatic-speech-recognition-asr-onnx_amd.engine")
shapes = [("wh qk", 12800, 2560, 1280), ("wh out", 12800, 1280, 1280), ("wh fc1", 12800, 5120, 1280), ("wh fc2", 12800, 1280, 5120),
          ("wh30 fc1", 48000, 5120, 1280), ("qw qkv", 8192, 4096, 1024), ("qw gate_up", 8192, 6144, 1024), ("qw down", 8192, 1024, 3072)]
for name, M, N, K in shapes:
    row = []
    for v in (4, 2, 6, 7):
        try:
            best = min(eng.op_gemm_bench(M, N, K, v, 0, 20) for _ in range(3))
            row.append(f"v{v}: {best*1e3:7.1f} us {2*M*N*K/best/1e9:6.0f} TF")
        except Exception as e:
            row.append(f"v{v}: n/a")
    print(f"{name:10s} M={M} N={N} K={K} | " + " | ".join(row), flush=True)
