"""GPU probe: A/B of the ping-pong GEMM's tuning variants (interleaved rounds, best of N), bf16-out epilogue."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
eng = importlib.import_module("automatic-speech-recognition-asr-onnx_amd.engine")
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8, 9, 10, 11]
shapes = [("sq8192", 8192, 8192, 8192), ("fc1 b64", 25600, 5120, 1280), ("fc2 b64", 25600, 1280, 5120), ("out b64", 25600, 1280, 1280)]
for name, M, N, K in shapes:
    best = {v: 1e9 for v in variants}
    for _ in range(4):
        for v in variants:
            try:
                best[v] = min(best[v], eng.op_gemm_bench(M, N, K, v, 0, 10))
            except Exception as e:
                best[v] = float("nan")
    print(f"{name:8s} | " + " | ".join(f"v{v}: {best[v]*1e3:7.1f} us {2*M*N*K/best[v]/1e9:5.0f} TF" for v in variants), flush=True)
