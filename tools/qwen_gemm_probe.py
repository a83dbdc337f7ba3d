"""GPU probe: decode-step GEMM shapes of Qwen3-ASR-0.6B (M = batch) across kernel variants (run via gpurun)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
eng = importlib.import_module("automatic-speech-recognition-asr-onnx_amd.engine")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
shapes = [("qkv", 4096, 1024), ("wo", 1024, 2048), ("gate_up", 6144, 1024), ("down", 1024, 3072), ("lm_head", 151936, 1024)]
for name, N, K in shapes:
    row = []
    for v in (-1, 2, 4):
        try:
            best = min(eng.op_gemm_bench(M, N, K, v, 0, 30) for _ in range(3))
            row.append(f"v{v}: {best*1e3:7.1f} us {N*K*2/best/1e6:7.0f} GB/s")
        except Exception as e:
            row.append(f"v{v}: n/a")
    print(f"{name:8s} M={M} N={N} K={K} | " + " | ".join(row), flush=True)
