"""GPU probe: streaming-Paraformer GEMM shapes (64 streams x 16-row slots = 1024 rows) across tilings (run via gpurun)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
eng = importlib.import_module("automatic-speech-recognition-asr-onnx_amd.engine")
shapes = [("qkv", 1024, 1536, 512), ("out", 1024, 512, 512), ("ffn1", 1024, 2048, 512), ("ffn2", 1024, 512, 2048), ("dec ffn", 1024, 512, 2048)]
for name, M, N, K in shapes:
    row = []
    for v in (1, 2, 3, 4, 5, 6):
        try:
            best = min(eng.op_gemm_bench(M, N, K, v, 0, 30) for _ in range(3))
            row.append(f"v{v}: {best*1e3:6.1f} us")
        except Exception as e:
            row.append(f"v{v}: n/a")
    print(f"{name:8s} M={M} N={N} K={K} | " + " | ".join(row), flush=True)
