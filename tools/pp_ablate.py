"""GPU probe: timing-only ablations of the ping-pong GEMM (dbg bits: 1 no operand stream, 2 no MFMA, 4 no fragment reads)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
eng = importlib.import_module("automatic-speech-recognition-asr-onnx_amd.engine")
var = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name, M, N, K in [("sq8192", 8192, 8192, 8192), ("fc1 b64", 25600, 5120, 1280), ("fc2 b64", 25600, 1280, 5120)]:
    row = []
    for dbg in (0, 1, 2, 4, 3, 5, 6, 7):
        best = min(eng.op_gemm_bench(M, N, K, var | (dbg << 8), 0, 10) for _ in range(3))
        row.append(f"dbg{dbg}: {best*1e3:7.1f} us")
    print(f"v{var} {name:8s} | " + " | ".join(row), flush=True)
