"""GPU probe: per-phase segment clocks of the ping-pong GEMM's timed instance (variant 16; ASR_PP_CLK=1 makes the bench hook print them)."""
import importlib, os, sys
os.environ["ASR_PP_CLK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
eng = importlib.import_module("automatic-speech-recognition-asr-onnx_amd.engine")
for v in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["16"])]:
    for name, M, N, K in [("sq8192", 8192, 8192, 8192), ("fc1 b64", 25600, 5120, 1280)]:
        ms = eng.op_gemm_bench(M, N, K, v, 0, 5)
        print(f"v{v} {name}: {ms*1e3:.1f} us", flush=True)
