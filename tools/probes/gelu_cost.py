import importlib, sys
sys.path.insert(0, ".")
eng = importlib.import_module("automatic-speech-recognition-asr-onnx_amd.engine")
for name, M, N, K in [("fc1 b64", 25600, 5120, 1280), ("fc1 b32x30", 48000, 5120, 1280)]:
    for ep, what in ((0, "bias -> bf16"), (1, "relu"), (7, "erf-GELU")):
        best = min(eng.op_gemm_bench(M, N, K, 8, ep, 20) for _ in range(3))
        print(f"{name} {what:14s} {best*1e3:7.1f} us {2*M*N*K/best/1e9:6.0f} TF", flush=True)
