import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
eng = importlib.import_module("automatic-speech-recognition-asr-onnx_amd.engine")
M = 9216
for name, N, K, epi in [("qkv", 1536, 512, 4), ("out", 512, 512, 3), ("ffn1", 2048, 512, 1), ("ffn2", 512, 2048, 2)]:
    for v in (1, 2, 3, 4):
        row = []
        for dbg, tag in ((0, "full"), (4, "noepi"), (5, "norefill+noepi"), (6, "nomfma+noepi"), (7, "launch only")):
            best = min(eng.op_gemm_bench(M, N, K, v | (dbg << 8), epi, 30) for _ in range(3))
            row.append(f"{tag}: {best*1e3:6.1f}us" + (f" {2*M*N*K/best/1e9:5.0f}TF" if dbg == 0 else ""))
        print(f"{name} N={N} K={K} v{v} | " + " | ".join(row), flush=True)
