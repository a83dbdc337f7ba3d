"""Run one GEMM shape/variant a few times (target for rocprofv3 --pmc)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
eng = importlib.import_module("automatic-speech-recognition-asr-onnx_amd.engine")
M, N, K, v, epi = [int(x) for x in sys.argv[1:6]]
ms = eng.op_gemm_bench(M, N, K, v, epi, 20)
print(f"M={M} N={N} K={K} v{v} epi{epi}: {ms*1e3:.1f} us {2*M*N*K/ms/1e9:.0f} TF")
