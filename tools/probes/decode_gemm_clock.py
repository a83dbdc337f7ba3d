"""Where a decode-step GEMM launch spends its time: the phase clock of decode_gemm_kernel (DecGemmArgs::dbg_clk) over cold-weight launch chains of the
Whisper-large-v3 decoder shapes. ASR_PROBE_CLK=1 makes asr_probe_gemm_chain print the breakdown of every chain to stderr."""
import importlib, os, sys
sys.path.insert(0, ".")
os.environ["ASR_PROBE_CLK"] = "1"
probe = importlib.import_module("automatic-speech-recognition-asr-onnx_amd._probe")
shapes = [("q|k|v (LN folded)", 3840, 1280, 13), ("out / cross-out (+ residual)", 1280, 1280, 12), ("cross-q (LN folded)", 1280, 1280, 13),
          ("fc1 (LN folded, GELU)", 5120, 1280, 11), ("fc2 (+ residual)", 1280, 5120, 12)]
for M in (32, 64, 1):
    for name, N, K, epi in shapes:
        for cold in (768, 0):          # 0: every launch of the chain streams the same weights (L2 / MALL / TLB warm)
            us, kern = probe.gemm_chain(M, N, K, epi, cold, 5)
            print(f"M={M:3d} {name:30s} N={N:5d} K={K:5d} {'cold' if cold else 'hot '} {us:7.2f} us per launch  [{kern}]", flush=True)
