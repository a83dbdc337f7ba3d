"""GPU probe: cold decode-GEMM chains with and without the side-branch weight prefetch (ASR_PROBE_PREFETCH=1), next to the hot chain (the bound)."""
import importlib, os, sys
sys.path.insert(0, ".")
probe = importlib.import_module("automatic-speech-recognition-asr-onnx_amd._probe")
shapes = [("qkv (LN fold)", 3840, 1280, 13), ("out / cross-out (+res)", 1280, 1280, 12), ("cross-q (LN fold)", 1280, 1280, 13),
          ("fc1 (LN fold, GELU)", 5120, 1280, 11), ("fc2 (+res)", 1280, 5120, 12)]
for M in (1, 32, 64):
    tot = [0.0, 0.0, 0.0]
    for name, N, K, epi in shapes:
        os.environ["ASR_PROBE_PREFETCH"] = "0"
        cold, kern = probe.gemm_chain(M, N, K, epi, 768, 5)
        hot, _ = probe.gemm_chain(M, N, K, epi, 0, 5)
        os.environ["ASR_PROBE_PREFETCH"] = "1"
        pre, _ = probe.gemm_chain(M, N, K, epi, 768, 5)
        w = 2 if "out" in name else 1
        for i, v in enumerate((cold, pre, hot)): tot[i] += v * w
        print(f"M={M:3d} {name:24s} cold {cold:6.2f} us  cold + prefetch {pre:6.2f} us  hot {hot:6.2f} us  ({kern})", flush=True)
    print(f"M={M:3d} six GEMMs of a layer: cold {tot[0]:.1f}, with prefetch {tot[1]:.1f}, hot {tot[2]:.1f} us", flush=True)
