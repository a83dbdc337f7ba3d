"""GPU probe: the decode GEMM chain with cold weights (rotating copies, 768 MB) against the same chain over ONE weight copy (hot in the XCDs' L2s) -- the upper
bound of what prefetching the next GEMM's weights into L2 on a side branch of the decode graph could buy."""
import importlib, sys
sys.path.insert(0, ".")
probe = importlib.import_module("automatic-speech-recognition-asr-onnx_amd._probe")
shapes = [("qkv (LN fold)", 3840, 1280, 13), ("out / cross-out (+res)", 1280, 1280, 12), ("cross-q (LN fold)", 1280, 1280, 13),
          ("fc1 (LN fold, GELU)", 5120, 1280, 11), ("fc2 (+res)", 1280, 5120, 12)]
for M in (1, 32, 64):
    tot = [0.0, 0.0]
    for name, N, K, epi in shapes:
        cold, kern = probe.gemm_chain(M, N, K, epi, 768, 5)
        hot, _ = probe.gemm_chain(M, N, K, epi, 0, 5)
        w = 2 if "out" in name else 1
        tot[0] += cold * w; tot[1] += hot * w
        print(f"M={M:3d} {name:24s} cold {cold:6.2f} us  hot {hot:6.2f} us  ({kern})", flush=True)
    print(f"M={M:3d} six GEMMs of a layer: cold {tot[0]:.1f} us, hot {tot[1]:.1f} us -> {32 * (tot[0] - tot[1]) / 1e3:.2f} ms per token over 32 layers", flush=True)
