"""GPU probe: the decode GEMM (csrc/decode_gemm.hip) on the Whisper-large-v3 decoder shapes as cold-weight launch chains (captured graph over
rotating weight copies): microseconds per launch including the ~1.6 us graph boundary, next to the time the weight bytes take at 5 TB/s."""
import importlib, sys
sys.path.insert(0, ".")
probe = importlib.import_module("automatic-speech-recognition-asr-onnx_amd._probe")
shapes = [("qkv (LN fold)", 3840, 1280, 13), ("out / cross-out (+res)", 1280, 1280, 12), ("cross-q (LN fold)", 1280, 1280, 13),
          ("fc1 (LN fold, GELU)", 5120, 1280, 11), ("fc2 (+res)", 1280, 5120, 12)]
tot = {}
for M in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["32", "64", "1"])]:
    for name, N, K, epi in shapes:
        us, kern = probe.gemm_chain(M, N, K, epi, 768, 5)
        mb = N * K * 2 / 1e6
        tot[M] = tot.get(M, 0.0) + us * (2 if "out" in name else 1)
        print(f"M={M:3d} {name:24s} N={N:5d} K={K:5d} {us:7.2f} us  {kern:14s} weights {mb:5.1f} MB = {mb / 5.0:5.2f} us at 5 TB/s", flush=True)
    print(f"M={M:3d} six GEMMs of a layer: {tot[M]:.1f} us", flush=True)
