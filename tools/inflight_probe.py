"""GPU probe: SenseVoiceSmall B = 64 x 8 s with N batches in flight on N sessions / HIP streams (run via gpurun)."""
import importlib, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-speech-recognition-asr-onnx_amd"
import torch
cfgm, ckm, arena, eng = (importlib.import_module(f"{PKG}.{m}") for m in ("config", "checkpoints", "arena", "engine"))
cfg = cfgm.sensevoice_small()
B, n = 64, 128000
blob = torch.from_numpy(arena.build_sensevoice_arena(cfg, ckm.synth_sensevoice_checkpoint(cfg, seed=0), 0)).cuda()
audio = torch.from_numpy(ckm.synth_audio("kaldi", B, n, seed=1)).cuda()
offs = np.arange(B + 1, dtype=np.int64) * n
lang = np.zeros(B, np.int32)
for N in (1, 2, 3):
    sessions = [eng.SenseVoiceSession(cfg, blob, 0, 0, arena_device_ptr=blob.data_ptr(), arena_bytes=blob.numel()) for _ in range(N)]
    def work(s, k):
        for _ in range(k):
            s.run_packed(None, offs, lang, audio_device_ptr=audio.data_ptr())
    for s in sessions:
        work(s, 3)
    torch.cuda.synchronize()
    per = 30
    ths = [threading.Thread(target=work, args=(s, per)) for s in sessions]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"in flight {N}: {B * 8 * per * N / el:9.0f} audio-s/s, {el / (per * N) * 1e3:6.2f} ms per batch", flush=True)
