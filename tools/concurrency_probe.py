"""Does running k concurrent sub-batches on k streams beat one big batch? (two sessions share one weight arena)"""
import importlib, os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = "automatic-speech-recognition-asr-onnx_amd"
cfgm, ckm, arena, eng = (importlib.import_module(f"{P}.{m}") for m in ("config", "checkpoints", "arena", "engine"))
cfg = cfgm.sensevoice_small()
ck = ckm.synth_sensevoice_checkpoint(cfg, 0)
blob = torch.from_numpy(arena.build_sensevoice_arena(cfg, ck, 0)).cuda()
n = 128000
for k in (1, 2, 4):
    B = 64 // k
    sess = [eng.SenseVoiceSession(cfg, blob, 0, 0, arena_device_ptr=blob.data_ptr(), arena_bytes=blob.numel()) for _ in range(k)]
    aud = [torch.from_numpy(ckm.synth_audio("kaldi", B, n, seed=1234 + i)).cuda() for i in range(k)]
    offs = np.arange(B + 1, dtype=np.int64) * n
    lang = np.zeros(B, np.int32)
    def work(i, steps):
        for _ in range(steps):
            sess[i].run_packed(None, offs, lang, audio_device_ptr=aud[i].data_ptr())
    for i in range(k): work(i, 2)
    torch.cuda.synchronize()
    steps = 10
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i, steps)) for i in range(k)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(f"k={k} streams x B={B}: {dt/steps*1e3:.2f} ms per 64 utterances  ({64*8*steps/dt:.0f} audio-s/s)", flush=True)
    del sess
