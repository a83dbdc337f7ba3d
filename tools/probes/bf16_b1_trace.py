"""GPU probe (under rocprofv3 --kernel-trace): SenseVoiceSmall bf16, ONE 8 s chunk -- BASELINE's batch-1 point: 20 timed chunks after warm-up."""
import importlib, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
PKG = "automatic-speech-recognition-asr-onnx_amd"
cfgm = importlib.import_module(PKG + ".config"); ckm = importlib.import_module(PKG + ".checkpoints"); eng = importlib.import_module(PKG + ".engine")
cfg = cfgm.sensevoice_small(); ck = ckm.synth_sensevoice_checkpoint(cfg, 0)
audio = ckm.synth_audio("kaldi", 1, 128000, seed=1234)
sess = eng.SenseVoiceSession.from_checkpoint(cfg, ck, precision=0)
a = [audio[0, 0]]
for _ in range(3): sess.run(a, [0])
t = time.perf_counter()
for _ in range(20): sess.run(a, [0])
print("bf16 B = 1: ms per chunk", (time.perf_counter() - t) / 20 * 1e3)
