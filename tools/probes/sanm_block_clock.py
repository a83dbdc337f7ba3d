import importlib, sys, os, numpy as np
sys.path.insert(0, ".")
PKG = "automatic-speech-recognition-asr-onnx_amd"
cfgm = importlib.import_module(PKG + ".config"); ckm = importlib.import_module(PKG + ".checkpoints"); eng = importlib.import_module(PKG + ".engine")
cfg = cfgm.sensevoice_small(); ck = ckm.synth_sensevoice_checkpoint(cfg, 0)
B = int(__import__("os").environ.get("CLOCK_B", "64"))
audio = ckm.synth_audio("kaldi", B, 128000, seed=1234)
sess = eng.SenseVoiceSession.from_checkpoint(cfg, ck, precision=0)
audios = [audio[i, 0] for i in range(B)]
for _ in range(3):
    sess.run(audios, [0] * B)
