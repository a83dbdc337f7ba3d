import importlib, sys, time, os, numpy as np
sys.path.insert(0, ".")
PKG = "automatic-speech-recognition-asr-onnx_amd"
cfgm = importlib.import_module(PKG + ".config"); ckm = importlib.import_module(PKG + ".checkpoints"); eng = importlib.import_module(PKG + ".engine")
cfg = cfgm.sensevoice_small(); ck = ckm.synth_sensevoice_checkpoint(cfg, 0)
audio = ckm.synth_audio("kaldi", 1, 128000, seed=1234)
sess = eng.SenseVoiceSession.from_checkpoint(cfg, ck, precision=0)
a = [audio[0, 0]]
for _ in range(3): sess.run(a, [0])
t = time.perf_counter()
for _ in range(20): sess.run(a, [0])
print(os.environ.get("TAG"), "ms per chunk", (time.perf_counter() - t) / 20 * 1e3)
sess.profile(True); sess.profile_reset(); sess.run(a, [0]); pr = sess.profile_read(); sess.profile(False)
print("   ", {k: (round(v["total_ms"], 3), v["launches"]) for k, v in sorted(pr.items(), key=lambda kv: -kv[1]["total_ms"])[:7]})
