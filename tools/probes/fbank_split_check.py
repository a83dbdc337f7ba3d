"""mel tap of a bf16 SenseVoice session with the split-operand DFT vs the exact-f32 DFT (ASR_FBANK_SPLIT=0), plus the step time of both."""
import importlib, os, sys, time, numpy as np
sys.path.insert(0, ".")
PKG = "automatic-speech-recognition-asr-onnx_amd"
cfgm = importlib.import_module(PKG + ".config"); ckm = importlib.import_module(PKG + ".checkpoints"); eng = importlib.import_module(PKG + ".engine")
cfg = cfgm.sensevoice_small(); ck = ckm.synth_sensevoice_checkpoint(cfg, 0)
B = 64
audio = ckm.synth_audio("kaldi", B, 128000, seed=1234)
# a speech-like dynamic range: a loud low tone + a quiet high tone + noise, 16-bit PCM values
t = np.arange(128000) / 16000.0
audio[1, 0] = np.round(12000 * np.sin(2 * np.pi * 180 * t) + 6 * np.sin(2 * np.pi * 6500 * t) + np.random.default_rng(0).normal(0, 1.5, t.size))
audios = [audio[i, 0] for i in range(B)]
mels = {}
for split in ("0", "1"):
    os.environ["ASR_FBANK_SPLIT"] = split
    sess = eng.SenseVoiceSession.from_checkpoint(cfg, ck, precision=0)
    sess.taps(True)
    sess.run(audios[:2], [0, 0])
    mels[split] = sess.tap("mel").copy()
    sess.taps(False)
    for _ in range(3): sess.run(audios, [0] * B)
    t0 = time.perf_counter()
    for _ in range(20): sess.run(audios, [0] * B)
    print("split", split, "ms per step (host audio in)", (time.perf_counter() - t0) / 20 * 1e3)
    sess.profile(True); sess.profile_reset(); sess.run(audios, [0] * B); pr = sess.profile_read(); sess.profile(False)
    print("   fbank ms", pr["fbank"]["total_ms"])
d = np.abs(mels["1"] - mels["0"])
n0 = mels["0"].shape[0] // 2
print("mel |max| diff: utt0 %.3e  utt1 (60 dB dynamic range) %.3e   mel range %.2f .. %.2f" % (d[:n0].max(), d[n0:].max(), mels["0"].min(), mels["0"].max()))
