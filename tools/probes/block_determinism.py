"""Run-to-run determinism of the SANM block kernel at the logits level (eager runs with taps), for a few batch shapes."""
import importlib, sys, os, numpy as np
sys.path.insert(0, ".")
PKG = "automatic-speech-recognition-asr-onnx_amd"
cfgm = importlib.import_module(PKG + ".config"); ckm = importlib.import_module(PKG + ".checkpoints"); eng = importlib.import_module(PKG + ".engine")
cfg = cfgm.sensevoice_small(); ck = ckm.synth_sensevoice_checkpoint(cfg, 0)
for lens in ([128000], [128000, 38880, 127000, 16000, 7777, 128000, 400, 64000, 100000, 128000, 3000], [128000] * 16):
    audios = [ckm.synth_audio("kaldi", 1, n, seed=300 + i)[0, 0] for i, n in enumerate(lens)]
    sess = eng.SenseVoiceSession.from_checkpoint(cfg, ck, precision=0)
    sess.taps(True)
    outs = []
    for _ in range(4):
        sess.run(audios, [0] * len(lens))
        outs.append((sess.tap("block0").copy(), sess.tap("enc_out").copy(), sess.tap("logits").copy()))
    for k, name in enumerate(("block0", "enc_out", "logits")):
        d = [float(np.abs(outs[r][k] - outs[0][k]).max()) for r in range(1, 4)]
        bad = np.argwhere(np.abs(outs[1][k] - outs[0][k]).max(axis=1) > 0).reshape(-1)
        print(len(lens), "utts", name, "max diff vs run 0:", d, "rows differing:", bad[:12], len(bad))
