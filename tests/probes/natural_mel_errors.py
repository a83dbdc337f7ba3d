"""Probe (GPU box): log-mel of the natural clips -- bf16 sessions' split-operand DFT, the exact-f32 MFMA DFT and the reference's own f32 result, each against
the float64 oracle (the truth none of the f32 forms reaches on 60-80 dB of in-frame range)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import sub
from helpers import load_golden, sensevoice_setup
from oracle import natural_audio as na
from oracle.sensevoice_oracle import SenseVoiceOracle

g = load_golden("sensevoice_tiny_natural")
cfg, ck = sensevoice_setup("sensevoice_tiny")
clips = na.load_clips()
audios = [na.kaldi_input(p) for p in clips.values()]
o64 = SenseVoiceOracle(cfg, ck, dtype=torch.float64)
truth = [o64.stages(a, 0)["mel"] for a in audios]
out = {}
for tag, prec, env in (("split (bf16 session)", 0, {}), ("exact f32 MFMA (bf16 session, ASR_FBANK_SPLIT=0)", 0, {"ASR_FBANK_SPLIT": "0"}), ("f32 session", 1, {})):
    for k, v in env.items():
        os.environ[k] = v
    sess = sub("engine").SenseVoiceSession.from_checkpoint(cfg, ck, precision=prec)
    sess.taps(True)
    sess.run(audios, [0] * len(audios))
    mel = sess.tap("mel")
    f = 0
    for (n, a), t in zip(clips.items(), truth):
        nf = cfg.n_frames(a.size)
        m = mel[f:f + nf]; f += nf
        out.setdefault(n, {})[tag] = (np.abs(m - t).max(), np.abs(m - g[n + "_mel"]).max())
    for k in env:
        del os.environ[k]
for (n, _), t in zip(clips.items(), truth):
    print(f"{n:11s} reference f32 vs f64 {np.abs(g[n + '_mel'] - t).max():.2e} |", " | ".join(f"{tag}: vs f64 {e[0]:.2e}, vs reference {e[1]:.2e}" for tag, e in out[n].items()))
