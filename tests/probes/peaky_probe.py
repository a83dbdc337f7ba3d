"""GPU probe: how large is the bf16 perturbation of the SenseVoice encoder output next to the distances between frames, and what margins does a
nearest-prototype CTC head built on the oracle's encoder outputs leave (tests/test_sensevoice_gpu.py: peaky-head parity test)."""
import importlib, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (checker-side probe: lives under tests/ because it runs the oracle)
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from conftest import sub
from helpers import sensevoice_setup, kaldi_audio
from oracle.sensevoice_oracle import SenseVoiceOracle

cfg, ck = sensevoice_setup("sensevoice_small")
eng = sub("engine")
B = int(os.environ.get("PEAKY_B", "64"))
audios = [kaldi_audio(7400 + i, 128000) for i in range(B)]
langs = [i % 7 for i in range(B)]
orc = SenseVoiceOracle(cfg, ck)
H = np.stack([orc.stages(a, l)["enc_out"] for a, l in zip(audios, langs)])
T, d = H.shape[1:]
sess = eng.SenseVoiceSession.from_checkpoint(cfg, ck, precision=0)
sess.taps(True)
sess.run(audios, langs)
rows = sess.utterance_rows([a.size for a in audios])
G = sess.tap("enc_out")
Hg = np.stack([G[r0:r0 + T] for r0, T_ in rows])
dh = Hg - H
print("enc_out |h| %.2f; bf16 perturbation: |dh| per frame mean %.3f max %.3f; per component rms %.4f max %.4f" % (
    np.linalg.norm(H, axis=2).mean(), np.linalg.norm(dh, axis=2).mean(), np.linalg.norm(dh, axis=2).max(), np.sqrt((dh ** 2).mean()), np.abs(dh).max()))
X = H.reshape(-1, d); mu = X.mean(0); C = X - mu
Gm = C @ C.T; n2 = np.diag(Gm).copy(); D2 = n2[:, None] + n2[None, :] - 2 * Gm; np.fill_diagonal(D2, 1e9)
nn = np.sqrt(np.maximum(D2.min(1), 0))
print("centred |c| %.2f; nearest other frame: min %.2f p1 %.2f median %.2f" % (np.sqrt(n2).mean(), nn.min(), np.percentile(nn, 1), np.median(nn)))
# every frame its own class: nearest-prototype head  logit_v(x) = (x - mu) . p_v - |p_v|^2 / 2
N = X.shape[0]
W = ck["ctc.ctc_lo.weight"].copy() * 0.1
b = ck["ctc.ctc_lo.bias"].copy() * 0.1
cls = 1 + np.arange(N)
W[cls] = C
b[cls] = -(C @ mu) - 0.5 * n2
ck2 = dict(ck); ck2["ctc.ctc_lo.weight"] = W.astype(np.float32); ck2["ctc.ctc_lo.bias"] = b.astype(np.float32)
lo = X @ ck2["ctc.ctc_lo.weight"].T + ck2["ctc.ctc_lo.bias"]
srt = np.sort(lo, 1); margin = srt[:, -1] - srt[:, -2]
print("own-class head: oracle argmax == own class for %d / %d frames; margin min %.3f p1 %.3f median %.3f; top logit mean %.2f" % (
    (lo.argmax(1) == cls).sum(), N, margin.min(), np.percentile(margin, 1), np.median(margin), srt[:, -1].mean()))
s2 = eng.SenseVoiceSession.from_checkpoint(cfg, ck2, precision=0)
s2.taps(True)
s2.run(audios, langs)
lg = s2.tap("logits"); ids = s2.tap("frame_ids", dtype=np.int32)[:, 0]
lgg = np.concatenate([lg[r0:r0 + T] for r0, _ in rows]); idg = np.concatenate([ids[r0:r0 + T] for r0, _ in rows])
err = np.abs(lgg[:, :cfg.vocab] - lo)
print("bf16 logits: max |err| %.4f (all classes), %.4f on the winning class; frames whose argmax differs from the oracle: %d / %d" % (
    err.max(), err[np.arange(N), lo.argmax(1)].max(), (idg != lo.argmax(1)).sum(), N))
bad = np.flatnonzero(idg != lo.argmax(1))
print("margins of the differing frames:", np.sort(margin[bad])[:20])
top = lo.argmax(1)
rel = (lgg[:, :cfg.vocab] - lgg[np.arange(N), top][:, None]) - (lo - lo[np.arange(N), top][:, None])
print("error of logit differences to the winner: max %.4f" % np.abs(rel).max())
for thr in (0.25, 0.5, 1.0):
    ok_utt = (margin.reshape(B, T) > thr).all(1)
    print("margin > %.2f on every frame: %d / %d utterances; frames %d / %d" % (thr, ok_utt.sum(), B, (margin > thr).sum(), N))
