"""GPU probe: Paraformer-large bf16 vs oracle on a nearest-prototype output layer -- where do picks flip, and how large is the per-token error?"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (checker-side probe: lives under tests/ because it runs the oracle)
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from conftest import sub
from helpers import kaldi_audio
from test_oracle_paraformer import paraformer_setup
from oracle.paraformer_oracle import ParaformerOracle
import test_paraformer_gpu as tp

cfg, ck = paraformer_setup("paraformer_large")
eng = sub("engine")
B, slack = 64, 0.25
orc = ParaformerOracle(cfg, ck)
audios, stages, tried = [], [], 0
while len(audios) < B:
    a = kaldi_audio(7400 + tried, 128000); st = orc.stages(a); tried += 1
    total = float(np.sum(st["alphas"].astype(np.float64)) + cfg.tail_threshold)
    if abs(total - round(total)) > slack: audios.append(a); stages.append(st)
hidden = [st["dec_hidden"][:int(st["num_id"][0])] for st in stages]
ck2, cls = tp._prototype_output_layer(cfg, ck, hidden)
orc2 = ParaformerOracle(cfg, ck2)
sess = eng.ParaformerSession.from_checkpoint(cfg, ck2, precision=0)
sess.taps(True)
toks = sess.run(audios)
rows = sess.utterance_rows([a.size for a in audios]); trow = sess.token_rows([t.size for t in toks])
alphas, logits = sess.tap("alphas")[:, 0], sess.tap("logits")
W, bvec = orc2.w_out.numpy(), orc2.b_out.numpy()
mu = np.concatenate(hidden).mean(0)
rec = []
for b, (st, (r0, T), t0) in enumerate(zip(stages, rows, trow)):
    n = int(st["num_id"][0])
    da = alphas[r0:r0 + T].astype(np.float64) - st["alphas"]
    if toks[b].size != n: print("count differs", b, toks[b].size, n, da.sum()); continue
    lo = hidden[b] @ W.T + bvec; want = lo.argmax(1)
    lg = logits[t0:t0 + n, :cfg.vocab]
    d_orc = lo[np.arange(n), want][:, None] - lo
    err = np.abs((lg[np.arange(n), want][:, None] - lg) - d_orc)
    m = np.partition(d_orc, 1, axis=1)[:, 1]
    for i in range(n):
        rec.append((b, i, n, m[i], err[i][d_orc[i] <= 1.5].max(), err[i].max(), int(toks[b][i] != want[i]), np.linalg.norm(hidden[b][i] - mu), abs(da.sum()), np.abs(da).max()))
rec = np.array(rec)
last = rec[:, 1] == rec[:, 2] - 1
print("tokens", len(rec), "flips", int(rec[:, 6].sum()), "flips on last token", int(rec[last, 6].sum()))
for name, sel in (("last", last), ("second last", rec[:, 1] == rec[:, 2] - 2), ("first", rec[:, 1] == 0), ("inner", ~last & (rec[:, 1] > 0))):
    r = rec[sel]
    print(f"{name:12s} n={len(r)} margin min {r[:,3].min():.3f} med {np.median(r[:,3]):.3f} | e_near max {r[:,4].max():.3f} med {np.median(r[:,4]):.3f} | e_all max {r[:,5].max():.3f} | |c| mean {r[:,7].mean():.2f}")
print("alpha sum err max", rec[:, 8].max(), "alpha err max", rec[:, 9].max())
worst = rec[np.argsort(rec[:, 3] - 2 * rec[:, 4])[:12]]
print("worst margin - 2 e_near:"); print(np.round(worst[:, :8], 3))
