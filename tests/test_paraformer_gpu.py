"""GPU parity: the HIP Paraformer path (Kaldi fbank, SANM encoder, CIF predictor, NAR decoder) through the C ABI vs
goldens minted from the reference's PARAFORMER class and vs the oracle."""
import numpy as np
import pytest

from conftest import sub
from helpers import golden_cases, kaldi_audio, load_golden
from oracle.paraformer_oracle import ParaformerOracle
from test_oracle_paraformer import paraformer_setup

pytestmark = pytest.mark.gpu

BF16, F32 = 0, 1
TOL_F32 = 1e-3


@pytest.mark.parametrize("fixture", ["paraformer_tiny", "paraformer_large"])
def test_f32_mode_matches_reference_goldens(fixture):
    g = load_golden(fixture)
    cfg, ck = paraformer_setup(str(g["cfg_name"]))
    sess = sub("engine").ParaformerSession.from_checkpoint(cfg, ck, precision=F32)
    cases = [c for _, c in golden_cases(g)]
    audios = [kaldi_audio(c["audio_seed"], c["n_samples"]) for c in cases]
    sess.taps(True)
    toks = sess.run(audios)                                   # one ragged batch, incl. the zero-token clip
    rows = sess.utterance_rows([a.size for a in audios])
    enc, alphas, logits = sess.tap("enc_out"), sess.tap("alphas")[:, 0], sess.tap("logits")
    trow = sess.token_rows([t.size for t in toks])                 # decoder-side taps follow the packed token rows
    for c, tok, (r0, T), t0 in zip(cases, toks, rows, trow):
        n = int(c["num_id"][0])
        assert np.abs(alphas[r0:r0 + T] - c["alphas"]).max() < TOL_F32
        if "logits" in c:
            assert np.abs(enc[r0:r0 + T] - c["enc_out"]).max() < TOL_F32
        else:
            assert np.abs(enc[r0:r0 + T][::8] - c["enc_out"]).max() < TOL_F32
        if c["cif_slack"] > 2e-4:                              # fire count is only defined away from an integer boundary
            assert tok.size == n, (tok.size, n)
            lg = logits[t0:t0 + max(n, 1)]
            ref = c["logits"] if "logits" in c else None
            if ref is not None:
                assert np.abs(lg - ref).max() < TOL_F32
            else:
                assert np.abs(lg[:, ::37] - c["logits_cols"]).max() < TOL_F32
            if n and (c["margin"] > 2 * TOL_F32).all():
                assert np.array_equal(tok, c["token_ids"])


def test_bf16_batch_vs_oracle_and_determinism():
    cfg, ck = paraformer_setup("paraformer_large")
    sess = sub("engine").ParaformerSession.from_checkpoint(cfg, ck, precision=BF16)
    orc = ParaformerOracle(cfg, ck)
    lens = [128000, 38880, 128000, 16000]
    audios = [kaldi_audio(400 + i, n) for i, n in enumerate(lens)]
    audios[2] = audios[0].copy()
    sess.taps(True)
    t1 = sess.run(audios)
    rows = sess.utterance_rows(lens)
    alphas = sess.tap("alphas")[:, 0]
    assert np.array_equal(t1[0], t1[2])                        # batching does not change per-utterance results
    for a, (r0, T), tok in zip(audios, rows, t1):
        st = orc.stages(a)
        assert np.abs(alphas[r0:r0 + T] - st["alphas"]).max() < 0.05
        cs = np.cumsum(np.concatenate([st["alphas"].astype(np.float64), [cfg.tail_threshold]]))
        if np.min(np.abs(cs - np.round(cs))) > 0.2:             # far from a boundary even for bf16 alphas
            assert tok.size == int(st["num_id"][0])
    sess.taps(False)
    t2 = sess.run(audios)                                      # eager, then captured graph replay
    t3 = sess.run(audios)
    for x, y, z in zip(t1, t2, t3):
        assert np.array_equal(x, y) and np.array_equal(y, z)


def test_long_and_maximum_windows_f32():
    """13.6 s and 30 s windows (beyond the 144-row fused kernels) beside short ones: every utterance vs the batch-1 oracle."""
    cfg, ck = paraformer_setup("paraformer_tiny")
    sess = sub("engine").ParaformerSession.from_checkpoint(cfg, ck, precision=F32)
    orc = ParaformerOracle(cfg, ck)
    lens = [218080, cfg.max_audio_len, 16000, 128000]
    audios = [kaldi_audio(760 + i, n) for i, n in enumerate(lens)]
    sess.taps(True)
    toks = sess.run(audios)
    rows = sess.utterance_rows(lens)
    enc, alphas, logits = sess.tap("enc_out"), sess.tap("alphas")[:, 0], sess.tap("logits")
    trow = sess.token_rows([t.size for t in toks])
    for a, (r0, T), tok, t0 in zip(audios, rows, toks, trow):
        st = orc.stages(a)
        assert np.abs(enc[r0:r0 + T] - st["enc_out"]).max() < TOL_F32
        assert np.abs(alphas[r0:r0 + T] - st["alphas"]).max() < TOL_F32
        cs = np.cumsum(np.concatenate([st["alphas"].astype(np.float64), [cfg.tail_threshold]]))
        if np.min(np.abs(cs - np.round(cs))) > 1e-3:                 # fire count defined away from an integer boundary
            n = int(st["num_id"][0])
            assert tok.size == n
            assert np.abs(logits[t0:t0 + max(n, 1)] - st["logits"]).max() < TOL_F32
            srt = np.sort(st["logits"][:max(n, 1)], axis=1)
            if n and ((srt[:, -1] - srt[:, -2]) > 2 * TOL_F32).all():
                assert np.array_equal(tok, st["token_ids"])
    with pytest.raises(Exception, match="max_audio_len"):
        sess.run([kaldi_audio(1, cfg.max_audio_len + 160)])
