"""CPU: the SenseVoice oracle is pinned against golden vectors minted from the real reference modules
(oracle/gen_golden.py), and -- in the build container only -- against the reference run live."""
import numpy as np
import pytest
import torch

from helpers import golden_cases, kaldi_audio, load_golden, sensevoice_setup
from oracle.sensevoice_oracle import SenseVoiceOracle

F32_TOL = 5e-5   # f32 roundoff between two orderings of the same arithmetic (observed <= 2e-5)


@pytest.mark.parametrize("fixture,cfg_name", [("sensevoice_tiny", "sensevoice_tiny"), ("sensevoice_small", "sensevoice_small")])
def test_oracle_matches_reference_goldens(fixture, cfg_name):
    g = load_golden(fixture)
    cfg, ck = sensevoice_setup(cfg_name, int(g["ckpt_seed"]))
    orc = SenseVoiceOracle(cfg, ck)
    for i, c in golden_cases(g):
        st = orc.stages(kaldi_audio(c["audio_seed"], c["n_samples"]), int(c["lang"]))
        assert np.array_equal(st["token_ids"], c["token_ids"]), f"case {i}"
        assert np.array_equal(st["num_id"], c["num_id"])
        assert np.array_equal(st["frame_ids"], c["frame_ids"])
        srt = np.sort(st["logits"], axis=1)
        assert np.abs(srt[:, -1] - c["top1"]).max() < F32_TOL
        if "logits" in c:
            for k in ("mel", "enc_in", "block0", "enc_out", "logits"):
                assert st[k].shape == c[k].shape
                assert np.abs(st[k] - c[k]).max() < F32_TOL * max(1.0, np.abs(c[k]).max()), (i, k)
        else:
            for k in ("mel", "enc_in", "block0", "enc_out"):
                assert np.abs(st[k][::8] - c[k]).max() < F32_TOL * max(1.0, np.abs(c[k]).max()), (i, k)
            assert np.abs(st["logits"][:, ::97] - c["logits_cols"]).max() < F32_TOL


def test_oracle_f64_agrees_with_f32():
    cfg, ck = sensevoice_setup("sensevoice_tiny")
    a = kaldi_audio(7, 20000)
    s32 = SenseVoiceOracle(cfg, ck).stages(a, 1)
    s64 = SenseVoiceOracle(cfg, ck, dtype=torch.float64).stages(a, 1)
    assert np.abs(s32["logits"] - s64["logits"]).max() < 1e-4
    assert np.array_equal(s32["token_ids"], s64["token_ids"])


def test_ctc_collapse_is_circular():
    """Export_SenseVoice.py:291-292: ids[t] is compared with ids[(t+1) % T]; the last frame of a run is kept."""
    f = lambda ids: SenseVoiceOracle.ctc_collapse(torch.tensor(ids), 0).tolist()
    assert f([0, 0, 0]) == []
    assert f([5, 5, 5]) == []                  # a single run wraps onto itself
    assert f([7, 0, 3, 3, 0, 7]) == [7, 3]     # trailing 7 equals frame 0 -> dropped
    assert f([1, 2, 2, 0, 2, 3]) == [1, 2, 2, 3]
    assert f([4]) == []


def test_sequence_geometry():
    cfg, _ = sensevoice_setup("sensevoice_tiny")
    assert cfg.n_frames(128000) == 798 and cfg.n_lfr(128000) == 133 and cfg.seq_len(128000) == 137
    assert cfg.n_frames(400) == 1 and cfg.seq_len(400) == 5
    assert cfg.n_frames(480000) == 2998 and cfg.n_lfr(480000) == 500


def test_oracle_matches_reference_live():
    from oracle import kaldi_mel, reference_harness as rh
    if not rh.reference_available():
        pytest.skip("/root/reference is only mounted in the build container")
    cfg, ck = sensevoice_setup("sensevoice_tiny")
    ref = rh.build_reference_sensevoice(cfg, ck, kaldi_mel.get_mel_banks)
    orc = SenseVoiceOracle(cfg, ck)
    for seed, n, lang in ((11, 16000, 4), (12, 801, 5)):
        a = kaldi_audio(seed, n)
        r = rh.reference_sensevoice_stages(ref, a, lang)
        o = orc.stages(a, lang)
        assert np.array_equal(r["token_ids"], o["token_ids"])
        for k in ("mel", "enc_in", "block0", "enc_out", "logits"):
            assert np.abs(r[k] - o[k]).max() < F32_TOL * max(1.0, np.abs(r[k]).max())
