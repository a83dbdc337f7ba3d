"""CPU: the Whisper oracle is pinned against goldens minted from the real reference classes
(oracle/gen_golden_whisper.py: WHISPER_ENCODER / WHISPER_DECODER + the unmodified STFT_Process)."""
import numpy as np
import pytest
import torch

from conftest import sub
from helpers import golden_cases, load_golden
from oracle.whisper_oracle import WhisperOracle, slaney_mel_filterbank

F32_TOL = 1e-4


def whisper_setup(cfg_name, seed=0):
    cfg = getattr(sub("config"), cfg_name)()
    ckm = sub("checkpoints")
    return cfg, ckm.synth_whisper_checkpoint(cfg, seed), ckm.whisper_suppress_tokens(cfg), ckm.whisper_begin_suppress_tokens(cfg)


def unit_audio(seed, n):
    return sub("checkpoints").synth_audio("unit", 1, int(n), seed=int(seed))[0, 0]


@pytest.mark.parametrize("fixture", ["whisper_tiny", "whisper_mid"])
def test_oracle_matches_reference_goldens(fixture):
    g = load_golden(fixture)
    cfg, ck, sup, beg = whisper_setup(str(g["cfg_name"]), int(g["ckpt_seed"]))
    orc = WhisperOracle(cfg, ck, sup, beg)
    n_new = int(g["n_new"])
    for i, c in golden_cases(g):
        r = orc.greedy([unit_audio(c["audio_seed"], c["n_samples"])], [c["prompt"].tolist()], n_new)
        k, v = r["cross"][0]
        if "top1" in c:
            k, v, lg = k[:, ::5, ::16], v[:, ::5, ::16], r["logits"][0][:, ::53]
            assert np.abs(np.sort(r["logits"][0], axis=1)[:, -1] - c["top1"]).max() < F32_TOL
        else:
            lg = r["logits"][0]
        assert np.abs(k - c["cross_k"]).max() < F32_TOL and np.abs(v - c["cross_v"]).max() < F32_TOL, i
        assert np.abs(lg - c["logits"]).max() < F32_TOL, i
        assert np.array_equal(r["token_ids"][0], c["token_ids"])


def test_slaney_filterbank_matches_transformers():
    from transformers.audio_utils import mel_filter_bank
    for n_mels in (80, 128):
        a = slaney_mel_filterbank(201, 0.0, 8000.0, n_mels, 16000).numpy()
        b = mel_filter_bank(201, n_mels, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney").astype(np.float32)
        assert a.shape == b.shape == (201, n_mels) and np.abs(a - b).max() < 1e-7


def test_log_mel_matches_hf_feature_extractor():
    """Known-answer: STFT power (reflect pad, last frame dropped) + log-mel vs transformers' WhisperFeatureExtractor."""
    from transformers import WhisperFeatureExtractor
    cfg, ck, sup, beg = whisper_setup("whisper_tiny_test")
    orc = WhisperOracle(cfg, ck, sup, beg)
    a = unit_audio(5, 32000)
    fe = WhisperFeatureExtractor(feature_size=cfg.n_mels, sampling_rate=16000, hop_length=160, chunk_length=2, n_fft=400)
    want = fe(a, sampling_rate=16000, return_tensors="np", padding="max_length").input_features[0]   # (n_mels, 200)
    got = orc.log_mel(torch.from_numpy(a)).numpy()
    assert got.shape == want.shape and np.abs(got - want).max() < 2e-4


def test_no_speech_and_begin_suppress_heads():
    cfg, ck, sup, beg = whisper_setup("whisper_tiny_test")
    orc = WhisperOracle(cfg, ck, sup, beg)
    logits = torch.zeros(1, cfg.vocab) + orc.suppress_penalty
    p = orc.no_speech_prob(logits)                       # uniform after the +128 un-suppress => 1/V
    assert abs(float(p) - 1.0 / cfg.vocab) < 1e-6
    assert torch.isinf(orc.begin_bias[beg]).all() and float(orc.suppress_penalty[sup].max()) == -128.0


def test_penalty_greedy_matches_reference_heads():
    """APPLY_PENALTY + GREEDY_SEARCH (the reference host's default strategy) vs goldens minted with the reference's own head
    modules: the multiplier kicks in once `range` ids were generated and redirects the random-weight model's repeats."""
    g = load_golden("whisper_tiny")
    cfg, ck, sup, beg = whisper_setup(str(g["cfg_name"]), int(g["ckpt_seed"]))
    orc = WhisperOracle(cfg, ck, sup, beg)
    value, rng = float(g["penalty_value"]), int(g["penalty_range"])
    differs = 0
    for i, c in golden_cases(g):
        want = c["penalty_token_ids"]
        audio = [unit_audio(c["audio_seed"], c["n_samples"])]
        r = orc.greedy(audio, [c["prompt"].tolist()], want.size, repeat_penalty=value, penalty_range=rng)
        safe = c["penalty_margin"] > 1e-3
        first_unsafe = int(np.argmin(safe)) if not safe.all() else want.size      # after a near-tie the histories may fork
        assert np.array_equal(r["token_ids"][0][:first_unsafe], want[:first_unsafe]), i
        plain = orc.greedy(audio, [c["prompt"].tolist()], want.size)["token_ids"][0]
        assert np.array_equal(plain[:rng], want[:rng])                            # inactive until the window is full
        differs += int(not np.array_equal(c["plain_token_ids"], want))
    assert differs >= 2


def test_sampling_head_matches_reference_module():
    """TOPK_TOPP_SAMPLING restated with explicit uniforms vs goldens drawn from the reference module (its torch.rand_like stream
    captured by re-seeding): repetition penalty, temperature, top-k, top-p cut, Gumbel-max."""
    g = load_golden("whisper_tiny")
    cfg, ck, sup, beg = whisper_setup(str(g["cfg_name"]), int(g["ckpt_seed"]))
    orc = WhisperOracle(cfg, ck, sup, beg)
    t, k, p, rp = (float(v) for v in g["sampling_params"])
    for i, c in golden_cases(g):
        got = orc.sample(unit_audio(c["audio_seed"], c["n_samples"]), c["prompt"].tolist(), c["sampling_noise"], t, int(k), p, rp)
        assert np.array_equal(got, c["sampling_token_ids"]), i
    # head-level properties: top_k = 1 is arg-max whatever the noise; a tiny top_p keeps only the best candidate
    lg = torch.linspace(-3, 3, 50)
    assert WhisperOracle.sample_head(lg, [], [0.3], 1.0, 1, 0.9, 1.0) == 49
    assert WhisperOracle.sample_head(lg, [], [1e-6] + [0.999999] * 4, 1.0, 5, 1e-6, 1.0) == 49
    assert WhisperOracle.sample_head(lg, [49], [0.5] * 2, 1.0, 2, 1.0, 100.0) == 48        # penalised id drops out of first place
