"""CPU: the Qwen3-ASR oracle is pinned against goldens minted from the real reference classes (oracle/gen_golden_qwen_asr.py:
QWEN3_ASR_ENCODER / ROTARY_MASK_* / DECODER_EMBED / DECODER_MAIN / CONCAT_EMBED + the unmodified STFT_Process)."""
import numpy as np
import pytest

from conftest import sub
from helpers import golden_cases, load_golden
from oracle.qwen_asr_oracle import QwenAsrOracle, feat_lengths

F32_TOL = 2e-4


def qwen_setup(g):
    cfg = getattr(sub("config"), str(g["cfg_name"]))()
    ck = sub("checkpoints").synth_qwen_asr_checkpoint(cfg, int(g["ckpt_seed"]))
    return cfg, ck


def unit_audio(seed, n):
    return sub("checkpoints").synth_audio("unit", 1, int(n), seed=int(seed))[0, 0]


@pytest.mark.parametrize("fixture", ["qwen_asr_tiny", "qwen_asr_mid"])
def test_oracle_matches_reference_goldens(fixture):
    g = load_golden(fixture)
    cfg, ck = qwen_setup(g)
    orc = QwenAsrOracle(cfg, ck, g["head_ids"].tolist(), g["tail_ids"].tolist(), g["suffix_ids"].tolist())
    for i, c in golden_cases(g):
        r = orc.greedy(unit_audio(c["audio_seed"], c["n_samples"]), int(g["n_new"]), c["query_ids"].tolist(), c["language_tail_ids"].tolist())
        assert r["audio_hidden"].shape == c["audio_hidden"].shape and r["ids_len"] == int(c["ids_len"]), i
        assert np.abs(r["audio_hidden"] - c["audio_hidden"]).max() < F32_TOL, i
        assert np.abs(r["logits"] - c["logits"]).max() < 5 * F32_TOL, i
        assert np.array_equal(r["token_ids"], c["token_ids"]), i


def test_feat_lengths_table():
    assert [feat_lengths(n) for n in (0, 1, 2, 8, 9, 99, 100, 101, 250, 800, 813)] == [0, 1, 1, 1, 2, 13, 13, 14, 33, 104, 106]


@pytest.mark.parametrize("fixture", ["qwen_asr_tiny", "qwen_asr_mid"])
def test_oracle_heads_match_reference_goldens(fixture):
    """penalty-greedy (APPLY_PENALTY over save_id[-range:] from the first decode step + GREEDY_SEARCH) and TOPK_TOPP_SAMPLING with the
    committed uniforms, against the reference's own head classes."""
    g = load_golden(fixture)
    cfg, ck = qwen_setup(g)
    orc = QwenAsrOracle(cfg, ck, g["head_ids"].tolist(), g["tail_ids"].tolist(), g["suffix_ids"].tolist())
    pen = (float(g["penalty"][0]), int(g["penalty"][1]))
    t, k, p, rp = (float(v) for v in g["sampling_params"])
    for i, c in golden_cases(g):
        if "penalty_token_ids" not in c:
            continue
        audio = unit_audio(c["audio_seed"], c["n_samples"])
        r = orc.heads(audio, len(c["penalty_token_ids"]), c["query_ids"].tolist(), c["language_tail_ids"].tolist(), penalty=pen)
        assert np.abs(r["logits"][:, ::7] - c["penalty_logits"]).max() < 5 * F32_TOL, i
        assert np.array_equal(r["token_ids"], c["penalty_token_ids"]), i
        r = orc.heads(audio, len(c["sampling_token_ids"]), c["query_ids"].tolist(), c["language_tail_ids"].tolist(), sampling=(t, int(k), p, rp),
                      noise=c["sampling_noise"])
        assert np.array_equal(r["token_ids"], c["sampling_token_ids"]), i
