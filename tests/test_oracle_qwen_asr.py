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


def _beam_rows(hyps, max_new):
    toks = np.full((len(hyps), max_new), -1, np.int32)
    for r, (t, _) in enumerate(hyps):
        toks[r, :t.size] = t
    return toks, np.asarray([s for _, s in hyps], np.float32)


@pytest.mark.parametrize("fixture", ["qwen_asr_tiny", "qwen_asr_mid"])
def test_oracle_beam_search_matches_goldens_from_reference_logits(fixture):
    """beam_search_core over THIS restatement's decoder == the same search over the reference classes' logits (the reference has no beam
    code: what is pinned here is the decoder under a branching cache, not the search rule)."""
    g = load_golden(fixture)
    cfg, ck = qwen_setup(g)
    orc = QwenAsrOracle(cfg, ck, g["head_ids"].tolist(), g["tail_ids"].tolist(), g["suffix_ids"].tolist())
    width, max_new = (int(v) for v in g["beam"])
    for i, c in golden_cases(g):
        if "beam_tokens" not in c:
            continue
        audio = unit_audio(c["audio_seed"], c["n_samples"])
        for tag in ("beam", "beamstop"):
            hyps = orc.beam(audio, width, max_new, c["query_ids"].tolist(), c["language_tail_ids"].tolist(), c[tag + "_stop"].tolist())
            toks, scores = _beam_rows(hyps, max_new)
            assert np.array_equal(toks, c[tag + "_tokens"]), (i, tag)
            assert np.abs(scores - c[tag + "_scores"]).max() < 5 * F32_TOL, (i, tag)


def test_beam_search_core_rules():
    """Known-answer cases of the search rule on a hand-made 4-word model (no decoder involved)."""
    from oracle.qwen_asr_oracle import beam_search_core, log_softmax_f32
    table = {(): [2.0, 1.9, 0.0, -5.0], (0,): [0.0, 0.0, 0.0, 3.0], (1,): [5.0, 0.0, 0.0, 0.0], (1, 0): [0.0, 0.0, 4.0, 0.0]}

    def step(state, token):
        seq = state + (token,)
        return np.asarray(table.get(seq, [0.0, 0.0, 0.0, 0.0]), np.float32), seq
    lp = lambda seq: log_softmax_f32(np.asarray(table.get(tuple(seq), [0.0] * 4), np.float32))
    # width 1 == greedy: 0, then 3
    (t, s), = beam_search_core(table[()], (), step, 1, 2)
    assert t.tolist() == [0, 3] and abs(s - (lp([])[0] + lp([0])[3])) < 1e-6
    # width 2 overtakes greedy: 1 -> 0 scores higher than 0 -> 3
    hyps = beam_search_core(table[()], (), step, 2, 2)
    assert hyps[0][0].tolist() == [1, 0] and hyps[1][0].tolist() == [0, 3] and hyps[0][1] > hyps[1][1]
    # a stop id ends a hypothesis without being emitted; the search stops once the ended hypothesis leads
    hyps = beam_search_core(table[()], (), step, 2, 4, stop_ids=[0])
    assert hyps[0][0].tolist() == [] and abs(hyps[0][1] - lp([])[0]) < 1e-6      # "0" first: ended at once, nothing overtakes it
    hyps = beam_search_core(table[()], (), step, 2, 4, stop_ids=[3])
    # [0] + stop (-0.90) trails [1, 0] (-0.88) at first, leads once [1, 0] is extended, and the search ends there
    assert hyps[0][0].tolist() == [0] and abs(hyps[0][1] - (lp([])[0] + lp([0])[3])) < 1e-6 and hyps[1][0].tolist()[:2] == [1, 0]
    # ties -> lower id / earlier parent
    (t, _), (t2, _) = beam_search_core([1.0, 1.0, 1.0, 1.0], (), step, 2, 1)
    assert t.tolist() == [0] and t2.tolist() == [1]
