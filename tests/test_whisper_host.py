"""Whisper host loop (probe -> language / no-speech -> prefill -> greedy decode): CPU pieces + GPU end-to-end vs the oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import sub
from oracle.whisper_oracle import WhisperOracle
from test_oracle_whisper import unit_audio, whisper_setup

REF = "/root/reference/Whisper/Inference_Whisper_ONNX.py"


def test_remove_repeated_parts_and_audio_prep_cpu():
    wh = sub("whisper")
    f = lambda ids: list(wh.remove_repeated_parts(ids, 3, len(ids)))
    assert f([1, 2, 3]) == [1, 2, 3]
    assert f([5, 6, 7, 8, 9, 1, 6, 7, 8, 2]) == [5, 6, 7, 8, 9, 1]            # window (6,7,8) re-occurs -> cut before it
    assert f([1, 2, 3, 4, 5, 6, 7]) == [1, 2, 3, 4, 5, 6, 7]
    pcm = np.array([-32768, 0, 16384, 32767], np.int16)
    a = wh.prepare_audio_input(pcm)
    assert a.dtype == np.float32 and np.allclose(a, [-1.0, 0.0, 0.5, 32767 / 32768])
    assert wh.prepare_audio_input(pcm, np.int16).dtype == np.int16


@pytest.mark.skipif(not os.path.isfile(REF), reason="reference only mounted in the build container")
def test_remove_repeated_parts_matches_reference_function():
    import ast
    src = open(REF).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "remove_repeated_parts"][0]
    ns = {}
    exec(compile(ast.Module([fn], []), "ref", "exec"), ns)
    wh = sub("whisper")
    rng = np.random.default_rng(0)
    for _ in range(200):
        ids = rng.integers(0, 4, size=int(rng.integers(0, 14))).tolist()
        assert list(wh.remove_repeated_parts(ids, 3, len(ids))) == list(ns["remove_repeated_parts"](ids, 3, len(ids)))


def test_no_speech_probability_matches_oracle_head_cpu():
    cfg, ck, sup, beg = whisper_setup("whisper_tiny_test")
    orc = WhisperOracle(cfg, ck, sup, beg)
    rng = np.random.default_rng(1)
    logits = rng.standard_normal((3, cfg.vocab)).astype(np.float32) + orc.suppress_penalty.numpy()
    want = orc.no_speech_prob(torch.from_numpy(logits)).numpy()
    got = sub("whisper").no_speech_probability(logits, sup, cfg.no_speech_id)
    assert np.abs(got - want).max() < 1e-6


@pytest.mark.gpu
def test_transcriber_follows_the_reference_host_loop():
    cfg, ck, sup, beg = whisper_setup("whisper_mid_test")
    sess = sub("engine").WhisperSession.from_checkpoint(cfg, ck, precision=1, suppress_tokens=sup, begin_suppress_tokens=beg)
    tr = sub("whisper").WhisperTranscriber(cfg, sess, suppress_tokens=sup, remove_repeats=False, no_speech_threshold=2.0)
    clips = [(unit_audio(61, 48000) * 32768).astype(np.int16), (unit_audio(62, 20000) * 32768).astype(np.int16)]
    out, stats = tr.transcribe(clips, max_new=5)
    orc = WhisperOracle(cfg, ck, sup, beg)
    wh = sub("whisper")
    lang_ids = np.arange(cfg.first_language_id, cfg.first_language_id + cfg.n_languages)
    for clip, o in zip(clips, out):
        a = wh.prepare_audio_input(clip)
        probe = orc.greedy([a], [[cfg.sot_id]], 1)["logits"][0][0]
        srt = np.sort(probe[lang_ids])
        if srt[-1] - srt[-2] > 2e-3:
            assert o["language_id"] == int(lang_ids[np.argmax(probe[lang_ids])])
        want_p = float(orc.no_speech_prob(torch.from_numpy(probe[None]))[0])
        assert abs(o["no_speech_prob"] - want_p) < 1e-4
        ref = orc.greedy([a], [[cfg.sot_id, o["language_id"], cfg.transcribe_id, cfg.no_timestamps_id]], 5, eos_id=cfg.eot_id)
        margins = np.sort(ref["logits"][0], axis=1)
        if ((margins[:, -1] - margins[:, -2]) > 2e-3).all():
            want = [t for t in ref["token_ids"][0].tolist() if t != cfg.eot_id]
            assert o["tokens"].tolist() == want[:len(o["tokens"])] and len(o["tokens"]) >= min(len(want), 5) - 1
    assert stats["rtf"] > 0
    # no-speech gate: threshold 0 skips every clip
    tr0 = wh.WhisperTranscriber(cfg, sess, suppress_tokens=sup, no_speech_threshold=0.0)
    out0, _ = tr0.transcribe(clips, max_new=3)
    assert all(o["skipped"] and o["tokens"].size == 0 for o in out0)
