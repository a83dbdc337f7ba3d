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


# ---- the reference's per-file window loop (Inference_Whisper_ONNX.py:741-829; VERDICT r05 missing #5)
def _reference_window_plan(audio_len, input_audio_length, sliding_window):
    """The reference's arithmetic, restated line for line from :745-757 (test-side checker)."""
    stride = input_audio_length if sliding_window <= 0 else sliding_window
    if audio_len <= input_audio_length:
        windows = 1
    else:
        windows = int(np.ceil((audio_len - input_audio_length) / stride)) + 1
    return windows, stride, (windows - 1) * stride + input_audio_length


def test_window_planner_cpu():
    wh = sub("whisper")
    for audio_len in (1, 31999, 32000, 32001, 56000, 80000, 96000, 96001, 480000, 1200000):
        for win in (32000, 480000):
            for sl in (0, -1, 24000, 32000, 40000):
                n, stride, window, aligned = wh.plan_windows(audio_len, win, sl)
                assert (n, stride, aligned) == _reference_window_plan(audio_len, win, sl) and window == win
                assert aligned >= min(audio_len, aligned) and (n - 1) * stride + window == aligned
    # dynamic-axis export: the window is the file
    assert wh.plan_windows(70000, None, 24000) == (1, 24000, 70000, 70000)
    assert wh.plan_windows(80000, 32000, 0) == (3, 32000, 32000, 96000)       # 2.5 windows -> 3, tail zero-padded by 16000
    assert wh.plan_windows(80000, 32000, 24000) == (3, 24000, 32000, 80000)   # overlapping stride, no padding


@pytest.mark.gpu
def test_file_loop_probes_window_zero_only_and_pads_the_tail():
    """2.5 windows of 2 s; window 1 is silence. The reference probes language / no-speech on window 0 only, later windows reuse its
    language id and are never gated; a no-speech verdict on window 0 aborts the file; the tail window is zero-padded."""
    cfg, ck, sup, beg = whisper_setup("whisper_mid_test")
    wh = sub("whisper")
    sess = sub("engine").WhisperSession.from_checkpoint(cfg, ck, precision=1, suppress_tokens=sup, begin_suppress_tokens=beg)
    orc = WhisperOracle(cfg, ck, sup, beg)
    W = 32000
    pcm = (unit_audio(71, 80000) * 32768).astype(np.int16)
    pcm[W:2 * W] = 0                                                       # window 1: silence
    lang_ids = np.arange(cfg.first_language_id, cfg.first_language_id + cfg.n_languages)
    n_new = 5

    def oracle_window(a, lang):
        ref = orc.greedy([a], [[cfg.sot_id, lang, cfg.transcribe_id, cfg.no_timestamps_id]], n_new, eos_id=cfg.eot_id)
        margins = np.sort(ref["logits"][0], axis=1)
        return [t for t in ref["token_ids"][0].tolist() if t != cfg.eot_id], bool(((margins[:, -1] - margins[:, -2]) > 2e-3).all())

    for sliding, want_n, want_aligned in ((0, 3, 96000), (24000, 3, 80000)):
        tr = wh.WhisperTranscriber(cfg, sess, suppress_tokens=sup, remove_repeats=False, no_speech_threshold=2.0)
        r, stat = tr.transcribe_file(pcm, sliding_window=sliding, input_audio_length=W, max_new=n_new)
        assert r["n_windows"] == want_n == len(r["windows"]) and not r["no_speech"] and stat["rtf"] > 0
        audio = wh.prepare_audio_input(pcm)
        audio = np.concatenate([audio, np.zeros(want_aligned - audio.size, np.float32)])
        stride = r["stride"]
        clips = [audio[w * stride:w * stride + W] for w in range(want_n)]
        assert all(c.size == W for c in clips)
        probe = orc.greedy([clips[0]], [[cfg.sot_id]], 1)["logits"][0][0]
        srt = np.sort(probe[lang_ids])
        if srt[-1] - srt[-2] > 2e-3:
            assert r["language_id"] == int(lang_ids[np.argmax(probe[lang_ids])])       # window 0's language ...
        assert abs(r["no_speech_prob"] - float(orc.no_speech_prob(torch.from_numpy(probe[None]))[0])) < 1e-4
        for w, clip in enumerate(clips):                                     # ... carried into EVERY window's prompt
            want, clear = oracle_window(clip, r["language_id"])
            if clear:
                assert r["windows"][w] == want[:len(r["windows"][w])] and len(r["windows"][w]) >= min(len(want), n_new) - 1, (sliding, w)
        assert r["tokens"].tolist() == [t for w in r["windows"] for t in w]
    # the silent window has its own (different) no-speech probability, and the reference never looks at it: a threshold between
    # window 0's and the silent window's probability must NOT skip anything when window 0 passes ...
    out, _ = wh.WhisperTranscriber(cfg, sess, suppress_tokens=sup, remove_repeats=False).transcribe(
        [pcm[:W], pcm[W:2 * W]], max_new=1)
    p0, p1 = out[0]["no_speech_prob"], out[1]["no_speech_prob"]
    assert abs(p0 - r["no_speech_prob"]) < 1e-5
    if p1 > p0 + 1e-4:
        thr = 0.5 * (p0 + p1)
        tr = wh.WhisperTranscriber(cfg, sess, suppress_tokens=sup, remove_repeats=False, no_speech_threshold=thr)
        r2, _ = tr.transcribe_file(pcm, input_audio_length=W, max_new=n_new)
        assert not r2["no_speech"] and len(r2["windows"]) == 3
    # ... and a verdict on window 0 aborts the WHOLE file (:801-805)
    tr0 = wh.WhisperTranscriber(cfg, sess, suppress_tokens=sup, no_speech_threshold=0.0)
    r0, _ = tr0.transcribe_file(pcm, input_audio_length=W, max_new=n_new)
    assert r0["no_speech"] and r0["windows"] == [] and r0["tokens"].size == 0
    # the repeat guard sees the concatenation of the windows (:705-708), not each window
    tr3 = wh.WhisperTranscriber(cfg, sess, suppress_tokens=sup, remove_repeats=True, no_speech_threshold=2.0)
    r3, _ = tr3.transcribe_file(np.concatenate([pcm[:W], pcm[:W]]), input_audio_length=W, max_new=n_new)
    flat = [t for w in r3["windows"] for t in w]
    assert r3["windows"][0] == r3["windows"][1] and r3["tokens"].tolist() == list(wh.remove_repeated_parts(flat, 3, len(flat)))
