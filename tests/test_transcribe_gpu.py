"""GPU: tools/transcribe.py end to end on synthetic models -- wav files in, JSON token dump out, the dump compared with the session
driven directly (the checkpoint-agnostic harness that closes real-checkpoint transcript parity elsewhere)."""
import importlib.util
import json
import os
import types

import numpy as np
import pytest

from conftest import ROOT, sub
from helpers import kaldi_audio, sensevoice_setup
from test_oracle_whisper import unit_audio, whisper_setup

pytestmark = pytest.mark.gpu


def _tool():
    spec = importlib.util.spec_from_file_location("transcribe_tool", os.path.join(ROOT, "tools", "transcribe.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_sensevoice_wavs_to_dump_and_compare(tmp_path):
    t, aio = _tool(), sub("audio_io")
    cfg, ck = sensevoice_setup("sensevoice_tiny")
    folder = str(tmp_path / "SenseVoice_MI355X")
    sub("sensevoice").export_sensevoice(folder, cfg, ck, precision=1)
    wavs = []
    for i, n in enumerate((32000, 9000)):
        p = str(tmp_path / f"clip{i}.wav")
        aio.write_wav_int16(p, kaldi_audio(70 + i, n).astype(np.int16), 16000)
        wavs.append(p)
    args = types.SimpleNamespace(family="sensevoice", model=folder, wav=wavs, language="en", tokenizer=None, precision="f32", sliding_window=0, strict_wav=True,
                                 repeat_penalty=1.0, beam=1)
    dump = t.run(args)
    sess = sub("engine").SenseVoiceSession.from_checkpoint(cfg, ck, precision=1)
    direct = sess.run([kaldi_audio(70 + i, n) for i, n in enumerate((32000, 9000))], [2, 2])
    for f, d in zip(dump["files"], direct):
        assert f["windows"] == [d.astype(int).tolist()] and f["text"] is None and f["language"] == "en"
    out = str(tmp_path / "ours.json")
    json.dump(dump, open(out, "w"))
    assert t.compare(json.load(open(out)), dump)["token_for_token"]


def test_whisper_wavs_to_dump(tmp_path):
    t, aio, shim, arena, ckm = _tool(), sub("audio_io"), sub("ort_shim"), sub("arena"), sub("checkpoints")
    cfg, ck, sup, beg = whisper_setup("whisper_tiny_test")
    folder = str(tmp_path / "Whisper_MI355X")
    os.makedirs(folder)
    shim.save_model(os.path.join(folder, "Whisper.asrmodel"), "whisper", cfg.to_dict(), arena.build_whisper_arena(cfg, ck, 1, sup, beg), {}, 1)
    pcm = np.clip(np.round(unit_audio(41, 24000) * 32768.0), -32768, 32767).astype(np.int16)
    p = str(tmp_path / "clip.wav")
    aio.write_wav_int16(p, pcm, 16000)
    args = types.SimpleNamespace(family="whisper", model=folder, wav=[p], language="auto", tokenizer=None, precision="f32", sliding_window=0, strict_wav=True,
                                 repeat_penalty=1.0, beam=1)
    dump = t.run(args)
    f = dump["files"][0]
    assert f["n_samples"] == 24000 and len(f["windows"]) == 1 and f["text"] is None
    sess = sub("engine").WhisperSession.from_checkpoint(cfg, ck, precision=1, suppress_tokens=sup, begin_suppress_tokens=beg)
    tr = sub("whisper").WhisperTranscriber(cfg, sess, suppress_tokens=sup, detect_language=True, remove_repeats=False)
    out, _ = tr.transcribe([pcm])                  # the dump holds the windows' raw ids (the repeat guard belongs to detokenisation, Inference_Whisper_ONNX.py:705-708)
    assert f["windows"][0] == out[0]["tokens"].astype(int).tolist() and f["language_ids"][0] == out[0]["language_id"]


# ---- the same harness WITH detokeniser assets (built in the test: tests/tiny_tokenizers.py): `text` must be what the reference's
# detokenisation call makes of the dumped ids
def test_sensevoice_run_with_sentencepiece_model(tmp_path):
    from sentencepiece import SentencePieceProcessor
    from tiny_tokenizers import sentencepiece_model
    t, aio = _tool(), sub("audio_io")
    cfg, ck = sensevoice_setup("sensevoice_tiny")
    folder = str(tmp_path / "SenseVoice_MI355X")
    sub("sensevoice").export_sensevoice(folder, cfg, ck, precision=1)
    spm_path = sentencepiece_model(str(tmp_path / "tiny.model"), cfg.vocab)
    p = str(tmp_path / "clip.wav")
    aio.write_wav_int16(p, kaldi_audio(81, 40000).astype(np.int16), 16000)
    args = types.SimpleNamespace(family="sensevoice", model=folder, wav=[p], language="en", tokenizer=spm_path, precision="f32", sliding_window=0, strict_wav=True,
                                 repeat_penalty=1.0, beam=1)
    f = t.run(args)["files"][0]
    sp = SentencePieceProcessor(); sp.Load(spm_path)
    assert len(f["windows"]) == 1 and len(f["windows"][0]) > 0
    assert isinstance(f["text"], str) and f["text"] == sp.decode([f["windows"][0]])[0]          # Inference_SenseVoice_ONNX.py:305


def test_whisper_run_with_tokenizer_directory(tmp_path):
    from transformers import AutoTokenizer
    from tiny_tokenizers import whisper_tokenizer_dir
    t, aio, shim, arena = _tool(), sub("audio_io"), sub("ort_shim"), sub("arena")
    cfg, ck, sup, beg = whisper_setup("whisper_tiny_test")
    folder = str(tmp_path / "Whisper_MI355X")
    os.makedirs(folder)
    shim.save_model(os.path.join(folder, "Whisper.asrmodel"), "whisper", cfg.to_dict(), arena.build_whisper_arena(cfg, ck, 1, sup, beg), {}, 1)
    tok_dir = whisper_tokenizer_dir(str(tmp_path / "wtok"), cfg)
    pcm = np.clip(np.round(unit_audio(43, 24000) * 32768.0), -32768, 32767).astype(np.int16)
    p = str(tmp_path / "clip.wav")
    aio.write_wav_int16(p, pcm, 16000)
    args = types.SimpleNamespace(family="whisper", model=folder, wav=[p], language="en", tokenizer=tok_dir, precision="f32", sliding_window=0, strict_wav=True,
                                 repeat_penalty=1.0, beam=1)
    f = t.run(args)["files"][0]
    tok = AutoTokenizer.from_pretrained(tok_dir)
    assert f["language_ids"] == [tok.convert_tokens_to_ids("<|en|>")] == [cfg.first_language_id]        # --language en resolved through the tokenizer
    ids = f["windows"][0]
    assert len(ids) > 0 and isinstance(f["text"], str) and f["text"] == t.whisper_text(tok, ids) and "<|" not in f["text"]


def test_qwen_run_with_tokenizer_directory(tmp_path):
    from transformers import AutoTokenizer
    from tiny_tokenizers import qwen_tokenizer_dir
    from helpers import load_golden
    from test_oracle_qwen_asr import qwen_setup
    t, aio, q = _tool(), sub("audio_io"), sub("qwen_asr")
    cfg, ck = qwen_setup(load_golden("qwen_asr_tiny"))
    tok_dir = qwen_tokenizer_dir(str(tmp_path / "qtok"), cfg.vocab)
    tok = AutoTokenizer.from_pretrained(tok_dir)
    folder = str(tmp_path / "Qwen_ASR_MI355X")
    os.makedirs(folder)
    q.export_qwen_asr(cfg, ck, os.path.join(folder, "Qwen_ASR.asrmodel"), q.build_metadata(tok, ["English", "Chinese"], cfg), precision=1)
    pcm = np.clip(np.round(unit_audio(44, 20000) * 32768.0), -32768, 32767).astype(np.int16)
    p = str(tmp_path / "clip.wav")
    aio.write_wav_int16(p, pcm, 16000)
    for lang in ("auto", "English"):
        args = types.SimpleNamespace(family="qwen_asr", model=folder, wav=[p], language=lang, tokenizer=tok_dir, precision="f32", sliding_window=0, strict_wav=True,
                                     repeat_penalty=1.0, beam=1)
        f = t.run(args)["files"][0]
        ids = f["windows"][0]
        raw = tok.decode(ids, skip_special_tokens=True).strip()                       # Inference_Qwen_ASR_ONNX.py:746-752
        if lang == "auto":
            want_lang, want_text = q.parse_asr_output(q.LANG_PREFIX + raw if raw else raw)
            assert f["text"] == want_text and (f["language"] or "auto") == (want_lang or "auto")
        else:
            assert f["text"] == q.parse_asr_output(raw)[1] and f["language"] == "English"
        assert isinstance(f["text"], str)
