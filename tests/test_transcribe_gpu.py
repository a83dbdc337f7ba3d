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
    args = types.SimpleNamespace(family="sensevoice", model=folder, wav=wavs, language="en", tokenizer=None, precision="f32", sliding_window=0,
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
    args = types.SimpleNamespace(family="whisper", model=folder, wav=[p], language="auto", tokenizer=None, precision="f32", sliding_window=0,
                                 repeat_penalty=1.0, beam=1)
    dump = t.run(args)
    f = dump["files"][0]
    assert f["n_samples"] == 24000 and len(f["windows"]) == 1 and f["text"] is None
    sess = sub("engine").WhisperSession.from_checkpoint(cfg, ck, precision=1, suppress_tokens=sup, begin_suppress_tokens=beg)
    tr = sub("whisper").WhisperTranscriber(cfg, sess, suppress_tokens=sup, detect_language=True)
    out, _ = tr.transcribe([pcm])
    assert f["windows"][0] == out[0]["tokens"].astype(int).tolist() and f["language_ids"][0] == out[0]["language_id"]
