"""CPU: wav ingest (the reference's pydub calls restated on stdlib wave / audioop) and the transcript dump comparator of
tools/transcribe.py. No GPU: the `run` leg is exercised on the GPU box by tests/test_transcribe_gpu.py."""
import importlib.util
import json
import os
import wave

import numpy as np

from conftest import ROOT, sub


def _tool():
    spec = importlib.util.spec_from_file_location("transcribe_tool", os.path.join(ROOT, "tools", "transcribe.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _write(path, data, n_ch, width, rate):
    with wave.open(path, "wb") as w:
        w.setnchannels(n_ch); w.setsampwidth(width); w.setframerate(rate)
        w.writeframes(data)


def test_wav_ingest_matches_the_pydub_conversions(tmp_path):
    aio = sub("audio_io")
    rng = np.random.default_rng(0)
    mono = rng.integers(-20000, 20000, 16000, dtype=np.int16)
    p = str(tmp_path / "mono16k.wav")
    aio.write_wav_int16(p, mono, 16000)
    assert np.array_equal(aio.read_wav_int16(p, 16000), mono)                       # 16 kHz mono passes through untouched
    # stereo -> mono: pydub's set_channels(1) = audioop.tomono(data, 2, 0.5, 0.5): each channel halved (floor), then added
    left, right = rng.integers(-20000, 20000, 4800, dtype=np.int16), rng.integers(-20000, 20000, 4800, dtype=np.int16)
    st = np.stack([left, right], 1).reshape(-1)
    p2 = str(tmp_path / "stereo16k.wav")
    _write(p2, st.astype("<i2").tobytes(), 2, 2, 16000)
    got = aio.read_wav_int16(p2, 16000)
    want = np.floor(left * 0.5) + np.floor(right * 0.5)
    assert got.shape == (4800,) and np.abs(got - want).max() <= 1
    # 48 kHz -> 16 kHz (the reference's en/test_sample.wav is 48 kHz stereo): audioop.ratecv keeps every third sample of a
    # linearly interpolated stream: length / 3, a constant stays constant, a slow ramp stays monotone
    ramp = (np.arange(48000) // 8).astype(np.int16)
    p3 = str(tmp_path / "ramp48k.wav")
    _write(p3, ramp.astype("<i2").tobytes(), 1, 2, 48000)
    r = aio.read_wav_int16(p3, 16000)
    assert abs(r.size - 16000) <= 1 and (np.diff(r.astype(np.int32)) >= 0).all() and abs(int(r[-1]) - int(ramp[-1])) <= 4
    const = np.full(24000, 1234, np.int16)
    p4 = str(tmp_path / "const24k.wav")
    _write(p4, const.astype("<i2").tobytes(), 1, 2, 24000)
    c = aio.read_wav_int16(p4, 16000)
    assert abs(c.size - 16000) <= 1 and (np.abs(c[8:].astype(np.int32) - 1234) <= 1).all()
    # 8-bit unsigned and 32-bit wavs are widened / narrowed to 16 bit
    p5 = str(tmp_path / "u8.wav")
    _write(p5, (np.full(1600, 128 + 64, np.uint8)).tobytes(), 1, 1, 16000)
    assert (aio.read_wav_int16(p5, 16000) == 64 * 256).all()


def test_dump_comparator_reports_first_difference(tmp_path):
    t = _tool()
    ours = {"family": "whisper", "precision": "f32", "files": [
        {"path": "a/en.wav", "n_samples": 10, "language": "en", "windows": [[1, 2, 3, 4], [7, 8]], "text": "hi"},
        {"path": "a/zh.wav", "n_samples": 10, "language": "zh", "windows": [[5, 6]], "text": None}]}
    same = json.loads(json.dumps(ours))
    same["files"][0]["path"] = "elsewhere/en.wav"                                    # matched by base name
    rep = t.compare(ours, same)
    assert rep["token_for_token"] and all(f["status"] == "equal" for f in rep["files"]) and rep["files"][0]["text_equal"]
    diff = json.loads(json.dumps(ours))
    diff["files"][0]["windows"][0][2] = 99
    diff["files"].pop(1)
    rep = t.compare(ours, diff)
    assert not rep["token_for_token"]
    m = rep["files"][0]["mismatches"][0]
    assert m["window"] == 0 and m["first_difference_at"] == 2 and m["ours"][0] == 3 and m["reference"][0] == 99
    assert rep["files"][1]["status"] == "missing in reference"


def test_the_parity_harness_rejects_wav_widths_whose_samples_differ_from_the_references(tmp_path):
    """A 24-bit wav is rescaled by the ingest (audioop.lin2lin) where the reference's dump truncates: tools/transcribe.py `run` takes 16-bit wav only unless
    --any-wav-width is given, so a dump that `compare` will judge cannot be made from samples the reference never saw."""
    import pytest
    aio = sub("audio_io")
    p24 = str(tmp_path / "mono24.wav")
    _write(p24, np.zeros(3 * 1600, np.uint8).tobytes(), 1, 3, 16000)
    assert aio.read_wav_int16(p24, 16000).shape == (1600,)                          # the library call converts ...
    with pytest.raises(ValueError, match="24-bit"):
        aio.read_wav_int16(p24, 16000, exact_width=True)                            # ... the strict form refuses
    src = open(os.path.join(ROOT, "tools", "transcribe.py")).read()
    assert src.count("exact_width=a.strict_wav") == 3 and "read_wav_int16(p, cfg.sample_rate)" not in src     # every ingest of `run` honours the switch
    t = _tool()
    ap_defaults = [a for a in ("--any-wav-width",) if a in src]
    assert ap_defaults and hasattr(t, "main")
