"""Test infrastructure (never imported by the product path): the NON-NOISE audio clips of the parity suite.

Every other fixture of this repo is white noise (checkpoints.synth_audio): a flat spectrum never reaches the branches the reference's
front-ends exist to pin -- the clamp(FLT_EPS) in front of the Kaldi log (SenseVoice/Export_SenseVoice.py:275-278), Whisper's
clamp(1e-10) -> log10 -> max(x, global_max - 8) (Whisper/Export_Whisper.py:424-427) and the per-utterance global maximum under
batching. These clips do:

  silence      digital silence, 8 s                                    -> every mel bin at the log floor
  dc_clip      DC offset + a full-scale clipped 50 Hz square, 8 s     -> DC removal / pre-emphasis on saturated int16, huge low bins over an empty top
  chirp80      100 Hz -> 7 kHz sweep fading 80 dB, 8 s                -> 60-80 dB of bin-to-bin range in one frame sequence (the split-operand bf16 DFT's hard case)
  zh_1         /root/reference/Test_Examples/zh/zh_1.wav, 2.4 s       \
  shanghai8    .../zh/zh-Shanghai.wav, first 8 s                       > the reference's own example speech (Example_Audio.py:6-18); 16 kHz mono int16
  multitalk8   .../en/test_sample_multitalker_overlap.wav, 1 s .. 9 s /

Audio is data: the three speech clips are committed as int16 arrays in tests/golden/audio_natural.npz (written by oracle/gen_golden_natural.py in the
build container); the three synthetic ones are rebuilt from the formulas below, bit for bit (integer arithmetic after one rounding).
"""
from __future__ import annotations

import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLIPS_NPZ = os.path.join(ROOT, "tests", "golden", "audio_natural.npz")
SPEECH = {
    "zh_1": ("zh/zh_1.wav", 0, 38910),
    "shanghai8": ("zh/zh-Shanghai.wav", 0, 128000),
    "multitalk8": ("en/test_sample_multitalker_overlap.wav", 16000, 128000),
}
ORDER = ("silence", "dc_clip", "chirp80", "zh_1", "shanghai8", "multitalk8")
N8 = 128000


def synthetic_clips() -> dict:
    t = np.arange(N8, dtype=np.float64) / 16000.0
    square = np.where(np.sin(2 * np.pi * 50.0 * t) >= 0, 1.0, -1.0)
    dc_clip = np.clip(np.round(12000.0 + 40000.0 * square), -32768, 32767).astype(np.int16)          # saturates on both sides
    phase = 2 * np.pi * (100.0 * t + 0.5 * (7000.0 - 100.0) / 8.0 * t * t)
    amp = 30000.0 * 10.0 ** (-4.0 * t / 8.0)                                                           # 80 dB down at the end
    chirp = np.round(amp * np.sin(phase)).astype(np.int16)
    return {"silence": np.zeros(N8, np.int16), "dc_clip": dc_clip, "chirp80": chirp}


def read_reference_wavs(ref_root="/root/reference/Test_Examples") -> dict:
    """Build container only (the generator): the reference's example speech as int16."""
    import wave
    out = {}
    for name, (rel, start, n) in SPEECH.items():
        with wave.open(os.path.join(ref_root, rel)) as w:
            assert w.getnchannels() == 1 and w.getsampwidth() == 2 and w.getframerate() == 16000, rel
            pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
        out[name] = pcm[start:start + n].copy()
        assert out[name].size == n, (name, out[name].size)
    return out


def load_clips() -> dict:
    """name -> int16 array, in ORDER (tests and the oracle checks; reads the committed speech arrays)."""
    clips = synthetic_clips()
    z = np.load(CLIPS_NPZ)
    for name in SPEECH:
        clips[name] = z[name]
    return {k: clips[k] for k in ORDER}


def kaldi_input(pcm: np.ndarray) -> np.ndarray:
    """int16-range float32, the Kaldi front-ends' input (SenseVoice / Paraformer)."""
    return pcm.astype(np.float32)


def unit_input(pcm: np.ndarray) -> np.ndarray:
    """[-1, 1) float32, the Whisper / Qwen3-ASR front-ends' input."""
    return (pcm.astype(np.float32) / np.float32(32768.0)).astype(np.float32)


STREAM_CHUNK = 8000


def streaming_clips(clips: dict | None = None) -> dict:
    """Composite clips for the STREAMING graphs (whole 0.5 s chunks; the carried CIF / K-V / FSMN state has to survive stretches of digital silence in which
    nothing fires, then fire again): name -> int16.
      sil_speech_sil   1 s of silence, zh_1 (2.43 s, zero-padded to whole chunks), 1.5 s of silence                      = 10 chunks
      talk_gap_talk    multitalk8[0 : 2 s], 1 s of silence, shanghai8[2 s : 4 s], chirp80[0 : 0.5 s]                     = 11 chunks"""
    c = clips or load_clips()
    z = lambda n: np.zeros(n, np.int16)
    zh = c["zh_1"]
    pad = (-zh.size) % STREAM_CHUNK
    out = {
        "sil_speech_sil": np.concatenate([z(16000), zh, z(pad), z(24000)]),
        "talk_gap_talk": np.concatenate([c["multitalk8"][:32000], z(16000), c["shanghai8"][32000:64000], c["chirp80"][:8000]]),
    }
    for k, v in out.items():
        assert v.size % STREAM_CHUNK == 0 and v.dtype == np.int16, k
    return out
