"""Audio ingest for the host loops: a .wav file -> int16 mono PCM at the model's sample rate.

The reference decodes with pydub (`AudioSegment.from_file(p).set_channels(1).set_frame_rate(SAMPLE_RATE)
.get_array_of_samples()`, SenseVoice/Inference_SenseVoice_ONNX.py:236-242, Whisper/Inference_Whisper_ONNX.py:735-741). For PCM
.wav input pydub's two conversions are thin wrappers over the standard library: `set_channels(1)` = `audioop.tomono(data, width,
0.5, 0.5)` (stereo) and `set_frame_rate(r)` = `audioop.ratecv(data, width, channels, rate, r, None)`; this module performs the same
two calls in the same order on the frames `wave` reads, so a **16-bit** wav file yields exactly the samples the reference would feed
its graphs -- without pydub / ffmpeg (neither ships here). Other sample widths are accepted but are NOT sample-exact against the
reference: it casts pydub's sample array straight to int16 (no rescale of 24- / 32-bit data), whereas this module rescales with
`audioop.lin2lin`; the parity harness (tools/transcribe.py compare) therefore takes 16-bit wav only (`exact_width=True`).
Compressed inputs (.mp3 ...) are out of scope: convert them to .wav first.
"""
from __future__ import annotations

import wave

try:
    import audioop                                   # standard library up to Python 3.12
except ImportError as e:                             # removed in 3.13 (PEP 594)
    raise ImportError("audio_io needs the standard-library `audioop` module (Python <= 3.12); on newer interpreters install the "
                      "`audioop-lts` backport or convert audio to 16 kHz mono int16 yourself and pass arrays") from e

import numpy as np


def read_wav_int16(path: str, sample_rate: int = 16000, exact_width: bool = False) -> np.ndarray:
    """Mono int16 samples at `sample_rate`. PCM wav of any width / channel count / rate; `exact_width` rejects everything but 16-bit
    input (the only width for which the samples equal the reference's)."""
    with wave.open(path, "rb") as w:
        n_ch, width, rate, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        if w.getcomptype() != "NONE":
            raise ValueError(f"{path}: compressed wav ({w.getcomptype()}) is not supported")
        if exact_width and width != 2:
            raise ValueError(f"{path}: {8 * width}-bit samples; sample-exact parity with the reference holds for 16-bit wav only")
        data = w.readframes(n)
    if width == 1:                                   # 8-bit wav is unsigned: pydub biases it to signed before any conversion
        data = audioop.bias(data, 1, -128)
    if n_ch == 2:
        data = audioop.tomono(data, width, 0.5, 0.5)
        n_ch = 1
    elif n_ch != 1:
        raise ValueError(f"{path}: {n_ch} channels (mono or stereo expected)")
    if rate != sample_rate:
        data, _ = audioop.ratecv(data, width, 1, rate, sample_rate, None)
    if width != 2:
        data = audioop.lin2lin(data, width, 2)
    return np.frombuffer(data, dtype="<i2").astype(np.int16)


def write_wav_int16(path: str, samples: np.ndarray, sample_rate: int = 16000) -> None:
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sample_rate)
        w.writeframes(np.ascontiguousarray(samples, dtype="<i2").tobytes())
