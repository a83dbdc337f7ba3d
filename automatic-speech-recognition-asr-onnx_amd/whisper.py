"""Whisper host loop: audio in -> token ids out, RTF -- the call surface of `Whisper/Inference_Whisper_ONNX.py`
(:721-842) on the native session (the three merged graphs collapse into `encode` / `prefill` / `generate`):

  prepare_audio_input()   = :103-126  int16 PCM -> model dtype, PCM scale 32768 (floats in [-1, 1])
  remove_repeated_parts() = :129-139  tail-repeat guard applied before detokenisation (:705-708)
  probe                   = _probe_prefill (:493-550): encoder + cross-KV + prefill([SOT]); language = arg-max over the
                            language-token logits (:793-798); no-speech gate: softmax(logits + 128 on suppressed
                            ids)[<|nospeech|>] >= 0.6 => skip (:799-805, NO_SPEECH_DETECTION Export_Whisper.py:334-348)
  prefill                 = _prefill (:437-490) with [SOT, language, task, <|notimestamps|>] (:807)
  decode                  = _decode_tokens (:584-663): greedy (REPEAT_PENALTY = 1.0) or penalty-greedy (any other value; the
                            multiplier applies once PENALTY_RANGE ids were generated, :630-632), limit MAX_SEQ_LEN - 4 (:821)
  windows of one file    = :745-806: stride SLIDING_WINDOW (or the window length), `ceil((len - window) / stride) + 1` windows, the
                            tail window zero-padded to the aligned length; the [SOT] probe (language, no-speech) runs on window 0
                            ONLY, a no-speech verdict aborts the whole file, later windows reuse window 0's language id; the
                            windows' ids are concatenated and the repeat guard sees the concatenation (:705-708)
Batch extension: `transcribe` takes a list of independent clips as one batch (language detection / no-speech per clip);
`transcribe_file` is the reference's per-file behaviour (the windows of one file form the batch).
"""
from __future__ import annotations

import time
from typing import Sequence

import numpy as np

from .config import WhisperConfig
from .engine import WhisperSession


def prepare_audio_input(audio_int16: np.ndarray, target_dtype=np.float32, *, audio_pcm_scale: int = 32768,
                        normalise: bool = False, target_rms: float = 4096.0) -> np.ndarray:
    target_dtype = np.dtype(target_dtype)
    if not normalise and target_dtype == np.int16:
        return np.ascontiguousarray(audio_int16, dtype=np.int16)
    audio = np.asarray(audio_int16).astype(np.float32)
    if normalise:
        rms = np.sqrt(np.mean(audio * audio, dtype=np.float32), dtype=np.float32)
        if rms > 0:
            audio *= target_rms / (rms + 1e-7)
            np.clip(audio, -float(audio_pcm_scale), float(audio_pcm_scale) - 1.0, out=audio)
    if target_dtype == np.int16:
        return np.ascontiguousarray(audio, dtype=np.int16)
    audio *= np.float32(1.0 / audio_pcm_scale)
    return np.ascontiguousarray(audio, dtype=target_dtype)


def remove_repeated_parts(ids: Sequence[int], repeat_words_threshold: int, ids_len: int):
    """Cut the sequence where a window of `threshold` ids re-occurs later (the reference's loop, :129-139)."""
    if ids_len <= repeat_words_threshold:
        return ids
    left = repeat_words_threshold // 2
    right = left + 1
    end = ids_len - left
    for i in range(left, end):
        for j in range(i + repeat_words_threshold, end):
            if all(ids[j + k] == ids[i + k] for k in range(-left, right)):
                return ids[:j - left]
    return ids


def plan_windows(audio_len: int, input_audio_length: int | None, sliding_window: int = 0):
    """(windows, stride, window_length, aligned_length) of one file, Inference_Whisper_ONNX.py:741-757.
    `input_audio_length=None` is the dynamic-axis export: the window is the whole file."""
    window = int(audio_len) if input_audio_length is None else int(input_audio_length)
    stride = window if sliding_window <= 0 else int(sliding_window)
    if audio_len <= window:
        windows = 1
    else:
        windows = int(np.ceil((audio_len - window) / stride)) + 1
    return windows, stride, window, (windows - 1) * stride + window


def no_speech_probability(logits: np.ndarray, suppress_tokens: Sequence[int], no_speech_id: int) -> np.ndarray:
    """softmax(logits + 128 on the permanently suppressed ids)[<|nospeech|>]  (Export_Whisper.py:334-348)."""
    x = np.asarray(logits, dtype=np.float32).copy()
    if suppress_tokens is not None:
        x[:, list(suppress_tokens)] += np.float32(128.0)
    x -= x.max(axis=1, keepdims=True)
    e = np.exp(x)
    return e[:, no_speech_id] / e.sum(axis=1)


class WhisperTranscriber:
    def __init__(self, cfg: WhisperConfig, session: WhisperSession, suppress_tokens=None, task: str = "transcribe",
                 detect_language: bool = True, no_speech_detection: bool = True, no_speech_threshold: float = 0.6,
                 remove_repeats: bool = True, repeat_penalty: float = 1.0, penalty_range: int = 20,
                 use_sampling: bool = False, temperature: float = 0.8, top_k: int = 10, top_p: float = 0.95,
                 sampling_repetition_penalty: float = 1.0, seed: int = 0):
        self.cfg, self.sess = cfg, session
        self.suppress_tokens = list(suppress_tokens) if suppress_tokens is not None else None
        self.task_token = cfg.transcribe_id if task == "transcribe" else cfg.translate_id
        self.detect_language, self.no_speech_detection = detect_language, no_speech_detection
        self.no_speech_threshold, self.remove_repeats = no_speech_threshold, remove_repeats
        self.language_token_ids = np.arange(cfg.first_language_id, cfg.first_language_id + cfg.n_languages, dtype=np.int64)
        self.stop_tokens = {cfg.eot_id}
        # REPEAT_PENALTY / PENALTY_RANGE (:77-79): 1.0 selects greedy, any other value penalty-greedy (the reference default is 0.8)
        self.repeat_penalty, self.penalty_range = float(repeat_penalty), int(penalty_range)
        # USE_SAMPLING / TEMPERATURE / TOP_K / TOP_P / SAMPLING_REPETITION_PENALTY (:71-75)
        self.sampling = (bool(use_sampling), float(temperature), int(top_k), float(top_p), float(sampling_repetition_penalty), int(seed))

    def transcribe(self, clips_int16: Sequence[np.ndarray], language_ids: Sequence[int] | None = None, max_new: int | None = None):
        """List of int16 mono 16 kHz clips (each <= 30 s) -> per clip dict(tokens, language_id, no_speech_prob, skipped)."""
        cfg = self.cfg
        audios = [prepare_audio_input(np.asarray(c, dtype=np.int16).reshape(-1)) for c in clips_int16]
        B = len(audios)
        lang = np.asarray(language_ids if language_ids is not None else [cfg.first_language_id] * B, dtype=np.int64)
        t0 = time.time()
        self.sess.encode(audios)                                         # STFT + encoder + cross-KV, once per window
        probs = np.zeros(B, dtype=np.float32)
        if self.detect_language or self.no_speech_detection:
            self.sess.set_sampling(False)
            self.sess.set_penalty(1.0, self.penalty_range)
            _, logits = self.sess.prefill(np.full((B, 1), cfg.sot_id, dtype=np.int32))      # probe with [SOT]
            if self.detect_language:
                lang = self.language_token_ids[np.argmax(logits[:, self.language_token_ids], axis=1)]
            if self.no_speech_detection:
                probs = self.sess.no_speech_prob(cfg.no_speech_id)          # device head over the probe's logits
        skipped = probs >= self.no_speech_threshold if self.no_speech_detection else np.zeros(B, dtype=bool)
        prompt = np.stack([[cfg.sot_id, int(l), self.task_token, cfg.no_timestamps_id] for l in lang]).astype(np.int32)
        limit = max(0, cfg.max_target_positions - prompt.shape[1])
        if max_new is not None:
            limit = min(limit, max_new)
        self.sess.set_penalty(self.repeat_penalty, self.penalty_range)
        self.sess.set_sampling(*self.sampling)
        self.sess.prefill(prompt, want_logits=False)
        toks = self.sess.generate(limit, eos_id=cfg.eot_id) if limit > 0 else [np.zeros(0, np.int32)] * B
        wall = time.time() - t0
        out = []
        for b in range(B):
            ids = [] if skipped[b] else toks[b].tolist()
            if self.remove_repeats:
                ids = list(remove_repeated_parts(ids, 3, len(ids)))
            out.append({"tokens": np.asarray(ids, dtype=np.int32), "language_id": int(lang[b]), "no_speech_prob": float(probs[b]),
                        "skipped": bool(skipped[b])})
        total_s = sum(a.size for a in audios) / cfg.sample_rate
        return out, {"rtf": wall / total_s, "wall_s": wall}

    def transcribe_file(self, pcm_int16: np.ndarray, language_id: int | None = None, sliding_window: int = 0,
                        input_audio_length: int | None = -1, max_new: int | None = None):
        """One file the way the reference's loop walks it (Inference_Whisper_ONNX.py:741-829). input_audio_length: the encoder's static audio
        dimension, None = dynamic axis (one unpadded window), -1 = dynamic when the file fits the encoder, else windows of cfg.max_audio_len.
        -> dict(tokens = the windows' ids concatenated (repeat guard applied when enabled), windows = per-window ids,
                language_id, no_speech_prob, no_speech), stats.
        The windows are independent once window 0's probe has fixed the language, so they run as ONE batch; the probe's [SOT]
        prefill is evaluated for the batch but only window 0's row is read (the reference never probes a later window)."""
        cfg = self.cfg
        raw = np.asarray(pcm_int16, dtype=np.int16).reshape(-1)
        audio_len = int(raw.size)
        if input_audio_length == -1:
            # the reference's default export keeps the audio axis dynamic (Export_Whisper.py:743): the window is the file, unpadded. A file longer than
            # the encoder's position table can only run through the static-axis export: windows of max_audio_len, the tail zero-padded.
            input_audio_length = None if audio_len <= cfg.max_audio_len else cfg.max_audio_len
        audio = prepare_audio_input(raw)
        n_win, stride, window, aligned = plan_windows(audio_len, input_audio_length, sliding_window)
        if audio.size < aligned:                                         # zero-padded tail (:751-757)
            audio = np.concatenate([audio, np.zeros(aligned - audio.size, dtype=audio.dtype)])
        clips = [np.ascontiguousarray(audio[w * stride:w * stride + window]) for w in range(n_win)]
        lang = cfg.first_language_id if language_id is None else int(language_id)
        t0 = time.time()
        self.sess.encode(clips)
        prob, no_speech = 0.0, False
        if self.detect_language or self.no_speech_detection:            # needs_probe: window 0 only (:768)
            self.sess.set_sampling(False)
            self.sess.set_penalty(1.0, self.penalty_range)
            _, logits = self.sess.prefill(np.full((n_win, 1), cfg.sot_id, dtype=np.int32))
            if self.detect_language:
                lang = int(self.language_token_ids[np.argmax(logits[0, self.language_token_ids])])
            if self.no_speech_detection:
                prob = float(self.sess.no_speech_prob(cfg.no_speech_id)[0])
                no_speech = prob >= self.no_speech_threshold             # aborts the file (:801-805)
        windows: list[list[int]] = []
        if not no_speech:
            prompt = np.tile(np.asarray([[cfg.sot_id, lang, self.task_token, cfg.no_timestamps_id]], dtype=np.int32), (n_win, 1))
            limit = max(0, cfg.max_target_positions - prompt.shape[1])
            if max_new is not None:
                limit = min(limit, max_new)
            self.sess.set_penalty(self.repeat_penalty, self.penalty_range)
            self.sess.set_sampling(*self.sampling)
            self.sess.prefill(prompt, want_logits=False)
            toks = self.sess.generate(limit, eos_id=cfg.eot_id) if limit > 0 else [np.zeros(0, np.int32)] * n_win
            windows = [t.astype(int).tolist() for t in toks]
        wall = time.time() - t0
        ids = [t for w in windows for t in w]
        if self.remove_repeats:
            ids = list(remove_repeated_parts(ids, 3, len(ids)))
        res = {"tokens": np.asarray(ids, dtype=np.int32), "windows": windows, "language_id": lang, "no_speech_prob": prob,
               "no_speech": bool(no_speech), "n_windows": n_win, "stride": stride, "window": window}
        return res, {"rtf": wall / max(audio_len / cfg.sample_rate, 1e-9), "wall_s": wall}
