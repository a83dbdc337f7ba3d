"""SenseVoice host loop: audio in -> token ids / text out, RTF -- the call surface of
`SenseVoice/Inference_SenseVoice_ONNX.py`, written against the onnxruntime-API shim so the structure
(bind -> run_with_iobinding -> read ids) is the reference's own (:236-310).

  export_sensevoice()      ~ Export_SenseVoice.py tail (:371-405): writes SenseVoiceSmall.asrmodel + ASR_Metadata.asrmodel
  prepare_audio_input()    = :62-86   (int16 PCM -> model dtype, optional RMS normalisation, immutable PCM scale)
  plan_windows()           = :243-260 (pad / sliding-window geometry)
  transcribe()             = :262-310 (per-window bind / run / decode, RTF = wall / audio seconds)
"""
from __future__ import annotations

import os
import time

import numpy as np

from . import ort_shim as onnxruntime
from .arena import build_sensevoice_arena
from .config import SenseVoiceConfig
from .ort_io import array_for, filled_for, is_dynamic_dim, load_supported_languages, numpy_dtype, resolve_supported_language

C = onnxruntime  # `C.OrtDevice` in the reference scripts


def export_sensevoice(folder: str, cfg: SenseVoiceConfig, ck: dict, precision: int = 0) -> None:
    os.makedirs(folder, exist_ok=True)
    meta = onnxruntime.sensevoice_metadata(cfg)
    onnxruntime.save_model(os.path.join(folder, "SenseVoiceSmall.asrmodel"), "sensevoice", cfg.to_dict(),
                           build_sensevoice_arena(cfg, ck, precision), {}, precision)
    onnxruntime.save_model(os.path.join(folder, "ASR_Metadata.asrmodel"), "metadata", None, None, meta)


def prepare_audio_input(audio_int16: np.ndarray, input_audio_dtype: str, *, audio_pcm_scale: int, normalise: bool = False,
                        target_rms: float = 4096.0) -> np.ndarray:
    if not normalise and input_audio_dtype == "INT16":
        return np.ascontiguousarray(audio_int16, dtype=np.int16)
    audio = audio_int16.astype(np.float32)
    if normalise:
        rms = np.sqrt(np.mean(audio * audio, dtype=np.float32), dtype=np.float32)
        if rms > 0:
            audio *= (target_rms / (rms + 1e-7))
            np.clip(audio, -32768.0, 32767.0, out=audio)
    if input_audio_dtype == "INT16":
        return audio.astype(np.int16)
    audio *= np.float32(1.0 / audio_pcm_scale)
    return audio.astype(np.float16) if input_audio_dtype == "F16" else audio


def plan_windows(audio: np.ndarray, audio_len: int, window: int, stride: int) -> np.ndarray:
    """Zero-pad (1,1,L) audio so that windows of `window` samples every `stride` cover it."""
    if audio_len > window:
        n = int(np.ceil((audio_len - window) / stride)) + 1
        need = (n - 1) * stride + window
        audio = np.concatenate((audio, np.zeros((1, 1, need - audio_len), dtype=audio.dtype)), axis=-1)
    elif audio_len < window:
        audio = np.concatenate((audio, np.zeros((1, 1, window - audio_len), dtype=audio.dtype)), axis=-1)
    return audio


class SenseVoiceTranscriber:
    def __init__(self, model_folder: str, target_language: str = "en", tokenizer_path: str | None = None, device_id: int = 0,
                 device_type: str = "cpu"):
        opts = onnxruntime.SessionOptions()
        opts.execution_mode = onnxruntime.ExecutionMode.ORT_SEQUENTIAL
        self.run_options = onnxruntime.RunOptions()
        self.run_options.add_run_config_entry("disable_synchronize_execution_providers", "0")
        self.session_meta = onnxruntime.InferenceSession(os.path.join(model_folder, "ASR_Metadata.onnx"), sess_options=opts)
        self.session = onnxruntime.InferenceSession(os.path.join(model_folder, "SenseVoiceSmall.onnx"), sess_options=opts,
                                                    device_id=device_id)
        ins, outs = self.session.get_inputs(), self.session.get_outputs()
        self.audio_meta, self.lang_meta = ins[0], ins[1]
        self.out_name0 = outs[0].name
        self.device_type, self.device_id = device_type, device_id
        self.ort_device = C.OrtDevice(C.OrtDevice.cuda() if device_type != "cpu" else C.OrtDevice.cpu(),
                                      C.OrtDevice.default_memory(), device_id)
        dt = numpy_dtype(self.audio_meta)
        self.input_audio_dtype = "INT16" if dt == np.int16 else "F16" if dt == np.float16 else "F32"
        meta = self.session_meta.get_modelmeta().custom_metadata_map or {}
        self.sample_rate = int(meta["sample_rate"])
        self.audio_pcm_scale = int(meta["audio_pcm_scale"])
        self.languages = load_supported_languages(meta)
        self.language, entry = resolve_supported_language(self.languages, target_language)
        self.selector_index = entry.get("selector_index")
        self.tokenizer = None
        if tokenizer_path and os.path.isfile(tokenizer_path):
            from sentencepiece import SentencePieceProcessor
            self.tokenizer = SentencePieceProcessor()
            self.tokenizer.Load(tokenizer_path)

    def transcribe(self, audio_int16: np.ndarray, sliding_window: int = 0, normalise: bool = False):
        """int16 mono PCM at `sample_rate` -> dict(token_ids, text, rtf, windows)."""
        audio_len = int(np.asarray(audio_int16).size)
        audio = prepare_audio_input(np.asarray(audio_int16, dtype=np.int16).reshape(1, 1, -1), self.input_audio_dtype,
                                    audio_pcm_scale=self.audio_pcm_scale, normalise=normalise)
        shape_in = self.audio_meta.shape[-1]
        window = audio_len if is_dynamic_dim(shape_in) else int(shape_in)
        stride = window if sliding_window <= 0 else sliding_window
        audio = plan_windows(audio, audio_len, window, stride)
        aligned = audio.shape[-1]
        binding = self.session.io_binding()
        language_idx = filled_for(self.lang_meta, self.selector_index, axes={0: 1})
        if self.device_type == "cpu":
            audio_buffer = None
            binding.bind_cpu_input(self.lang_meta.name, language_idx)
        else:
            audio_buffer = onnxruntime.OrtValue.ortvalue_from_numpy(
                filled_for(self.audio_meta, axes={0: 1, 1: 1, 2: window}), self.device_type, self.device_id)
            language_buffer = onnxruntime.OrtValue.ortvalue_from_numpy(language_idx, "cpu", 0)
            binding.bind_ortvalue_input(self.audio_meta.name, audio_buffer)
            binding.bind_ortvalue_input(self.lang_meta.name, language_buffer)
        ids_all, text = [], ""
        start, end = 0, window
        t0 = time.time()
        while end <= aligned:
            win = array_for(self.audio_meta, audio[:, :, start:end], axes={0: 1, 1: 1, 2: window})
            if audio_buffer is None:
                binding.bind_cpu_input(self.audio_meta.name, win)
            else:
                audio_buffer.update_inplace(win)
            binding._iobinding.bind_output(self.out_name0, self.ort_device)     # token count is data dependent: re-bind per run
            self.session.run_with_iobinding(binding, run_options=self.run_options)
            token_ids = binding.get_outputs()[0].numpy()
            ids_all.append(token_ids.copy())
            if self.tokenizer is not None:
                text += self.tokenizer.decode([token_ids.tolist()])[0]
            start += stride
            end = start + window
        wall = time.time() - t0
        return {"token_ids": ids_all, "text": text if self.tokenizer is not None else None,
                "rtf": wall / (audio_len / self.sample_rate), "windows": len(ids_all), "language": self.language}
