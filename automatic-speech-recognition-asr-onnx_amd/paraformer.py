"""Paraformer (non-streaming) host loop: audio in -> token ids / text out, RTF -- the call surface of
`Paraformer/Non-Streaming/Inference_Paraformer_ONNX.py`, written against the onnxruntime-API shim:

  export_paraformer()   ~ Export_Paraformer.py tail (:583-640): Paraformer.asrmodel + ASR_Metadata.asrmodel + Vocab_Paraformer.txt
  build_tokenizer_metadata() = :136-159 (blank/eos/stop ids by token role, one artefact language with its decode mode)
  decode_tokens()       = Inference_Paraformer_ONNX.py:86-89 ("en": BPE "@@ " joins; otherwise plain concatenation)
  transcribe()          = :236-297 (window/pad, one pre-bound audio buffer updated in place, per-window run, stop ids stripped)
prepare_audio_input / plan_windows are the SenseVoice ones (identical code in the reference, :62-84, :243-259).
"""
from __future__ import annotations

import json
import os
import time
from typing import Sequence

import numpy as np

from . import ort_shim as onnxruntime
from .arena import build_paraformer_arena
from .config import ParaformerConfig
from .ort_io import (array_for, filled_for, is_dynamic_dim, load_special_token_ids, load_supported_languages, numpy_dtype,
                     resolve_supported_language)
from .sensevoice import plan_windows, prepare_audio_input

C = onnxruntime

_ROLE_TOKENS = {"blank": ("<blank>", "<blk>", "<eps>"), "eos": ("</s>", "<eos>"), "unknown": ("<unk>", "<unknown>", "[UNK]"),
                "pad": ("<pad>", "[PAD]"), "bos": ("<s>", "<bos>", "<sos>")}
_LANGUAGES = {"zh": ("Chinese", ["Chinese", "Mandarin", "zh-CN", "中文"]), "en": ("English", ["English", "en-US"])}


def build_tokenizer_metadata(token_list: Sequence[str], language: str, decode_mode: str):
    tokens = list(token_list)

    def find(role):
        return next((i for i, t in enumerate(tokens) if t in _ROLE_TOKENS[role]), None)

    special = {"blank": find("blank"), "eos": find("eos"), "stop": [find("eos")]}
    for role in ("unknown", "pad", "bos"):
        if find(role) is not None:
            special[role] = find(role)
    name, aliases = _LANGUAGES[language]
    return special, {language: {"name": name, "aliases": aliases, "prompt_token_ids": [], "decode_mode": decode_mode}}


def export_paraformer(folder: str, cfg: ParaformerConfig, ck: dict, token_list: Sequence[str], language: str = "zh",
                      decode_mode: str = "zh", precision: int = 0) -> None:
    os.makedirs(folder, exist_ok=True)
    special, langs = build_tokenizer_metadata(token_list, language, decode_mode)
    meta = {"sample_rate": str(cfg.sample_rate), "audio_pcm_scale": "1",
            "special_token_ids": json.dumps(special, separators=(",", ":")),
            "supported_languages": json.dumps(langs, ensure_ascii=False, sort_keys=True, separators=(",", ":"))}
    onnxruntime.save_model(os.path.join(folder, "Paraformer.asrmodel"), "paraformer", cfg.to_dict(),
                           build_paraformer_arena(cfg, ck, precision), {}, precision)
    onnxruntime.save_model(os.path.join(folder, "ASR_Metadata.asrmodel"), "metadata", None, None, meta)
    with open(os.path.join(folder, "Vocab_Paraformer.txt"), "w", encoding="utf-8") as f:
        for t in token_list:
            f.write(f"{t}\n")


def decode_tokens(tokens: Sequence[str], mode: str) -> str:
    if mode == "en":
        return " ".join(tokens).replace("@@ ", "").strip()
    return "".join(tokens).strip()


class ParaformerTranscriber:
    def __init__(self, model_folder: str, vocab_path: str | None = None, device_id: int = 0, device_type: str = "cpu"):
        opts = onnxruntime.SessionOptions()
        opts.execution_mode = onnxruntime.ExecutionMode.ORT_SEQUENTIAL
        self.run_options = onnxruntime.RunOptions()
        self.run_options.add_run_config_entry("disable_synchronize_execution_providers", "0")
        self.session_meta = onnxruntime.InferenceSession(os.path.join(model_folder, "ASR_Metadata.onnx"), sess_options=opts)
        self.session = onnxruntime.InferenceSession(os.path.join(model_folder, "Paraformer.onnx"), sess_options=opts, device_id=device_id)
        self.audio_meta = self.session.get_inputs()[0]
        self.out_names = [o.name for o in self.session.get_outputs()]
        self.device_type, self.device_id = device_type, device_id
        self.ort_device = C.OrtDevice(C.OrtDevice.cuda() if device_type != "cpu" else C.OrtDevice.cpu(),
                                      C.OrtDevice.default_memory(), device_id)
        dt = numpy_dtype(self.audio_meta)
        self.input_audio_dtype = "INT16" if dt == np.int16 else "F16" if dt == np.float16 else "F32"
        meta = self.session_meta.get_modelmeta().custom_metadata_map or {}
        self.sample_rate = int(meta["sample_rate"])
        self.audio_pcm_scale = int(meta["audio_pcm_scale"])
        stop = load_special_token_ids(meta)["stop"]
        self.stop_token_ids = stop if isinstance(stop, list) else [stop]
        languages = load_supported_languages(meta)
        self.language, entry = resolve_supported_language(languages, next(iter(languages)))
        self.decode_mode = entry.get("decode_mode")
        with open(vocab_path or os.path.join(model_folder, "Vocab_Paraformer.txt"), "r", encoding="UTF-8") as f:
            self.tokenizer = np.array([line.rstrip("\n") for line in f], dtype=np.str_)

    def transcribe(self, audio_int16: np.ndarray, sliding_window: int = 0, normalise: bool = False):
        """int16 mono PCM at `sample_rate` -> dict(token_ids per window, text, rtf, windows)."""
        audio_len = int(np.asarray(audio_int16).size)
        audio = prepare_audio_input(np.asarray(audio_int16, dtype=np.int16).reshape(1, 1, -1), self.input_audio_dtype,
                                    audio_pcm_scale=self.audio_pcm_scale, normalise=normalise)
        shape_in = self.audio_meta.shape[-1]
        window = audio_len if is_dynamic_dim(shape_in) else int(shape_in)
        stride = window if sliding_window <= 0 else sliding_window
        audio = plan_windows(audio, audio_len, window, stride)
        aligned = audio.shape[-1]
        binding = self.session.io_binding()
        audio_buffer = onnxruntime.OrtValue.ortvalue_from_numpy(
            filled_for(self.audio_meta, axes={0: 1, 1: 1, 2: window}), self.device_type, self.device_id)
        binding.bind_ortvalue_input(self.audio_meta.name, audio_buffer)
        ids_all, pieces = [], []
        start, end = 0, window
        t0 = time.time()
        while end <= aligned:
            audio_buffer.update_inplace(array_for(self.audio_meta, audio[:, :, start:end], axes={0: 1, 1: 1, 2: window}))
            for name in self.out_names:                                         # data-dependent token count: re-arm per window
                binding._iobinding.bind_output(name, self.ort_device)
            self.session.run_with_iobinding(binding, run_options=self.run_options)
            token_ids = binding.get_outputs()[0].numpy().reshape(-1)
            token_ids = token_ids[~np.isin(token_ids, self.stop_token_ids)]
            ids_all.append(token_ids.copy())
            pieces.extend(self.tokenizer[token_ids].tolist())
            start += stride
            end = start + window
        wall = time.time() - t0
        return {"token_ids": ids_all, "text": decode_tokens(pieces, self.decode_mode), "rtf": wall / (audio_len / self.sample_rate),
                "windows": len(ids_all), "language": self.language}
