"""The onnxruntime Python-API subset the in-scope `Inference_*_ONNX.py` scripts use, backed by the C ABI.

This is the drop-in boundary of SURVEY.md section 8(b): the reference's host loops talk to
`onnxruntime.InferenceSession / SessionOptions / RunOptions / OrtValue / IOBinding / C.OrtDevice`
(SenseVoice/Inference_SenseVoice_ONNX.py:96-199,262-305; Whisper/Inference_Whisper_ONNX.py:142-277,
430-663). The same calls here drive `libasr_mi355x.so`:

    import importlib; shim = importlib.import_module("automatic-speech-recognition-asr-onnx_amd.ort_shim")
    shim.install_as_onnxruntime()       # `import onnxruntime`, `from onnxruntime.capi import _pybind_state as C`

A "model file" is an `.asrmodel` bundle written by `save_model()` (JSON header + weight arena); a path ending
in `.onnx` resolves to the sibling `.asrmodel`, so the reference's folder layout and file names carry over.
Graph I/O names / shapes / dtypes are the reference's (Export_SenseVoice.py:375-379). Extension: a leading
batch axis > 1 on `audio` / `language_idx` runs independent utterances in one call and returns
`token_ids` as (B, max_tokens) + `num_id` (B,).

Errors are not swallowed: C-ABI failures raise `_lib.AsrError` (ORT raises too), binding mistakes raise ValueError.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import struct
import sys
import types
from typing import Any, Sequence

import numpy as np

from . import _lib
from .config import ParaformerConfig, SenseVoiceConfig

MAGIC = b"ASRMODEL"
_ORT_TYPES = {np.dtype(np.float32): "tensor(float)", np.dtype(np.float16): "tensor(float16)", np.dtype(np.int16): "tensor(int16)",
              np.dtype(np.int32): "tensor(int32)", np.dtype(np.int64): "tensor(int64)"}


# ------------------------------------------------------------------------------------- model bundles
def save_model(path: str, kind: str, config: dict | None, arena: np.ndarray | None, metadata: dict[str, str],
               precision: int = 0) -> None:
    header = json.dumps({"kind": kind, "config": config, "metadata": {str(k): str(v) for k, v in metadata.items()},
                         "precision": int(precision)}, ensure_ascii=False).encode("utf-8")
    pad = (-(16 + len(header))) % 256
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<Q", len(header)) + header + b"\0" * pad)
        if arena is not None:
            f.write(np.ascontiguousarray(arena, dtype=np.uint8).tobytes())


def load_model(path: str):
    if not os.path.isfile(path) and path.endswith(".onnx"):
        path = path[:-5] + ".asrmodel"
    with open(path, "rb") as f:
        head = f.read(16)
        if head[:8] != MAGIC:
            raise ValueError(f"{path!r} is not an .asrmodel bundle (this engine does not execute ONNX graphs)")
        (n,) = struct.unpack("<Q", head[8:])
        info = json.loads(f.read(n).decode("utf-8"))
        f.seek((16 + n + 255) // 256 * 256)
        blob = np.frombuffer(f.read(), dtype=np.uint8)
    return info, (blob if blob.size else None)


def sensevoice_metadata(cfg: SenseVoiceConfig) -> dict[str, str]:
    """Metadata carried by ASR_Metadata.onnx (Export_SenseVoice.py:36-50,298-310,395-403)."""
    profiles = (("auto", "Automatic language detection", ["automatic", "detect"]), ("zh", "Chinese", ["Chinese", "Mandarin", "zh-CN", "中文"]),
                ("en", "English", ["English", "en-US"]), ("yue", "Cantonese", ["Cantonese", "zh-yue", "粤语", "粵語"]),
                ("ja", "Japanese", ["Japanese", "jp", "日本語"]), ("ko", "Korean", ["Korean", "kr", "한국어"]),
                ("nospeech", "No speech", ["no-speech", "silence"]))
    langs = {code: {"name": name, "aliases": aliases, "selector_index": i, "prompt_token_ids": [cfg.language_prompt_token_ids[i]]}
             for i, (code, name, aliases) in enumerate(profiles)}
    return {"sample_rate": str(cfg.sample_rate), "audio_pcm_scale": "1",
            "supported_languages": json.dumps(langs, ensure_ascii=False, sort_keys=True, separators=(",", ":"))}


# ------------------------------------------------------------------------------------- API objects
class ExecutionMode:
    ORT_SEQUENTIAL, ORT_PARALLEL = 0, 1


class GraphOptimizationLevel:
    ORT_DISABLE_ALL, ORT_ENABLE_BASIC, ORT_ENABLE_EXTENDED, ORT_ENABLE_ALL = 0, 1, 2, 99


class SessionOptions:
    """Attributes are accepted for source compatibility; the HIP engine has no graph optimiser or CPU thread pool."""

    def __init__(self):
        self.log_severity_level = 2
        self.log_verbosity_level = 0
        self.inter_op_num_threads = 0
        self.intra_op_num_threads = 0
        self.enable_cpu_mem_arena = True
        self.execution_mode = ExecutionMode.ORT_SEQUENTIAL
        self.graph_optimization_level = GraphOptimizationLevel.ORT_ENABLE_ALL
        self.config_entries: dict[str, str] = {}
        self.initializers: dict[str, "OrtValue"] = {}

    def add_session_config_entry(self, key: str, value: str):
        self.config_entries[str(key)] = str(value)

    def add_initializer(self, name: str, value: "OrtValue"):
        self.initializers[name] = value          # externally owned weights; must outlive the session


class RunOptions:
    def __init__(self):
        self.log_severity_level = 2
        self.log_verbosity_level = 0
        self.config_entries: dict[str, str] = {}

    def add_run_config_entry(self, key: str, value: str):
        self.config_entries[str(key)] = str(value)   # runs are always synchronous in this engine


class OrtDevice:
    _CPU, _HIP = 0, 1

    def __init__(self, device_type=0, memory_type=0, device_id=0):
        self.type, self.memory_type, self.device_id = device_type, memory_type, device_id

    @staticmethod
    def cpu():
        return OrtDevice._CPU

    @staticmethod
    def cuda():          # reference scripts say "cuda" for any accelerator handle; it is the MI355X here
        return OrtDevice._HIP

    hip = cuda
    dml = cuda

    @staticmethod
    def default_memory():
        return 0


class NodeArg:
    def __init__(self, name: str, shape: Sequence[Any], dtype):
        self.name, self.shape, self.type = name, list(shape), _ORT_TYPES[np.dtype(dtype)]

    def __repr__(self):
        return f"NodeArg(name={self.name!r}, type={self.type!r}, shape={self.shape!r})"


class ModelMetadata:
    def __init__(self, custom: dict[str, str]):
        self.custom_metadata_map = dict(custom)


class OrtValue:
    """Host array or an HBM allocation owned by this object."""

    def __init__(self, array: np.ndarray | None, device_type="cpu", device_id=0):
        self._shape, self._dtype = tuple(array.shape), array.dtype
        self._device_type, self._device_id = device_type, device_id
        self._host, self._dptr = None, C.c_void_p(None)
        if device_type == "cpu":
            self._host = np.ascontiguousarray(array)
        else:
            _lib.check(_lib.load().asr_mem_alloc(device_id, array.nbytes, C.byref(self._dptr)))
            self.update_inplace(array)

    @staticmethod
    def ortvalue_from_numpy(array: np.ndarray, device_type: str = "cpu", device_id: int = 0) -> "OrtValue":
        return OrtValue(np.asarray(array), device_type, device_id)

    def update_inplace(self, array: np.ndarray):
        a = np.ascontiguousarray(array, dtype=self._dtype)
        if a.shape != self._shape:
            raise ValueError(f"update_inplace: shape {a.shape} != allocation {self._shape}")
        if self._host is not None:
            self._host[...] = a
        else:
            _lib.check(_lib.load().asr_mem_copy(self._device_id, self._dptr, a.ctypes.data_as(C.c_void_p), a.nbytes, 0))

    def numpy(self) -> np.ndarray:
        if self._host is not None:
            return self._host
        out = np.empty(self._shape, dtype=self._dtype)
        _lib.check(_lib.load().asr_mem_copy(self._device_id, out.ctypes.data_as(C.c_void_p), self._dptr, out.nbytes, 1))
        return out

    def shape(self):
        return list(self._shape)

    def device_name(self):
        return self._device_type

    def data_ptr(self):
        return self._dptr.value if self._host is None else self._host.ctypes.data

    def __del__(self):
        try:
            if self._host is None and self._dptr:
                _lib.load().asr_mem_free(self._device_id, self._dptr)
        except Exception:
            pass


class _RawBinding:
    def __init__(self, owner: "IOBinding"):
        self._owner = owner

    def bind_output(self, name: str, device):          # device-auto allocation for data-dependent shapes
        self._owner._out_requests[name] = None


class IOBinding:
    def __init__(self, session: "InferenceSession"):
        self._session = session
        self._inputs: dict[str, OrtValue] = {}
        self._out_requests: dict[str, OrtValue | None] = {}
        self._outputs: list[OrtValue] = []
        self._iobinding = _RawBinding(self)

    def _check_input(self, name):
        if name not in self._session._input_names:
            raise ValueError(f"{name!r} is not an input of this graph (inputs: {self._session._input_names})")

    def bind_cpu_input(self, name: str, array: np.ndarray):
        self._check_input(name)
        self._inputs[name] = OrtValue(np.asarray(array), "cpu", 0)

    def bind_ortvalue_input(self, name: str, value: OrtValue):
        self._check_input(name)
        self._inputs[name] = value

    def bind_ortvalue_output(self, name: str, value: OrtValue):
        self._out_requests[name] = value

    def bind_output(self, name: str, device_type="cpu", device_id=0, *unused):
        self._out_requests[name] = None

    def clear_binding_outputs(self):
        self._out_requests.clear()
        self._outputs = []

    def clear_binding_inputs(self):
        self._inputs.clear()

    def get_outputs(self) -> list[OrtValue]:
        return self._outputs                      # ordered like the graph outputs that were bound


class InferenceSession:
    def __init__(self, path_or_bytes, sess_options: SessionOptions | None = None, providers=None, provider_options=None,
                 disabled_optimizers=None, device_id: int = 0):
        info, blob = load_model(str(path_or_bytes))
        self._info, self._kind = info, info["kind"]
        self._options = sess_options          # keeps add_initializer() values alive
        self._meta = ModelMetadata(info.get("metadata", {}))
        self._native = None
        if self._kind == "metadata":
            self._inputs = [NodeArg("metadata_marker", [1], np.int64)]
            self._outputs = [NodeArg("metadata_marker_out", [1], np.int64)]
        elif self._kind == "sensevoice":
            from .engine import SenseVoiceSession
            cfg = dict(info["config"])
            cfg["language_prompt_token_ids"] = tuple(cfg["language_prompt_token_ids"])
            self._cfg = SenseVoiceConfig(**cfg)
            self._native = SenseVoiceSession(self._cfg, blob, info["precision"], device_id)
            self._inputs = [NodeArg("audio", [1, 1, "audio_len"], np.float32), NodeArg("language_idx", [1], np.int32)]
            self._outputs = [NodeArg("token_ids", ["num_token"], np.int32), NodeArg("num_id", [1], np.int32)]
        elif self._kind == "paraformer":
            from .engine import ParaformerSession
            self._cfg = ParaformerConfig(**info["config"])
            self._native = ParaformerSession(self._cfg, blob, info["precision"], device_id)
            self._inputs = [NodeArg("audio", [1, 1, "audio_len"], np.float32)]
            self._outputs = [NodeArg("token_ids", [1, "num_token"], np.int32), NodeArg("num_id", [1], np.int32)]
        elif self._kind in ("whisper_graph", "qwen_graph", "paraformer_stream_graph"):    # state lives in the folder's shared native session
            path = str(path_or_bytes)
            if not os.path.isfile(path) and path.endswith(".onnx"):
                path = path[:-5] + ".asrmodel"
            if self._kind == "whisper_graph":
                from .ort_shim_whisper import WhisperGraph
                self._graph = WhisperGraph(path, info, device_id, load_model)
            elif self._kind == "qwen_graph":
                from .ort_shim_qwen import QwenGraph
                self._graph = QwenGraph(path, info, device_id, load_model)
            else:
                from .ort_shim_paraformer_streaming import ParaformerStreamGraph
                self._graph = ParaformerStreamGraph(path, info, device_id, load_model)
            self._native = self._graph.sh.native
            self._inputs = [NodeArg(n, sh, dt) for n, sh, dt in self._graph.inputs]
            self._outputs = [NodeArg(n, sh, dt) for n, sh, dt in self._graph.outputs]
        else:
            raise ValueError(f"unknown model kind {self._kind!r}")
        self._input_names = [a.name for a in self._inputs]
        self._output_names = [a.name for a in self._outputs]

    # -- introspection
    def get_inputs(self):
        return list(self._inputs)

    def get_outputs(self):
        return list(self._outputs)

    def get_providers(self):
        return ["MI355XExecutionProvider"]

    def get_modelmeta(self):
        return self._meta

    def io_binding(self):
        return IOBinding(self)

    # -- execution
    def _run_sensevoice(self, feeds: dict[str, OrtValue]) -> dict[str, np.ndarray]:
        for name in self._input_names:
            if name not in feeds:
                raise ValueError(f"input {name!r} is not bound")
        audio, lang = feeds["audio"], feeds["language_idx"]
        shape = tuple(audio._shape)
        if len(shape) != 3 or shape[1] != 1:
            raise ValueError(f"audio must have shape (batch, 1, audio_len), got {shape}")
        if np.dtype(audio._dtype) != np.float32:
            raise ValueError(f"audio must be tensor(float) carrying int16-range values, got {audio._dtype}")
        B, L = shape[0], shape[2]
        lang_np = np.asarray(lang.numpy(), dtype=np.int32).reshape(-1)
        if lang_np.size != B:
            raise ValueError(f"language_idx has {lang_np.size} entries for a batch of {B}")
        offsets = np.arange(B + 1, dtype=np.int64) * L
        if audio._host is not None:
            tok, num = self._native.run_packed(audio._host.reshape(-1), offsets, lang_np)
        else:
            tok, num = self._native.run_packed(None, offsets, lang_np, audio_device_ptr=audio._dptr.value)
        if B == 1:        # the reference graph's exact output shapes: token_ids (num_token,), num_id (1,)
            return {"token_ids": tok[0, :num[0]].copy(), "num_id": num.copy()}
        return {"token_ids": tok, "num_id": num}

    def _run_paraformer(self, feeds: dict[str, OrtValue]) -> dict[str, np.ndarray]:
        if "audio" not in feeds:
            raise ValueError("input 'audio' is not bound")
        audio = feeds["audio"]
        shape = tuple(audio._shape)
        if len(shape) != 3 or shape[1] != 1:
            raise ValueError(f"audio must have shape (batch, 1, audio_len), got {shape}")
        if np.dtype(audio._dtype) != np.float32:
            raise ValueError(f"audio must be tensor(float) carrying int16-range values, got {audio._dtype}")
        B, L = shape[0], shape[2]
        offsets = np.arange(B + 1, dtype=np.int64) * L
        if audio._host is not None:
            tok, num = self._native.run_packed(audio._host.reshape(-1), offsets)
        else:
            tok, num = self._native.run_packed(None, offsets, audio_device_ptr=audio._dptr.value)
        if B == 1:        # the reference graph's exact output shapes: token_ids (1, num_token), num_id (1,)
            return {"token_ids": tok[:, :num[0]].copy(), "num_id": num.copy()}
        return {"token_ids": tok, "num_id": num}

    def _execute(self, feeds: dict[str, OrtValue]) -> dict[str, np.ndarray]:
        if self._kind == "sensevoice":
            return self._run_sensevoice(feeds)
        if self._kind == "paraformer":
            return self._run_paraformer(feeds)
        if self._kind == "metadata":
            return {"metadata_marker_out": np.asarray(feeds["metadata_marker"].numpy())}
        if self._kind in ("whisper_graph", "qwen_graph", "paraformer_stream_graph"):
            return self._graph.execute(feeds, OrtValue)
        raise ValueError(self._kind)

    def run_with_iobinding(self, binding: IOBinding, run_options: RunOptions | None = None):
        results = self._execute(binding._inputs)
        outs = []
        for name in self._output_names:               # graph order
            if name not in binding._out_requests:
                continue
            target = binding._out_requests[name]
            if isinstance(results[name], OrtValue):                # state handles / values the engine keeps a reference to
                outs.append(results[name])
            elif target is None:
                outs.append(OrtValue(results[name], "cpu", 0))     # ownership passes to the returned value
            else:
                target.update_inplace(results[name])
                outs.append(target)
        binding._outputs = outs

    def run(self, output_names: Sequence[str] | None, input_feed: dict[str, np.ndarray], run_options=None):
        feeds = {k: OrtValue(np.asarray(v, dtype=self._dtype_of(k)), "cpu", 0) for k, v in input_feed.items()}
        results = self._execute(feeds)
        names = list(output_names) if output_names else self._output_names
        return [results[n].numpy() if isinstance(results[n], OrtValue) else results[n] for n in names]

    def _dtype_of(self, name):
        from .ort_io import numpy_dtype
        for a in self._inputs:
            if a.name == name:
                return numpy_dtype(a)
        raise ValueError(f"{name!r} is not an input of this graph")


def install_as_onnxruntime():
    """Register this module as `onnxruntime` (+ `onnxruntime.capi._pybind_state`) in sys.modules."""
    me = sys.modules[__name__]
    capi = types.ModuleType("onnxruntime.capi")
    state = types.ModuleType("onnxruntime.capi._pybind_state")
    state.OrtDevice = OrtDevice
    capi._pybind_state = state
    me.capi = capi
    sys.modules["onnxruntime"] = me
    sys.modules["onnxruntime.capi"] = capi
    sys.modules["onnxruntime.capi._pybind_state"] = state
    return me
