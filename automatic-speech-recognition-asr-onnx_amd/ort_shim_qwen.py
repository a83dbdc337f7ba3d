"""Qwen3-ASR's graphs behind the onnxruntime API subset: `Qwen3_ASR_Decoder_Embed.onnx`, the merged
`Qwen3_ASR_Prefill_*.onnx` / `Qwen3_ASR_Decode_*.onnx` (Greedy / Penalty_Greedy / Sampling; merge recipes Qwen_ASR/Shared_Merged.py) and
`ASR_Metadata.onnx`, with the I/O names and ORDER the reference host plans from (`_derive_merged_kv_layout` / `_plan_merged_io`,
Qwen_ASR/Inference_Qwen_ASR_ONNX.py:315-366: leading `past_*` state block; outputs = state, max id, (save ids), kv_seq_len), so that its
main loop (:424-745) runs unchanged on `ort_shim.InferenceSession`.

What stays in the native session instead of travelling through Python:
  * the 2 x n_layers KV tensors -- `present_*` outputs are zero-size placeholders with a generation stamp; a stale one raises;
  * the embeddings. The reference embeds the system prompt / language tail / every decoded token with the Embed graph and feeds the
    float tensors to the prefill / decode graphs. The native session gathers embedding rows on the device from token ids, so the Embed
    graph here returns an (n, H) float tensor that CARRIES the ids -- element 0 of every row is the id (exact in f32 below 2^24),
    element 1 a tag -- and the consumer graphs read them back. Host code treats the tensor as opaque data either way
    (`_persistent_embed`, :394-420; `hidden_states_buffer`, :545-551); anything that is not such a tensor is refused.
"""
from __future__ import annotations

import json
import os

import numpy as np

from .config import QwenAsrConfig

STRATEGIES = ("greedy", "penalty_greedy", "sampling")
GRAPH_FILES = {"prefill_greedy": "Qwen3_ASR_Prefill_Greedy", "prefill_penalty_greedy": "Qwen3_ASR_Prefill_Penalty_Greedy", "prefill_sampling": "Qwen3_ASR_PrefillSampling",
               "decode_greedy": "Qwen3_ASR_Decode_Greedy", "decode_penalty_greedy": "Qwen3_ASR_Decode_Penalty_Greedy", "decode_sampling": "Qwen3_ASR_DecodeSampling"}
EMBED_FILE, WEIGHTS_FILE, METADATA_FILE = "Qwen3_ASR_Decoder_Embed", "Qwen3_ASR", "ASR_Metadata"
MAX_OUT = {"greedy": "greedy_max_logits_idx", "penalty_greedy": "penalty_greedy_max_logits_idx", "sampling": "sampling_sampled_id"}
SAVE_OUT = {"greedy": None, "penalty_greedy": "penalty_greedy_save_id_out", "sampling": "sampling_save_id_out"}
SAMPLING_INPUTS = ("sampling_temperature", "sampling_top_k", "sampling_top_p", "sampling_repetition_penalty")
_TAG = np.float32(-7.25e18)         # element 1 of an id-carrying embedding row

_SHARED: dict = {}


class _Shared:
    def __init__(self, cfg, native, meta):
        from .qwen_asr import prompt_ids
        self.cfg, self.native = cfg, native
        special = json.loads(meta["special_token_ids"])
        self.head_ids, self.suffix_ids, self.tail_ids = prompt_ids(special)
        self.kv_gen = 0
        self.seq_len = 0
        self.saved: list[int] = []
        self.last_next = None          # id the native session picked last (kept on the device)
        self.head = None


def ids_as_embedding(ids, hidden: int) -> np.ndarray:
    ids = np.asarray(ids, dtype=np.int64).reshape(-1)
    out = np.zeros((1, ids.size, hidden), dtype=np.float32)
    out[0, :, 0] = ids.astype(np.float32)
    out[0, :, 1] = _TAG
    return out


def embedding_as_ids(arr, what: str) -> list:
    a = np.asarray(arr, dtype=np.float32)
    if a.ndim != 3 or a.shape[0] != 1:
        raise ValueError(f"{what}: expected an embedding tensor of shape (1, n, hidden), got {a.shape}")
    if a.shape[1] == 0:
        return []
    if not np.all(a[0, :, 1] == _TAG):
        raise ValueError(f"{what}: not an output of this engine's Embed graph (the native session embeds token ids on the device; "
                         "feed what Qwen3_ASR_Decoder_Embed returned)")
    return [int(v) for v in a[0, :, 0]]


def graph_io(cfg: QwenAsrConfig, role: str, strategy: str, kv_dtype=np.float16):
    if role == "embed":
        return [("input_ids", [1, "ids_len"], np.int32)], [("hidden_states", [1, "ids_len", cfg.d_model], np.float32)]
    L, KV, hd = cfg.n_layers, cfg.n_kv_heads, cfg.d_head
    ins = [(f"past_key_{i}", ["batch", KV, 1, hd, "history_len"], kv_dtype) for i in range(L)]
    ins += [(f"past_value_{i}", ["batch", KV, 1, "history_len", hd], kv_dtype) for i in range(L)]
    if role == "prefill":
        ins += [("audio", [1, 1, "audio_len"], np.float32), ("query_embed", [1, "query_len", cfg.d_model], np.float32),
                ("language_tail_embed", [1, "language_tail_len", cfg.d_model], np.float32), ("prefill_history_len", [1], np.int64)]
    else:
        ins += [("hidden_states", [1, 1, cfg.d_model], np.float32), ("decode_kv_seq_len", [1], np.int64)]
    if strategy == "penalty_greedy":
        if role == "decode":
            ins += [("penalty_save_id_in", ["batch", "history_len_ids"], np.int32), ("penalty_penalty_value", [1], np.float32),
                    ("penalty_penalty_range", [1], np.int64)]
        ins.append(("penalty_greedy_save_id_in", ["batch", "history_len_ids"], np.int32))
    if strategy == "sampling":
        ins += [("sampling_temperature", [1], np.float32), ("sampling_top_k", [1], np.int64), ("sampling_top_p", [1], np.float32),
                ("sampling_repetition_penalty", [1], np.float32), ("sampling_previous_ids", ["batch", "history_len_ids"], np.int32)]
    outs = [(f"present_key_{i}", ["batch", KV, 1, hd, "kv_seq_len"], kv_dtype) for i in range(L)]
    outs += [(f"present_value_{i}", ["batch", KV, 1, "kv_seq_len", hd], kv_dtype) for i in range(L)]
    outs.append((MAX_OUT[strategy], ["batch", 1], np.int32))
    if SAVE_OUT[strategy]:
        outs.append((SAVE_OUT[strategy], ["batch", "history_len_out"], np.int32))
    outs.append(("decode_kv_seq_len_next" if role == "decode" else "prefill_kv_seq_len", [1], np.int64))
    return ins, outs


class QwenGraph:
    def __init__(self, stub_path: str, info: dict, device_id: int, load_model):
        from .engine import QwenAsrSession
        conf = info["config"]
        self.role, self.strategy = conf["role"], conf.get("strategy", "greedy")
        wpath = os.path.join(os.path.dirname(os.path.abspath(stub_path)), conf["weights"])
        key = (wpath, device_id)
        if key not in _SHARED:
            winfo, blob = load_model(wpath)
            cfg = QwenAsrConfig(**winfo["config"])
            _SHARED[key] = _Shared(cfg, QwenAsrSession(cfg, blob, int(winfo.get("precision", 0)), device_id), winfo["metadata"])
        self.sh: _Shared = _SHARED[key]
        self.cfg = self.sh.cfg
        self.kv_dtype = np.float16
        self.inputs, self.outputs = graph_io(self.cfg, self.role, self.strategy, self.kv_dtype)

    def _configure_head(self, feeds, is_decode):
        sh, n = self.sh, self.sh.native
        value, rng, samp = 1.0, 10, None
        if self.strategy == "penalty_greedy" and is_decode:
            value = float(np.asarray(feeds["penalty_penalty_value"].numpy()).reshape(-1)[0])
            rng = int(np.asarray(feeds["penalty_penalty_range"].numpy()).reshape(-1)[0])
        if self.strategy == "sampling":
            samp = tuple(float(np.asarray(feeds[k].numpy()).reshape(-1)[0]) for k in SAMPLING_INPUTS)
        head = (self.strategy, value, rng, samp)
        if head == sh.head:
            return
        n.set_sampling(False)
        n.track_history(self.strategy == "penalty_greedy")
        n.set_penalty(value, min(max(rng, 1), 64))
        if samp is not None:
            n.set_sampling(True, samp[0], int(samp[1]), samp[2], samp[3], seed=0)
        sh.head = head

    def _check_state(self, feeds, required):
        for i in range(self.cfg.n_layers):
            for part in ("key", "value"):
                name = f"past_{part}_{i}"
                if name not in feeds:
                    raise ValueError(f"input {name!r} is not bound")
                h = getattr(feeds[name], "_asr_handle", None)
                if h is None:
                    if required:
                        raise ValueError(f"{name!r}: expected the value a previous run returned (the KV cache lives in the native session)")
                    if int(np.prod(feeds[name]._shape)) != 0:
                        raise ValueError(f"{name!r}: a prefill starts from an empty cache (the reference binds zero-length tensors, :600-610)")
                elif h != (id(self.sh), self.sh.kv_gen):
                    raise ValueError(f"{name!r} is a stale handle: it belongs to an earlier run than the cache it is bound to")

    def _outputs(self, OrtValue, results, nxt, kv_name):
        sh, cfg = self.sh, self.cfg
        sh.kv_gen += 1
        for i in range(cfg.n_layers):
            for part, shape in (("key", (1, cfg.n_kv_heads, 1, cfg.d_head, 0)), ("value", (1, cfg.n_kv_heads, 1, 0, cfg.d_head))):
                v = OrtValue(np.zeros(shape, dtype=self.kv_dtype), "cpu", 0)
                v._asr_handle = (id(sh), sh.kv_gen)
                results[f"present_{part}_{i}"] = v
        sh.last_next = int(nxt)
        sh.saved.append(sh.last_next)
        results[MAX_OUT[self.strategy]] = np.asarray([[sh.last_next]], dtype=np.int32)
        if SAVE_OUT[self.strategy]:
            results[SAVE_OUT[self.strategy]] = np.asarray([sh.saved], dtype=np.int32)
        results[kv_name] = np.asarray([sh.seq_len], dtype=np.int64)

    def execute(self, feeds: dict, OrtValue) -> dict:
        sh, cfg = self.sh, self.cfg
        if self.role == "embed":
            ids = np.asarray(feeds["input_ids"].numpy(), dtype=np.int64).reshape(-1)
            if ids.size and (ids.min() < 0 or ids.max() >= cfg.vocab):
                raise ValueError(f"input_ids out of range [0, {cfg.vocab})")
            return {"hidden_states": ids_as_embedding(ids, cfg.d_model)}
        for name, _, _ in self.inputs:
            if name not in feeds:
                raise ValueError(f"input {name!r} is not bound")
        results: dict = {}
        if self.role == "prefill":
            self._check_state(feeds, required=False)
            if int(np.asarray(feeds["prefill_history_len"].numpy()).reshape(-1)[0]) != 0:
                raise ValueError("prefill_history_len must be 0 (the reference's one prefill per clip, :586-588)")
            audio = feeds["audio"]
            shape = tuple(audio._shape)
            if len(shape) != 3 or shape[0] != 1 or shape[1] != 1 or np.dtype(audio._dtype) != np.float32:
                raise ValueError(f"audio must be tensor(float) of shape (1, 1, audio_len) in [-1, 1], got {audio._dtype} {shape}")
            query = embedding_as_ids(feeds["query_embed"].numpy(), "query_embed")
            tail = embedding_as_ids(feeds["language_tail_embed"].numpy(), "language_tail_embed")
            self._configure_head(feeds, False)
            pre, post = [sh.head_ids + query + sh.suffix_ids], [sh.tail_ids + tail]
            offsets = np.array([0, shape[2]], dtype=np.int64)
            if audio._host is not None:
                nxt, _, ids_len = sh.native.prefill_packed(audio._host.reshape(-1), offsets, pre, post, want_logits=False)
            else:
                nxt, _, ids_len = sh.native.prefill_packed(None, offsets, pre, post, want_logits=False, audio_device_ptr=audio._dptr.value)
            sh.seq_len, sh.saved = int(ids_len[0]), []
            self._outputs(OrtValue, results, nxt[0], "prefill_kv_seq_len")
            return results
        # decode
        self._check_state(feeds, required=True)
        seq = int(np.asarray(feeds["decode_kv_seq_len"].numpy()).reshape(-1)[0])
        if seq != sh.seq_len:
            raise ValueError(f"decode_kv_seq_len = {seq} but the cache holds {sh.seq_len} positions")
        ids = embedding_as_ids(feeds["hidden_states"].numpy(), "hidden_states")
        if len(ids) != 1:
            raise ValueError("hidden_states: one position per decode step")
        self._configure_head(feeds, True)
        nxt, _ = sh.native.decode(None if ids[0] == sh.last_next else np.asarray(ids, dtype=np.int32))
        sh.seq_len += 1
        self._outputs(OrtValue, results, nxt[0], "decode_kv_seq_len_next")
        return results


def export_qwen_asr_folder(folder: str, cfg: QwenAsrConfig, ck: dict, metadata: dict, precision: int = 0) -> str:
    """Model folder with the reference's file names (Qwen_ASR/Shared_Merged.DEFAULT_MODEL_FILE_NAMES): `Qwen3_ASR.asrmodel` (arena + the
    exporter's metadata map), one stub per merged graph, the Embed stub and `ASR_Metadata.asrmodel`."""
    from .ort_shim import save_model
    from .qwen_asr import export_qwen_asr
    os.makedirs(folder, exist_ok=True)
    export_qwen_asr(cfg, ck, os.path.join(folder, WEIGHTS_FILE + ".asrmodel"), metadata, precision)
    for key, stem in GRAPH_FILES.items():
        role, strategy = key.split("_", 1)
        save_model(os.path.join(folder, stem + ".asrmodel"), "qwen_graph", {"role": role, "strategy": strategy, "weights": WEIGHTS_FILE + ".asrmodel"}, None, {}, precision)
    save_model(os.path.join(folder, EMBED_FILE + ".asrmodel"), "qwen_graph", {"role": "embed", "weights": WEIGHTS_FILE + ".asrmodel"}, None, {}, precision)
    save_model(os.path.join(folder, METADATA_FILE + ".asrmodel"), "metadata", None, None, dict(metadata))
    return folder
