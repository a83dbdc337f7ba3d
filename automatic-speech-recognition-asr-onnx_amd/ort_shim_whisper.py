"""Whisper's merged graphs behind the onnxruntime API subset: `Whisper_ProbePrefill*.onnx`, `Whisper_Prefill*.onnx`,
`Whisper_Decode*.onnx` (Greedy / PenaltyGreedy / Sampling, merge recipes Whisper/Shared_Merged.py:755-905) and
`Whisper_No_Speech_Detection.onnx`, with the I/O names, order and shapes the reference host plans its bindings from
(`_plan_merged_io`, Whisper/Inference_Whisper_ONNX.py:323-392) -- so its `_probe_prefill / _prefill / _decode_tokens` loops (:437-663)
run unchanged on `ort_shim.InferenceSession`.

The reference shuttles 2 x n_layers self-KV and 2 x n_layers cross-KV tensors through Python on every call. Here they are state of ONE
native `WhisperSession` shared by the graph sessions of a model folder; the KV outputs are zero-size placeholder OrtValues carrying a
generation stamp (`_asr_handle`). Binding them back as inputs is what the host's ping-pong loop does; a stale stamp (outputs of an
older run fed again) raises instead of silently decoding from the wrong state.

  role           native calls                                   inputs beyond the KV state                      outputs beyond the KV state
  probe_prefill  encode(audio) + prefill(ids, logits)           audio, embed_input_ids, prefill_ids_len,        encoder_en_{key,value}_layer_i, max id,
                                                                prefill_history_len                             (save ids), logits, prefill_kv_seq_len
  prefill        prefill(ids, logits)                           en_{key,value}_layer_i, same scalars            max id, (save ids), logits, prefill_kv_seq_len
  decode         decode(ids | device-resident)                  en_*, embed_input_ids, decode_kv_seq_len        max id, (save ids), decode_kv_seq_len_next
                                                                (+ penalty_* / sampling_* controls)
Strategy heads: greedy = BEGIN_SUPPRESS + ARGMAX; penalty_greedy = GREEDY_SEARCH every step (history always appended,
`track_history`) + APPLY_PENALTY with the value the host binds for THAT step (1.0 until PENALTY_RANGE ids exist, :630-632);
sampling = TOPK_TOPP_SAMPLING with the bound scalars (the in-graph RandomUniformLike is the device generator here).
"""
from __future__ import annotations

import json
import os

import numpy as np

from .config import WhisperConfig

STRATEGIES = ("greedy", "penalty_greedy", "sampling")
_SUFFIX = {"greedy": "Greedy", "penalty_greedy": "PenaltyGreedy", "sampling": "Sampling"}
GRAPH_FILES = {f"{role}_{st}": f"Whisper_{stem}{_SUFFIX[st]}" for role, stem in (("probe_prefill", "ProbePrefill"), ("prefill", "Prefill"),
                                                                                  ("decode", "Decode")) for st in STRATEGIES}
NO_SPEECH_FILE, WEIGHTS_FILE, METADATA_FILE = "Whisper_No_Speech_Detection", "Whisper", "ASR_Metadata"
MAX_OUT = {"greedy": "argmax_max_logits_idx", "penalty_greedy": "greedy_max_logits_idx", "sampling": "sampling_sampled_id"}
SAVE_OUT = {"greedy": None, "penalty_greedy": "greedy_save_id_out", "sampling": "sampling_save_id_out"}
SAMPLING_INPUTS = ("sampling_temperature", "sampling_top_k", "sampling_top_p", "sampling_repetition_penalty")

_SHARED: dict = {}          # (weights bundle path, device) -> _Shared: the one native session of a model folder


class _Shared:
    def __init__(self, cfg: WhisperConfig, native, meta: dict):
        self.cfg, self.native = cfg, native
        self.suppress = json.loads(meta.get("suppress_tokens", "[]"))
        self.enc_gen = 0            # bumped by every encode: stamps the cross-KV handles
        self.kv_gen = 0             # bumped by every prefill / decode: stamps the self-KV handles
        self.hist = 0               # positions in the self-KV cache
        self.batch = 0
        self.saved: list[list[int]] = []
        self.last_next = None       # the OrtValue handed out as max id by the previous run (fed back => ids stay on the device)
        self.head = None            # (strategy, penalty value, penalty range, sampling scalars) configured on the native session


def graph_io(cfg: WhisperConfig, role: str, strategy: str, kv_dtype=np.float16):
    """-> (inputs, outputs) as (name, shape, dtype) lists, in the order the merged graphs declare them: self-KV state first."""
    L, H = cfg.n_dec_layers, cfg.n_heads
    if role == "no_speech":
        return [("logits", ["batch", cfg.vocab], np.float32)], [("no_speech_prob", ["batch"], np.float32)]
    hist = "history_len"
    ins = [(f"in_de_key_layer_{i}", ["batch", H, cfg.d_head, hist], kv_dtype) for i in range(L)]
    ins += [(f"in_de_value_layer_{i}", ["batch", H, hist, cfg.d_head], kv_dtype) for i in range(L)]
    if role == "probe_prefill":
        ins.append(("audio", [1, 1, "audio_len"], np.float32))
    else:
        ins += [(f"en_key_layer_{i}", [H, cfg.d_head, "signal_len"], kv_dtype) for i in range(L)]
        ins += [(f"en_value_layer_{i}", [H, "signal_len", cfg.d_head], kv_dtype) for i in range(L)]
    ins.append(("embed_input_ids", ["batch", "ids_len" if role != "decode" else 1], np.int32))
    if role == "decode":
        ins.append(("decode_kv_seq_len", [1], np.int64))
    else:
        ins += [("prefill_ids_len", [1], np.int64), ("prefill_history_len", [1], np.int64)]
    if strategy == "penalty_greedy":
        if role == "decode":
            ins += [("penalty_save_id_in", ["batch", "save_len"], np.int32), ("penalty_penalty_value", [1], np.float32),
                    ("penalty_penalty_range", [1], np.int64)]
        ins.append(("greedy_save_id_in", ["batch", "save_len"], np.int32))
    if strategy == "sampling":
        ins += [("sampling_temperature", [1], np.float32), ("sampling_top_k", [1], np.int64), ("sampling_top_p", [1], np.float32),
                ("sampling_repetition_penalty", [1], np.float32), ("sampling_previous_ids", ["batch", "save_len"], np.int32)]
    new = "kv_seq_len"
    outs = [(f"out_de_key_layer_{i}", ["batch", H, cfg.d_head, new], kv_dtype) for i in range(L)]
    outs += [(f"out_de_value_layer_{i}", ["batch", H, new, cfg.d_head], kv_dtype) for i in range(L)]
    if role == "probe_prefill":
        outs += [(f"encoder_en_key_layer_{i}", [H, cfg.d_head, "signal_len"], kv_dtype) for i in range(L)]
        outs += [(f"encoder_en_value_layer_{i}", [H, "signal_len", cfg.d_head], kv_dtype) for i in range(L)]
    outs.append((MAX_OUT[strategy], ["batch", 1], np.int32))
    if SAVE_OUT[strategy]:
        outs.append((SAVE_OUT[strategy], ["batch", "save_len_next"], np.int32))
    if role != "decode":
        outs.append(("logits", ["batch", cfg.vocab], np.float32))           # raw logits incl. the -128 suppress penalty (probe: language / no-speech)
    outs.append(("decode_kv_seq_len_next" if role == "decode" else "prefill_kv_seq_len", [1], np.int64))
    return ins, outs


class WhisperGraph:
    """What ort_shim.InferenceSession delegates to for a Whisper graph bundle."""

    def __init__(self, stub_path: str, info: dict, device_id: int, load_model):
        from .engine import WhisperSession
        conf = info["config"]
        self.role, self.strategy = conf["role"], conf.get("strategy", "greedy")
        wpath = os.path.join(os.path.dirname(os.path.abspath(stub_path)), conf["weights"])
        key = (wpath, device_id)
        if key not in _SHARED:
            winfo, blob = load_model(wpath)
            cfg = WhisperConfig(**winfo["config"])
            native = WhisperSession(cfg, blob, int(winfo.get("precision", 0)), device_id, gelu_tanh=bool(winfo["metadata"].get("gelu_tanh", "1") == "1"))
            _SHARED[key] = _Shared(cfg, native, winfo["metadata"])
        self.sh: _Shared = _SHARED[key]
        self.cfg = self.sh.cfg
        self.kv_dtype = np.float16
        self.inputs, self.outputs = graph_io(self.cfg, self.role, self.strategy, self.kv_dtype)

    # ------------------------------------------------------------------ helpers
    def _placeholder(self, OrtValue, shape, kind, gen):
        v = OrtValue(np.zeros(shape, dtype=self.kv_dtype), "cpu", 0)
        v._asr_handle = (kind, id(self.sh), gen)
        return v

    def _check_handles(self, feeds, prefix, kind, gen, required):
        for i in range(self.cfg.n_dec_layers):
            for part in ("key", "value"):
                name = f"{prefix}{part}_layer_{i}"
                if name not in feeds:
                    raise ValueError(f"input {name!r} is not bound")
                h = getattr(feeds[name], "_asr_handle", None)
                if h is None:
                    if required:
                        raise ValueError(f"{name!r}: expected the value a previous run returned (the KV state lives in the native session)")
                    if int(np.prod(feeds[name]._shape)) != 0:
                        raise ValueError(f"{name!r}: a prefill starts from an empty history (the reference binds zero-length tensors, :445-456)")
                elif h != (kind, id(self.sh), gen):
                    raise ValueError(f"{name!r} is a stale handle: it belongs to an earlier run than the state it is bound to")

    def _configure_head(self, feeds, is_decode):
        sh, n = self.sh, self.sh.native
        value, rng, samp = 1.0, 20, None
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

    def _state_outputs(self, OrtValue, results, B, new_len):
        H, hd = self.cfg.n_heads, self.cfg.d_head
        for i in range(self.cfg.n_dec_layers):
            results[f"out_de_key_layer_{i}"] = self._placeholder(OrtValue, (B, H, hd, 0), "self", self.sh.kv_gen)
            results[f"out_de_value_layer_{i}"] = self._placeholder(OrtValue, (B, H, 0, hd), "self", self.sh.kv_gen)

    def _head_outputs(self, OrtValue, results, nxt):
        sh = self.sh
        ids = np.asarray(nxt, dtype=np.int32).reshape(-1, 1)
        results[MAX_OUT[self.strategy]] = sh.last_next = OrtValue(ids, "cpu", 0)
        for b in range(sh.batch):
            sh.saved[b].append(int(ids[b, 0]))
        if SAVE_OUT[self.strategy]:
            results[SAVE_OUT[self.strategy]] = OrtValue(np.asarray(sh.saved, dtype=np.int32), "cpu", 0)

    # ------------------------------------------------------------------ execution
    def execute(self, feeds: dict, OrtValue) -> dict:
        if self.role == "no_speech":
            # the host feeds the probe's `logits` output straight back (:799-805): the device head runs on the resident copy of those logits
            if "logits" not in feeds:
                raise ValueError("input 'logits' is not bound")
            if tuple(feeds["logits"]._shape) != (self.sh.batch, self.cfg.vocab):
                raise ValueError(f"logits must be the (batch, vocab) output of the probe-prefill graph, got {tuple(feeds['logits']._shape)}")
            return {"no_speech_prob": self.sh.native.no_speech_prob(self.cfg.no_speech_id)}
        for name, _, _ in self.inputs:
            if name not in feeds:
                raise ValueError(f"input {name!r} is not bound")
        sh, cfg, results = self.sh, self.cfg, {}
        ids_v = feeds["embed_input_ids"]
        if self.role == "decode":
            self._check_handles(feeds, "in_de_", "self", sh.kv_gen, required=True)
            self._check_handles(feeds, "en_", "cross", sh.enc_gen, required=True)
            seq = int(np.asarray(feeds["decode_kv_seq_len"].numpy()).reshape(-1)[0])
            if seq != sh.hist:
                raise ValueError(f"decode_kv_seq_len = {seq} but the cache holds {sh.hist} positions")
            self._configure_head(feeds, True)
            if ids_v is sh.last_next:
                nxt, _ = sh.native.decode(None)                        # the previous pick never left the device
            else:
                nxt, _ = sh.native.decode(np.asarray(ids_v.numpy(), dtype=np.int32).reshape(-1))
            sh.hist += 1
            sh.kv_gen += 1
            self._state_outputs(OrtValue, results, sh.batch, sh.hist)
            self._head_outputs(OrtValue, results, nxt)
            results["decode_kv_seq_len_next"] = OrtValue(np.asarray([sh.hist], dtype=np.int64), "cpu", 0)
            return results
        # probe_prefill / prefill
        ids = np.asarray(ids_v.numpy(), dtype=np.int32)
        if ids.ndim != 2:
            raise ValueError(f"embed_input_ids must have shape (batch, ids_len), got {ids.shape}")
        if int(np.asarray(feeds["prefill_ids_len"].numpy()).reshape(-1)[0]) != ids.shape[1]:
            raise ValueError("prefill_ids_len does not match embed_input_ids")
        if int(np.asarray(feeds["prefill_history_len"].numpy()).reshape(-1)[0]) != 0:
            raise ValueError("prefill_history_len must be 0: the reference always prefills with an empty self-KV (:476-480)")
        self._check_handles(feeds, "in_de_", "self", -1, required=False)
        if self.role == "probe_prefill":
            audio = feeds["audio"]
            shape = tuple(audio._shape)
            if len(shape) != 3 or shape[1] != 1 or np.dtype(audio._dtype) != np.float32:
                raise ValueError(f"audio must be tensor(float) of shape (batch, 1, audio_len) in [-1, 1], got {audio._dtype} {shape}")
            offsets = np.arange(shape[0] + 1, dtype=np.int64) * shape[2]
            if audio._host is not None:
                sh.native.encode_packed(audio._host.reshape(-1), offsets)
            else:
                sh.native.encode_packed(None, offsets, audio_device_ptr=audio._dptr.value)
            sh.enc_gen += 1
            sh.batch = shape[0]
            T = cfg.n_enc_pos(shape[2])
            for i in range(cfg.n_dec_layers):
                results[f"encoder_en_key_layer_{i}"] = self._placeholder(OrtValue, (cfg.n_heads, cfg.d_head, 0), "cross", sh.enc_gen)
                results[f"encoder_en_value_layer_{i}"] = self._placeholder(OrtValue, (cfg.n_heads, 0, cfg.d_head), "cross", sh.enc_gen)
            del T
        else:
            self._check_handles(feeds, "en_", "cross", sh.enc_gen, required=True)
        if ids.shape[0] != sh.batch:
            raise ValueError(f"embed_input_ids has batch {ids.shape[0]} but {sh.batch} clips are encoded")
        self._configure_head(feeds, False)
        nxt, logits = sh.native.prefill(ids, want_logits=True)
        sh.hist = ids.shape[1]
        sh.kv_gen += 1
        sh.saved = [[] for _ in range(sh.batch)]
        self._state_outputs(OrtValue, results, sh.batch, sh.hist)
        self._head_outputs(OrtValue, results, nxt)
        results["logits"] = logits
        results["prefill_kv_seq_len"] = OrtValue(np.asarray([sh.hist], dtype=np.int64), "cpu", 0)
        return results


def export_whisper(folder: str, cfg: WhisperConfig, ck: dict, precision: int = 0, suppress_tokens=None, begin_suppress_tokens=(),
                   supported_languages: dict | None = None, gelu_tanh: bool = True) -> str:
    """Model folder with the reference's file names (Shared_Merged.DEFAULT_MODEL_FILE_NAMES): `Whisper.asrmodel` (arena), one stub per
    merged graph and for the no-speech graph, and `ASR_Metadata.asrmodel` with the exporter's metadata map (Export_Whisper.py:1064-1074)."""
    from .arena import build_whisper_arena
    from .ort_shim import save_model
    os.makedirs(folder, exist_ok=True)
    meta_w = {"suppress_tokens": json.dumps([int(t) for t in (suppress_tokens or [])]), "gelu_tanh": "1" if gelu_tanh else "0"}
    save_model(os.path.join(folder, WEIGHTS_FILE + ".asrmodel"), "whisper", cfg.to_dict(),
               build_whisper_arena(cfg, ck, precision, suppress_tokens, begin_suppress_tokens), meta_w, precision)
    for key, stem in GRAPH_FILES.items():
        role, strategy = key.rsplit("_", 1) if not key.endswith("penalty_greedy") else (key[:-len("_penalty_greedy")], "penalty_greedy")
        save_model(os.path.join(folder, stem + ".asrmodel"), "whisper_graph", {"role": role, "strategy": strategy, "weights": WEIGHTS_FILE + ".asrmodel"},
                   None, {}, precision)
    save_model(os.path.join(folder, NO_SPEECH_FILE + ".asrmodel"), "whisper_graph", {"role": "no_speech", "weights": WEIGHTS_FILE + ".asrmodel"}, None, {}, precision)
    special = {"bos": cfg.eot_id, "decoder_start": cfg.sot_id, "eos": cfg.eot_id, "pad": cfg.eot_id, "unknown": cfg.eot_id, "stop": [cfg.eot_id],
               "no_speech": cfg.no_speech_id, "no_timestamps": cfg.no_timestamps_id, "tasks": {"transcribe": cfg.transcribe_id, "translate": cfg.translate_id}}
    langs = supported_languages if supported_languages is not None else {
        f"l{i:02d}": {"name": f"Language {i}", "aliases": [], "token_id": cfg.first_language_id + i, "prompt_token_ids": []} for i in range(cfg.n_languages)}
    meta = {"audio_pcm_scale": "32768", "max_seq_len": str(cfg.max_target_positions), "sample_rate": str(cfg.sample_rate),
            "special_token_ids": json.dumps(special), "supported_languages": json.dumps(langs)}
    save_model(os.path.join(folder, METADATA_FILE + ".asrmodel"), "metadata", None, None, meta)
    return folder
