"""Streaming Paraformer's two graphs behind the onnxruntime API subset: `Paraformer_Streaming_Encoder.onnx` and
`Paraformer_Streaming_Decoder.onnx` with the I/O names the reference host wires together (encoder_feedback / decoder_feedback /
encoder_decoder_bridge, Paraformer/Streaming/Inference_Paraformer_Streaming_ONNX.py:296-338) and its per-window call sequence (:401-449):
encoder run; if `list_frame_len != 0` decoder run on the bridged outputs; outputs fed back as the next window's inputs.

The 100 encoder K/V histories, the carried LFR rows, the CIF state and the 48 decoder caches are state of ONE native
`ParaformerStreamSession` (stream 0) per model folder. The graph outputs that stand for them are placeholder OrtValues with a generation
stamp: feeding back anything but the latest ones raises; feeding fresh (unstamped) caches -- what the host allocates before the first
window (:366-392) -- starts a new utterance. The native step runs the decoder inside the same launch sequence whenever the CIF fired, so
the Decoder graph here only hands out what that step produced (`max_logit_ids`, `num_id`) after checking that it is being fed the
bridge values of the same window.
"""
from __future__ import annotations

import os

import numpy as np

from .config import ParaformerConfig

ENCODER_FILE, DECODER_FILE, WEIGHTS_FILE = "Paraformer_Streaming_Encoder", "Paraformer_Streaming_Decoder", "Paraformer_Streaming"
_SHARED: dict = {}


class _Shared:
    def __init__(self, cfg, native, chunk):
        self.cfg, self.native, self.chunk = cfg, native, chunk
        self.enc_gen = 0          # stamps the encoder's outputs (state + bridge) of the latest window
        self.dec_gen = 0          # stamps the decoder's state outputs
        self.tokens = np.zeros(0, np.int32)
        self.tokens_taken = True


def graph_io(cfg: ParaformerConfig, role: str, chunk: int, rows_per_chunk: int, carried: int, kv_dtype=np.float16):
    H, hd, d, feat = cfg.n_heads, cfg.d_head, cfg.d_model, cfg.n_mels * cfg.lfr_m
    Le, Ld, pad = cfg.n_enc0 + cfg.n_enc, cfg.n_dec, cfg.fsmn_kernel - 1
    bridge = [("encoder_out", [1, rows_per_chunk + carried, d], np.float32), ("list_frame", [1, "num_frame", d], np.float32), ("list_frame_len", [], np.int64)]
    if role == "encoder":
        ins = [(f"in_en_key_{i}", [H, hd, "history_len"], kv_dtype) for i in range(Le)] + [(f"in_en_value_{i}", [H, "history_len", hd], kv_dtype) for i in range(Le)]
        ins += [("in_previous_mel_features", [1, carried, feat], np.float32), ("in_cif_hidden", [1, 1, d], np.float32), ("in_cif_alphas", [1], np.float32),
                ("start_idx", [1], np.int64), ("audio", [1, 1, chunk], np.float32)]
        outs = [(f"out_en_key_{i}", [H, hd, "history_len_out"], kv_dtype) for i in range(Le)] + [(f"out_en_value_{i}", [H, "history_len_out", hd], kv_dtype) for i in range(Le)]
        outs += [("out_previous_mel_features", [1, carried, feat], np.float32), ("out_cif_hidden", [1, 1, d], np.float32), ("out_cif_alphas", [1], np.float32),
                 ("end_idx", [1], np.int64)] + bridge
        return ins, outs
    ins = [(f"in_de_fsmn_{i}", [1, d, pad], np.float32) for i in range(Ld)] + [(f"in_de_key_{i}", [H, hd, "history_len"], kv_dtype) for i in range(Ld)]
    ins += [(f"in_de_value_{i}", [H, "history_len", hd], kv_dtype) for i in range(Ld)] + bridge
    outs = [(f"out_de_fsmn_{i}", [1, d, pad], np.float32) for i in range(Ld)] + [(f"out_de_key_{i}", [H, hd, "history_len_out"], kv_dtype) for i in range(Ld)]
    outs += [(f"out_de_value_{i}", [H, "history_len_out", hd], kv_dtype) for i in range(Ld)]
    outs += [("max_logit_ids", [1, "num_frame"], np.int32), ("num_id", [1], np.int32)]
    return ins, outs


class ParaformerStreamGraph:
    def __init__(self, stub_path: str, info: dict, device_id: int, load_model):
        from .engine import ParaformerStreamSession
        conf = info["config"]
        self.role = conf["role"]
        wpath = os.path.join(os.path.dirname(os.path.abspath(stub_path)), conf["weights"])
        key = (wpath, device_id)
        if key not in _SHARED:
            winfo, blob = load_model(wpath)
            cfg = ParaformerConfig(**winfo["config"])
            chunk = int(winfo["metadata"].get("chunk", 8000))
            native = ParaformerStreamSession(cfg, blob, int(winfo.get("precision", 0)), device_id, chunk=chunk, max_streams=1)
            _SHARED[key] = _Shared(cfg, native, chunk)
        self.sh: _Shared = _SHARED[key]
        self.cfg = self.sh.cfg
        self.rows = self.sh.native.rows_per_chunk
        self.carried = self.rows // 2
        self.inputs, self.outputs = graph_io(self.cfg, self.role, self.sh.chunk, self.rows, self.carried)

    @staticmethod
    def _stamp(feeds, names):
        stamps = {getattr(feeds[n], "_asr_handle", None) for n in names}
        if len(stamps) != 1:
            raise ValueError("state inputs mix values of different runs (or fresh tensors with fed-back ones)")
        return stamps.pop()

    def _placeholders(self, OrtValue, results, specs, stamp, dyn=0):
        for name, shape, dt in specs:
            v = OrtValue(np.zeros([dyn if isinstance(s, str) else s for s in shape], dtype=dt), "cpu", 0)
            v._asr_handle = stamp
            results[name] = v

    def execute(self, feeds: dict, OrtValue) -> dict:
        sh = self.sh
        for name, _, _ in self.inputs:
            if name not in feeds:
                raise ValueError(f"input {name!r} is not bound")
        results: dict = {}
        if self.role == "encoder":
            state = [n for n, _, _ in self.inputs if n != "audio"]
            stamp = self._stamp(feeds, state)
            if stamp is None:                                      # fresh caches: a new utterance (:366-392)
                if any(int(np.prod(feeds[n]._shape)) != 0 for n in state if n.startswith(("in_en_key_", "in_en_value_"))):
                    raise ValueError("a new utterance starts from zero-length K/V histories")
                sh.native.reset(0)
                sh.dec_gen = 0
            elif stamp != ("enc", id(sh), sh.enc_gen):
                raise ValueError("stale encoder state: feed back the outputs of the latest window")
            audio = feeds["audio"]
            if tuple(audio._shape) != (1, 1, sh.chunk) or np.dtype(audio._dtype) != np.float32:
                raise ValueError(f"audio must be tensor(float) of shape (1, 1, {sh.chunk}) carrying int16-range values, got {audio._dtype} {tuple(audio._shape)}")
            if audio._host is not None:
                fired = sh.native.step(audio._host.reshape(1, sh.chunk), [0])[0]
            else:
                fired = sh.native.step(None, [0], audio_device_ptr=audio._dptr.value)[0]
            sh.enc_gen += 1
            sh.tokens, sh.tokens_taken = np.asarray(fired, dtype=np.int32), False
            start = int(np.asarray(feeds["start_idx"].numpy()).reshape(-1)[0])
            specs = [s for s in self.outputs if s[0] not in ("list_frame_len", "end_idx", "list_frame")]
            self._placeholders(OrtValue, results, specs, ("enc", id(sh), sh.enc_gen))
            self._placeholders(OrtValue, results, [s for s in self.outputs if s[0] == "list_frame"], ("enc", id(sh), sh.enc_gen), dyn=int(sh.tokens.size))
            end = OrtValue(np.asarray([start + self.rows], dtype=np.int64), "cpu", 0)
            end._asr_handle = ("enc", id(sh), sh.enc_gen)
            results["end_idx"] = end
            n = OrtValue(np.asarray(sh.tokens.size, dtype=np.int64), "cpu", 0)      # the one value the host reads: the CIF fire count (:415-416)
            n._asr_handle = ("enc", id(sh), sh.enc_gen)
            results["list_frame_len"] = n
            return results
        # decoder
        bridge = ("encoder_out", "list_frame", "list_frame_len")
        if self._stamp(feeds, bridge) != ("enc", id(sh), sh.enc_gen):
            raise ValueError("the decoder must be fed encoder_out / list_frame / list_frame_len of the latest encoder run")
        state = [n for n, _, _ in self.inputs if n not in bridge]
        stamp = self._stamp(feeds, state)
        if stamp is not None and stamp != ("dec", id(sh), sh.dec_gen):
            raise ValueError("stale decoder state: feed back the outputs of the latest decoder run")
        if stamp is None and sh.dec_gen != 0:
            raise ValueError("fresh decoder caches in the middle of an utterance")
        if sh.tokens.size == 0 or sh.tokens_taken:
            raise ValueError("the decoder runs once per window, and only when the CIF fired (list_frame_len != 0, :417-419)")
        sh.tokens_taken = True
        sh.dec_gen += 1
        self._placeholders(OrtValue, results, [s for s in self.outputs if s[0] not in ("max_logit_ids", "num_id")], ("dec", id(sh), sh.dec_gen))
        results["max_logit_ids"] = sh.tokens.reshape(1, -1).copy()
        results["num_id"] = np.asarray([sh.tokens.size], dtype=np.int32)
        return results


def export_paraformer_streaming_folder(folder: str, cfg: ParaformerConfig, ck: dict, metadata: dict, precision: int = 0, chunk: int = 8000,
                                       max_continue: int = 502) -> str:
    """`Paraformer_Streaming.asrmodel` (streaming arena) + stubs for the two graphs + `ASR_Metadata.asrmodel`."""
    import dataclasses
    from .arena import build_paraformer_arena
    from .ort_shim import save_model
    os.makedirs(folder, exist_ok=True)
    n_pos = max_continue - 1
    scfg = dataclasses.replace(cfg, max_audio_len=cfg.win_length + cfg.hop_length * (n_pos * cfg.lfr_n - 1))       # as ParaformerStreamSession sizes it
    save_model(os.path.join(folder, WEIGHTS_FILE + ".asrmodel"), "paraformer_streaming", cfg.to_dict(), build_paraformer_arena(scfg, ck, precision, streaming=True),
               {"chunk": str(int(chunk))}, precision)
    for stem, role in ((ENCODER_FILE, "encoder"), (DECODER_FILE, "decoder")):
        save_model(os.path.join(folder, stem + ".asrmodel"), "paraformer_stream_graph", {"role": role, "weights": WEIGHTS_FILE + ".asrmodel"}, None, {}, precision)
    save_model(os.path.join(folder, "ASR_Metadata.asrmodel"), "metadata", None, None, dict(metadata))
    return folder
