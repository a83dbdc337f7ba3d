"""Python handles over the C ABI: sessions, taps, profiling, operator hooks."""
from __future__ import annotations

import ctypes as C
import warnings
from typing import Sequence

import numpy as np

from . import _lib
from .arena import PRECISION_BF16, PRECISION_F32, PRECISION_FP8MM, PRECISION_FP8W, PRECISION_MXFP4W, build_sensevoice_arena
from .config import SenseVoiceConfig

MEM_HOST, MEM_DEVICE = 0, 1


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def sensevoice_config_c(cfg: SenseVoiceConfig) -> _lib.SenseVoiceConfigC:
    c = _lib.SenseVoiceConfigC()
    c.sample_rate, c.n_mels, c.nfft, c.win_length, c.hop_length = cfg.sample_rate, cfg.n_mels, cfg.nfft, cfg.win_length, cfg.hop_length
    c.lfr_m, c.lfr_n = cfg.lfr_m, cfg.lfr_n
    c.d_model, c.n_heads, c.d_head, c.d_ffn = cfg.d_model, cfg.n_heads, cfg.d_head, cfg.d_ffn
    c.n_blocks, c.n_main = cfg.n_blocks, cfg.n_enc0 + cfg.n_enc
    c.fsmn_kernel, c.vocab, c.blank_id = cfg.fsmn_kernel, cfg.vocab, cfg.blank_id
    c.n_prompt, c.n_languages, c.max_audio_len = cfg.n_prompt, len(cfg.language_prompt_token_ids), cfg.max_audio_len
    return c


class _Session:
    """Common session utilities (stream, profiling, taps)."""

    def __init__(self):
        self._h = C.c_void_p(None)
        self._keep = None

    def close(self):
        if self._h:
            _lib.check(_lib.load().asr_session_destroy(self._h))
            self._h = C.c_void_p(None)
        self._keep = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream: int):
        _lib.check(_lib.load().asr_session_set_stream(self._h, C.c_void_p(hip_stream)))

    def profile(self, enable: bool):
        _lib.check(_lib.load().asr_session_profile_enable(self._h, int(enable)))

    def profile_reset(self):
        _lib.check(_lib.load().asr_session_profile_reset(self._h))

    def profile_read(self) -> dict:
        cap = 64
        names = C.create_string_buffer(32 * cap)
        ms = (C.c_double * cap)()
        cnt = (C.c_int64 * cap)()
        n = C.c_int(0)
        _lib.check(_lib.load().asr_session_profile_read(self._h, cap, names, ms, cnt, C.byref(n)))
        out = {}
        for i in range(n.value):
            nm = names.raw[i * 32:(i + 1) * 32].split(b"\0", 1)[0].decode()
            out[nm] = {"total_ms": ms[i], "launches": int(cnt[i])}
        return out

    def sanm_stats(self) -> dict:
        """SenseVoice / Paraformer sessions: counters of the cluster kernels (asr_sanm_stats)."""
        out = np.zeros(8, dtype=np.int32)
        _lib.check(_lib.load().asr_sanm_stats(self._h, _ip(out)))
        return {"giveups": int(out[0]), "cooldown": int(out[1]), "foreign_diverted": int(out[2]), "block_kernel": bool(out[3])}

    def taps(self, enable: bool):
        _lib.check(_lib.load().asr_session_taps_enable(self._h, int(enable)))

    def tap(self, name: str, dtype=np.float32) -> np.ndarray:
        rows, cols = C.c_int64(0), C.c_int64(0)
        _lib.check(_lib.load().asr_session_tap_shape(self._h, name.encode(), C.byref(rows), C.byref(cols)))
        out = np.empty((rows.value, cols.value), dtype=dtype)
        _lib.check(_lib.load().asr_session_tap_read(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out


class SenseVoiceSession(_Session):
    """HIP replacement of `SenseVoiceSmall.onnx` (SENSE_VOICE.forward, Export_SenseVoice.py:271-296)."""

    def __init__(self, cfg: SenseVoiceConfig, arena, precision: int = PRECISION_BF16, device_id: int = 0,
                 arena_device_ptr: int | None = None, arena_bytes: int | None = None):
        super().__init__()
        self.cfg, self.precision, self.device_id = cfg, precision, device_id
        self._cfg_c = sensevoice_config_c(cfg)
        lib = _lib.load()
        if arena_device_ptr is not None:
            self._keep = arena          # whatever owns the device memory (e.g. a torch tensor)
            _lib.check(lib.asr_sensevoice_create(C.byref(self._cfg_c), C.c_void_p(arena_device_ptr), arena_bytes, MEM_DEVICE,
                                                 device_id, precision, C.byref(self._h)))
        else:
            blob = np.ascontiguousarray(arena, dtype=np.uint8)
            _lib.check(lib.asr_sensevoice_create(C.byref(self._cfg_c), blob.ctypes.data_as(C.c_void_p), blob.nbytes, MEM_HOST,
                                                 device_id, precision, C.byref(self._h)))

    @classmethod
    def from_checkpoint(cls, cfg, ck, precision=PRECISION_BF16, device_id=0):
        return cls(cfg, build_sensevoice_arena(cfg, ck, precision), precision, device_id)

    def seq_len(self, n_samples: int) -> int:
        return self.cfg.seq_len(n_samples)

    def run_packed(self, audio, offsets: np.ndarray, language_idx: np.ndarray, audio_device_ptr: int | None = None):
        """audio: packed f32 samples (host ndarray) or None with `audio_device_ptr` (HBM-resident).
        Returns (token_ids [B, max_T] int32, num_id [B] int32)."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        lang = np.ascontiguousarray(language_idx, dtype=np.int32)
        B = lang.shape[0]
        assert offsets.shape[0] == B + 1
        max_t = max(self.cfg.seq_len(int(n)) for n in np.diff(offsets)) if B else 1
        tok = np.zeros((B, max_t), dtype=np.int32)
        num = np.zeros((B,), dtype=np.int32)
        lib = _lib.load()
        if audio_device_ptr is not None:
            ap, mem = C.c_void_p(audio_device_ptr), MEM_DEVICE
        else:
            audio = _f32(audio).reshape(-1)
            ap, mem = audio.ctypes.data_as(C.c_void_p), MEM_HOST
        _lib.check(lib.asr_sensevoice_run(self._h, ap, mem, offsets.ctypes.data_as(C.POINTER(C.c_int64)), B, _ip(lang),
                                          _ip(tok), max_t, _ip(num)))
        return tok, num

    def run(self, audios: Sequence[np.ndarray], language_idx: Sequence[int]):
        """List of 1-D utterances -> list of int32 token-id arrays (one per utterance)."""
        flat = [_f32(a).reshape(-1) for a in audios]
        offs = np.zeros(len(flat) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([a.size for a in flat])
        tok, num = self.run_packed(np.concatenate(flat), offs, np.asarray(language_idx, dtype=np.int32))
        return [tok[b, :num[b]].copy() for b in range(len(flat))]

    def utterance_rows(self, lengths: Sequence[int]):
        """(row_off, T) of each utterance inside the packed tap tensors (16-row aligned)."""
        out, r = [], 0
        for n in lengths:
            t = self.cfg.seq_len(int(n))
            out.append((r, t))
            r += (t + 15) // 16 * 16
        return out


# ------------------------------------------------------------------------------- operator hooks
def op_gemm(a, w, bias=None, act=0, precision=PRECISION_BF16):
    a, w = _f32(a), _f32(w)
    M, K = a.shape
    N = w.shape[0]
    out = np.empty((M, N), dtype=np.float32)
    b = _f32(bias) if bias is not None else None
    _lib.check(_lib.load().asr_op_gemm(precision, _fp(a), _fp(w), _fp(b), M, N, K, act, _fp(out)))
    return out


def op_gemm_ln(x, w, bias=None, gamma=None, beta=None):
    x, w = _f32(x), _f32(w)
    out = np.empty((x.shape[0], w.shape[0]), dtype=np.float32)
    b, g, be = (_f32(a) if a is not None else None for a in (bias, gamma, beta))
    _lib.check(_lib.load().asr_op_gemm_ln(_fp(x), _fp(w), _fp(b), _fp(g), _fp(be), x.shape[0], w.shape[0], x.shape[1], _fp(out)))
    return out


def op_layernorm(x, gamma=None, beta=None, eps=1e-5, precision=PRECISION_F32):
    x = _f32(x)
    rows, D = x.shape
    out = np.empty_like(x)
    g = _f32(gamma) if gamma is not None else None
    b = _f32(beta) if beta is not None else None
    _lib.check(_lib.load().asr_op_layernorm(precision, _fp(x), rows, D, _fp(g), _fp(b), eps, _fp(out)))
    return out


def op_attention(q, k, v, seq_lens, n_heads, d_head, precision=PRECISION_BF16):
    q, k, v = _f32(q), _f32(k), _f32(v)
    sl = np.ascontiguousarray(seq_lens, dtype=np.int32)
    out = np.empty_like(q)
    _lib.check(_lib.load().asr_op_attention(precision, _fp(q), _fp(k), _fp(v), _ip(sl), sl.size, n_heads, d_head, _fp(out)))
    return out


def op_fsmn(v, w, b, seq_lens, precision=PRECISION_F32):
    v, w, b = _f32(v), _f32(w), _f32(b)
    sl = np.ascontiguousarray(seq_lens, dtype=np.int32)
    out = np.empty_like(v)
    _lib.check(_lib.load().asr_op_fsmn(precision, _fp(v), _fp(w), _fp(b), _ip(sl), sl.size, v.shape[1], w.shape[1], _fp(out)))
    return out


def op_ctc_collapse(frame_ids, seq_lens, blank_id=0):
    ids = np.ascontiguousarray(frame_ids, dtype=np.int32)
    sl = np.ascontiguousarray(seq_lens, dtype=np.int32)
    max_t = int(sl.max())
    tok = np.zeros((sl.size, max_t), dtype=np.int32)
    num = np.zeros((sl.size,), dtype=np.int32)
    _lib.check(_lib.load().asr_op_ctc_collapse(_ip(ids), _ip(sl), sl.size, blank_id, _ip(tok), max_t, _ip(num)))
    return [tok[b, :num[b]].copy() for b in range(sl.size)]


def op_gemm_bench(M, N, K, variant=-1, epilogue=0, iters=50) -> float:
    """Tuning hook (probe library, not the product ABI): average milliseconds per launch of the bf16 GEMM."""
    from . import _probe
    return _probe.gemm_bench(M, N, K, variant, epilogue, iters)


def op_gemm_set_variant(variant: int = -1) -> None:
    """Pin the bf16 GEMM kernel variant for the following op_gemm calls (-1 = heuristic); probe-library hook."""
    from . import _probe
    _probe.gemm_set_variant(variant)


# =============================================================================== Whisper
def whisper_config_c(cfg, gelu_tanh: bool = False) -> _lib.WhisperConfigC:
    c = _lib.WhisperConfigC()
    c.sample_rate, c.n_mels, c.nfft, c.hop_length = cfg.sample_rate, cfg.n_mels, cfg.nfft, cfg.hop_length
    c.d_model, c.n_heads, c.d_head, c.d_ffn = cfg.d_model, cfg.n_heads, cfg.d_head, cfg.d_ffn
    c.n_enc_layers, c.n_dec_layers, c.vocab = cfg.n_enc_layers, cfg.n_dec_layers, cfg.vocab
    c.max_source_positions, c.max_target_positions, c.max_audio_len = cfg.max_source_positions, cfg.max_target_positions, cfg.max_audio_len
    c.gelu_tanh = int(gelu_tanh)
    return c


class WhisperSession(_Session):
    """HIP replacement of the merged Whisper graphs (encoder + KV-cache decoder + greedy heads)."""

    def __init__(self, cfg, arena, precision: int = PRECISION_BF16, device_id: int = 0, gelu_tanh: bool = False,
                 arena_device_ptr: int | None = None, arena_bytes: int | None = None):
        super().__init__()
        self.cfg, self.precision, self.device_id = cfg, precision, device_id
        self._cfg_c = whisper_config_c(cfg, gelu_tanh)
        lib = _lib.load()
        if arena_device_ptr is not None:
            self._keep = arena
            _lib.check(lib.asr_whisper_create(C.byref(self._cfg_c), C.c_void_p(arena_device_ptr), arena_bytes, MEM_DEVICE,
                                              device_id, precision, C.byref(self._h)))
        else:
            blob = np.ascontiguousarray(arena, dtype=np.uint8)
            _lib.check(lib.asr_whisper_create(C.byref(self._cfg_c), blob.ctypes.data_as(C.c_void_p), blob.nbytes, MEM_HOST,
                                              device_id, precision, C.byref(self._h)))
        self.batch = 0

    @classmethod
    def from_checkpoint(cls, cfg, ck, precision=PRECISION_BF16, device_id=0, suppress_tokens=None, begin_suppress_tokens=(),
                        gelu_tanh=False):
        from .arena import build_whisper_arena
        arena_precision = PRECISION_BF16 if precision in (PRECISION_FP8W, PRECISION_FP8MM, PRECISION_MXFP4W) else precision
        return cls(cfg, build_whisper_arena(cfg, ck, arena_precision, suppress_tokens, begin_suppress_tokens), precision, device_id, gelu_tanh)

    def encode_packed(self, audio, offsets, audio_device_ptr: int | None = None) -> np.ndarray:
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        B = offsets.size - 1
        npos = np.zeros(B, dtype=np.int32)
        if audio_device_ptr is not None:
            ap, mem = C.c_void_p(audio_device_ptr), MEM_DEVICE
        else:
            audio = _f32(audio).reshape(-1)
            ap, mem = audio.ctypes.data_as(C.c_void_p), MEM_HOST
        _lib.check(_lib.load().asr_whisper_encode(self._h, ap, mem, offsets.ctypes.data_as(C.POINTER(C.c_int64)), B, _ip(npos)))
        self.batch = B
        if self.precision == PRECISION_FP8MM:                     # a GELU operand that met the e4m3 clamp: loud, not silent (the scale is static)
            n, shift = self.fp8_stats()
            if n > getattr(self, "_fp8_saturated", 0):
                warnings.warn(f"Whisper FP8MM: {n - getattr(self, '_fp8_saturated', 0)} activation elements saturated at 448 * 2^{shift} in this encode; "
                              f"raise the shift (set_fp8_act_shift / ASR_FP8MM_ACT_SHIFT)", RuntimeWarning, stacklevel=2)
            self._fp8_saturated = n
        return npos

    def fp8_stats(self) -> tuple[int, int]:
        """(activation elements that met the e4m3 clamp since creation, activation shift in use) -- asr_whisper_fp8_stats; zeros outside FP8MM."""
        st = (C.c_uint64 * 2)()
        _lib.check(_lib.load().asr_whisper_fp8_stats(self._h, st))
        return int(st[0]), int(st[1])

    def set_fp8_act_shift(self, shift: int):
        """FP8MM: store fc2's GELU operand as value * 2^-shift (asr_whisper_set_fp8_act_shift)."""
        _lib.check(_lib.load().asr_whisper_set_fp8_act_shift(self._h, int(shift)))

    def encode(self, audios: Sequence[np.ndarray]) -> np.ndarray:
        flat = [_f32(a).reshape(-1) for a in audios]
        offs = np.zeros(len(flat) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([a.size for a in flat])
        return self.encode_packed(np.concatenate(flat), offs)

    def prefill(self, ids, want_logits: bool = True):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        assert ids.ndim == 2
        if self.batch and ids.shape[0] != self.batch:
            raise ValueError(f"prompt batch {ids.shape[0]} != encoded batch {self.batch}")
        nxt = np.zeros(ids.shape[0], dtype=np.int32)
        logits = np.empty((ids.shape[0], self.cfg.vocab), dtype=np.float32) if want_logits else None
        _lib.check(_lib.load().asr_whisper_prefill(self._h, _ip(ids), ids.shape[1], _ip(nxt), _fp(logits)))
        return nxt, logits

    def decode(self, ids=None, want_logits: bool = False, sync: bool = True):
        nxt = np.zeros(self.batch, dtype=np.int32) if sync else None
        logits = np.empty((self.batch, self.cfg.vocab), dtype=np.float32) if want_logits else None
        idp = _ip(np.ascontiguousarray(ids, dtype=np.int32)) if ids is not None else None
        _lib.check(_lib.load().asr_whisper_decode(self._h, idp, _ip(nxt) if nxt is not None else None, _fp(logits)))
        return nxt, logits

    def generate(self, max_new: int, eos_id: int):
        tok = np.zeros((self.batch, max_new), dtype=np.int32)
        n = np.zeros(self.batch, dtype=np.int32)
        _lib.check(_lib.load().asr_whisper_generate(self._h, max_new, eos_id, _ip(tok), _ip(n)))
        return [tok[b, :n[b]].copy() for b in range(self.batch)]

    def set_penalty(self, repeat_penalty: float = 1.0, penalty_range: int = 20):
        """Decode head: 1.0 = plain arg-max; else penalty-greedy (APPLY_PENALTY + GREEDY_SEARCH, the reference host's default)."""
        _lib.check(_lib.load().asr_whisper_set_penalty(self._h, C.c_float(repeat_penalty), int(penalty_range)))

    def no_speech_prob(self, no_speech_id: int | None = None) -> np.ndarray:
        """NO_SPEECH_DETECTION on the device-resident logits of the last prefill (the probe): (B,) probabilities."""
        out = np.zeros(self.batch, dtype=np.float32)
        _lib.check(_lib.load().asr_whisper_no_speech_prob(self._h, int(self.cfg.no_speech_id if no_speech_id is None else no_speech_id), _fp(out)))
        return out

    def track_history(self, enable: bool):
        """Append every pick to the device-side id history even while the penalty value is 1.0 (what the *PenaltyGreedy graphs do)."""
        _lib.check(_lib.load().asr_whisper_track_history(self._h, int(enable)))

    def set_sampling(self, enable: bool, temperature: float = 0.8, top_k: int = 10, top_p: float = 0.95,
                     repetition_penalty: float = 1.0, seed: int = 0):
        """TOPK_TOPP_SAMPLING head (USE_SAMPLING in the reference host); enable=False restores arg-max / penalty-greedy."""
        _lib.check(_lib.load().asr_whisper_set_sampling(self._h, int(enable), C.c_float(temperature), int(top_k), C.c_float(top_p),
                                                        C.c_float(repetition_penalty), C.c_uint64(seed)))

    def set_sampling_noise(self, uniforms):
        """Parity hook: uniforms [batch, top_k] for the next prefill / decode step (otherwise the device generator is used)."""
        u = _f32(uniforms).reshape(-1)
        _lib.check(_lib.load().asr_whisper_set_sampling_noise(self._h, _fp(u), u.size))

    def cross_kv(self, lengths_pos: Sequence[int]):
        """Debug: (K, V) per utterance as (L, H, T, 64) arrays from the 'cross' tap (f32 mode)."""
        cfg = self.cfg
        raw = self.tap("cross", dtype=np.float32 if self.precision == PRECISION_F32 else np.uint16)
        G = 2 * cfg.n_dec_layers * cfg.n_heads
        mpad = raw.shape[0] // G
        slabs = raw.reshape(2, cfg.n_dec_layers, cfg.n_heads, mpad, 64)
        out, r = [], 0
        for t in lengths_pos:
            out.append((slabs[0, :, :, r:r + t], slabs[1, :, :, r:r + t]))
            r += (t + 15) // 16 * 16
        return out


# =============================================================================== Paraformer
class ParaformerSession(_Session):
    """HIP replacement of `Paraformer.onnx` (PARAFORMER.forward, Export_Paraformer.py:474-563)."""

    def __init__(self, cfg, arena, precision: int = PRECISION_BF16, device_id: int = 0, arena_device_ptr: int | None = None,
                 arena_bytes: int | None = None):
        super().__init__()
        self.cfg, self.precision, self.device_id = cfg, precision, device_id
        c = _lib.ParaformerConfigC()
        for f in ("sample_rate", "n_mels", "nfft", "win_length", "hop_length", "lfr_m", "lfr_n", "d_model", "n_heads", "d_head", "d_ffn",
                  "fsmn_kernel", "n_dec", "n_dec3", "d_dec_ffn", "cif_kernel", "vocab", "max_audio_len"):
            setattr(c, f, getattr(cfg, f))
        c.n_blocks = cfg.n_enc0 + cfg.n_enc
        c.tail_threshold = cfg.tail_threshold
        self._cfg_c = c
        lib = _lib.load()
        if arena_device_ptr is not None:
            self._keep = arena
            _lib.check(lib.asr_paraformer_create(C.byref(c), C.c_void_p(arena_device_ptr), arena_bytes, MEM_DEVICE, device_id, precision,
                                                 C.byref(self._h)))
        else:
            blob = np.ascontiguousarray(arena, dtype=np.uint8)
            _lib.check(lib.asr_paraformer_create(C.byref(c), blob.ctypes.data_as(C.c_void_p), blob.nbytes, MEM_HOST, device_id, precision,
                                                 C.byref(self._h)))

    @classmethod
    def from_checkpoint(cls, cfg, ck, precision=PRECISION_BF16, device_id=0):
        from .arena import build_paraformer_arena
        return cls(cfg, build_paraformer_arena(cfg, ck, precision), precision, device_id)

    def run_packed(self, audio, offsets, audio_device_ptr: int | None = None):
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        B = offsets.size - 1
        max_t = max(self.cfg.seq_len(int(n)) for n in np.diff(offsets)) if B else 1
        tok = np.zeros((B, max_t), dtype=np.int32)
        num = np.zeros((B,), dtype=np.int32)
        if audio_device_ptr is not None:
            ap, mem = C.c_void_p(audio_device_ptr), MEM_DEVICE
        else:
            audio = _f32(audio).reshape(-1)
            ap, mem = audio.ctypes.data_as(C.c_void_p), MEM_HOST
        _lib.check(_lib.load().asr_paraformer_run(self._h, ap, mem, offsets.ctypes.data_as(C.POINTER(C.c_int64)), B, _ip(tok), max_t, _ip(num)))
        return tok, num

    def run(self, audios: Sequence[np.ndarray]):
        flat = [_f32(a).reshape(-1) for a in audios]
        offs = np.zeros(len(flat) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([a.size for a in flat])
        tok, num = self.run_packed(np.concatenate(flat), offs)
        return [tok[b, :num[b]].copy() for b in range(len(flat))]

    def utterance_rows(self, lengths: Sequence[int]):
        out, r = [], 0
        for n in lengths:
            t = self.cfg.seq_len(int(n))
            out.append((r, t))
            r += (t + 15) // 16 * 16
        return out

    @staticmethod
    def token_rows(num_tokens: Sequence[int]):
        """First row of each utterance's tokens in the decoder-side taps ('logits'): the fired frames are packed, 16-row aligned
        per utterance (a zero-token utterance keeps one dummy row)."""
        out, r = [], 0
        for n in num_tokens:
            out.append(r)
            r += (max(int(n), 1) + 15) // 16 * 16
        return out


class ParaformerStreamSession(_Session):
    """Streaming Paraformer: per-stream recurrent state (encoder K/V histories, carried LFR rows, CIF state, decoder FSMN / cross
    K/V histories) lives in the session; `step` advances a set of streams by one chunk (Export_Paraformer_Streaming.py:386-553)."""

    def __init__(self, cfg, ck_or_arena, precision: int = PRECISION_BF16, device_id: int = 0, chunk: int = 8000, look_back_encoder: int = 4,
                 look_back_decoder: int = 1, max_streams: int = 8, max_continue: int = 502):
        super().__init__()
        import dataclasses
        from .arena import build_paraformer_arena
        n_pos = max_continue - 1                                          # rows of the position table (positions 1 .. max_continue - 1)
        cfg = dataclasses.replace(cfg, max_audio_len=cfg.win_length + cfg.hop_length * (n_pos * cfg.lfr_n - 1))
        assert cfg.seq_len(cfg.max_audio_len) == n_pos
        self.cfg, self.precision, self.chunk, self.max_streams = cfg, precision, int(chunk), int(max_streams)
        n_frames = (chunk - cfg.win_length) // cfg.hop_length + 1
        self.rows_per_chunk = ((cfg.lfr_m - 1) // 2 + n_frames) // cfg.lfr_n + 1
        blob = build_paraformer_arena(cfg, ck_or_arena, precision, streaming=True) if isinstance(ck_or_arena, dict) else ck_or_arena
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        c = _lib.ParaformerConfigC()
        for f in ("sample_rate", "n_mels", "nfft", "win_length", "hop_length", "lfr_m", "lfr_n", "d_model", "n_heads", "d_head", "d_ffn",
                  "fsmn_kernel", "n_dec", "n_dec3", "d_dec_ffn", "cif_kernel", "vocab", "max_audio_len"):
            setattr(c, f, getattr(cfg, f))
        c.n_blocks = cfg.n_enc0 + cfg.n_enc
        c.tail_threshold = cfg.tail_threshold
        _lib.check(_lib.load().asr_paraformer_stream_create(C.byref(c), blob.ctypes.data_as(C.c_void_p), blob.nbytes, MEM_HOST, device_id, precision,
                                                            int(chunk), look_back_encoder, look_back_decoder, int(max_streams), C.byref(self._h)))

    def reset(self, stream_id: int = -1):
        _lib.check(_lib.load().asr_paraformer_stream_reset(self._h, int(stream_id)))

    def step(self, chunks, stream_ids, audio_device_ptr: int | None = None):
        """chunks: (n, chunk) int16-range float32 (or None with `audio_device_ptr`: HBM-resident [n][chunk] floats); stream_ids: n
        distinct ids -> list of n int32 arrays (tokens fired by this chunk)."""
        sid = np.ascontiguousarray(stream_ids, dtype=np.int32)
        if audio_device_ptr is not None:
            ap, mem = C.c_void_p(audio_device_ptr), MEM_DEVICE
        else:
            a = _f32(chunks).reshape(sid.size, self.chunk)
            ap, mem = a.ctypes.data_as(C.c_void_p), MEM_HOST
        cap = self.rows_per_chunk + 1
        tok = np.zeros((sid.size, cap), dtype=np.int32)
        num = np.zeros(sid.size, dtype=np.int32)
        _lib.check(_lib.load().asr_paraformer_stream_step(self._h, ap, mem, _ip(sid), sid.size, _ip(tok), cap, _ip(num)))
        return [tok[i, :num[i]].copy() for i in range(sid.size)]

    def stream_stats(self) -> dict:
        """Which path the chunk steps took (asr_paraformer_stream_stats): give-ups recovered, steps moved to the per-launch path because the GPU was shared,
        snapshots taken, the stream count above which steps stay on the per-launch path, cool-down steps left, whether the session fuses at all."""
        out = np.zeros(8, dtype=np.int32)
        _lib.check(_lib.load().asr_paraformer_stream_stats(self._h, _ip(out)))
        return {"giveups": int(out[0]), "shared_steps": int(out[1]), "snapshots": int(out[2]), "fused_max": int(out[3]), "cooldown": int(out[4]),
                "can_fuse": bool(out[5])}


# =============================================================================== Qwen3-ASR
class QwenAsrSession(_Session):
    """HIP replacement of the merged Qwen3-ASR graphs (audio encoder + prompt assembly + Qwen3 decoder prefill / decode + arg-max;
    Qwen_ASR/Inference_Qwen_ASR_ONNX.py:424-760 drives them)."""

    def __init__(self, cfg, arena, precision: int = PRECISION_BF16, device_id: int = 0, arena_device_ptr: int | None = None,
                 arena_bytes: int | None = None):
        super().__init__()
        self.cfg, self.precision, self.device_id = cfg, precision, device_id
        c = _lib.QwenConfigC()
        for f in ("sample_rate", "n_mels", "nfft", "hop_length", "enc_d", "enc_heads", "enc_ffn", "n_enc_layers", "conv_channels", "n_window",
                  "n_window_infer", "max_source_positions", "d_model", "n_heads", "n_kv_heads", "d_head", "d_ffn", "n_layers", "vocab",
                  "max_seq_len", "max_audio_len", "rms_eps", "rope_theta"):
            setattr(c, f, getattr(cfg, f))
        self._cfg_c = c
        if arena_device_ptr is not None:
            self._keep = arena
            _lib.check(_lib.load().asr_qwen_create(C.byref(c), C.c_void_p(arena_device_ptr), arena_bytes, MEM_DEVICE, device_id, precision,
                                                   C.byref(self._h)))
        else:
            blob = np.ascontiguousarray(arena, dtype=np.uint8)
            _lib.check(_lib.load().asr_qwen_create(C.byref(c), blob.ctypes.data_as(C.c_void_p), blob.nbytes, MEM_HOST, device_id, precision,
                                                   C.byref(self._h)))
        self.batch = 0

    @classmethod
    def from_checkpoint(cls, cfg, ck, precision=PRECISION_BF16, device_id=0):
        from .arena import build_qwen_asr_arena
        arena_precision = PRECISION_BF16 if precision in (PRECISION_FP8W, PRECISION_MXFP4W) else precision
        return cls(cfg, build_qwen_asr_arena(cfg, ck, arena_precision), precision, device_id)

    @staticmethod
    def _ragged(seqs, B):
        seqs = [np.asarray(x, dtype=np.int32).reshape(-1) for x in seqs]
        if len(seqs) == 1 and B > 1:
            seqs = seqs * B
        if len(seqs) != B:
            raise ValueError(f"{len(seqs)} prompts for a batch of {B}")
        offs = np.zeros(B + 1, dtype=np.int32)
        offs[1:] = np.cumsum([x.size for x in seqs])
        flat = np.concatenate(seqs) if offs[-1] else np.zeros(1, dtype=np.int32)
        return np.ascontiguousarray(flat, dtype=np.int32), offs

    def prefill_packed(self, audio, offsets, pre_ids, post_ids, want_logits: bool = True, audio_device_ptr: int | None = None):
        """pre_ids / post_ids: one id list per utterance (or one shared list): the prompt is [pre | audio embeddings | post].
        -> (next_ids (B,), logits (B, vocab) | None, ids_len (B,))"""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        B = offsets.size - 1
        pre, pre_off = self._ragged(pre_ids, B)
        post, post_off = self._ragged(post_ids, B)
        nxt = np.zeros(B, dtype=np.int32)
        ids_len = np.zeros(B, dtype=np.int32)
        logits = np.empty((B, self.cfg.vocab), dtype=np.float32) if want_logits else None
        if audio_device_ptr is not None:
            ap, mem = C.c_void_p(audio_device_ptr), MEM_DEVICE
        else:
            audio = _f32(audio).reshape(-1)
            ap, mem = audio.ctypes.data_as(C.c_void_p), MEM_HOST
        _lib.check(_lib.load().asr_qwen_prefill(self._h, ap, mem, offsets.ctypes.data_as(C.POINTER(C.c_int64)), B, _ip(pre), _ip(pre_off),
                                                _ip(post), _ip(post_off), _ip(nxt), _fp(logits), _ip(ids_len)))
        self.batch = B
        return nxt, logits, ids_len

    def prefill(self, audios: Sequence[np.ndarray], pre_ids, post_ids, want_logits: bool = True):
        flat = [_f32(a).reshape(-1) for a in audios]
        offs = np.zeros(len(flat) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([a.size for a in flat])
        return self.prefill_packed(np.concatenate(flat), offs, pre_ids, post_ids, want_logits)

    def decode(self, ids=None, want_logits: bool = False, sync: bool = True):
        nxt = np.zeros(self.batch, dtype=np.int32) if sync else None
        logits = np.empty((self.batch, self.cfg.vocab), dtype=np.float32) if want_logits else None
        idp = _ip(np.ascontiguousarray(ids, dtype=np.int32)) if ids is not None else None
        _lib.check(_lib.load().asr_qwen_decode(self._h, idp, _ip(nxt) if nxt is not None else None, _fp(logits)))
        return nxt, logits

    def set_penalty(self, repeat_penalty: float = 1.0, penalty_range: int = 10):
        """Decode head: 1.0 = plain arg-max; else penalty-greedy (the reference host's default is 0.8 over the last 10 ids)."""
        _lib.check(_lib.load().asr_qwen_set_penalty(self._h, C.c_float(repeat_penalty), int(penalty_range)))

    def track_history(self, enable: bool):
        """Append every pick to the device-side id history whatever the penalty value is (what the *_Penalty_Greedy graphs do)."""
        _lib.check(_lib.load().asr_qwen_track_history(self._h, int(enable)))

    def set_sampling(self, enable: bool, temperature: float = 0.8, top_k: int = 10, top_p: float = 0.95, repetition_penalty: float = 1.0, seed: int = 0):
        _lib.check(_lib.load().asr_qwen_set_sampling(self._h, int(enable), C.c_float(temperature), int(top_k), C.c_float(top_p),
                                                     C.c_float(repetition_penalty), C.c_uint64(seed)))

    def set_sampling_noise(self, uniforms):
        """Parity hook: uniforms [batch, top_k] for the next prefill / decode step (otherwise the device generator is used)."""
        u = _f32(uniforms).reshape(-1)
        _lib.check(_lib.load().asr_qwen_set_sampling_noise(self._h, _fp(u), u.size))

    def audio_tokens(self, n_samples: int) -> int:
        """_get_feat_extract_output_lengths (Export_Qwen_ASR.py:519-527) of a clip's mel frames."""
        n = int(n_samples) // self.cfg.hop_length
        f = n % self.cfg.chunk
        for _ in range(3):
            f = (max(f - 1, 0) // 2 + 1) if f > 0 else 0
        return f + (n // self.cfg.chunk) * 13

    def audio_hidden(self, n_samples: Sequence[int]):
        """Debug: per-utterance audio embeddings (tokens, d_model) from the 'audio_hidden' tap (rows live in window slots)."""
        cfg = self.cfg
        raw = self.tap("audio_hidden", dtype=np.float32)
        cpw = cfg.chunks_per_window
        rpw = (cpw * 13 + 15) // 16 * 16
        out, win = [], 0
        for n in n_samples:
            frames = int(n) // cfg.hop_length
            n_win = ((frames + cfg.chunk - 1) // cfg.chunk + cpw - 1) // cpw
            rows = raw[win * rpw:(win + n_win) * rpw].reshape(n_win, rpw, -1)[:, :cpw * 13].reshape(n_win * cpw * 13, -1)
            out.append(rows[:self.audio_tokens(n)].copy())
            win += n_win
        return out

    def generate(self, max_new: int, stop_ids=()):
        tok = np.zeros((self.batch, max_new), dtype=np.int32)
        n = np.zeros(self.batch, dtype=np.int32)
        stop = np.ascontiguousarray(list(stop_ids), dtype=np.int32)
        _lib.check(_lib.load().asr_qwen_generate(self._h, max_new, _ip(stop) if stop.size else None, stop.size, _ip(tok), _ip(n)))
        return [tok[b, :n[b]].copy() for b in range(self.batch)]

    def kv_stats(self) -> dict:
        """The KV cache's page accounting (asr_qwen_kv_stats): paged?, pages in the pool, pages held now, high-water mark since the prefill."""
        out = np.zeros(4, dtype=np.int32)
        _lib.check(_lib.load().asr_qwen_kv_stats(self._h, _ip(out)))
        return {"paged": bool(out[0]), "pool_pages": int(out[1]), "held": int(out[2]), "high_water": int(out[3])}

    def beam_search(self, beam: int, max_new: int, stop_ids=()):
        """Width-`beam` search after a prefill -> per utterance a best-first list of (token ids, summed log-probability)."""
        tok = np.zeros((self.batch, beam, max_new), dtype=np.int32)
        n = np.zeros((self.batch, beam), dtype=np.int32)
        score = np.zeros((self.batch, beam), dtype=np.float32)
        stop = np.ascontiguousarray(list(stop_ids), dtype=np.int32)
        _lib.check(_lib.load().asr_qwen_beam_search(self._h, int(beam), int(max_new), _ip(stop) if stop.size else None, stop.size, _ip(tok), _ip(n),
                                                    _fp(score)))
        return [[(tok[b, r, :n[b, r]].copy(), float(score[b, r])) for r in range(beam)] for b in range(self.batch)]


def load_session(path: str, device_id: int = 0):
    """Open an `.asrmodel` bundle (tools/convert_checkpoint.py, export_*) as the matching native session."""
    from . import config as cfgm
    from .ort_shim import load_model
    info, blob = load_model(path)
    kind, conf, prec = info["kind"], dict(info["config"] or {}), int(info.get("precision", 0))
    if kind == "sensevoice":
        conf["language_prompt_token_ids"] = tuple(conf["language_prompt_token_ids"])
        return SenseVoiceSession(cfgm.SenseVoiceConfig(**conf), blob, prec, device_id)
    if kind == "paraformer":
        return ParaformerSession(cfgm.ParaformerConfig(**conf), blob, prec, device_id)
    if kind == "whisper":
        return WhisperSession(cfgm.WhisperConfig(**conf), blob, prec, device_id)
    if kind == "qwen_asr":
        return QwenAsrSession(cfgm.QwenAsrConfig(**conf), blob, prec, device_id)
    raise ValueError(f"{path!r}: no native session for bundle kind {kind!r}")
