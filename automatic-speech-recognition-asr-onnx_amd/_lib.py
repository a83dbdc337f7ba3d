"""ctypes binding of libasr_mi355x.so (the C ABI declared in include/asr_mi355x.h).

There is no CPU fallback: if the shared library is missing, `load()` raises, and every entry
point of the library itself fails with ASR_STATUS_NO_DEVICE when no GPU is visible.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libasr_mi355x.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

_lib = None


class AsrError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"[asr_mi355x status {code}] {message}")
        self.code = code


class SenseVoiceConfigC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "sample_rate", "n_mels", "nfft", "win_length", "hop_length", "lfr_m", "lfr_n", "d_model", "n_heads", "d_head",
        "d_ffn", "n_blocks", "n_main", "fsmn_kernel", "vocab", "blank_id", "n_prompt", "n_languages", "max_audio_len")] + [
        ("reserved", C.c_int32 * 8)]


class ParaformerConfigC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "sample_rate", "n_mels", "nfft", "win_length", "hop_length", "lfr_m", "lfr_n", "d_model", "n_heads", "d_head", "d_ffn",
        "n_blocks", "fsmn_kernel", "n_dec", "n_dec3", "d_dec_ffn", "cif_kernel", "vocab", "max_audio_len")] + [
        ("tail_threshold", C.c_float), ("reserved", C.c_int32 * 8)]


class WhisperConfigC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "sample_rate", "n_mels", "nfft", "hop_length", "d_model", "n_heads", "d_head", "d_ffn", "n_enc_layers",
        "n_dec_layers", "vocab", "max_source_positions", "max_target_positions", "max_audio_len", "gelu_tanh")] + [
        ("reserved", C.c_int32 * 9)]


class QwenConfigC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "sample_rate", "n_mels", "nfft", "hop_length", "enc_d", "enc_heads", "enc_ffn", "n_enc_layers", "conv_channels", "n_window",
        "n_window_infer", "max_source_positions", "d_model", "n_heads", "n_kv_heads", "d_head", "d_ffn", "n_layers", "vocab", "max_seq_len",
        "max_audio_len")] + [("rms_eps", C.c_float), ("rope_theta", C.c_float), ("reserved", C.c_int32 * 9)]


# name -> (restype, argtypes); every symbol include/asr_mi355x.h declares
_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
_fp, _ip, _lp, _dp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double)
SIGNATURES = {
    "asr_abi_version": (_i, []),
    "asr_last_error": (C.c_char_p, []),
    "asr_device_count": (_i, [C.POINTER(C.c_int)]),
    "asr_device_foreign_begin": (_i, [_i]),
    "asr_device_foreign_end": (_i, [_i]),
    "asr_device_foreign_stats": (_i, [_i, _lp]),
    "asr_sensevoice_create": (_i, [C.POINTER(SenseVoiceConfigC), _vp, _sz, _i, _i, _i, C.POINTER(_vp)]),
    "asr_sensevoice_run": (_i, [_vp, _vp, _i, _lp, _i, _ip, _ip, _i, _ip]),
    "asr_sanm_stats": (_i, [_vp, _ip]),
    "asr_sensevoice_seq_len": (_i, [C.POINTER(SenseVoiceConfigC), _i, C.POINTER(C.c_int)]),
    "asr_paraformer_create": (_i, [C.POINTER(ParaformerConfigC), _vp, _sz, _i, _i, _i, C.POINTER(_vp)]),
    "asr_paraformer_run": (_i, [_vp, _vp, _i, _lp, _i, _ip, _i, _ip]),
    "asr_paraformer_stream_create": (_i, [C.POINTER(ParaformerConfigC), _vp, _sz, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "asr_paraformer_stream_reset": (_i, [_vp, _i]),
    "asr_paraformer_stream_step": (_i, [_vp, _vp, _i, _ip, _i, _ip, _i, _ip]),
    "asr_paraformer_stream_stats": (_i, [_vp, _ip]),
    "asr_whisper_create": (_i, [C.POINTER(WhisperConfigC), _vp, _sz, _i, _i, _i, C.POINTER(_vp)]),
    "asr_whisper_encode": (_i, [_vp, _vp, _i, _lp, _i, _ip]),
    "asr_whisper_prefill": (_i, [_vp, _ip, _i, _ip, _fp]),
    "asr_whisper_decode": (_i, [_vp, _ip, _ip, _fp]),
    "asr_whisper_generate": (_i, [_vp, _i, _i, _ip, _ip]),
    "asr_whisper_set_penalty": (_i, [_vp, C.c_float, _i]),
    "asr_whisper_track_history": (_i, [_vp, _i]),
    "asr_whisper_set_fp8_act_shift": (_i, [_vp, _i]),
    "asr_whisper_fp8_stats": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "asr_whisper_no_speech_prob": (_i, [_vp, _i, _fp]),
    "asr_whisper_set_sampling": (_i, [_vp, _i, C.c_float, _i, C.c_float, C.c_float, C.c_uint64]),
    "asr_whisper_set_sampling_noise": (_i, [_vp, _fp, _i]),
    "asr_qwen_create": (_i, [C.POINTER(QwenConfigC), _vp, _sz, _i, _i, _i, C.POINTER(_vp)]),
    "asr_qwen_prefill": (_i, [_vp, _vp, _i, _lp, _i, _ip, _ip, _ip, _ip, _ip, _fp, _ip]),
    "asr_qwen_decode": (_i, [_vp, _ip, _ip, _fp]),
    "asr_qwen_generate": (_i, [_vp, _i, _ip, _i, _ip, _ip]),
    "asr_qwen_kv_stats": (_i, [_vp, _ip]),
    "asr_qwen_beam_search": (_i, [_vp, _i, _i, _ip, _i, _ip, _ip, _fp]),
    "asr_qwen_set_penalty": (_i, [_vp, C.c_float, _i]),
    "asr_qwen_track_history": (_i, [_vp, _i]),
    "asr_qwen_set_sampling": (_i, [_vp, _i, C.c_float, _i, C.c_float, C.c_float, C.c_uint64]),
    "asr_qwen_set_sampling_noise": (_i, [_vp, _fp, _i]),
    "asr_mem_alloc": (_i, [_i, _sz, C.POINTER(_vp)]),
    "asr_mem_free": (_i, [_i, _vp]),
    "asr_mem_copy": (_i, [_i, _vp, _vp, _sz, _i]),
    "asr_session_destroy": (_i, [_vp]),
    "asr_session_set_stream": (_i, [_vp, _vp]),
    "asr_session_device": (_i, [_vp, C.POINTER(C.c_int)]),
    "asr_session_profile_enable": (_i, [_vp, _i]),
    "asr_session_profile_reset": (_i, [_vp]),
    "asr_session_profile_read": (_i, [_vp, _i, C.c_char_p, _dp, _lp, C.POINTER(C.c_int)]),
    "asr_session_taps_enable": (_i, [_vp, _i]),
    "asr_session_tap_shape": (_i, [_vp, C.c_char_p, _lp, _lp]),
    "asr_session_tap_read": (_i, [_vp, C.c_char_p, _vp, _sz]),
    "asr_op_gemm": (_i, [_i, _fp, _fp, _fp, _i, _i, _i, _i, _fp]),
    "asr_op_layernorm": (_i, [_i, _fp, _i, _i, _fp, _fp, C.c_float, _fp]),
    "asr_op_attention": (_i, [_i, _fp, _fp, _fp, _ip, _i, _i, _i, _fp]),
    "asr_op_fsmn": (_i, [_i, _fp, _fp, _fp, _ip, _i, _i, _i, _fp]),
    "asr_op_gemm_ln": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _fp]),
    "asr_op_ctc_collapse": (_i, [_ip, _ip, _i, _i, _ip, _i, _ip]),
}


def build(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libasr_mi355x.so (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC_DIR, "clean"], check=True, capture_output=True)
    res = subprocess.run(["make", "-C", CSRC_DIR, "-j8"], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libasr_mi355x.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the engine has no CPU fallback)")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError here == missing export
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int):
    if status != 0:
        raise AsrError(status, load().asr_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    n = C.c_int(0)
    check(load().asr_device_count(C.byref(n)))
    return n.value
