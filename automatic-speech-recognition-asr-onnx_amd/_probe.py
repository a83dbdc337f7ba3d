"""ctypes binding of libasr_mi355x_probe.so (include/asr_mi355x_probe.h): test / tuning hooks, NOT the product ABI.

Used by tests/ and tools/ only; the transcribers, the onnxruntime shim and bench.py's timed path never import this."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib

PROBE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libasr_mi355x_probe.so")
_fp, _ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)


class GemmDesc(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("a", _fp), ("w", _fp), ("bias", _fp), ("add", _fp),
                ("act", C.c_int32), ("ln", C.c_int32), ("ln_eps", C.c_float), ("argmax", C.c_int32), ("n_valid", C.c_int32),
                ("variant", C.c_int32), ("out_lo", _fp), ("out_f32", _fp), ("out_stats", _fp), ("out_ids", _ip),
                ("kernel", C.c_char * 32)]


SIGNATURES = {
    "asr_probe_gemm": (C.c_int, [C.POINTER(GemmDesc)]),
    "asr_probe_gemm_chain": (C.c_int, [C.c_int] * 6 + [_fp]),
    "asr_probe_last_kernel": (C.c_char_p, []),
    "asr_probe_quantize_fp8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, _fp, C.c_void_p]),
    "asr_probe_quantize_mxfp4": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "asr_probe_decode_gemm_mxfp4": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _fp, C.c_int, _fp]),
    "asr_probe_decode_gemm": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _fp, _fp, C.c_int, _fp]),
    "asr_probe_gemm_counts": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "asr_probe_gemm_fp8": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, _fp, C.c_float, _fp, _fp, C.c_int, C.c_void_p, _fp, C.c_int, _fp]),
    "asr_probe_gemm_bench": (C.c_int, [C.c_int] * 6 + [_fp]),
    "asr_probe_grid_barrier": (C.c_int, [C.c_int, C.c_int, _fp]),
    "asr_probe_grid_barrier2": (C.c_int, [C.c_int, C.c_int, C.c_int, _fp]),
}
_plib = None


def load():
    global _plib
    if _plib is None:
        _lib.load()                       # the probe library resolves the product library's launchers
        if not os.path.isfile(PROBE_PATH):
            raise ImportError(f"{PROBE_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(PROBE_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _plib = lib
    return _plib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def gemm(a, w, bias=None, add=None, act=0, ln=False, ln_eps=1e-5, argmax=False, n_valid=0, variant=-1, want=("lo",)):
    """One bf16 GEMM through the product dispatcher; returns (dict of outputs, kernel family name)."""
    a, w = _f32(a), _f32(w)
    M, K = a.shape
    N = w.shape[0]
    d = GemmDesc()
    d.M, d.N, d.K, d.act, d.ln, d.ln_eps, d.argmax, d.n_valid, d.variant = M, N, K, act, int(ln), ln_eps, int(argmax), n_valid, variant
    keep = [a, w]
    d.a, d.w = a.ctypes.data_as(_fp), w.ctypes.data_as(_fp)
    if bias is not None:
        bias = _f32(bias); keep.append(bias); d.bias = bias.ctypes.data_as(_fp)
    if add is not None:
        add = _f32(add); keep.append(add); d.add = add.ctypes.data_as(_fp)
    out = {}
    if argmax:
        out["ids"] = np.zeros((M,), dtype=np.int32); d.out_ids = out["ids"].ctypes.data_as(_ip)
    else:
        if "lo" in want:
            out["lo"] = np.zeros((M, N), dtype=np.float32); d.out_lo = out["lo"].ctypes.data_as(_fp)
        if "f32" in want:
            out["f32"] = np.zeros((M, N), dtype=np.float32); d.out_f32 = out["f32"].ctypes.data_as(_fp)
        if "stats" in want:
            out["stats"] = np.zeros((M, N // 32, 2), dtype=np.float32); d.out_stats = out["stats"].ctypes.data_as(_fp)
    _lib.check(load().asr_probe_gemm(C.byref(d)))
    return out, d.kernel.decode()


def gemm_bench(M, N, K, variant=-1, epilogue=0, iters=50) -> float:
    """Average milliseconds per launch of the bf16 GEMM (device-resident operands)."""
    ms = C.c_float(0.0)
    _lib.check(load().asr_probe_gemm_bench(variant, M, N, K, epilogue, iters, C.byref(ms)))
    return ms.value


def gemm_set_variant(variant: int = -1) -> None:
    """Pin the bf16 GEMM kernel variant for the following op_gemm calls (-1 = heuristic)."""
    gemm_bench(1152, 256, 64, variant, 0, 1)


def gemm_counts(reset: bool = False) -> dict:
    """Launches per GEMM kernel family since the last reset (host-side; a hipGraph replay does not count)."""
    buf = C.create_string_buffer(1024)
    _lib.check(load().asr_probe_gemm_counts(int(reset), buf, 1024))
    return {k: int(v) for k, v in (kv.split("=") for kv in buf.value.decode().split(";") if kv)}


def gemm_chain(M, N, K, epilogue=0, cold_mb=768, replays=5):
    """(microseconds per launch, kernel family) of a captured chain of decode-shaped GEMMs over cold weights."""
    us = C.c_float(0.0)
    _lib.check(load().asr_probe_gemm_chain(M, N, K, epilogue, cold_mb, replays, C.byref(us)))
    return us.value, load().asr_probe_last_kernel().decode()


def _bf16_bits(x):
    """f32 array -> bf16 bit patterns (round to nearest even), as uint16."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


def _bf16_to_f32(bits):
    return (np.ascontiguousarray(bits, np.uint16).astype(np.uint32) << 16).view(np.float32)


def quantize_fp8(w):
    """Row quantiser of the FP8 mode on an f32 array (rounded to bf16 first): (bytes [N][K] uint8, scale [N] f32, dequantised f32 [N][K])."""
    wb = _bf16_bits(w)
    N, K = wb.shape
    q = np.zeros((N, K), np.uint8); sc = np.zeros((N,), np.float32); dq = np.zeros((N, K), np.uint16)
    _lib.check(load().asr_probe_quantize_fp8(wb.ctypes.data, N, K, q.ctypes.data, sc.ctypes.data_as(_fp), dq.ctypes.data))
    return q, sc, _bf16_to_f32(dq)


def quantize_mxfp4(w):
    """Block quantiser of the MXFP4 mode on an f32 array (rounded to bf16 first): (nibbles [N][K / 2] uint8, e8m0 scales [N][K / 32] uint8, dequantised f32 [N][K])."""
    wb = _bf16_bits(w)
    N, K = wb.shape
    q = np.zeros((N, K // 2), np.uint8); sc = np.zeros((N, K // 32), np.uint8); dq = np.zeros((N, K), np.uint16)
    _lib.check(load().asr_probe_quantize_mxfp4(wb.ctypes.data, N, K, q.ctypes.data, sc.ctypes.data, dq.ctypes.data))
    return q, sc, _bf16_to_f32(dq)


def decode_gemm_mxfp4(a, w4, scale8, w_dq=None, bias=None, fold=False):
    """Decode GEMM (<= 64 rows) over MXFP4 weights; w_dq (f32, exact in bf16) supplies the column sums of the LayerNorm fold."""
    ab = _bf16_bits(a)
    M, K = ab.shape
    N = w4.shape[0]
    out = np.zeros((M, N), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    dq = None if w_dq is None else _bf16_bits(w_dq)
    q, s = np.ascontiguousarray(w4, np.uint8), np.ascontiguousarray(scale8, np.uint8)
    _lib.check(load().asr_probe_decode_gemm_mxfp4(M, N, K, ab.ctypes.data, q.ctypes.data, s.ctypes.data, None if dq is None else dq.ctypes.data,
                                                  None if b is None else b.ctypes.data_as(_fp), int(fold), out.ctypes.data_as(_fp)))
    return out


def decode_gemm(a, w=None, w8=None, scale=None, bias=None, fold=False):
    """Decode GEMM (<= 64 rows) on host arrays: a [M][K] f32 (rounded to bf16), weights either f32 `w` (rounded to bf16) or (w8, scale)."""
    ab = _bf16_bits(a)
    M, K = ab.shape
    wb = None if w is None else _bf16_bits(w)
    N = (wb if wb is not None else w8).shape[0]
    out = np.zeros((M, N), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    s = None if scale is None else np.ascontiguousarray(scale, np.float32)
    q = None if w8 is None else np.ascontiguousarray(w8, np.uint8)
    _lib.check(load().asr_probe_decode_gemm(M, N, K, ab.ctypes.data, None if wb is None else wb.ctypes.data, None if q is None else q.ctypes.data,
                                            None if s is None else s.ctypes.data_as(_fp), None if b is None else b.ctypes.data_as(_fp), int(fold),
                                            out.ctypes.data_as(_fp)))
    return out


def e4m3_table():
    """The 256 OCP e4m3 values (index = byte; 0x7f / 0xff are NaN)."""
    t = np.zeros(256, np.float64)
    for v in range(256):
        s, e, m = v >> 7, (v >> 3) & 15, v & 7
        x = (m / 8.0) * 2.0 ** -6 if e == 0 else (1.0 + m / 8.0) * 2.0 ** (e - 7)
        if e == 15 and m == 7:
            x = np.nan
        t[v] = -x if s else x
    return t


def gemm_fp8(a8, w8, w_scale, bias, a_scale=1.0, add=None, act=0, iters=0):
    """FP8 matrix-pipe GEMM on host arrays: a8 [M][K], w8 [N][K] uint8 (e4m3). Returns (out, us): out uint8 [M][N] (bytes of act(...)) without `add`,
    else float32 [M][N]."""
    a8, w8 = np.ascontiguousarray(a8, np.uint8), np.ascontiguousarray(w8, np.uint8)
    M, K = a8.shape
    N = w8.shape[0]
    sc, b = _f32(w_scale), _f32(bias)
    us = C.c_float(0.0)
    if add is None:
        out = np.zeros((M, N), np.uint8)
        _lib.check(load().asr_probe_gemm_fp8(M, N, K, a8.ctypes.data, w8.ctypes.data, sc.ctypes.data_as(_fp), a_scale, b.ctypes.data_as(_fp), None, act,
                                             out.ctypes.data, None, iters, C.byref(us)))
    else:
        r = _f32(add)
        out = np.zeros((M, N), np.float32)
        _lib.check(load().asr_probe_gemm_fp8(M, N, K, a8.ctypes.data, w8.ctypes.data, sc.ctypes.data_as(_fp), a_scale, b.ctypes.data_as(_fp), r.ctypes.data_as(_fp), act,
                                             None, out.ctypes.data_as(_fp), iters, C.byref(us)))
    return out, us.value
