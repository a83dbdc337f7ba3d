"""CPU: the converter's arena layout, and that the C-ABI library loads and exports every symbol
include/asr_mi355x.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re
import struct

import numpy as np
import pytest

from conftest import ROOT, sub
from helpers import sensevoice_setup


def test_library_exports_every_declared_symbol():
    _lib = sub("_lib")
    if not os.path.isfile(_lib.LIB_PATH):
        _lib.build()
    hdr = open(os.path.join(ROOT, "include", "asr_mi355x.h")).read()
    declared = set(re.findall(r"\b(asr_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes found"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/asr_mi355x.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    loaded = _lib.load()
    assert loaded.asr_abi_version() == 1


def test_precision_modes_agree_between_header_and_host():
    import re
    hdr = open(os.path.join(ROOT, "include", "asr_mi355x.h")).read()
    m = re.search(r"enum asr_precision \{([^}]*)\}", hdr)
    vals = dict((k.strip(), int(v)) for k, v in (item.split("=") for item in m.group(1).split(",")))
    arena = sub("arena")
    assert vals == {"ASR_PRECISION_BF16": arena.PRECISION_BF16, "ASR_PRECISION_F32": arena.PRECISION_F32, "ASR_PRECISION_FP8W": arena.PRECISION_FP8W,
                    "ASR_PRECISION_FP8MM": arena.PRECISION_FP8MM, "ASR_PRECISION_MXFP4W": arena.PRECISION_MXFP4W}


def test_probe_library_is_separate_from_the_product_abi():
    """The tuning / test hooks live in libasr_mi355x_probe.so: every symbol its header declares is exported there,
    and the product library exports none of them (and declares no probe / debug entry)."""
    _lib, _probe = sub("_lib"), sub("_probe")
    if not os.path.isfile(_probe.PROBE_PATH):
        _lib.build()
    hdr = open(os.path.join(ROOT, "include", "asr_mi355x_probe.h")).read()
    declared = set(re.findall(r"\b(asr_probe_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_probe.SIGNATURES), declared ^ set(_probe.SIGNATURES)
    plib = _probe.load()
    product = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(plib, name), f"{name} not exported by the probe library"
        assert not hasattr(product, name), f"{name} leaked into the product library"
    phdr = open(os.path.join(ROOT, "include", "asr_mi355x.h")).read()
    assert not re.search(r"asr_(debug|probe)_|gemm_bench", phdr)


def test_no_cpu_fallback_without_gpu():
    _lib = sub("_lib")
    if not os.path.isfile(_lib.LIB_PATH):
        _lib.build()
    if _lib.device_count() > 0:
        pytest.skip("GPU present")
    eng = sub("engine")
    with pytest.raises(_lib.AsrError) as e:
        eng.op_gemm(np.zeros((4, 64), np.float32), np.zeros((128, 64), np.float32))
    assert e.value.code == 5 and "no CPU fallback" in str(e.value)


def test_arena_layout_roundtrip():
    arena = sub("arena")
    cfg, ck = sensevoice_setup("sensevoice_tiny")
    blob = arena.build_sensevoice_arena(cfg, ck, arena.PRECISION_BF16)
    magic, ver, n, data_off, total = struct.unpack("<8sIIQQ", blob[:32].tobytes())
    assert magic == b"ASRARENA" and ver == 1 and total == blob.nbytes
    recs = {}
    for i in range(n):
        name, dt, nd, s0, s1, s2, s3, off = struct.unpack("<80sII4qQ", blob[32 + 128 * i: 160 + 128 * i].tobytes())
        recs[name.rstrip(b"\0").decode()] = (dt, (s0, s1, s2, s3)[:nd], off)
        assert off % 256 == 0 and off >= data_off
    d = cfg.d_model
    assert recs["blk0.wqkv"][0] == arena.DT_BF16 and recs["blk0.wqkv"][1] == (3 * d, 576)   # K padded 560 -> 576
    assert recs["blk1.wqkv"][1] == (3 * d, d)
    assert recs["ctc.w"][1] == (1024, d)
    # folds (Export_SenseVoice.py:208-220): q,k rows scaled by d_k^-1/4, v rows untouched, FSMN centre tap + 1
    dt, shape, off = recs["blk1.wqkv"]
    w = blob[off: off + 2 * shape[0] * shape[1]].view(np.uint16).reshape(shape)
    as_f32 = lambda u: (u.astype(np.uint32) << 16).view(np.float32)
    raw = ck["encoder.encoders.0.self_attn.linear_q_k_v.weight"]
    s = np.float32(cfg.d_head ** -0.25)
    # bf16 (performance) arenas also absorb the LayerNorm affine: W' = W diag(gamma), b' = b + W beta, plus the column sums the
    # engine needs to evaluate the normalisation inside the GEMM; f32 (verification) arenas keep the reference's layout
    gamma, beta = ck["encoder.encoders.0.norm1.weight"], ck["encoder.encoders.0.norm1.bias"]
    assert np.abs(as_f32(w[:2 * d]) - raw[:2 * d] * s * gamma[None, :]).max() < 1e-2
    assert np.abs(as_f32(w[2 * d:]) - raw[2 * d:] * gamma[None, :]).max() < 1e-2
    assert "blk1.ln1_g" not in recs and recs["blk1.cqkv"][1] == (3 * d,) and recs["blk1.c1"][1] == (cfg.d_ffn,)
    dt, shape, off = recs["blk1.cqkv"]
    cq = blob[off: off + 4 * shape[0]].view(np.float32)
    assert np.allclose(cq, as_f32(w).astype(np.float64).sum(1), rtol=1e-6, atol=1e-6)
    dt, shape, off = recs["blk1.bqkv"]
    bq = blob[off: off + 4 * shape[0]].view(np.float32)
    braw = ck["encoder.encoders.0.self_attn.linear_q_k_v.bias"].astype(np.float64) + raw.astype(np.float64) @ beta.astype(np.float64)
    braw[:2 * d] *= float(s)
    assert np.allclose(bq, braw, rtol=1e-5, atol=1e-6)
    blob32 = arena.build_sensevoice_arena(cfg, ck, arena.PRECISION_F32)
    n32 = struct.unpack("<8sIIQQ", blob32[:32].tobytes())[2]
    names32 = {struct.unpack("<80s", blob32[32 + 128 * i: 112 + 128 * i].tobytes())[0].rstrip(b"\0").decode() for i in range(n32)}
    assert "blk1.ln1_g" in names32 and "blk1.cqkv" not in names32
    dt, shape, off = recs["blk1.wfsmn"]
    wf = blob[off: off + 4 * shape[0] * shape[1]].view(np.float32).reshape(shape)
    rawf = ck["encoder.encoders.0.self_attn.fsmn_block.weight"][:, 0, :]
    assert np.allclose(wf[:, 5], rawf[:, 5] + 1.0) and np.array_equal(wf[:, :5], rawf[:, :5])


def test_dft_fragment_packing_matches_dense_matrix():
    arena = sub("arena")
    cfg, _ = sensevoice_setup("sensevoice_tiny")
    kmat = arena.kaldi_fbank_matrix(cfg).numpy()
    packed = arena.pack_dft_for_mfma(kmat, 257, 400).reshape(17, 2, 25, 64, 4)
    rng = np.random.default_rng(0)
    for _ in range(200):
        t, part, kc, lane, j = rng.integers(17), rng.integers(2), rng.integers(25), rng.integers(64), rng.integers(4)
        b = t * 16 + (lane & 15)
        k = kc * 16 + 4 * j + (lane >> 4)
        want = kmat[part * 257 + b, k] if b < 257 else 0.0
        assert packed[t, part, kc, lane, j] == want


def test_kaldi_mel_banks_known_answers():
    """The product's mel banks vs the oracle's restatement and analytic properties of Kaldi's MelBanks."""
    import torch
    arena = sub("arena")
    from oracle.kaldi_mel import get_mel_banks
    a = arena.kaldi_mel_banks(80, 512, 16000.0)
    b, centers = get_mel_banks(80, 512, 16000.0, 20.0, 0.0, 100.0, -500.0, 1.0)
    assert a.shape == (80, 256) and torch.equal(a, b)
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0
    # triangles: centre frequencies increase, each filter is unimodal and non-empty, bin 0 (DC, below 20 Hz) is unused
    assert torch.all(centers[1:] > centers[:-1])
    assert float(a[:, 0].max()) == 0.0
    peaks = a.argmax(dim=1)
    assert torch.all(peaks[1:] >= peaks[:-1]) and torch.all(a.sum(dim=1) > 0)
    # adjacent triangles partition unity between the first and last centre
    inner = (torch.arange(256) * 31.25 > float(centers[0])) & (torch.arange(256) * 31.25 < float(centers[-1]))
    assert torch.allclose(a.sum(dim=0)[inner], torch.ones(int(inner.sum())), atol=1e-4)


def test_kaldi_mel_banks_against_an_independent_statement_of_kaldis_loop():
    """Third, code-independent statement of Kaldi's `MelBanks::MelBanks` (feat/mel-computations.cc: per bin, per FFT bin, scalar double
    arithmetic, strict `mel > left && mel < right` membership, the two-slope formula) -- the product (arena.py) and the oracle
    (oracle/kaldi_mel.py) were written by the same hand in vectorised float32; a shared misreading of the published algorithm would
    have to be repeated here in a different form to go unnoticed. torchaudio itself is absent: parity with ITS output
    stays unpinned until a dumped (80, 256) matrix can be committed."""
    import math
    arena = sub("arena")
    from oracle.kaldi_mel import get_mel_banks
    num_bins, padded, sr, low = 80, 512, 16000.0, 20.0
    n_fft_bins, high = padded // 2, 0.5 * sr                     # high_freq = 0 -> Nyquist
    bin_width = sr / padded

    def mel(f):
        return 1127.0 * math.log(1.0 + f / 700.0)
    mel_low, mel_high = mel(low), mel(high)
    delta = (mel_high - mel_low) / (num_bins + 1)
    want = np.zeros((num_bins, n_fft_bins), dtype=np.float64)
    for b in range(num_bins):
        left, center, right = mel_low + b * delta, mel_low + (b + 1) * delta, mel_low + (b + 2) * delta
        for i in range(n_fft_bins):
            m = mel(bin_width * i)
            if left < m < right:
                want[b, i] = (m - left) / (center - left) if m <= center else (right - m) / (right - center)
    got_product = arena.kaldi_mel_banks(num_bins, padded, sr).numpy().astype(np.float64)
    got_oracle = get_mel_banks(num_bins, padded, sr, low, 0.0, 100.0, -500.0, 1.0)[0].numpy().astype(np.float64)
    assert np.abs(got_product - want).max() < 5e-5 and np.abs(got_oracle - want).max() < 5e-5      # float32 mel arithmetic (mel ~ 2840, slope 1 / 34) vs double
    assert ((want > 1e-4) == (got_product > 1e-4)).mean() > 0.9995      # same support up to float32 ties at a triangle's foot
