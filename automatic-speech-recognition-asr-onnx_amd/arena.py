"""Checkpoint -> engine weight arena (the offline "export" step of this engine).

The reference bakes its export-time weight folds and constant tables into the
.onnx files (`Export_*.py`); this module performs the same folds and writes one
flat arena that `libasr_mi355x.so` maps in HBM (layout: csrc/engine.h).

SenseVoice folds follow SenseVoice/Export_SenseVoice.py:
  :139-155  Kaldi fbank as one folded DFT matrix (DC removal, pre-emphasis, Hamming)
  :157-160  Kaldi mel banks (+ zero Nyquist column)
  :170-206  prompt embeddings / sinusoidal positions (f16-rounded), position folded into prompts
  :208-220  d_k^-1/4 into q,k rows; FSMN identity tap; linear_out.bias moved into the FSMN conv
  :361-364  embed.weight and cmvn_vars scaled by sqrt(d_model)
Constant tables are additionally re-ordered into MFMA fragment order for the HIP kernels.
"""
from __future__ import annotations

import math
import struct

import numpy as np
import torch

from .config import SenseVoiceConfig

DT_F32, DT_BF16, DT_I32, DT_F16 = 0, 1, 2, 3
PRECISION_BF16, PRECISION_F32 = 0, 1


def _to_bf16_bits(a: np.ndarray) -> np.ndarray:
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16)
    return t.view(torch.int16).numpy().view(np.uint16)


class ArenaWriter:
    def __init__(self):
        self._items = []

    def add(self, name: str, array: np.ndarray, dtype: int):
        assert len(name) < 80, name
        a = np.ascontiguousarray(array)
        if dtype == DT_F32:
            a = a.astype(np.float32, copy=False)
        elif dtype == DT_BF16:
            shape = a.shape
            a = _to_bf16_bits(a).reshape(shape)
        elif dtype == DT_I32:
            a = a.astype(np.int32, copy=False)
        elif dtype == DT_F16:
            a = a.astype(np.float16, copy=False)
        assert a.ndim <= 4
        self._items.append((name, a, dtype))

    def weight(self, name: str, array: np.ndarray, precision: int):
        self.add(name, array, DT_BF16 if precision == PRECISION_BF16 else DT_F32)

    def finish(self) -> np.ndarray:
        n = len(self._items)
        data_off = (32 + 128 * n + 255) // 256 * 256
        offs, cur = [], data_off
        for _, a, _ in self._items:
            offs.append(cur)
            cur = (cur + a.nbytes + 255) // 256 * 256
        total = cur
        blob = np.zeros(total, dtype=np.uint8)
        blob[:32] = np.frombuffer(struct.pack("<8sIIQQ", b"ASRARENA", 1, n, data_off, total), dtype=np.uint8)
        for i, ((name, a, dtype), off) in enumerate(zip(self._items, offs)):
            shape = list(a.shape) + [0] * (4 - a.ndim)
            rec = struct.pack("<80sII4qQ", name.encode(), dtype, a.ndim, *shape, off)
            blob[32 + 128 * i: 32 + 128 * (i + 1)] = np.frombuffer(rec, dtype=np.uint8)
            blob[off: off + a.nbytes] = a.view(np.uint8).reshape(-1)
        return blob


# --------------------------------------------------------------------------- Kaldi mel banks
def kaldi_mel_banks(num_bins: int, padded_window: int, sample_freq: float, low_freq: float = 20.0,
                    high_freq: float = 0.0) -> torch.Tensor:
    """Kaldi `MelBanks` triangular filters over the first padded_window/2 FFT bins, float32.

    Stands in for `torchaudio.compliance.kaldi.get_mel_banks(n, nfft, sr, 20., 0., 100., -500., 1.)`
    (called at SenseVoice/Export_SenseVoice.py:159; torchaudio is a third-party dependency that is
    not vendored in the reference): mel(f) = 1127 ln(1 + f/700), triangles equally spaced in mel
    between low_freq and Nyquist + high_freq, no VTLN warp.
    """
    n_fft_bins = padded_window // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    bin_width = sample_freq / padded_window
    mel_lo = 1127.0 * math.log(1.0 + low_freq / 700.0)
    mel_hi = 1127.0 * math.log(1.0 + high_freq / 700.0)
    step = (mel_hi - mel_lo) / (num_bins + 1)
    idx = torch.arange(num_bins).unsqueeze(1)
    left, center, right = mel_lo + idx * step, mel_lo + (idx + 1.0) * step, mel_lo + (idx + 2.0) * step
    mel = (1127.0 * (1.0 + bin_width * torch.arange(n_fft_bins) / 700.0).log()).unsqueeze(0)
    rising = (mel - left) / (center - left)
    falling = (right - mel) / (right - center)
    return torch.max(torch.zeros(1), torch.min(rising, falling))            # (num_bins, n_fft_bins)


def kaldi_fbank_matrix(cfg) -> torch.Tensor:
    """(2*(nfft/2+1), win) folded DFT matrix: rows [0, F) real part, [F, 2F) imaginary part."""
    nfreq = cfg.nfft // 2 + 1
    window = torch.hamming_window(cfg.win_length, periodic=False, alpha=0.54, beta=0.46, dtype=torch.float32)
    k = torch.arange(nfreq, dtype=torch.float32).unsqueeze(1)
    n = torch.arange(cfg.win_length, dtype=torch.float32).unsqueeze(0)
    omega = (2.0 * torch.pi / cfg.nfft) * k * n
    c = cfg.pre_emphasis

    def fold(basis):
        nxt = torch.cat([basis[:, 1:], torch.zeros_like(basis[:, :1])], dim=1)
        g = basis - c * nxt                      # pre-emphasis moved onto the basis
        g[:, 0] = g[:, 0] - c * basis[:, 0]      # replicate boundary
        return g - g.mean(dim=1, keepdim=True)   # per-frame DC removal

    return torch.cat([fold(torch.cos(omega) * window), fold(-torch.sin(omega) * window)], dim=0)


def pack_dft_for_mfma(kmat: np.ndarray, nfreq: int, win: int) -> np.ndarray:
    """[n_bin_tiles][re|im][win/16][64 lanes][4]: lane l of k-chunk kc holds, for j=0..3,
    K[row(tile, part, l & 15)][kc*16 + 4*j + (l >> 4)] -- the B fragment of v_mfma_f32_16x16x4_f32."""
    n_tiles = (nfreq + 15) // 16
    padded = np.zeros((2, n_tiles * 16, win), dtype=np.float32)
    padded[0, :nfreq] = kmat[:nfreq]
    padded[1, :nfreq] = kmat[nfreq:]
    lane = np.arange(64)
    out = np.zeros((n_tiles, 2, win // 16, 64, 4), dtype=np.float32)
    for j in range(4):
        kidx = (np.arange(win // 16)[:, None] * 16 + 4 * j + (lane >> 4)[None, :])        # (kc, lane)
        for t in range(n_tiles):
            rows = t * 16 + (lane & 15)                                                   # (lane,)
            for part in range(2):
                out[t, part, :, :, j] = padded[part][rows[None, :], kidx]
    return out.reshape(-1)


def pack_mel_for_mfma(mel_t: np.ndarray, n_mels: int) -> np.ndarray:
    """mel_t: (nfreq, n_mels). [n_mel_tiles][n_bin_tiles][64 lanes][4]: lane l holds
    melT[kc*16 + 4*j + (l >> 4)][nt*16 + (l & 15)]."""
    nfreq = mel_t.shape[0]
    n_bin_tiles = (nfreq + 15) // 16
    padded = np.zeros((n_bin_tiles * 16, n_mels), dtype=np.float32)
    padded[:nfreq] = mel_t
    lane = np.arange(64)
    out = np.zeros((n_mels // 16, n_bin_tiles, 64, 4), dtype=np.float32)
    for nt in range(n_mels // 16):
        for kc in range(n_bin_tiles):
            for j in range(4):
                out[nt, kc, :, j] = padded[kc * 16 + 4 * j + (lane >> 4), nt * 16 + (lane & 15)]
    return out.reshape(-1)


def _pad_cols(a: np.ndarray, cols: int) -> np.ndarray:
    if a.shape[1] == cols:
        return a
    out = np.zeros((a.shape[0], cols), dtype=a.dtype)
    out[:, :a.shape[1]] = a
    return out


def build_sensevoice_arena(cfg: SenseVoiceConfig, ck: dict, precision: int = PRECISION_BF16) -> np.ndarray:
    """Fold a source-layout SenseVoice checkpoint into the engine arena (uint8 blob)."""
    w = ArenaWriter()
    feat, d = cfg.feat_dim, cfg.d_model
    nfreq = cfg.nfft // 2 + 1
    # ---- front-end constants
    kmat = kaldi_fbank_matrix(cfg).numpy()
    w.add("fe.dft", pack_dft_for_mfma(kmat, nfreq, cfg.win_length), DT_F32)
    banks = kaldi_mel_banks(cfg.n_mels, cfg.nfft, float(cfg.sample_rate))
    mel_t = torch.nn.functional.pad(banks, (0, 1), value=0.0).transpose(0, 1).contiguous().numpy()   # (nfreq, n_mels)
    w.add("fe.mel", pack_mel_for_mfma(mel_t, cfg.n_mels), DT_F32)
    factor = float(d) ** 0.5
    w.add("fe.cmvn_means", ck["frontend.cmvn_means"].reshape(feat), DT_F32)
    w.add("fe.cmvn_vars", (torch.from_numpy(ck["frontend.cmvn_vars"]) * factor).numpy().reshape(feat), DT_F32)
    # ---- prompts + positions
    embed = torch.from_numpy(ck["embed.weight"]) * factor
    sys_ids = [1, 2, 14] if cfg.use_emo else [5, 14]
    n_prompt = 1 + len(sys_ids)
    lfr_len = cfg.n_lfr(cfg.max_audio_len)
    positions = torch.arange(1, lfr_len + n_prompt + 1, dtype=torch.float32)
    log_inc = torch.log(torch.tensor([10000.0], dtype=torch.float32)) / (feat / 2 - 1)
    inv_ts = torch.exp(torch.arange(feat / 2, dtype=torch.float32) * (-log_inc)).reshape(1, -1)
    scaled = positions.reshape(-1, 1) * inv_ts
    pos = torch.cat([torch.sin(scaled), torch.cos(scaled)], dim=1).half().float()
    lang = embed[list(cfg.language_prompt_token_ids)].half().float() + pos[:1]
    sysm = embed[sys_ids] + pos[1:n_prompt]
    w.add("fe.language_embed", lang.numpy(), DT_F32)
    w.add("fe.system_embed", sysm.numpy(), DT_F32)
    w.add("fe.speech_pos", pos[n_prompt:].contiguous().numpy(), DT_F32)
    # ---- SANM blocks
    names = ([f"encoder.encoders0.{i}." for i in range(cfg.n_enc0)] + [f"encoder.encoders.{i}." for i in range(cfg.n_enc)]
             + [f"encoder.tp_encoders.{i}." for i in range(cfg.n_tp)])
    scale = np.float32(float(cfg.d_head ** (-0.25)))
    pad = (cfg.fsmn_kernel - 1) // 2
    for i, p in enumerate(names):
        q = f"blk{i}."
        wqkv = ck[p + "self_attn.linear_q_k_v.weight"].copy()
        bqkv = ck[p + "self_attn.linear_q_k_v.bias"].copy()
        wqkv[:-d] *= scale
        bqkv[:-d] *= scale
        in_size = wqkv.shape[1]
        kpad = (in_size + 63) // 64 * 64
        wf = ck[p + "self_attn.fsmn_block.weight"][:, 0, :].copy()
        wf[:, pad] += np.float32(1.0)
        w.add(q + "ln1_g", ck[p + "norm1.weight"], DT_F32)
        w.add(q + "ln1_b", ck[p + "norm1.bias"], DT_F32)
        w.weight(q + "wqkv", _pad_cols(wqkv, kpad), precision)
        w.add(q + "bqkv", bqkv, DT_F32)
        w.add(q + "wfsmn", wf, DT_F32)
        w.add(q + "bfsmn", ck[p + "self_attn.linear_out.bias"], DT_F32)
        w.weight(q + "wout", ck[p + "self_attn.linear_out.weight"], precision)
        w.add(q + "ln2_g", ck[p + "norm2.weight"], DT_F32)
        w.add(q + "ln2_b", ck[p + "norm2.bias"], DT_F32)
        w.weight(q + "w1", ck[p + "feed_forward.w_1.weight"], precision)
        w.add(q + "b1", ck[p + "feed_forward.w_1.bias"], DT_F32)
        w.weight(q + "w2", ck[p + "feed_forward.w_2.weight"], precision)
        w.add(q + "b2", ck[p + "feed_forward.w_2.bias"], DT_F32)
    w.add("after_norm_g", ck["encoder.after_norm.weight"], DT_F32)
    w.add("after_norm_b", ck["encoder.after_norm.bias"], DT_F32)
    w.add("tp_norm_g", ck["encoder.tp_norm.weight"], DT_F32)
    w.add("tp_norm_b", ck["encoder.tp_norm.bias"], DT_F32)
    vpad = (cfg.vocab + 127) // 128 * 128
    cw = np.zeros((vpad, d), dtype=np.float32)
    cw[:cfg.vocab] = ck["ctc.ctc_lo.weight"]
    cb = np.zeros((vpad,), dtype=np.float32)
    cb[:cfg.vocab] = ck["ctc.ctc_lo.bias"]
    w.weight("ctc.w", cw, precision)
    w.add("ctc.b", cb, DT_F32)
    return w.finish()
