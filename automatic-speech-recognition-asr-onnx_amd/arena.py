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
PRECISION_FP8W = 2          # Whisper sessions only: bf16 arena, decoder weights / cross-K/V quantised to e4m3 at session creation (asr_mi355x.h)
PRECISION_MXFP4W = 4        # Whisper sessions: FP8W with the decoder projections as OCP MXFP4 (e2m1 + one e8m0 scale per 32 input channels) instead of e4m3
PRECISION_FP8MM = 3         # FP8W + the encoder's FFN pair on the FP8 matrix pipe (e4m3 weights and activations, v_mfma_scale_f32_16x16x128_f8f6f4)


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
        blob = np.empty(total, dtype=np.uint8)               # gigabytes for the large models: zero only the header and the alignment gaps
        blob[:data_off] = 0
        blob[:32] = np.frombuffer(struct.pack("<8sIIQQ", b"ASRARENA", 1, n, data_off, total), dtype=np.uint8)
        for i, ((name, a, dtype), off) in enumerate(zip(self._items, offs)):
            shape = list(a.shape) + [0] * (4 - a.ndim)
            rec = struct.pack("<80sII4qQ", name.encode(), dtype, a.ndim, *shape, off)
            blob[32 + 128 * i: 32 + 128 * (i + 1)] = np.frombuffer(rec, dtype=np.uint8)
            blob[off: off + a.nbytes] = a.view(np.uint8).reshape(-1)
            end = (off + a.nbytes + 255) // 256 * 256
            blob[off + a.nbytes: end] = 0
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


def _operand_colsum(wmat: np.ndarray, precision: int) -> np.ndarray:
    """c[n] = sum_k W[n][k] over the weights AS THE GEMM SEES THEM (bf16-rounded in bf16 mode), summed in float64.
    With it the engine evaluates LayerNorm algebraically inside the GEMM: LN(x) W^T = rstd (x W^T - mean c)."""
    wm = np.asarray(wmat, dtype=np.float32)
    if precision == PRECISION_BF16:
        wm = torch.from_numpy(wm).to(torch.bfloat16).float().numpy()
    return wm.astype(np.float64).sum(axis=1).astype(np.float32)


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
        w1, b1 = ck[p + "feed_forward.w_1.weight"], ck[p + "feed_forward.w_1.bias"]
        if precision == PRECISION_BF16:
            # performance mode: LayerNorm affines are absorbed into the following Linear (float64, rounded once) so the
            # normalisation itself can be evaluated inside the GEMM from row statistics (column sums below)
            wqkv, bqkv = _fold64(ck[p + "norm1.weight"], ck[p + "norm1.bias"], wqkv, bqkv)
            w1, b1 = _fold64(ck[p + "norm2.weight"], ck[p + "norm2.bias"], w1, b1)
            w.add(q + "cqkv", _operand_colsum(wqkv, precision), DT_F32)
            w.add(q + "c1", _operand_colsum(w1, precision), DT_F32)
        else:
            w.add(q + "ln1_g", ck[p + "norm1.weight"], DT_F32)
            w.add(q + "ln1_b", ck[p + "norm1.bias"], DT_F32)
            w.add(q + "ln2_g", ck[p + "norm2.weight"], DT_F32)
            w.add(q + "ln2_b", ck[p + "norm2.bias"], DT_F32)
        w.weight(q + "wqkv", _pad_cols(wqkv, kpad), precision)
        w.add(q + "bqkv", bqkv, DT_F32)
        w.add(q + "wfsmn", wf, DT_F32)
        w.add(q + "bfsmn", ck[p + "self_attn.linear_out.bias"], DT_F32)
        w.weight(q + "wout", ck[p + "self_attn.linear_out.weight"], precision)
        w.weight(q + "w1", w1, precision)
        w.add(q + "b1", b1, DT_F32)
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


# =========================================================================== Whisper
def slaney_mel_filterbank(n_freqs: int, n_mels: int, sample_rate: int) -> np.ndarray:
    """(n_mels, n_freqs) slaney-scale, slaney-normalised triangular filters over [0, sr/2].

    Stands in for `torchaudio.functional.melscale_fbanks(n_freqs, 0, sr/2, n_mels, sr, "slaney", "slaney")`
    (Whisper/Export_Whisper.py:359-361; third-party, not vendored): mel = f / (200/3) below 1 kHz, logarithmic
    (step ln(6.4)/27) above; each triangle scaled by 2 / (f_right - f_left)."""
    def to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) * (27.0 / np.log(6.4)), f * 3.0 / 200.0)

    def to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), m * 200.0 / 3.0)

    freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    edges = to_hz(np.linspace(to_mel(0.0), to_mel(sample_rate / 2.0), n_mels + 2))
    width = np.diff(edges)
    ramps = edges[None, :] - freqs[:, None]
    tri = np.maximum(0.0, np.minimum(-ramps[:, :-2] / width[:-1], ramps[:, 2:] / width[1:]))
    tri = tri * (2.0 / (edges[2:] - edges[:-2]))[None, :]
    return np.ascontiguousarray(tri.T.astype(np.float32))


def whisper_dft_matrix(nfft: int) -> np.ndarray:
    """(2*(nfft/2+1), nfft): periodic-Hann-windowed cos rows then -sin rows (Whisper/STFT_Process.py:136-150)."""
    bins = nfft // 2 + 1
    t = torch.arange(nfft, dtype=torch.float32).unsqueeze(0)
    f = torch.arange(bins, dtype=torch.float32).unsqueeze(1)
    omega = (2.0 * torch.pi / nfft) * f * t
    win = torch.hann_window(nfft, periodic=True).float().unsqueeze(0)
    return torch.cat([torch.cos(omega) * win, -torch.sin(omega) * win], dim=0).numpy()


def _absorb_ln(gamma, beta, w, b):
    """Linear(gamma * xhat + beta) = (W * gamma) xhat + (W beta + b)   (Export_Whisper.py:215-225)."""
    return w * gamma[None, :], b + w @ beta


def build_whisper_arena(cfg, ck: dict, precision: int = PRECISION_BF16, suppress_tokens=None, begin_suppress_tokens=()) -> np.ndarray:
    """Fold an HF-layout Whisper checkpoint into the engine arena (Export_Whisper.py:376-420,527-550)."""
    w = ArenaWriter()
    d, Ld = cfg.d_model, cfg.n_dec_layers
    bins = cfg.nfft // 2 + 1
    w.add("fe.dft", pack_dft_for_mfma(whisper_dft_matrix(cfg.nfft), bins, cfg.nfft), DT_F32)
    w.add("fe.mel", pack_mel_for_mfma(slaney_mel_filterbank(bins, cfg.n_mels, cfg.sample_rate).T.copy(), cfg.n_mels), DT_F32)
    # conv stem as GEMMs over time-major strided views: weight column index = tap * C_in + c_in
    c1 = ck["model.encoder.conv1.weight"]                       # (d, n_mels, 3)
    c2 = ck["model.encoder.conv2.weight"]                       # (d, d, 3)
    w.weight("enc.conv1_w", np.ascontiguousarray(c1.transpose(0, 2, 1)).reshape(d, 3 * cfg.n_mels), precision)
    w.add("enc.conv1_b", ck["model.encoder.conv1.bias"], DT_F32)
    w.weight("enc.conv2_w", np.ascontiguousarray(c2.transpose(0, 2, 1)).reshape(d, 3 * d), precision)
    w.add("enc.conv2_b", ck["model.encoder.conv2.bias"], DT_F32)
    w.add("enc.pos", ck["model.encoder.embed_positions.weight"], DT_F32)
    scale = np.float32(float(cfg.d_head ** -0.25))
    zero = np.zeros(d, dtype=np.float32)

    def fused_qkv(p, gamma, beta):
        wq = np.concatenate([ck[p + "q_proj.weight"], ck[p + "k_proj.weight"], ck[p + "v_proj.weight"]], 0)     # a fresh array
        bq = np.concatenate([ck[p + "q_proj.bias"], zero, ck[p + "v_proj.bias"]], 0)
        wq[:2 * d] *= scale                                      # d^-1/4 on q and k; k has no bias
        bq[:d] *= scale
        return _absorb_ln(gamma, beta, wq, bq)

    for i in range(cfg.n_enc_layers):
        p, q = f"model.encoder.layers.{i}.", f"enc{i}."
        wq, bq = fused_qkv(p + "self_attn.", ck[p + "self_attn_layer_norm.weight"], ck[p + "self_attn_layer_norm.bias"])
        w1, b1 = _absorb_ln(ck[p + "final_layer_norm.weight"], ck[p + "final_layer_norm.bias"], ck[p + "fc1.weight"], ck[p + "fc1.bias"])
        w.weight(q + "wqkv", wq, precision); w.add(q + "bqkv", bq, DT_F32)
        w.weight(q + "wo", ck[p + "self_attn.out_proj.weight"], precision); w.add(q + "bo", ck[p + "self_attn.out_proj.bias"], DT_F32)
        w.weight(q + "w1", w1, precision); w.add(q + "b1", b1, DT_F32)
        w.weight(q + "w2", ck[p + "fc2.weight"], precision); w.add(q + "b2", ck[p + "fc2.bias"], DT_F32)
    w.add("enc.ln_g", ck["model.encoder.layer_norm.weight"], DT_F32)
    w.add("enc.ln_b", ck["model.encoder.layer_norm.bias"], DT_F32)
    # fused cross-KV: every layer's (scaled) K projection, then every layer's V projection (:393-417)
    kw = [ck[f"model.decoder.layers.{i}.encoder_attn.k_proj.weight"] * scale for i in range(Ld)]
    vw = [ck[f"model.decoder.layers.{i}.encoder_attn.v_proj.weight"] for i in range(Ld)]
    vb = [ck[f"model.decoder.layers.{i}.encoder_attn.v_proj.bias"] for i in range(Ld)]
    w.weight("ckv.w", np.concatenate(kw + vw, 0), precision)
    w.add("ckv.b", np.concatenate([zero] * Ld + vb, 0), DT_F32)
    # decoder
    vpad = (cfg.vocab + 127) // 128 * 128
    emb = np.zeros((vpad, d), dtype=np.float32)
    emb[:cfg.vocab] = ck["model.decoder.embed_tokens.weight"]
    w.weight("dec.embed", emb, precision)                        # token embedding == tied proj_out
    w.add("dec.pos", ck["model.decoder.embed_positions.weight"], DT_F32)
    sup = np.zeros(vpad, dtype=np.float32)
    if suppress_tokens is not None:
        sup[list(suppress_tokens)] = -128.0                      # -128, not -inf (:517-520)
    beg = np.zeros(vpad, dtype=np.float32)
    ids = [int(i) for i in begin_suppress_tokens if 0 <= int(i) < cfg.vocab]
    if ids:
        beg[ids] = -np.inf                                       # BEGIN_SUPPRESS (:228-240)
    w.add("dec.suppress", sup, DT_F32)
    w.add("dec.begin", beg, DT_F32)
    for i in range(Ld):
        p, q = f"model.decoder.layers.{i}.", f"dec{i}."
        wq, bq = fused_qkv(p + "self_attn.", ck[p + "self_attn_layer_norm.weight"], ck[p + "self_attn_layer_norm.bias"])
        wcq, bcq = _absorb_ln(ck[p + "encoder_attn_layer_norm.weight"], ck[p + "encoder_attn_layer_norm.bias"],
                              ck[p + "encoder_attn.q_proj.weight"] * scale, ck[p + "encoder_attn.q_proj.bias"] * scale)
        w1, b1 = _absorb_ln(ck[p + "final_layer_norm.weight"], ck[p + "final_layer_norm.bias"], ck[p + "fc1.weight"], ck[p + "fc1.bias"])
        w.weight(q + "wqkv", wq, precision); w.add(q + "bqkv", bq, DT_F32)
        w.weight(q + "wo", ck[p + "self_attn.out_proj.weight"], precision); w.add(q + "bo", ck[p + "self_attn.out_proj.bias"], DT_F32)
        w.weight(q + "wcq", wcq, precision); w.add(q + "bcq", bcq, DT_F32)
        w.weight(q + "wco", ck[p + "encoder_attn.out_proj.weight"], precision); w.add(q + "bco", ck[p + "encoder_attn.out_proj.bias"], DT_F32)
        w.weight(q + "w1", w1, precision); w.add(q + "b1", b1, DT_F32)
        w.weight(q + "w2", ck[p + "fc2.weight"], precision); w.add(q + "b2", ck[p + "fc2.bias"], DT_F32)
    w.add("dec.ln_g", ck["model.decoder.layer_norm.weight"], DT_F32)
    w.add("dec.ln_b", ck["model.decoder.layer_norm.bias"], DT_F32)
    return w.finish()


# =========================================================================== Paraformer (non-streaming)
def paraformer_fbank_matrix(cfg) -> np.ndarray:
    """(2*(nfft/2+1), win): windowed DFT basis @ (pre-emphasis matrix @ DC-removal matrix) -- the Paraformer exporter's
    formulation of the Kaldi front-end (Export_Paraformer.py:326-343; same operator as SenseVoice's fold, other rounding)."""
    W = cfg.win_length
    window = torch.hamming_window(W, periodic=False, alpha=0.54, beta=0.46)
    k = torch.arange(cfg.nfft // 2 + 1, dtype=torch.float32).unsqueeze(1)
    n = torch.arange(W, dtype=torch.float32).unsqueeze(0)
    omega = (2.0 * torch.pi / cfg.nfft) * k * n
    dc = torch.eye(W) - torch.full((W, W), 1.0 / W)
    prev = torch.zeros(W, W)
    prev[0, 0] = 1.0
    prev[1:, :-1] = torch.eye(W - 1)
    t = (torch.eye(W) - float(cfg.pre_emphasis) * prev) @ dc
    return torch.cat([(torch.cos(omega) * window) @ t, (-torch.sin(omega) * window) @ t], 0).numpy()


def _fold64(norm_w, norm_b, w, b, out_scale=1.0):
    """absorb_layer_norm_affine / fold_linear_output_scale in float64, rounded once (Export_Paraformer.py:214-258)."""
    w64 = np.asarray(w, dtype=np.float64)
    b64 = np.asarray(b, dtype=np.float64) if b is not None else np.zeros(w64.shape[0])
    s = np.asarray(out_scale, dtype=np.float64)
    w64 = w64 * (s[:, None] if s.ndim else s)
    b64 = b64 * s
    if norm_w is not None:
        b64 = b64 + w64 @ np.asarray(norm_b, dtype=np.float64)
        w64 = w64 * np.asarray(norm_w, dtype=np.float64)[None, :]
    return w64.astype(np.float32), b64.astype(np.float32)


def build_paraformer_arena(cfg, ck: dict, precision: int = PRECISION_BF16, streaming: bool = False) -> np.ndarray:
    """streaming=True (Export_Paraformer_Streaming.py): the decoder FSMN is a valid convolution over [10 history | tokens], so its
    identity rides on the LAST tap; `cfg.max_audio_len` sizes the position table (MAX_CONTINUE_STREAMING - 1 rows for 30 s)."""
    w = ArenaWriter()
    d, feat, dd = cfg.d_model, cfg.feat_dim, cfg.d_dec_ffn
    nfreq = cfg.nfft // 2 + 1
    w.add("fe.dft", pack_dft_for_mfma(paraformer_fbank_matrix(cfg), nfreq, cfg.win_length), DT_F32)
    banks = kaldi_mel_banks(cfg.n_mels, cfg.nfft, float(cfg.sample_rate))
    w.add("fe.mel", pack_mel_for_mfma(torch.nn.functional.pad(banks, (0, 1), value=0.0).t().contiguous().numpy(), cfg.n_mels), DT_F32)
    # encoder input: x * vars + (means * vars + positions), the bias built in float64 (:459-465, 580-584)
    vars_ = torch.from_numpy(ck["frontend.cmvn_vars"]) * (float(d) ** 0.5)
    lfr_len = cfg.seq_len(cfg.max_audio_len)
    positions = torch.arange(1, lfr_len + 1, dtype=torch.float32)
    log_inc = torch.log(torch.tensor([10000.0])) / (feat / 2 - 1)
    inv_ts = torch.exp(torch.arange(feat / 2).float() * (-log_inc))
    st = positions.reshape(-1, 1) * inv_ts.reshape(1, -1)
    pos = torch.cat([torch.sin(st), torch.cos(st)], 1)
    bias = (torch.from_numpy(ck["frontend.cmvn_means"]).double().reshape(1, feat) * vars_.double().reshape(1, feat) + pos.double()).float()
    w.add("fe.cmvn_vars", vars_.numpy().reshape(feat), DT_F32)
    w.add("fe.speech_pos", bias.numpy(), DT_F32)
    factor = float(cfg.d_head ** -0.25)
    pad = (cfg.fsmn_kernel - 1) // 2

    def fsmn_w(p, tap=pad):
        wf = ck[p + "self_attn.fsmn_block.weight"][:, 0, :].astype(np.float64)
        wf[:, tap] += 1.0
        return wf.astype(np.float32)

    names = [f"encoder.encoders0.{i}." for i in range(cfg.n_enc0)] + [f"encoder.encoders.{i}." for i in range(cfg.n_enc)]
    for i, p in enumerate(names):
        q = f"blk{i}."
        scale = np.ones(3 * d)
        scale[:-d] = factor
        wqkv, bqkv = _fold64(ck[p + "norm1.weight"], ck[p + "norm1.bias"], ck[p + "self_attn.linear_q_k_v.weight"], ck[p + "self_attn.linear_q_k_v.bias"], scale)
        w1, b1 = _fold64(ck[p + "norm2.weight"], ck[p + "norm2.bias"], ck[p + "feed_forward.w_1.weight"], ck[p + "feed_forward.w_1.bias"])
        kpad = (wqkv.shape[1] + 63) // 64 * 64
        if precision == PRECISION_BF16:
            w.add(q + "cqkv", _operand_colsum(wqkv, precision), DT_F32)
            w.add(q + "c1", _operand_colsum(w1, precision), DT_F32)
        w.weight(q + "wqkv", _pad_cols(wqkv, kpad), precision)
        w.add(q + "bqkv", bqkv, DT_F32)
        w.add(q + "wfsmn", fsmn_w(p), DT_F32)
        w.add(q + "bfsmn", ck[p + "self_attn.linear_out.bias"], DT_F32)       # linear_out.bias rides with the FSMN term of the out-projection
        w.weight(q + "wout", ck[p + "self_attn.linear_out.weight"], precision)
        w.weight(q + "w1", w1, precision)
        w.add(q + "b1", b1, DT_F32)
        w.weight(q + "w2", ck[p + "feed_forward.w_2.weight"], precision)
        w.add(q + "b2", ck[p + "feed_forward.w_2.bias"], DT_F32)
    w.add("after_norm_g", ck["encoder.after_norm.weight"], DT_F32)
    w.add("after_norm_b", ck["encoder.after_norm.bias"], DT_F32)
    # CIF predictor: conv k=3 as a GEMM over [x[t-1] | x[t] | x[t+1]] (column = tap * d + channel)
    cw = ck["predictor.cif_conv1d.weight"]                                     # (d, d, 3)
    w.weight("cif.conv_w", np.ascontiguousarray(cw.transpose(0, 2, 1)).reshape(d, 3 * d), precision)
    w.add("cif.conv_b", ck["predictor.cif_conv1d.bias"], DT_F32)
    w.add("cif.out_w", ck["predictor.cif_output.weight"].reshape(d), DT_F32)
    w.add("cif.out_b", ck["predictor.cif_output.bias"].reshape(1), DT_F32)
    for j in range(cfg.n_dec + cfg.n_dec3):
        full = j < cfg.n_dec
        p = f"decoder.decoders.{j}." if full else f"decoder.decoders3.{j - cfg.n_dec}."
        q = f"dec{j}."
        w1, b1 = _fold64(ck[p + "norm1.weight"], ck[p + "norm1.bias"], ck[p + "feed_forward.w_1.weight"], ck[p + "feed_forward.w_1.bias"])
        w2, b2 = _fold64(ck[p + "feed_forward.norm.weight"], ck[p + "feed_forward.norm.bias"], ck[p + "feed_forward.w_2.weight"], None)
        w.weight(q + "w1", w1, precision); w.add(q + "b1", b1, DT_F32)
        w.weight(q + "w2", w2, precision); w.add(q + "b2", b2, DT_F32)
        if full:
            wq, bq = _fold64(ck[p + "norm3.weight"], ck[p + "norm3.bias"], ck[p + "src_attn.linear_q.weight"], ck[p + "src_attn.linear_q.bias"], factor)
            kv_scale = np.ones(2 * d)
            kv_scale[:d] = factor
            wkv, bkv = _fold64(None, None, ck[p + "src_attn.linear_k_v.weight"], ck[p + "src_attn.linear_k_v.bias"], kv_scale)
            w.add(q + "n2_g", ck[p + "norm2.weight"], DT_F32); w.add(q + "n2_b", ck[p + "norm2.bias"], DT_F32)
            w.add(q + "wfsmn", fsmn_w(p, cfg.fsmn_kernel - 1 if streaming else pad), DT_F32)
            w.weight(q + "wq", wq, precision); w.add(q + "bq", bq, DT_F32)
            w.weight(q + "wkv", wkv, precision); w.add(q + "bkv", bkv, DT_F32)
            w.weight(q + "wo", ck[p + "src_attn.linear_out.weight"], precision); w.add(q + "bo", ck[p + "src_attn.linear_out.bias"], DT_F32)
    wo, bo = _fold64(ck["decoder.after_norm.weight"], ck["decoder.after_norm.bias"], ck["decoder.output_layer.weight"], ck["decoder.output_layer.bias"])
    vpad = (cfg.vocab + 127) // 128 * 128
    ow = np.zeros((vpad, d), np.float32); ow[:cfg.vocab] = wo
    ob = np.zeros((vpad,), np.float32); ob[:cfg.vocab] = bo
    w.weight("out.w", ow, precision)
    w.add("out.b", ob, DT_F32)
    return w.finish()


# =========================================================================== Qwen3-ASR
def build_qwen_asr_arena(cfg, ck: dict, precision: int = PRECISION_BF16) -> np.ndarray:
    """Fold an HF-layout Qwen3-ASR checkpoint (thinker.audio_tower.* / thinker.model.* / thinker.lm_head.weight) into the engine
    arena. Folds follow the exporter: encoder LayerNorm affines into q|k|v / fc1 / proj1 and d^-1/4 on q and k
    (Export_Qwen_ASR.py:381-398); decoder RMSNorm weights into q|k|v and gate|up, d^-1/4 into the q / k norm weights (:1141-1190).
    The three stride-2 Conv2d become GEMMs over channel-last 3 x 3 patches: weight column = (kh * 3 + kw) * Cpad + c_in with
    channels zero-padded to Cpad = 128-multiple; conv_out's columns are re-ordered from (c, f) to (f, c)."""
    w = ArenaWriter()
    a, t = "thinker.audio_tower.", "thinker.model."
    bins = cfg.nfft // 2 + 1
    w.add("fe.dft", pack_dft_for_mfma(whisper_dft_matrix(cfg.nfft), bins, cfg.nfft), DT_F32)
    w.add("fe.mel", pack_mel_for_mfma(slaney_mel_filterbank(bins, cfg.n_mels, cfg.sample_rate).T.copy(), cfg.n_mels), DT_F32)
    C, de, d = cfg.conv_channels, cfg.enc_d, cfg.d_model
    cpad = (C + 127) // 128 * 128
    f32 = lambda x: np.asarray(x, dtype=np.float32)

    c1 = np.zeros((cpad, 64), dtype=np.float32)
    c1[:C, :9] = f32(ck[a + "conv2d1.weight"]).reshape(C, 9)
    w.weight("enc.conv1_w", c1, precision)
    for name, key in (("enc.conv2_w", "conv2d2"), ("enc.conv3_w", "conv2d3")):
        cw = np.zeros((cpad, 3, 3, cpad), dtype=np.float32)
        cw[:C, :, :, :C] = f32(ck[a + key + ".weight"]).transpose(0, 2, 3, 1)
        w.weight(name, cw.reshape(cpad, 9 * cpad), precision)
    for name, key in (("enc.conv1_b", "conv2d1"), ("enc.conv2_b", "conv2d2"), ("enc.conv3_b", "conv2d3")):
        b = np.zeros(cpad, dtype=np.float32)
        b[:C] = f32(ck[a + key + ".bias"])
        w.add(name, b, DT_F32)
    n_f = f32(ck[a + "conv_out.weight"]).shape[1] // C                        # 16 frequency rows left after the stem
    co = np.zeros((de, n_f, cpad), dtype=np.float32)
    co[:, :, :C] = f32(ck[a + "conv_out.weight"]).reshape(de, C, n_f).transpose(0, 2, 1)
    w.weight("enc.conv_out_w", co.reshape(de, n_f * cpad), precision)
    half = de // 2
    inv = np.exp(-(np.log(10000.0) / (half - 1)) * np.arange(half, dtype=np.float32)).astype(np.float32)
    st = np.arange(13, dtype=np.float32)[:, None] * inv[None, :]               # SinusoidsPositionEmbedding rows 0..12 (:869-872)
    w.add("enc.pos", np.concatenate([np.sin(st), np.cos(st)], 1).astype(np.float32), DT_F32)

    s = np.float64((de // cfg.enc_heads) ** -0.25)
    for i in range(cfg.n_enc_layers):
        p, q = f"{a}layers.{i}.", f"enc{i}."
        wq = np.concatenate([ck[p + "self_attn.q_proj.weight"], ck[p + "self_attn.k_proj.weight"], ck[p + "self_attn.v_proj.weight"]], 0).astype(np.float64)
        bq = np.concatenate([ck[p + "self_attn.q_proj.bias"], ck[p + "self_attn.k_proj.bias"], ck[p + "self_attn.v_proj.bias"]], 0).astype(np.float64)
        g, be = ck[p + "self_attn_layer_norm.weight"].astype(np.float64), ck[p + "self_attn_layer_norm.bias"].astype(np.float64)
        bq = bq + wq @ be
        wq = wq * g[None, :]
        wq[:2 * de] *= s
        bq[:2 * de] *= s
        g2, be2 = ck[p + "final_layer_norm.weight"].astype(np.float64), ck[p + "final_layer_norm.bias"].astype(np.float64)
        w1 = ck[p + "fc1.weight"].astype(np.float64)
        b1 = ck[p + "fc1.bias"].astype(np.float64) + w1 @ be2
        w1 = w1 * g2[None, :]
        w.weight(q + "wqkv", f32(wq), precision); w.add(q + "bqkv", f32(bq), DT_F32)
        w.weight(q + "wo", f32(ck[p + "self_attn.out_proj.weight"]), precision); w.add(q + "bo", f32(ck[p + "self_attn.out_proj.bias"]), DT_F32)
        w.weight(q + "w1", f32(w1), precision); w.add(q + "b1", f32(b1), DT_F32)
        w.weight(q + "w2", f32(ck[p + "fc2.weight"]), precision); w.add(q + "b2", f32(ck[p + "fc2.bias"]), DT_F32)
    gp, bp = ck[a + "ln_post.weight"].astype(np.float64), ck[a + "ln_post.bias"].astype(np.float64)
    wp = ck[a + "proj1.weight"].astype(np.float64)
    w.weight("enc.proj1_w", f32(wp * gp[None, :]), precision)
    w.add("enc.proj1_b", f32(ck[a + "proj1.bias"].astype(np.float64) + wp @ bp), DT_F32)
    w.weight("enc.proj2_w", f32(ck[a + "proj2.weight"]), precision)
    w.add("enc.proj2_b", f32(ck[a + "proj2.bias"]), DT_F32)

    vpad = (cfg.vocab + 127) // 128 * 128
    emb = np.zeros((vpad, d), dtype=np.float32)
    emb[:cfg.vocab] = ck[t + "embed_tokens.weight"]
    w.weight("dec.embed", emb, precision)
    head = np.zeros((vpad, d), dtype=np.float32)
    head[:cfg.vocab] = ck["thinker.lm_head.weight"]
    w.weight("dec.lm_head", head, precision)
    # rotary table [position][cos | sin] in f32, like the exporter's precomputed cos / sin buffers (:933-960)
    inv_freq = (1.0 / (cfg.rope_theta ** (np.arange(0, cfg.d_head, 2, dtype=np.float32) / cfg.d_head))).astype(np.float32)
    theta = np.arange(cfg.max_seq_len, dtype=np.float32)[:, None] * inv_freq[None, :]
    w.add("dec.rope", np.concatenate([np.cos(theta), np.sin(theta)], 1).astype(np.float32), DT_F32)
    w.add("dec.final_norm", f32(ck[t + "norm.weight"]), DT_F32)
    sc = np.float32(float(cfg.d_head ** -0.25))
    for i in range(cfg.n_layers):
        p, q = f"{t}layers.{i}.", f"dec{i}."
        wqkv = np.concatenate([ck[p + "self_attn.q_proj.weight"], ck[p + "self_attn.k_proj.weight"], ck[p + "self_attn.v_proj.weight"]], 0)
        w.weight(q + "wqkv", f32(wqkv * ck[p + "input_layernorm.weight"][None, :]), precision)
        w.weight(q + "wo", f32(ck[p + "self_attn.o_proj.weight"]), precision)
        # rows interleaved (gate_0, up_0, gate_1, up_1, ...): the GEMM's SwiGLU epilogue finds each pair in one lane
        gu = np.stack([ck[p + "mlp.gate_proj.weight"], ck[p + "mlp.up_proj.weight"]], 1).reshape(2 * cfg.d_ffn, d)
        w.weight(q + "gate_up", f32(gu * ck[p + "post_attention_layernorm.weight"][None, :]), precision)
        w.weight(q + "down", f32(ck[p + "mlp.down_proj.weight"]), precision)
        w.add(q + "qn", f32(ck[p + "self_attn.q_norm.weight"] * sc), DT_F32)
        w.add(q + "kn", f32(ck[p + "self_attn.k_norm.weight"] * sc), DT_F32)
    return w.finish()
