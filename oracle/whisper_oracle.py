"""CPU oracle for the Whisper hot path (STFT/log-mel -> encoder + fused cross-KV -> KV-cache decoder -> heads).

TEST INFRASTRUCTURE. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

torch-CPU f32 (optionally f64) restatement of the arithmetic the reference freezes into its Whisper graphs,
consuming the RAW HF-layout checkpoint (checkpoints.py) and applying the export-time folds itself:
    Whisper/STFT_Process.py:136-150,224-246   periodic Hann, 400-pt DFT as a matrix, reflect pad 200 / 40
                                              (drop_last_frame), power = re^2 + im^2
    Whisper/Export_Whisper.py:215-225         LayerNorm affine absorbed into the following Linear
    :376-420                                  encoder q/k/v fusion (d^-1/4 on q,k), fused cross-KV Linear (K scaled)
    :422-447                                  encoder forward: mel, log10, global-max clamp, conv stem, layers, cross-KV
    :450-497                                  embed / position / -128 causal mask shells
    :527-550                                  decoder fusion (qkv, cross q scale + LN absorb, fc1 LN absorb)
    :614-667                                  decoder forward incl. tied proj_out and -128 suppress penalty
    :228-260,334-348                          BEGIN_SUPPRESS (-inf), ARGMAX, NO_SPEECH_DETECTION
KV tensors are kept in f32 (the reference's USE_FP16_KV=False configuration = the north-star's "ONNX CPU f32").
The slaney mel filterbank of the third-party `torchaudio.functional.melscale_fbanks` (absent here, unpinned) is
restated from the published (librosa / Slaney Auditory Toolbox) formula and cross-checked against
`transformers.audio_utils.mel_filter_bank` in tests.

Pinned against the real reference classes: tests/golden/whisper_*.npz (oracle/gen_golden_whisper.py).
"""
from __future__ import annotations

import math

import numpy as np
import torch

F = torch.nn.functional


def slaney_mel_filterbank(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """(n_freqs, n_mels) triangular filters, slaney mel scale + slaney area normalisation."""
    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        lin = f / (200.0 / 3)
        log = 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) * (27.0 / np.log(6.4))
        return np.where(f >= 1000.0, log, lin)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        lin = m * (200.0 / 3)
        log = 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0))
        return np.where(m >= 15.0, log, lin)

    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    m_pts = np.linspace(hz_to_mel(f_min), hz_to_mel(f_max), n_mels + 2)
    f_pts = mel_to_hz(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])
    return torch.from_numpy((fb * enorm[None, :]).astype(np.float32))


def _gelu(x, kind):
    return F.gelu(x, approximate="tanh") if kind == "tanh" else F.gelu(x)


class WhisperOracle:
    def __init__(self, cfg, ck: dict, suppress_tokens=None, begin_suppress_tokens=(), gelu: str = "erf", dtype=torch.float32):
        self.cfg, self.dtype, self.gelu = cfg, dtype, gelu
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
        self.ck = {k: t(v) for k, v in ck.items()}
        self.suppress_tokens = list(suppress_tokens) if suppress_tokens is not None else None
        self.begin_suppress_tokens = [int(i) for i in begin_suppress_tokens if 0 <= int(i) < cfg.vocab]
        self._build_frontend()
        self._fold_encoder()
        self._fold_decoder()

    def _c(self, x):
        return x.to(self.dtype)

    # ---- STFT_Process.py:136-150 + Export_Whisper.py:357-362
    def _build_frontend(self):
        c = self.cfg
        n_fft, bins = c.nfft, c.nfft // 2 + 1
        t = torch.arange(n_fft, dtype=torch.float32).unsqueeze(0)
        f = torch.arange(bins, dtype=torch.float32).unsqueeze(1)
        omega = (2.0 * torch.pi / n_fft) * f * t
        window = torch.hann_window(n_fft, periodic=True).float()
        self.dft = torch.cat([torch.cos(omega) * window.unsqueeze(0), -torch.sin(omega) * window.unsqueeze(0)], dim=0)  # (2*bins, n_fft)
        self.bins = bins
        self.mel_fb = slaney_mel_filterbank(bins, 0.0, c.sample_rate // 2, c.n_mels, c.sample_rate).t().contiguous()   # (n_mels, bins)

    @staticmethod
    def _absorb(gamma, beta, w, b):
        b = b + w @ beta
        return w * gamma.unsqueeze(0), b

    def _qkv(self, p, scale, gamma, beta):
        d = self.cfg.d_model
        z = torch.zeros(d)
        w = torch.cat([self.ck[p + "q_proj.weight"], self.ck[p + "k_proj.weight"], self.ck[p + "v_proj.weight"]], 0).clone()
        b = torch.cat([self.ck[p + "q_proj.bias"], z, self.ck[p + "v_proj.bias"]], 0).clone()
        w[:2 * d] *= scale
        b[:d] *= scale
        return self._absorb(gamma, beta, w, b)

    # ---- Export_Whisper.py:376-420
    def _fold_encoder(self):
        c, ck = self.cfg, self.ck
        scale = float(c.d_head ** -0.25)
        self.enc = []
        for i in range(c.n_enc_layers):
            p = f"model.encoder.layers.{i}."
            wqkv, bqkv = self._qkv(p + "self_attn.", scale, ck[p + "self_attn_layer_norm.weight"], ck[p + "self_attn_layer_norm.bias"])
            w1, b1 = self._absorb(ck[p + "final_layer_norm.weight"], ck[p + "final_layer_norm.bias"], ck[p + "fc1.weight"].clone(), ck[p + "fc1.bias"].clone())
            self.enc.append(dict(wqkv=wqkv, bqkv=bqkv, wo=ck[p + "self_attn.out_proj.weight"], bo=ck[p + "self_attn.out_proj.bias"],
                                 w1=w1, b1=b1, w2=ck[p + "fc2.weight"], b2=ck[p + "fc2.bias"]))
        kw, kb, vw, vb = [], [], [], []
        for i in range(c.n_dec_layers):
            p = f"model.decoder.layers.{i}.encoder_attn."
            kw.append(ck[p + "k_proj.weight"] * scale)
            kb.append(torch.zeros(c.d_model) * scale)
            vw.append(ck[p + "v_proj.weight"])
            vb.append(ck[p + "v_proj.bias"])
        self.w_ckv = torch.cat(kw + vw, 0)
        self.b_ckv = torch.cat(kb + vb, 0)

    # ---- Export_Whisper.py:527-550
    def _fold_decoder(self):
        c, ck = self.cfg, self.ck
        scale = float(c.d_head ** -0.25)
        self.dec = []
        for i in range(c.n_dec_layers):
            p = f"model.decoder.layers.{i}."
            wqkv, bqkv = self._qkv(p + "self_attn.", scale, ck[p + "self_attn_layer_norm.weight"], ck[p + "self_attn_layer_norm.bias"])
            wcq, bcq = self._absorb(ck[p + "encoder_attn_layer_norm.weight"], ck[p + "encoder_attn_layer_norm.bias"],
                                    ck[p + "encoder_attn.q_proj.weight"] * scale, ck[p + "encoder_attn.q_proj.bias"] * scale)
            w1, b1 = self._absorb(ck[p + "final_layer_norm.weight"], ck[p + "final_layer_norm.bias"], ck[p + "fc1.weight"].clone(), ck[p + "fc1.bias"].clone())
            self.dec.append(dict(wqkv=wqkv, bqkv=bqkv, wo=ck[p + "self_attn.out_proj.weight"], bo=ck[p + "self_attn.out_proj.bias"],
                                 wcq=wcq, bcq=bcq, wco=ck[p + "encoder_attn.out_proj.weight"], bco=ck[p + "encoder_attn.out_proj.bias"],
                                 w1=w1, b1=b1, w2=ck[p + "fc2.weight"], b2=ck[p + "fc2.bias"]))
        pen = torch.zeros(c.vocab)
        if self.suppress_tokens is not None:
            pen[self.suppress_tokens] = -128.0
        self.suppress_penalty = pen
        bs = torch.zeros(c.vocab)
        if self.begin_suppress_tokens:
            bs[self.begin_suppress_tokens] = float("-inf")
        self.begin_bias = bs

    # ---- STFT_Process.py:224-246 + Export_Whisper.py:424-427
    def log_mel(self, audio_1d: torch.Tensor):
        c = self.cfg
        x = audio_1d.to(self.dtype).reshape(1, 1, -1)
        half, right = c.nfft // 2, c.nfft // 2 - c.hop_length
        left = x[..., 1:half + 1].flip(2)
        rgt = x[..., -(right + 1):-1].flip(2)
        xp = torch.cat([left, x, rgt], dim=2)[0, 0]
        frames = xp.unfold(0, c.nfft, c.hop_length)                       # (T, n_fft)
        packed = frames @ self._c(self.dft).t()                            # (T, 2*bins)
        sq = packed * packed
        power = (sq[:, :self.bins] + sq[:, self.bins:]).t()                # (bins, T)
        mel = (self._c(self.mel_fb) @ power).clamp(min=1e-10).log10()
        mel = torch.maximum(mel, mel.max() - 8.0)
        return (mel + 4.0) * 0.25                                          # (n_mels, T)

    def encode(self, audio_1d, taps=None):
        c, ck = self.cfg, self.ck
        mel = self.log_mel(torch.as_tensor(np.asarray(audio_1d, dtype=np.float32).reshape(-1)))
        x = _gelu(F.conv1d(mel.unsqueeze(0), self._c(ck["model.encoder.conv1.weight"]), self._c(ck["model.encoder.conv1.bias"]), padding=1), self.gelu)
        x = _gelu(F.conv1d(x, self._c(ck["model.encoder.conv2.weight"]), self._c(ck["model.encoder.conv2.bias"]), stride=2, padding=1), self.gelu)
        x = x[0].t()                                                       # (T, d)
        T, d, H, hd = x.shape[0], c.d_model, c.n_heads, c.d_head
        x = x + self._c(ck["model.encoder.embed_positions.weight"][:T])
        if taps is not None:
            taps["mel"], taps["stem"] = mel, x
        for L in self.enc:
            hn = F.layer_norm(x, (d,))
            qkv = hn @ self._c(L["wqkv"]).t() + self._c(L["bqkv"])
            q, k, v = [z.reshape(T, H, hd).transpose(0, 1) for z in qkv.split(d, dim=1)]
            a = (torch.softmax(q @ k.transpose(1, 2), dim=-1) @ v).transpose(0, 1).reshape(T, d)
            x = a @ self._c(L["wo"]).t() + self._c(L["bo"]) + x
            hn = F.layer_norm(x, (d,))
            x = x + _gelu(hn @ self._c(L["w1"]).t() + self._c(L["b1"]), self.gelu) @ self._c(L["w2"]).t() + self._c(L["b2"])
        x = F.layer_norm(x, (d,), self._c(ck["model.encoder.layer_norm.weight"]), self._c(ck["model.encoder.layer_norm.bias"]))
        ckv = x @ self._c(self.w_ckv).t() + self._c(self.b_ckv)            # (T, 2*Ld*d)
        Ld = c.n_dec_layers
        keys = ckv[:, :Ld * d].reshape(T, Ld, H, hd).permute(1, 2, 0, 3)   # (Ld, H, T, hd)  (scaled by d^-1/4)
        vals = ckv[:, Ld * d:].reshape(T, Ld, H, hd).permute(1, 2, 0, 3)
        if taps is not None:
            taps["enc_out"] = x
        return keys, vals

    # ---- Export_Whisper.py:450-497,614-667
    def decoder(self, ids, hist: int, self_k, self_v, cross_k, cross_v):
        """ids: (B, n) int; self_k/self_v: lists per layer of (B, H, S, hd) or None; cross: (B, Ld, H, T, hd).
        Returns (logits (B, V) with the suppress penalty, new self_k, new self_v)."""
        c, ck = self.cfg, self.ck
        ids = torch.as_tensor(ids, dtype=torch.long)
        B, n = ids.shape
        d, H, hd = c.d_model, c.n_heads, c.d_head
        h = self._c(ck["model.decoder.embed_tokens.weight"])[ids] + self._c(ck["model.decoder.embed_positions.weight"][hist:hist + n])
        S = hist + n
        mask = torch.triu(torch.full((n, S), -128.0, dtype=self.dtype), diagonal=1 + hist)      # (:468-480): row i sees cols <= hist + i
        nk, nv = [], []
        for li, L in enumerate(self.dec):
            hn = F.layer_norm(h, (d,))
            qkv = hn @ self._c(L["wqkv"]).t() + self._c(L["bqkv"])
            q, k, v = [z.reshape(B, n, H, hd).transpose(1, 2) for z in qkv.split(d, dim=-1)]
            if self_k is not None and self_k[li] is not None:
                k = torch.cat([self_k[li], k], dim=2)
                v = torch.cat([self_v[li], v], dim=2)
            nk.append(k)
            nv.append(v)
            a = (torch.softmax(q @ k.transpose(2, 3) + mask, dim=-1) @ v).transpose(1, 2).reshape(B, n, d)
            h = a @ self._c(L["wo"]).t() + self._c(L["bo"]) + h
            hn = F.layer_norm(h, (d,))
            cq = (hn @ self._c(L["wcq"]).t() + self._c(L["bcq"])).reshape(B, n, H, hd).transpose(1, 2)
            a = (torch.softmax(cq @ cross_k[:, li].transpose(2, 3), dim=-1) @ cross_v[:, li]).transpose(1, 2).reshape(B, n, d)
            h = a @ self._c(L["wco"]).t() + self._c(L["bco"]) + h
            hn = F.layer_norm(h, (d,))
            h = h + _gelu(hn @ self._c(L["w1"]).t() + self._c(L["b1"]), self.gelu) @ self._c(L["w2"]).t() + self._c(L["b2"])
        last = F.layer_norm(h[:, -1], (d,), self._c(ck["model.decoder.layer_norm.weight"]), self._c(ck["model.decoder.layer_norm.bias"]))
        self.last_hidden = last                                    # test tap: the rows the tied projection sees (B, d)
        logits = last @ self._c(ck["model.decoder.embed_tokens.weight"]).t() + self._c(self.suppress_penalty)
        return logits, nk, nv

    def no_speech_prob(self, logits):
        """NO_SPEECH_DETECTION (:334-348): softmax over logits with the -128 suppress penalty removed."""
        unsup = -self.suppress_penalty
        return torch.softmax(logits + self._c(unsup), dim=-1)[:, self.cfg.no_speech_id]

    @staticmethod
    def sample_head(logits, previous_ids, noise, temperature, top_k, top_p, repetition_penalty):
        """TOPK_TOPP_SAMPLING (Export_Whisper.py:263-308) with the uniforms given: repetition penalty on every previous id
        (negative logits multiplied, others divided), temperature, top-k, soft-max + exclusive-cumsum top-p cut, Gumbel-max."""
        x = logits.clone()
        if len(previous_ids):
            idx = torch.tensor(list(previous_ids), dtype=torch.long)
            pv = logits[idx]
            x[idx] = torch.where(pv < 0, pv * repetition_penalty, pv / repetition_penalty)
        x = x * (1.0 / temperature)
        vals, inds = torch.topk(x, top_k)
        probs = torch.softmax(vals, dim=-1)
        keep = (torch.cumsum(probs, dim=-1) - probs) <= top_p
        vals = torch.where(keep, vals, torch.tensor(float("-inf")))
        u = torch.clamp(torch.as_tensor(noise, dtype=torch.float32), 1.0e-7, 1.0 - 1.0e-7)
        return int(inds[int(torch.argmax(vals - torch.log(-torch.log(u))))])

    def sample(self, audio, prompt, noise, temperature, top_k, top_p, repetition_penalty):
        """One utterance, len(noise) sampled tokens; noise[step] holds the top_k uniforms of that step."""
        with torch.inference_mode():
            ck_, cv_ = (t.unsqueeze(0) for t in self.encode(audio))
            ids = torch.tensor([list(prompt)], dtype=torch.long)
            logits, sk, sv = self.decoder(ids, 0, None, None, ck_, cv_)
            toks, hist = [], ids.shape[1]
            head = logits[0] + self._c(self.begin_bias)
            for step in range(len(noise)):
                if step:
                    logits, sk, sv = self.decoder(torch.tensor([[toks[-1]]]), hist, sk, sv, ck_, cv_)
                    hist += 1
                    head = logits[0]
                toks.append(self.sample_head(head.float(), toks, noise[step], temperature, top_k, top_p, repetition_penalty))
        return np.asarray(toks, dtype=np.int32)

    def greedy(self, audios, prompt_ids, n_new: int, eos_id=None, repeat_penalty: float = 1.0, penalty_range: int = 20):
        """Batch of utterances (list of 1-D arrays) -> dict of per-step logits / ids, following the reference host
        loop: prefill(prompt) -> arg-max(logits + begin_suppress) -> decode steps with plain arg-max, or -- repeat_penalty != 1,
        the host's default -- penalty-greedy: APPLY_PENALTY (Export_Whisper.py:312-325) multiplies the logits of the last
        `penalty_range` generated ids by the penalty once `penalty_range` tokens exist (Inference_Whisper_ONNX.py:606-632)."""
        with torch.inference_mode():
            enc = [self.encode(a) for a in audios]
            out_ids, out_logits = [], []
            for (ck_, cv_), prompt in zip(enc, prompt_ids):
                ck_, cv_ = ck_.unsqueeze(0), cv_.unsqueeze(0)
                ids = torch.tensor([list(prompt)], dtype=torch.long)
                logits, sk, sv = self.decoder(ids, 0, None, None, ck_, cv_)
                steps_logits = [logits[0]]
                tok = int(torch.argmax(logits[0] + self._c(self.begin_bias)))
                toks = [tok]
                hist = ids.shape[1]
                while len(toks) < n_new and (eos_id is None or tok != eos_id):
                    logits, sk, sv = self.decoder(torch.tensor([[tok]]), hist, sk, sv, ck_, cv_)
                    hist += 1
                    head = logits[0].clone()
                    if repeat_penalty != 1.0 and len(toks) >= penalty_range:
                        idx = torch.tensor(toks[-penalty_range:], dtype=torch.long)
                        head[idx] = logits[0][idx] * repeat_penalty
                    steps_logits.append(head)
                    tok = int(torch.argmax(head))
                    toks.append(tok)
                out_ids.append(np.asarray(toks, dtype=np.int32))
                out_logits.append(torch.stack(steps_logits).float().numpy())
        return dict(token_ids=out_ids, logits=out_logits, cross=[(k.float().numpy(), v.float().numpy()) for k, v in enc])
