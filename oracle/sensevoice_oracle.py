"""CPU oracle for the SenseVoiceSmall hot path (front-end -> 70 SANM blocks -> CTC).

TEST INFRASTRUCTURE. Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this; the product package never does.

A torch-CPU restatement (f32 by default, f64 for error attribution) of the
arithmetic the reference freezes into `SenseVoiceSmall.onnx`:
    SenseVoice/Export_SenseVoice.py:139-155  Kaldi fbank kernel (DC removal, pre-emphasis,
                                              Hamming, 512-pt one-sided DFT folded in one matrix)
    :157-160                                 Kaldi mel banks (+ zero Nyquist column)
    :162-206                                 LFR indices, prompt embeddings, sinusoidal positions
    :208-220                                 export-time SANM folds (scale split, FSMN identity tap,
                                              linear_out bias moved into the FSMN conv)
    :227-258                                 sanm_block
    :260-269                                 encode (after_norm after encoders, tp_norm at the end)
    :271-296                                 forward incl. circular CTC collapse
    :361-364                                 embed.weight and cmvn_vars pre-scaled by sqrt(d_model)
It consumes the RAW source-layout checkpoint (checkpoints.py) and applies the
folds itself, independently of the product's converter (arena.py).

Pinned against the real reference classes run in the build container:
tests/golden/sensevoice_*.npz (oracle/gen_golden.py) and
tests/test_oracle_vs_reference.py.
"""
from __future__ import annotations

import numpy as np
import torch

from .kaldi_mel import get_mel_banks

F = torch.nn.functional


class SenseVoiceOracle:
    def __init__(self, cfg, ck: dict, dtype=torch.float32):
        self.cfg = cfg
        self.dtype = dtype
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
        self.ck = {k: t(v) for k, v in ck.items()}
        self._build_frontend()
        self._build_prompts()
        self._fold_blocks()

    # ---- Export_SenseVoice.py:139-160 -------------------------------------------------
    def _build_frontend(self):
        c = self.cfg
        nfreq = c.nfft // 2 + 1
        window = torch.hamming_window(c.win_length, periodic=False, alpha=0.54, beta=0.46, dtype=torch.float32)
        k = torch.arange(nfreq, dtype=torch.float32).unsqueeze(1)
        n = torch.arange(c.win_length, dtype=torch.float32).unsqueeze(0)
        omega = (2.0 * torch.pi / c.nfft) * k * n

        def fold(basis):
            # y = sum_n basis[n] * pf[n], pf[n] = s[n] - c*s[n-1] (pf[0] = (1-c) s[0]), s = x - mean(x)
            nxt = torch.cat([basis[:, 1:], torch.zeros_like(basis[:, :1])], dim=1)
            g = basis - c.pre_emphasis * nxt
            g[:, 0] = g[:, 0] - c.pre_emphasis * basis[:, 0]
            return g - g.mean(dim=1, keepdim=True)

        self.fbank_kernel = torch.cat([fold(torch.cos(omega) * window), fold(-torch.sin(omega) * window)], dim=0)  # (2*nfreq, win)
        banks, _ = get_mel_banks(c.n_mels, c.nfft, float(c.sample_rate), 20.0, 0.0, 100.0, -500.0, 1.0)
        self.mel_filters = F.pad(banks, (0, 1), value=0.0).transpose(0, 1).contiguous()       # (nfreq, n_mels)
        self.log_eps = float(torch.finfo(torch.float32).eps)
        self.nfreq = nfreq

    # ---- Export_SenseVoice.py:170-206, 361-364 ----------------------------------------
    def _build_prompts(self):
        c = self.cfg
        factor = float(c.d_model) ** 0.5
        embed = self.ck["embed.weight"] * factor
        sys_ids = [1, 2, 14] if c.use_emo else [5, 14]
        system_embed = embed[sys_ids]
        language_embed = embed[list(c.language_prompt_token_ids)].half().float()
        feat = c.feat_dim
        lfr_len = c.n_lfr(c.max_audio_len)
        n_prompt = 1 + len(sys_ids)
        positions = torch.arange(1, lfr_len + n_prompt + 1, dtype=torch.float32)
        log_inc = torch.log(torch.tensor([10000.0], dtype=torch.float32)) / (feat / 2 - 1)
        inv_ts = torch.exp(torch.arange(feat / 2, dtype=torch.float32) * (-log_inc)).reshape(1, -1)
        scaled = positions.reshape(-1, 1) * inv_ts
        pos = torch.cat([torch.sin(scaled), torch.cos(scaled)], dim=1).half().float()
        self.language_embed = language_embed + pos[:1]
        self.system_embed = system_embed + pos[1:n_prompt]
        self.speech_position = pos[n_prompt:]
        self.cmvn_means = self.ck["frontend.cmvn_means"].reshape(1, feat)
        self.cmvn_vars = (self.ck["frontend.cmvn_vars"] * factor).reshape(1, feat)

    # ---- Export_SenseVoice.py:208-220 -------------------------------------------------
    def _fold_blocks(self):
        c = self.cfg
        s = float(c.d_head ** (-0.25))
        names = ([f"encoder.encoders0.{i}." for i in range(c.n_enc0)] + [f"encoder.encoders.{i}." for i in range(c.n_enc)]
                 + [f"encoder.tp_encoders.{i}." for i in range(c.n_tp)])
        self.blocks = []
        pad = (c.fsmn_kernel - 1) // 2
        for p in names:
            g = lambda n: self.ck[p + n].clone()
            wqkv, bqkv = g("self_attn.linear_q_k_v.weight"), g("self_attn.linear_q_k_v.bias")
            wqkv[:-c.d_model] *= s
            bqkv[:-c.d_model] *= s
            wf = g("self_attn.fsmn_block.weight")          # (d, 1, k)
            wf[:, 0, pad] += 1.0
            self.blocks.append(dict(
                ln1=(g("norm1.weight"), g("norm1.bias")), wqkv=wqkv, bqkv=bqkv, wfsmn=wf,
                bfsmn=g("self_attn.linear_out.bias"), wout=g("self_attn.linear_out.weight"),
                ln2=(g("norm2.weight"), g("norm2.bias")),
                w1=g("feed_forward.w_1.weight"), b1=g("feed_forward.w_1.bias"),
                w2=g("feed_forward.w_2.weight"), b2=g("feed_forward.w_2.bias"),
                in_size=wqkv.shape[1]))

    def _c(self, x):
        return x.to(self.dtype)

    # ---- Export_SenseVoice.py:275-287 -------------------------------------------------
    def frontend(self, audio_1d: torch.Tensor):
        c = self.cfg
        frames = audio_1d.to(self.dtype).unfold(0, c.win_length, c.hop_length)             # (frames, win) snip_edges
        spec = frames @ self._c(self.fbank_kernel).t()                                     # (frames, 2*nfreq)
        sq = spec * spec
        power = sq[:, :self.nfreq] + sq[:, self.nfreq:]
        mel = (power @ self._c(self.mel_filters)).clamp(min=self.log_eps).log()            # (frames, n_mels)
        return mel

    def lfr_cmvn(self, mel: torch.Tensor, language_idx: int):
        c = self.cfg
        n_frames = mel.shape[0]
        n_lfr = (n_frames + c.lfr_n - 1) // c.lfr_n
        left = (c.lfr_m - 1) // 2
        idx = (torch.arange(0, n_lfr * c.lfr_n, c.lfr_n).unsqueeze(1) + torch.arange(c.lfr_m) - left).clamp(min=0, max=n_frames - 1)
        x = mel[idx].reshape(n_lfr, c.feat_dim)
        x = (x + self._c(self.cmvn_means)) * self._c(self.cmvn_vars)
        x = x + self._c(self.speech_position[:n_lfr])
        return torch.cat([self._c(self.language_embed[language_idx:language_idx + 1]), self._c(self.system_embed), x], dim=0)

    # ---- Export_SenseVoice.py:227-258 -------------------------------------------------
    def sanm_block(self, x, blk, taps=None):
        c = self.cfg
        T = x.shape[0]
        h = F.layer_norm(x, (blk["in_size"],), self._c(blk["ln1"][0]), self._c(blk["ln1"][1]), 1e-5)
        qkv = h @ self._c(blk["wqkv"]).t() + self._c(blk["bqkv"])
        q, k, v = qkv.split(c.d_model, dim=1)
        qh = q.reshape(T, c.n_heads, c.d_head).transpose(0, 1)
        kh = k.reshape(T, c.n_heads, c.d_head).transpose(0, 1)
        vh = v.reshape(T, c.n_heads, c.d_head).transpose(0, 1)
        p = torch.softmax(qh @ kh.transpose(1, 2), dim=-1)
        ctx = (p @ vh).transpose(0, 1).reshape(T, c.d_model)
        pad = (c.fsmn_kernel - 1) // 2
        mem = F.conv1d(v.t().unsqueeze(0), self._c(blk["wfsmn"]), self._c(blk["bfsmn"]), padding=pad, groups=c.d_model)[0].t()
        att = ctx @ self._c(blk["wout"]).t() + mem
        if blk["in_size"] == c.d_model:
            att = att + x
        h2 = F.layer_norm(att, (c.d_model,), self._c(blk["ln2"][0]), self._c(blk["ln2"][1]), 1e-5)
        out = att + torch.relu(h2 @ self._c(blk["w1"]).t() + self._c(blk["b1"])) @ self._c(blk["w2"]).t() + self._c(blk["b2"])
        if taps is not None:
            taps.update(ln1=h, qkv=qkv, ctx=ctx, mem=mem, att=att, ln2=h2, out=out)
        return out

    def encode(self, x, taps=None):
        c = self.cfg
        n_main = c.n_enc0 + c.n_enc
        for i, blk in enumerate(self.blocks[:n_main]):
            x = self.sanm_block(x, blk, taps if (taps is not None and i == 0 and "block0_taps" in taps) else None)
            if taps is not None and i == 0:
                taps["block0"] = x
        x = F.layer_norm(x, (c.d_model,), self._c(self.ck["encoder.after_norm.weight"]), self._c(self.ck["encoder.after_norm.bias"]), 1e-5)
        for blk in self.blocks[n_main:]:
            x = self.sanm_block(x, blk)
        return F.layer_norm(x, (c.d_model,), self._c(self.ck["encoder.tp_norm.weight"]), self._c(self.ck["encoder.tp_norm.bias"]), 1e-5)

    # ---- Export_SenseVoice.py:290-296 -------------------------------------------------
    @staticmethod
    def ctc_collapse(ids: torch.Tensor, blank_id: int) -> torch.Tensor:
        """Circular next-neighbour collapse: keep t iff ids[t] != ids[(t+1) % T] and ids[t] != blank."""
        nxt = torch.roll(ids, -1, 0)
        return ids[(ids != nxt) & (ids != blank_id)].to(torch.int32)

    def ctc_head(self, enc_out):
        """CTC projection + greedy pick + collapse of one utterance's encoder rows (Export_SenseVoice.py:289-296)."""
        enc_out = self._c(torch.as_tensor(enc_out))
        logits = enc_out @ self._c(self.ck["ctc.ctc_lo.weight"]).t() + self._c(self.ck["ctc.ctc_lo.bias"])
        ids = logits.argmax(dim=-1)
        return logits, ids, self.ctc_collapse(ids, self.cfg.blank_id)

    def stages(self, audio_1d, language_idx: int) -> dict:
        with torch.inference_mode():
            a = torch.as_tensor(np.asarray(audio_1d, dtype=np.float32).reshape(-1))
            mel = self.frontend(a)
            enc_in = self.lfr_cmvn(mel, language_idx)
            taps = {}
            enc_out = self.encode(enc_in, taps)
            logits, ids, tok = self.ctc_head(enc_out)
        f = lambda z: z.float().numpy() if z.dtype != torch.float64 else z.numpy()
        return dict(mel=f(mel), enc_in=f(enc_in), block0=f(taps["block0"]), enc_out=f(enc_out), logits=f(logits),
                    frame_ids=ids.numpy().astype(np.int32), token_ids=tok.numpy(),
                    num_id=np.array([tok.numel()], dtype=np.int32))

    def __call__(self, audio_1d, language_idx: int):
        s = self.stages(audio_1d, language_idx)
        return s["token_ids"], s["num_id"]
