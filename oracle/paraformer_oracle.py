"""CPU oracle for the non-streaming Paraformer hot path (Kaldi fbank -> SANM encoder -> CIF predictor -> SANM decoder).

TEST INFRASTRUCTURE. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

torch-CPU f32 restatement of `Paraformer/Non-Streaming/Export_Paraformer.py`, consuming the RAW FunASR-layout
checkpoint (checkpoints.py) and applying the export-time folds itself (in float64, rounded once, like the reference):
    :326-343   Kaldi fbank kernel = windowed DFT basis @ (pre-emphasis matrix @ DC-removal matrix)
    :346-364   KaldiFbank.forward (power, Kaldi mel banks + zero Nyquist column, max(eps).log)
    :214-258   absorb_layer_norm_affine / fold_linear_output_scale (d_k^-1/4 on q,k / cross q,k)
    :290-305   FSMN identity tap folded into the depth-wise conv (symmetric zero pad folded into the conv)
    :459-465   encoder_input_bias = cmvn_means * cmvn_vars + positions (float64)
    :474-497   LFR + encoder blocks (residual only when in_size == size) + after_norm
    :499-519   CIF predictor: conv3 + ReLU + Linear + sigmoid, tail threshold, FLOAT64 prefix sum, fire / frames / remains
    :521-563   decoder: grouped cross-KV, per layer FFN -> LN -> FSMN + residual -> cross-attention + residual,
               FFN-only blocks, folded after_norm + output_layer, arg-max, first num_id tokens
Pinned against the real reference classes: tests/golden/paraformer_*.npz (oracle/gen_golden_paraformer.py).
"""
from __future__ import annotations

import numpy as np
import torch

from .kaldi_mel import get_mel_banks

F = torch.nn.functional


def _absorb(norm_w, norm_b, w, b, out_scale=1.0):
    """float64: W' = (W * scale) * gamma,  b' = b * scale + (W * scale) @ beta; rounded once to f32."""
    w64 = w.double()
    b64 = b.double() if b is not None else torch.zeros(w.shape[0], dtype=torch.float64)
    s = torch.as_tensor(out_scale, dtype=torch.float64)
    if s.ndim == 0:
        w64, b64 = w64 * s, b64 * s
    else:
        w64, b64 = w64 * s.unsqueeze(1), b64 * s
    b64 = b64 + w64 @ norm_b.double()
    w64 = w64 * norm_w.double().unsqueeze(0)
    return w64.float(), b64.float()


class ParaformerOracle:
    def __init__(self, cfg, ck: dict):
        self.cfg = cfg
        self.ck = {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in ck.items()}
        self._frontend()
        self._fold()

    def _frontend(self):
        c = self.cfg
        nfreq = c.nfft // 2 + 1
        window = torch.hamming_window(c.win_length, periodic=False, alpha=0.54, beta=0.46)
        k = torch.arange(nfreq, dtype=torch.float32).unsqueeze(1)
        n = torch.arange(c.win_length, dtype=torch.float32).unsqueeze(0)
        omega = (2.0 * torch.pi / c.nfft) * k * n
        real, imag = torch.cos(omega) * window, -torch.sin(omega) * window
        W = c.win_length
        dc = torch.eye(W) - torch.full((W, W), 1.0 / W)
        prev = torch.zeros(W, W)
        prev[0, 0] = 1.0
        prev[1:, :-1] = torch.eye(W - 1)
        frame_transform = (torch.eye(W) - float(c.pre_emphasis) * prev) @ dc
        self.kernel = torch.cat([real @ frame_transform, imag @ frame_transform], 0)       # (2*nfreq, W)
        banks, _ = get_mel_banks(c.n_mels, c.nfft, c.sample_rate, 20.0, 0.0, 100.0, -500.0, 1.0)
        self.mel_bins = F.pad(banks, (0, 1), value=0.0)                                      # (n_mels, nfreq)
        self.nfreq = nfreq
        self.eps = torch.tensor(torch.finfo(torch.float32).eps)

    def _fold(self):
        c, ck = self.cfg, self.ck
        d = c.d_model
        factor = float(c.d_head ** -0.25)
        pad = (c.fsmn_kernel - 1) // 2

        def fsmn(p):
            w = ck[p + "self_attn.fsmn_block.weight"].double().clone()
            w[:, 0, pad] += 1.0
            return w.float()

        self.enc = []
        for p in [f"encoder.encoders0.{i}." for i in range(c.n_enc0)] + [f"encoder.encoders.{i}." for i in range(c.n_enc)]:
            scale = torch.ones(3 * d, dtype=torch.float64)
            scale[:-d] = factor
            wqkv, bqkv = _absorb(ck[p + "norm1.weight"], ck[p + "norm1.bias"], ck[p + "self_attn.linear_q_k_v.weight"],
                                 ck[p + "self_attn.linear_q_k_v.bias"], scale)
            w1, b1 = _absorb(ck[p + "norm2.weight"], ck[p + "norm2.bias"], ck[p + "feed_forward.w_1.weight"], ck[p + "feed_forward.w_1.bias"])
            self.enc.append(dict(wqkv=wqkv, bqkv=bqkv, wf=fsmn(p), wo=ck[p + "self_attn.linear_out.weight"], bo=ck[p + "self_attn.linear_out.bias"],
                                 w1=w1, b1=b1, w2=ck[p + "feed_forward.w_2.weight"], b2=ck[p + "feed_forward.w_2.bias"], in_size=wqkv.shape[1]))
        self.dec, self.dec3 = [], []
        for i in range(c.n_dec):
            p = f"decoder.decoders.{i}."
            w1, b1 = _absorb(ck[p + "norm1.weight"], ck[p + "norm1.bias"], ck[p + "feed_forward.w_1.weight"], ck[p + "feed_forward.w_1.bias"])
            w2, b2 = _absorb(ck[p + "feed_forward.norm.weight"], ck[p + "feed_forward.norm.bias"], ck[p + "feed_forward.w_2.weight"], None)
            wq, bq = _absorb(ck[p + "norm3.weight"], ck[p + "norm3.bias"], ck[p + "src_attn.linear_q.weight"], ck[p + "src_attn.linear_q.bias"], factor)
            kv_scale = torch.ones(2 * d, dtype=torch.float64)
            kv_scale[:d] = factor
            wkv = (ck[p + "src_attn.linear_k_v.weight"].double() * kv_scale.unsqueeze(1)).float()
            bkv = (ck[p + "src_attn.linear_k_v.bias"].double() * kv_scale).float()
            self.dec.append(dict(w1=w1, b1=b1, w2=w2, b2=b2, n2=(ck[p + "norm2.weight"], ck[p + "norm2.bias"]), wf=fsmn(p), wq=wq, bq=bq,
                                 wkv=wkv, bkv=bkv, wo=ck[p + "src_attn.linear_out.weight"], bo=ck[p + "src_attn.linear_out.bias"]))
        for i in range(c.n_dec3):
            p = f"decoder.decoders3.{i}."
            w1, b1 = _absorb(ck[p + "norm1.weight"], ck[p + "norm1.bias"], ck[p + "feed_forward.w_1.weight"], ck[p + "feed_forward.w_1.bias"])
            w2, b2 = _absorb(ck[p + "feed_forward.norm.weight"], ck[p + "feed_forward.norm.bias"], ck[p + "feed_forward.w_2.weight"], None)
            self.dec3.append(dict(w1=w1, b1=b1, w2=w2, b2=b2))
        self.w_out, self.b_out = _absorb(ck["decoder.after_norm.weight"], ck["decoder.after_norm.bias"], ck["decoder.output_layer.weight"],
                                         ck["decoder.output_layer.bias"])
        # encoder input: x * vars + (means * vars + positions), bias built in float64 (:459-465, 580-584)
        feat = c.feat_dim
        vars_ = ck["frontend.cmvn_vars"] * (float(d) ** 0.5)
        lfr_len = (c.n_frames(c.max_audio_len) + c.lfr_n - 1) // c.lfr_n
        positions = torch.arange(1, lfr_len + 1, dtype=torch.float32)
        log_inc = torch.log(torch.tensor([10000.0])) / (feat / 2 - 1)
        inv_ts = torch.exp(torch.arange(feat / 2).float() * (-log_inc))
        st = positions.reshape(-1, 1) * inv_ts.reshape(1, -1)
        pos = torch.cat([torch.sin(st), torch.cos(st)], 1)
        self.cmvn_vars = vars_.reshape(1, feat)
        self.in_bias = (ck["frontend.cmvn_means"].double().reshape(1, feat) * vars_.double().reshape(1, feat) + pos.double()).float()

    # ---- :358-364
    def fbank(self, audio_1d):
        c = self.cfg
        frames = audio_1d.float().unfold(0, c.win_length, c.hop_length)
        st = frames @ self.kernel.t()
        sq = st * st
        power = sq[:, :self.nfreq] + sq[:, self.nfreq:]
        return torch.maximum(power @ self.mel_bins.t(), self.eps).log()                      # (frames, n_mels)

    def encode(self, mel):
        c = self.cfg
        d, H, hd = c.d_model, c.n_heads, c.d_head
        n_frames = mel.shape[0]
        T = (n_frames + c.lfr_n - 1) // c.lfr_n
        left = (c.lfr_m - 1) // 2
        idx = (torch.arange(0, T * c.lfr_n, c.lfr_n).unsqueeze(1) + torch.arange(c.lfr_m) - left).clamp(min=0, max=n_frames - 1)
        x = mel[idx].reshape(T, c.feat_dim) * self.cmvn_vars + self.in_bias[:T]
        pad = (c.fsmn_kernel - 1) // 2
        for L in self.enc:
            qkv = F.layer_norm(x, (L["in_size"],)) @ L["wqkv"].t() + L["bqkv"]
            q, k, v = qkv.split(d, dim=1)
            qh, kh, vh = [z.reshape(T, H, hd).transpose(0, 1) for z in (q, k, v)]
            ctx = (torch.softmax(qh @ kh.transpose(1, 2), dim=-1) @ vh).transpose(0, 1).reshape(T, d)
            mem = F.conv1d(v.t().unsqueeze(0), L["wf"], None, padding=pad, groups=d)[0].t()
            att = ctx @ L["wo"].t() + L["bo"] + mem
            x = x + att if L["in_size"] == d else att
            x = x + torch.relu(F.layer_norm(x, (d,)) @ L["w1"].t() + L["b1"]) @ L["w2"].t() + L["b2"]
        return F.layer_norm(x, (d,), self.ck["encoder.after_norm.weight"], self.ck["encoder.after_norm.bias"])

    # ---- :499-519
    def cif(self, enc_out):
        c, ck = self.cfg, self.ck
        conv = torch.relu(F.conv1d(enc_out.t().unsqueeze(0), ck["predictor.cif_conv1d.weight"], ck["predictor.cif_conv1d.bias"],
                                   padding=c.cif_kernel // 2))[0].t()
        alphas = torch.sigmoid(conv @ ck["predictor.cif_output.weight"].t() + ck["predictor.cif_output.bias"]).squeeze(-1)
        a = torch.cat([alphas, torch.tensor([c.tail_threshold])])
        hidden = torch.cat([enc_out, torch.zeros(1, c.d_model)], 0)
        prefix = torch.cumsum(a, dim=0, dtype=torch.float64).float()
        floor = torch.floor(prefix)
        prev = torch.cat([torch.zeros(1), floor[:-1]])
        fire = torch.nonzero(floor > prev).squeeze(1)
        prefix_hidden = torch.cumsum(a.unsqueeze(1) * hidden, dim=0)
        frames = prefix_hidden[fire]
        remains = (prefix - floor)[fire]
        completed = torch.cat([torch.zeros(1, c.d_model), frames - remains.unsqueeze(1) * hidden[fire]], 0)
        return alphas, completed[1:] - completed[:-1], int(floor[-1])

    # ---- :521-563
    def decode_hidden(self, acoustic, memory, num_id):
        """decoder rows up to (not including) the output projection: after_norm without its affine, which __init__ folded into w_out / b_out"""
        c = self.cfg
        d, H, hd = c.d_model, c.n_heads, c.d_head
        n = max(num_id, 1)
        dec = torch.cat([acoustic, torch.zeros(1, d)], 0)[:n]
        T = memory.shape[0]
        pad = (c.fsmn_kernel - 1) // 2
        for L in self.dec:
            kv = memory @ L["wkv"].t() + L["bkv"]
            k = kv[:, :d].reshape(T, H, hd).transpose(0, 1)
            v = kv[:, d:].reshape(T, H, hd).transpose(0, 1)
            x = F.layer_norm(torch.relu(F.layer_norm(dec, (d,)) @ L["w1"].t() + L["b1"]), (c.d_dec_ffn,)) @ L["w2"].t() + L["b2"]
            sa_in = F.layer_norm(x, (d,), L["n2"][0], L["n2"][1])
            x = dec + F.conv1d(sa_in.t().unsqueeze(0), L["wf"], None, padding=pad, groups=d)[0].t()
            q = (F.layer_norm(x, (d,)) @ L["wq"].t() + L["bq"]).reshape(n, H, hd).transpose(0, 1)
            ctx = (torch.softmax(q @ k.transpose(1, 2), dim=-1) @ v).transpose(0, 1).reshape(n, d)
            dec = x + ctx @ L["wo"].t() + L["bo"]
        for L in self.dec3:
            dec = F.layer_norm(torch.relu(F.layer_norm(dec, (d,)) @ L["w1"].t() + L["b1"]), (c.d_dec_ffn,)) @ L["w2"].t() + L["b2"]
        return F.layer_norm(dec, (d,))

    def decode(self, acoustic, memory, num_id):
        return self.decode_hidden(acoustic, memory, num_id) @ self.w_out.t() + self.b_out

    def stages(self, audio_1d):
        with torch.inference_mode():
            mel = self.fbank(torch.as_tensor(np.asarray(audio_1d, dtype=np.float32).reshape(-1)))
            enc_out = self.encode(mel)
            alphas, acoustic, num_id = self.cif(enc_out)
            hidden = self.decode_hidden(acoustic, enc_out, num_id)
            logits = hidden @ self.w_out.t() + self.b_out
            ids = logits.argmax(-1).int()[:num_id]
        return dict(dec_hidden=hidden.numpy(), mel=mel.numpy(), enc_out=enc_out.numpy(), alphas=alphas.numpy(), acoustic=acoustic.numpy(), logits=logits.numpy(),
                    token_ids=ids.numpy().astype(np.int32), num_id=np.array([num_id], np.int32))

    def __call__(self, audio_1d):
        s = self.stages(audio_1d)
        return s["token_ids"], s["num_id"]
