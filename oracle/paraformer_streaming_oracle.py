"""CPU oracle for the STREAMING Paraformer hot path (chunk = 8000 samples): Kaldi fbank -> LFR -> SANM encoder with per-layer
K/V history -> unrolled integrate-and-fire with carried state -> SANM decoder with FSMN / cross-K/V history.

TEST INFRASTRUCTURE. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

torch-CPU f32 restatement of `Paraformer/Streaming/Export_Paraformer_Streaming.py` on the RAW FunASR-layout checkpoint:
    :64-71     chunk geometry: 48 fbank frames, LFR_LENGTH = 9 rows per chunk (B), overlap C = B // 2 = 4, A = 0
    :386-399   front-end: fbank, LFR gather (indices clamped into the chunk), (x + means) * vars, + positions[start : start + B],
               x = [4 carried rows | 9 new rows], carry the last 4 rows
    :400-435   per layer: folded-LN q|k|v, K/V history concatenated in front (kept: [-40:-4] of the concatenation), soft-max
               attention of all 13 rows over history + chunk, FSMN over the 13 rows (zero padded) + v, linear_out, residual from
               the second layer on, FFN
    :436-462   after_norm, k = 3 CIF conv + ReLU + linear + sigmoid; integrate-and-fire UNROLLED over rows 0..8 with the carried
               (cif_hidden, cif_alphas); cif_hidden is stored divided by cif_alphas
    :508-553   decoder per fired chunk: FFN (folded norms) -> norm2 -> FSMN over [10 history | tokens] (valid conv) + x + residual
               -> cross-attention over [9 cached | 13 chunk rows] keys -> residual; FFN-only block; folded output layer; arg-max
    host loop  Inference_Paraformer_Streaming_ONNX.py:401-449: the decoder (and its caches) advance only when a frame fired
Pinned against the real reference classes: tests/golden/paraformer_streaming_*.npz (oracle/gen_golden_paraformer_streaming.py).
"""
from __future__ import annotations

import numpy as np
import torch

from .paraformer_oracle import ParaformerOracle

F = torch.nn.functional


class ParaformerStreamingOracle(ParaformerOracle):
    def __init__(self, cfg, ck: dict, chunk: int = 8000, look_back_encoder: int = 4, look_back_decoder: int = 1, max_continue: int = 502):
        super().__init__(cfg, ck)
        c = cfg
        self.chunk = chunk
        self.n_frames_chunk = (chunk - c.win_length) // c.hop_length + 1
        self.B = ((c.lfr_m - 1) // 2 + self.n_frames_chunk) // c.lfr_n + 1
        self.C, self.A = self.B // 2, 0
        self.en_drop, self.en_keep = self.C, look_back_encoder * self.B            # keep [-(keep + C) : -C]
        self.de_keep = look_back_decoder * self.B
        self.fsmn_hist = c.fsmn_kernel - 1
        feat = c.feat_dim
        positions = torch.arange(1, max_continue, dtype=torch.float32)
        log_inc = torch.log(torch.tensor([10000.0])) / (feat / 2 - 1)
        inv_ts = torch.exp(torch.arange(feat / 2).float() * (-log_inc))
        st = positions.reshape(-1, 1) * inv_ts.reshape(1, -1)
        self.pos = torch.cat([torch.sin(st), torch.cos(st)], 1)                   # row i <-> position i + 1
        self.means = self.ck["frontend.cmvn_means"].reshape(1, feat)
        self.vars = (self.ck["frontend.cmvn_vars"] * (float(c.d_model) ** 0.5)).reshape(1, feat)
        self.dec_wf_raw = [self.ck[f"decoder.decoders.{i}.self_attn.fsmn_block.weight"] for i in range(c.n_dec)]

    def init_state(self):
        c = self.cfg
        d, H, hd = c.d_model, c.n_heads, c.d_head
        n_en = c.n_enc0 + c.n_enc
        return dict(en_k=[torch.zeros(H, 0, hd) for _ in range(n_en)], en_v=[torch.zeros(H, 0, hd) for _ in range(n_en)],
                    prev=torch.zeros(self.A + self.C, c.feat_dim), cif_hidden=torch.zeros(d), cif_alphas=torch.zeros(()), start=0,
                    de_fsmn=[torch.zeros(d, self.fsmn_hist) for _ in range(c.n_dec)],
                    de_k=[torch.zeros(H, 0, hd) for _ in range(c.n_dec)], de_v=[torch.zeros(H, 0, hd) for _ in range(c.n_dec)])

    def encoder_step(self, st, audio_chunk):
        c = self.cfg
        d, H, hd, B = c.d_model, c.n_heads, c.d_head, self.B
        mel = self.fbank(torch.as_tensor(np.asarray(audio_chunk, dtype=np.float32).reshape(-1)))
        nf = mel.shape[0]
        idx = (torch.arange(0, B * c.lfr_n, c.lfr_n).unsqueeze(1) + torch.arange(c.lfr_m) - (c.lfr_m - 1) // 2).clamp(min=0, max=nf - 1)
        feats = (mel[idx].reshape(B, c.feat_dim) + self.means) * self.vars
        feats = feats + self.pos[st["start"]:st["start"] + B]
        st["start"] += B
        x = torch.cat([st["prev"], feats], 0)
        st["prev"] = x[-(self.A + self.C):].clone()
        T = x.shape[0]
        pad = (c.fsmn_kernel - 1) // 2
        for li, L in enumerate(self.enc):
            qkv = F.layer_norm(x, (L["in_size"],)) @ L["wqkv"].t() + L["bqkv"]
            q, k, v = qkv.split(d, dim=1)
            qh, kh, vh = [z.reshape(T, H, hd).transpose(0, 1) for z in (q, k, v)]
            k_all, v_all = torch.cat([st["en_k"][li], kh], 1), torch.cat([st["en_v"][li], vh], 1)
            st["en_k"][li] = k_all[:, -(self.en_keep + self.en_drop):-self.en_drop].clone()
            st["en_v"][li] = v_all[:, -(self.en_keep + self.en_drop):-self.en_drop].clone()
            ctx = (torch.softmax(qh @ k_all.transpose(1, 2), dim=-1) @ v_all).transpose(0, 1).reshape(T, d)
            mem = F.conv1d(v.t().unsqueeze(0), L["wf"], None, padding=pad, groups=d)[0].t()          # identity folded into the centre tap
            att = ctx @ L["wo"].t() + L["bo"] + mem
            x = x + att if li > 0 else att
            x = x + torch.relu(F.layer_norm(x, (d,)) @ L["w1"].t() + L["b1"]) @ L["w2"].t() + L["b2"]
        enc_out = F.layer_norm(x, (d,), self.ck["encoder.after_norm.weight"], self.ck["encoder.after_norm.bias"])
        conv = torch.relu(F.conv1d(enc_out.t().unsqueeze(0), self.ck["predictor.cif_conv1d.weight"], self.ck["predictor.cif_conv1d.bias"],
                                   padding=c.cif_kernel // 2))[0].t()
        alphas = torch.sigmoid(conv @ self.ck["predictor.cif_output.weight"].t() + self.ck["predictor.cif_output.bias"]).squeeze(-1)
        # ---- unrolled integrate-and-fire (:438-462)
        one = torch.tensor(1.0)
        ca, ch = st["cif_alphas"].clone(), st["cif_hidden"]
        cond_a = (ca < one).float()
        cond_b = one - cond_a
        saves = [cond_b]
        frames = ca * ch * cond_a + ch * cond_b
        fl = [frames]
        ca = ca - cond_b
        frames = frames * cond_a + ca * ch * cond_b
        for t in range(self.A, self.A + B):
            alpha, hidden = alphas[t], enc_out[t]
            thr = one - ca
            cond_a = (alpha < thr).float()
            cond_b = one - cond_a
            saves.append(cond_b)
            frames = (frames + alpha * hidden) * cond_a + (frames + thr * hidden) * cond_b
            fl.append(frames)
            ca = ca + alpha
            ca = ca - cond_b
            frames = frames * cond_a + ca * hidden * cond_b
        fl = torch.stack(fl)
        st["cif_hidden"] = fl[-1] / ca
        st["cif_alphas"] = ca
        fired = torch.nonzero(torch.stack(saves)).squeeze(1)
        return enc_out, fl[fired], alphas

    def decoder_step(self, st, enc_out, list_frame):
        c = self.cfg
        d, H, hd = c.d_model, c.n_heads, c.d_head
        n, T = list_frame.shape[0], enc_out.shape[0]
        dec = list_frame
        for li, L in enumerate(self.dec):
            x = F.layer_norm(torch.relu(F.layer_norm(dec, (d,)) @ L["w1"].t() + L["b1"]), (c.d_dec_ffn,)) @ L["w2"].t() + L["b2"]
            x = F.layer_norm(x, (d,), L["n2"][0], L["n2"][1])
            cat = torch.cat([st["de_fsmn"][li], x.t()], 1)
            st["de_fsmn"][li] = cat[:, -self.fsmn_hist:].clone()
            y = F.conv1d(cat.unsqueeze(0), self.dec_wf_raw[li], None, groups=d)[0].t() + x + dec
            q = (F.layer_norm(y, (d,)) @ L["wq"].t() + L["bq"]).reshape(n, H, hd).transpose(0, 1)
            kv = enc_out @ L["wkv"].t() + L["bkv"]
            k_all = torch.cat([st["de_k"][li], kv[:, :d].reshape(T, H, hd).transpose(0, 1)], 1)
            v_all = torch.cat([st["de_v"][li], kv[:, d:].reshape(T, H, hd).transpose(0, 1)], 1)
            st["de_k"][li], st["de_v"][li] = k_all[:, -self.de_keep:].clone(), v_all[:, -self.de_keep:].clone()
            ctx = (torch.softmax(q @ k_all.transpose(1, 2), dim=-1) @ v_all).transpose(0, 1).reshape(n, d)
            dec = y + ctx @ L["wo"].t() + L["bo"]
        for L in self.dec3:
            dec = F.layer_norm(torch.relu(F.layer_norm(dec, (d,)) @ L["w1"].t() + L["b1"]), (c.d_dec_ffn,)) @ L["w2"].t() + L["b2"]
        self.last_dec_hidden = F.layer_norm(dec, (d,))                 # (n, d): what the output layer multiplies (tests build heads with trained margins on these rows)
        return self.last_dec_hidden @ self.w_out.t() + self.b_out

    def run(self, audio_1d, state=None):
        """Whole clip (length a multiple of the chunk) -> list of per-chunk dicts; the decoder runs only for chunks that fired."""
        st = state or self.init_state()
        audio = np.asarray(audio_1d, dtype=np.float32).reshape(-1)
        out = []
        with torch.inference_mode():
            for s in range(0, audio.size - self.chunk + 1, self.chunk):
                enc_out, frames, alphas = self.encoder_step(st, audio[s:s + self.chunk])
                rec = dict(enc_out=enc_out.numpy(), alphas=alphas.numpy(), n=int(frames.shape[0]), list_frame=frames.numpy(),
                           cif_alphas=float(st["cif_alphas"]))
                if frames.shape[0]:
                    logits = self.decoder_step(st, enc_out, frames)
                    rec["logits"] = logits.numpy()
                    rec["dec_hidden"] = self.last_dec_hidden.numpy().copy()
                    rec["token_ids"] = logits.argmax(-1).int().numpy()
                else:
                    rec["token_ids"] = np.zeros(0, np.int32)
                out.append(rec)
        return out
