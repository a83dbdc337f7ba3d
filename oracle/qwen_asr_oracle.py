"""CPU oracle for the Qwen3-ASR hot path (Whisper-style log-mel -> Conv2d chunk stem -> windowed-attention audio encoder -> prompt
assembly -> Qwen3 decoder with KV cache -> greedy head).

TEST INFRASTRUCTURE. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

torch-CPU f32 restatement of `Qwen_ASR/Export_Qwen_ASR.py` on the RAW Hugging Face-layout checkpoint (folds applied here,
independently of arena.py):
    :850-857    front-end: reflect-padded STFT (STFT_Process stft_B, last frame dropped), power, Slaney mel (128), log10,
                clamp to (clip max - 8), x * 0.25 + 1
    :858-880    frames padded to whole 100-frame chunks; each chunk [1, 128, 100] -> 3 x (Conv2d k3 s2 p1 + tanh-GELU) -> [C, 16, 13]
                -> rows (t, c * 16 + f) -> conv_out (no bias) -> + positions[0..13)
    :881-922    chunks grouped into windows of n_window_infer / 100 chunks; encoder layers (LayerNorm folded into the fused q|k|v /
                fc1, q and k scaled by d^-1/4 each); keys >= the window's valid-token count get -128; ln_post -> proj1 -> tanh-GELU
                -> proj2; the first sum(aftercnn_lens) tokens are the audio embeddings
    :923-927    prompt = [head | query | suffix | audio | tail] (+ language tail, CONCAT_EMBED :1428-1435)
    :933-1028   rotary table (half-split convention, position = absolute index) and causal -128 mask
    :1265-1336  decoder layer: RMSNorm (weight folded into q|k|v) -> per-head RMSNorm of q and k with weight * d^-1/4 -> RoPE ->
                cache -> GQA soft-max attention -> o_proj + residual -> RMSNorm -> gate|up -> SiLU(gate) * up -> down + residual;
                final RMSNorm (learned weight) of the last position -> lm_head
Pinned against the real reference classes: tests/golden/qwen_asr_*.npz (oracle/gen_golden_qwen_asr.py).
"""
from __future__ import annotations

import numpy as np
import torch

from .whisper_oracle import slaney_mel_filterbank

F = torch.nn.functional


def feat_lengths(n: int) -> int:
    """_get_feat_extract_output_lengths (:519-527): tokens after the three stride-2 convolutions of n mel frames."""
    leave = n % 100
    f1 = (max(leave - 1, 0) // 2 + 1) * (leave > 0)
    f2 = (max(f1 - 1, 0) // 2 + 1) * (f1 > 0)
    f3 = (max(f2 - 1, 0) // 2 + 1) * (f2 > 0)
    return f3 + (n // 100) * 13


def log_softmax_f32(logits) -> np.ndarray:
    x = np.asarray(logits, dtype=np.float32)
    m = x.max()
    return (x - (m + np.log(np.exp(x - m, dtype=np.float32).sum(dtype=np.float32), dtype=np.float32))).astype(np.float32)


def beam_search_core(first_logits, state0, step, beam: int, max_new: int, stop_ids=(), margins=None):
    """Beam search as the build defines it (the reference repo names a beam mode for Qwen3-ASR in README.md:38 but ships no code for it,
    so there is NO reference behaviour to pin this against -- parity for this entry point is build-vs-this-restatement only).

    Width-`beam` search over summed log-soft-max scores, no length normalisation:
      * first ranking: the `beam` best ids of the prompt's logits (ties -> lower id) open the hypotheses;
      * every step, each live hypothesis offers its `beam` best extensions (score + log-prob); a hypothesis that has ended on a stop id
        (not emitted) offers itself unchanged; the best `beam` candidates are kept, ordered by score descending, then by parent
        hypothesis index, then by rank inside the parent;
      * the utterance is finished when the best hypothesis has ended (extensions only lower a score, so no live one can overtake it), or
        when max_new ids have been produced; the list then stands as it is.
    step(state, token) -> (logits, new_state) advances ONE hypothesis. -> best-first list of (token ids, score). `margins` (a list)
    collects the smallest score gap between neighbours of each ranking's first beam + 1 entries (how close the search came to a flip)."""
    stop = set(int(t) for t in stop_ids)
    lp = log_softmax_f32(first_logits)
    order = np.lexsort((np.arange(lp.size), -lp))[:beam + 1]
    if margins is not None:
        margins.append(float(np.min(-np.diff(lp[order]))))
    order = order[:beam]
    hyps = []
    for v in order:
        v = int(v)
        hyps.append(dict(tokens=[] if v in stop else [v], last=v, score=np.float32(lp[v]), fin=v in stop, state=state0))
    for _ in range(max_new - 1):
        if hyps[0]["fin"]:
            break
        cands = []
        for r, h in enumerate(hyps):
            if h["fin"]:
                cands.append((h["score"], r, 0, None, None))
                continue
            logits, st = step(h["state"], h["last"])
            lp = log_softmax_f32(logits)
            top = np.lexsort((np.arange(lp.size), -lp))[:beam]
            for k, v in enumerate(top):
                cands.append((np.float32(h["score"] + lp[v]), r, k, int(v), st))
        cands.sort(key=lambda q: (-float(q[0]), q[1], q[2]))
        if margins is not None and len(cands) > 1:
            sc = np.asarray([float(q[0]) for q in cands[:beam + 1]])
            margins.append(float(np.min(-np.diff(sc))))
        new = []
        for score, r, _, v, st in cands[:beam]:
            h = hyps[r]
            if v is None:
                new.append(h)
            else:
                new.append(dict(tokens=h["tokens"] + ([] if v in stop else [v]), last=v, score=score, fin=v in stop, state=st))
        hyps = new
    return [(np.asarray(h["tokens"], np.int32), float(h["score"])) for h in hyps]


class QwenAsrOracle:
    def __init__(self, cfg, ck: dict, head_ids, tail_ids, query_suffix_ids):
        self.cfg = cfg
        self.ck = {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in ck.items()}
        self.head_ids, self.tail_ids, self.suffix_ids = list(head_ids), list(tail_ids), list(query_suffix_ids)
        c = cfg
        n = torch.arange(c.nfft, dtype=torch.float32)
        self.window = 0.5 - 0.5 * torch.cos(2 * torch.pi * n / c.nfft)                    # periodic Hann
        self.mel = slaney_mel_filterbank(c.nfft // 2 + 1, 0.0, c.sample_rate / 2, c.n_mels, c.sample_rate)    # (n_freqs, n_mels)
        half = c.enc_d // 2
        inc = np.log(10000) / (half - 1)
        inv = torch.exp(-inc * torch.arange(half).float())
        st = torch.arange(c.max_source_positions)[:, None] * inv[None, :]
        self.enc_pos = torch.cat([torch.sin(st), torch.cos(st)], 1)
        self.inv_freq = 1.0 / (c.rope_theta ** (torch.arange(0, c.d_head, 2, dtype=torch.int64).float() / c.d_head))
        self._fold()

    # ---- :381-398 encoder folds, :1141-1190 decoder folds
    def _fold(self):
        c, ck = self.cfg, self.ck
        a = "thinker.audio_tower."
        self.enc = []
        s = float((c.enc_d // c.enc_heads) ** -0.25)
        for i in range(c.n_enc_layers):
            p = f"{a}layers.{i}."
            w = torch.cat([ck[p + "self_attn.q_proj.weight"], ck[p + "self_attn.k_proj.weight"], ck[p + "self_attn.v_proj.weight"]], 0).double()
            b = torch.cat([ck[p + "self_attn.q_proj.bias"], ck[p + "self_attn.k_proj.bias"], ck[p + "self_attn.v_proj.bias"]], 0).double()
            g, be = ck[p + "self_attn_layer_norm.weight"].double(), ck[p + "self_attn_layer_norm.bias"].double()
            b = b + w @ be
            w = w * g[None, :]
            w[:2 * c.enc_d] *= s
            b[:2 * c.enc_d] *= s
            g2, be2 = ck[p + "final_layer_norm.weight"].double(), ck[p + "final_layer_norm.bias"].double()
            w1 = ck[p + "fc1.weight"].double()
            b1 = ck[p + "fc1.bias"].double() + w1 @ be2
            w1 = w1 * g2[None, :]
            self.enc.append(dict(wqkv=w.float(), bqkv=b.float(), wo=ck[p + "self_attn.out_proj.weight"], bo=ck[p + "self_attn.out_proj.bias"],
                                 w1=w1.float(), b1=b1.float(), w2=ck[p + "fc2.weight"], b2=ck[p + "fc2.bias"]))
        gp, bp = ck[a + "ln_post.weight"].double(), ck[a + "ln_post.bias"].double()
        wp = ck[a + "proj1.weight"].double()
        self.proj1_b = (ck[a + "proj1.bias"].double() + wp @ bp).float()
        self.proj1_w = (wp * gp[None, :]).float()
        t = "thinker.model."
        self.dec = []
        for i in range(c.n_layers):
            p = f"{t}layers.{i}."
            wqkv = torch.cat([ck[p + "self_attn.q_proj.weight"], ck[p + "self_attn.k_proj.weight"], ck[p + "self_attn.v_proj.weight"]], 0)
            wqkv = wqkv * ck[p + "input_layernorm.weight"][None, :]
            gate_up = torch.cat([ck[p + "mlp.gate_proj.weight"], ck[p + "mlp.up_proj.weight"]], 0) * ck[p + "post_attention_layernorm.weight"][None, :]
            sc = float(c.d_head ** -0.25)
            self.dec.append(dict(wqkv=wqkv, qn=ck[p + "self_attn.q_norm.weight"] * sc, kn=ck[p + "self_attn.k_norm.weight"] * sc,
                                 wo=ck[p + "self_attn.o_proj.weight"], gate_up=gate_up, down=ck[p + "mlp.down_proj.weight"]))

    # ---- front-end
    def log_mel(self, audio_1d: torch.Tensor) -> torch.Tensor:
        c = self.cfg
        half = c.nfft // 2
        x = F.pad(audio_1d.reshape(1, 1, -1), (half, half), mode="reflect")[0, 0]
        frames = x.unfold(0, c.nfft, c.hop_length)[:-1]                                   # the trailing frame is dropped (drop_last_frame)
        spec = torch.fft.rfft(frames * self.window, dim=1)
        power = spec.real ** 2 + spec.imag ** 2
        mel = torch.clamp(power @ self.mel, min=1e-10).log10()
        mel = torch.maximum(mel, mel.max() - 8.0)
        return (mel * 0.25 + 1.0).t()                                                     # (n_mels, frames)

    def _rms(self, x, eps):
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)

    def encode(self, audio_1d, taps=None):
        """-> audio embeddings (n_tokens, d_model)."""
        c, ck = self.cfg, self.ck
        a = "thinker.audio_tower."
        feats = self.log_mel(torch.as_tensor(np.asarray(audio_1d, dtype=np.float32).reshape(-1)))
        n_frames = feats.shape[1]
        n_chunks = (n_frames + c.chunk - 1) // c.chunk
        feats = F.pad(feats, (0, n_chunks * c.chunk - n_frames))
        chunks = feats.reshape(c.n_mels, n_chunks, c.chunk).permute(1, 0, 2).unsqueeze(1)            # (chunks, 1, mels, 100)
        lens = [feat_lengths(min(max(n_frames - i * c.chunk, 0), c.chunk)) for i in range(n_chunks)]
        x = F.gelu(F.conv2d(chunks, ck[a + "conv2d1.weight"], ck[a + "conv2d1.bias"], stride=2, padding=1), approximate="tanh")
        x = F.gelu(F.conv2d(x, ck[a + "conv2d2.weight"], ck[a + "conv2d2.bias"], stride=2, padding=1), approximate="tanh")
        x = F.gelu(F.conv2d(x, ck[a + "conv2d3.weight"], ck[a + "conv2d3.bias"], stride=2, padding=1), approximate="tanh")
        nt = x.shape[3]                                                                               # 13 tokens per chunk
        x = x.permute(0, 3, 1, 2).reshape(n_chunks, nt, -1) @ ck[a + "conv_out.weight"].t() + self.enc_pos[:nt]
        cpw = c.chunks_per_window
        n_win = (n_chunks + cpw - 1) // cpw
        x = torch.cat([x, torch.zeros(n_win * cpw - n_chunks, nt, c.enc_d)], 0).reshape(n_win, cpw * nt, c.enc_d)
        lens_p = lens + [0] * (n_win * cpw - n_chunks)
        valid = [sum(lens_p[w * cpw:(w + 1) * cpw]) for w in range(n_win)]
        H, hd, T = c.enc_heads, c.enc_d // c.enc_heads, cpw * nt
        mask = torch.zeros(n_win, 1, 1, T)
        for w, v in enumerate(valid):
            mask[w, 0, 0, v:] = -128.0
        if taps is not None:
            taps["stem"] = x.clone()
        for L in self.enc:
            qkv = F.layer_norm(x, (c.enc_d,)) @ L["wqkv"].t() + L["bqkv"]
            q, k, v = [z.reshape(n_win, T, H, hd).transpose(1, 2) for z in qkv.split(c.enc_d, dim=-1)]
            att = torch.softmax(q @ k.transpose(-1, -2) + mask, dim=-1) @ v
            x = x + att.transpose(1, 2).reshape(n_win, T, c.enc_d) @ L["wo"].t() + L["bo"]
            x = x + F.gelu(F.layer_norm(x, (c.enc_d,)) @ L["w1"].t() + L["b1"], approximate="tanh") @ L["w2"].t() + L["b2"]
        x = F.gelu(F.layer_norm(x, (c.enc_d,)) @ self.proj1_w.t() + self.proj1_b, approximate="tanh") @ ck[a + "proj2.weight"].t() + ck[a + "proj2.bias"]
        return x.reshape(-1, c.d_model)[:sum(lens)]

    def embed(self, ids):
        return self.ck["thinker.model.embed_tokens.weight"][torch.as_tensor(list(ids), dtype=torch.long)]

    def prompt(self, audio_hidden, query_ids=(), language_tail_ids=()):
        return torch.cat([self.embed(self.head_ids), self.embed(query_ids), self.embed(self.suffix_ids), audio_hidden,
                          self.embed(self.tail_ids), self.embed(language_tail_ids)], 0)

    # ---- decoder over n new positions with history (k, v lists of (kv_heads, S, hd))
    def decoder(self, x, hist: int, keys, vals):
        c = self.cfg
        n, H, KV, hd = x.shape[0], c.n_heads, c.n_kv_heads, c.d_head
        pos = torch.arange(hist, hist + n, dtype=torch.float32)
        theta = pos[:, None] * self.inv_freq[None, :]
        cos, sin = torch.cat([torch.cos(theta)] * 2, -1), torch.cat([torch.sin(theta)] * 2, -1)       # (n, hd)
        rot = lambda z: torch.cat([-z[..., hd // 2:], z[..., :hd // 2]], -1)
        mask = torch.where(torch.arange(hist + n)[None, :] <= (hist + torch.arange(n))[:, None], 0.0, -128.0)
        new_k, new_v = [], []
        for li, L in enumerate(self.dec):
            qkv = self._rms(x, c.rms_eps) @ L["wqkv"].t()
            q = qkv[:, :H * hd].reshape(n, H, hd)
            k = qkv[:, H * hd:(H + KV) * hd].reshape(n, KV, hd)
            v = qkv[:, (H + KV) * hd:].reshape(n, KV, hd)
            q = self._rms(q, c.rms_eps) * L["qn"]
            k = self._rms(k, c.rms_eps) * L["kn"]
            q = q * cos[:, None, :] + rot(q) * sin[:, None, :]
            k = k * cos[:, None, :] + rot(k) * sin[:, None, :]
            k_all = torch.cat([keys[li], k.transpose(0, 1)], 1) if keys is not None else k.transpose(0, 1)
            v_all = torch.cat([vals[li], v.transpose(0, 1)], 1) if vals is not None else v.transpose(0, 1)
            new_k.append(k_all)
            new_v.append(v_all)
            G = H // KV
            qg = q.reshape(n, KV, G, hd).permute(1, 2, 0, 3)                                          # (KV, G, n, hd)
            att = torch.softmax(qg @ k_all[:, None].transpose(-1, -2) + mask, dim=-1) @ v_all[:, None]
            x = x + att.permute(2, 0, 1, 3).reshape(n, H * hd) @ L["wo"].t()
            gu = self._rms(x, c.rms_eps) @ L["gate_up"].t()
            x = x + (F.silu(gu[:, :c.d_ffn]) * gu[:, c.d_ffn:]) @ L["down"].t()
        last = self._rms(x[-1], c.rms_eps) * self.ck["thinker.model.norm.weight"]
        return last @ self.ck["thinker.lm_head.weight"].t(), new_k, new_v

    def heads(self, audio_1d, n_new: int, query_ids=(), language_tail_ids=(), penalty=None, sampling=None, noise=None):
        """The other decode strategies (Inference_Qwen_ASR_ONNX.py:369-376). penalty = (value, range): prefill picks arg-max of the raw
        logits; every decode step multiplies the logits of save_id[-range:] (whatever exists, APPLY_PENALTY :1403-1415) by value, then
        arg-max (GREEDY_SEARCH). sampling = (temperature, top_k, top_p, repetition_penalty) with noise[step] = the top_k uniforms:
        TOPK_TOPP_SAMPLING (:1348-1400) at the prefill too, history = every sampled id. -> dict(token_ids, logits (penalised heads))"""
        from .whisper_oracle import WhisperOracle
        with torch.inference_mode():
            x = self.prompt(self.encode(audio_1d), query_ids, language_tail_ids)
            logits, ks, vs = self.decoder(x, 0, None, None)
            hist, toks, heads = x.shape[0], [], []
            for step in range(n_new):
                if step:
                    logits, ks, vs = self.decoder(self.embed([toks[-1]]), hist, ks, vs)
                    hist += 1
                head = logits.clone()
                if sampling is not None:
                    toks.append(WhisperOracle.sample_head(head.float(), toks, noise[step], *sampling))
                    continue
                if penalty is not None and step and toks:
                    idx = torch.tensor(toks[-int(penalty[1]):], dtype=torch.long)
                    head[idx] = logits[idx] * float(penalty[0])
                heads.append(head)
                toks.append(int(head.argmax()))
        return dict(token_ids=np.asarray(toks, np.int32), logits=torch.stack(heads).numpy() if heads else None)

    def beam(self, audio_1d, beam: int, max_new: int, query_ids=(), language_tail_ids=(), stop_ids=()):
        """beam_search_core over this restatement's decoder -> best-first [(token ids, score)]"""
        with torch.inference_mode():
            x = self.prompt(self.encode(audio_1d), query_ids, language_tail_ids)
            logits, ks, vs = self.decoder(x, 0, None, None)

            def step(state, token):
                hist, k, v = state
                lg, k2, v2 = self.decoder(self.embed([token]), hist, k, v)
                return lg.numpy(), (hist + 1, k2, v2)
            return beam_search_core(logits.numpy(), (x.shape[0], ks, vs), step, beam, max_new, stop_ids)

    def greedy(self, audio_1d, n_new: int, query_ids=(), language_tail_ids=(), stop_ids=()):
        """-> dict(audio_hidden, ids_len, logits (steps, vocab), token_ids)"""
        with torch.inference_mode():
            audio_hidden = self.encode(audio_1d)
            x = self.prompt(audio_hidden, query_ids, language_tail_ids)
            logits, ks, vs = self.decoder(x, 0, None, None)
            steps, toks, hist = [logits], [int(logits.argmax())], x.shape[0]
            while len(toks) < n_new and toks[-1] not in stop_ids:
                logits, ks, vs = self.decoder(self.embed([toks[-1]]), hist, ks, vs)
                hist += 1
                steps.append(logits)
                toks.append(int(logits.argmax()))
        return dict(audio_hidden=audio_hidden.numpy(), ids_len=int(x.shape[0]), logits=torch.stack(steps).numpy(), token_ids=np.asarray(toks, np.int32))
