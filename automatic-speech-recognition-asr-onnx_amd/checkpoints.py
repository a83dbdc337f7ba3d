"""Seeded synthetic checkpoints in the *source* parameter layout.

No real checkpoints exist in the build or bench environment (no network), so
benchmarks and parity fixtures use random weights of the exact architecture.
The dict keys follow the source checkpoints the reference exporters read
(FunASR `SenseVoiceSmall` state-dict names: SenseVoice/Export_SenseVoice.py:
130-132,172-183,211-220; HF `WhisperForConditionalGeneration` names:
Whisper/Export_Whisper.py:376-420,527-550), so the same converter
(`arena.py`) would ingest a real state dict unchanged.

Initialisation is fan-in scaled (std = 1/sqrt(fan_in)) so that activations are
O(1) and the soft-max / LayerNorm paths are numerically exercised; LayerNorm
affines are perturbed around (1, 0).
"""
from __future__ import annotations

import os

import numpy as np

from .config import SenseVoiceConfig, WhisperConfig


def _lin(rng, out_f, in_f, bias=True, gain=1.0):
    w = rng.standard_normal((out_f, in_f), dtype=np.float32) * np.float32(gain / np.sqrt(in_f))
    b = rng.standard_normal((out_f,), dtype=np.float32) * np.float32(0.1) if bias else None
    return w, b


def _ln(rng, n):
    g = (1.0 + 0.1 * rng.standard_normal((n,), dtype=np.float32)).astype(np.float32)
    b = (0.1 * rng.standard_normal((n,), dtype=np.float32)).astype(np.float32)
    return g, b


def synth_sensevoice_checkpoint(cfg: SenseVoiceConfig, seed: int = 0) -> dict:
    """Random SenseVoiceSmall-shaped checkpoint (raw, before any export-time fold)."""
    rng = np.random.default_rng(seed)
    ck: dict[str, np.ndarray] = {}
    d, dff, feat = cfg.d_model, cfg.d_ffn, cfg.feat_dim

    def block(prefix, in_size):
        ck[prefix + "norm1.weight"], ck[prefix + "norm1.bias"] = _ln(rng, in_size)
        w, b = _lin(rng, 3 * d, in_size)
        ck[prefix + "self_attn.linear_q_k_v.weight"], ck[prefix + "self_attn.linear_q_k_v.bias"] = w, b
        ck[prefix + "self_attn.fsmn_block.weight"] = (
            rng.standard_normal((d, 1, cfg.fsmn_kernel), dtype=np.float32) * np.float32(0.2))
        w, b = _lin(rng, d, d)
        ck[prefix + "self_attn.linear_out.weight"], ck[prefix + "self_attn.linear_out.bias"] = w, b
        ck[prefix + "norm2.weight"], ck[prefix + "norm2.bias"] = _ln(rng, d)
        w, b = _lin(rng, dff, d, gain=1.4)
        ck[prefix + "feed_forward.w_1.weight"], ck[prefix + "feed_forward.w_1.bias"] = w, b
        w, b = _lin(rng, d, dff)
        ck[prefix + "feed_forward.w_2.weight"], ck[prefix + "feed_forward.w_2.bias"] = w, b

    for i in range(cfg.n_enc0):
        block(f"encoder.encoders0.{i}.", feat)
    for i in range(cfg.n_enc):
        block(f"encoder.encoders.{i}.", d)
    for i in range(cfg.n_tp):
        block(f"encoder.tp_encoders.{i}.", d)
    ck["encoder.after_norm.weight"], ck["encoder.after_norm.bias"] = _ln(rng, d)
    ck["encoder.tp_norm.weight"], ck["encoder.tp_norm.bias"] = _ln(rng, d)
    # prompt embedding table; the exporter multiplies it by sqrt(d_model) (Export_SenseVoice.py:361-362)
    ck["embed.weight"] = rng.standard_normal((cfg.embed_rows, feat), dtype=np.float32) * np.float32(0.05)
    w, b = _lin(rng, cfg.vocab, d)
    ck["ctc.ctc_lo.weight"], ck["ctc.ctc_lo.bias"] = w, b
    # Frontend CMVN (FunASR `WavFrontend.cmvn`): additive means, multiplicative inverse-std.
    # Plausible magnitudes for int16-range Kaldi log-mel (ln power ~ 15..25).
    ck["frontend.cmvn_means"] = (-18.0 + rng.standard_normal((feat,), dtype=np.float32)).astype(np.float32)
    ck["frontend.cmvn_vars"] = (0.02 * (1.0 + 0.1 * rng.standard_normal((feat,), dtype=np.float32))).astype(np.float32)
    return ck


class _KeyedRng:
    """Stand-in for one sequential generator: every draw gets its own stream seeded by (seed, draw index), and the draws are
    filled by a thread pool afterwards (numpy releases the GIL) -- 1.5 G normals in seconds instead of a minute. Values differ
    from the sequential stream, so only configurations without sequential-stream goldens use it."""

    def __init__(self, seed):
        self.seed, self.jobs = seed, []

    def standard_normal(self, shape, dtype=np.float32):
        out = np.empty(shape, dtype=dtype)
        self.jobs.append(out)
        return out

    def fill(self, post):
        from concurrent.futures import ThreadPoolExecutor

        def run(i):
            a = self.jobs[i]
            np.random.default_rng([self.seed, i]).standard_normal(a.shape, dtype=a.dtype, out=a)

        with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 8)) as ex:
            list(ex.map(run, sorted(range(len(self.jobs)), key=lambda i: -self.jobs[i].size)))
        for fn in post:
            fn()


def synth_whisper_checkpoint(cfg: WhisperConfig, seed: int = 0) -> dict:
    """Random Whisper-shaped checkpoint with HF `model.*` / `proj_out` key names. Full-size configurations (d_model >= 1024) draw
    every tensor from its own stream in parallel; the small test configurations keep the single sequential stream their goldens
    were minted from."""
    if cfg.d_model >= 1024:
        return _synth_whisper_checkpoint_parallel(cfg, seed)
    rng = np.random.default_rng(seed)
    ck: dict[str, np.ndarray] = {}
    d, dff = cfg.d_model, cfg.d_ffn

    def attn(prefix):
        for name, has_bias in (("q_proj", True), ("k_proj", False), ("v_proj", True), ("out_proj", True)):
            w, b = _lin(rng, d, d, bias=has_bias)
            ck[f"{prefix}{name}.weight"] = w
            if has_bias:
                ck[f"{prefix}{name}.bias"] = b

    ck["model.encoder.conv1.weight"] = rng.standard_normal((d, cfg.n_mels, 3), dtype=np.float32) * np.float32(1.0 / np.sqrt(3 * cfg.n_mels))
    ck["model.encoder.conv1.bias"] = rng.standard_normal((d,), dtype=np.float32) * np.float32(0.1)
    ck["model.encoder.conv2.weight"] = rng.standard_normal((d, d, 3), dtype=np.float32) * np.float32(1.0 / np.sqrt(3 * d))
    ck["model.encoder.conv2.bias"] = rng.standard_normal((d,), dtype=np.float32) * np.float32(0.1)
    ck["model.encoder.embed_positions.weight"] = rng.standard_normal((cfg.max_source_positions, d), dtype=np.float32) * np.float32(0.1)
    for i in range(cfg.n_enc_layers):
        p = f"model.encoder.layers.{i}."
        ck[p + "self_attn_layer_norm.weight"], ck[p + "self_attn_layer_norm.bias"] = _ln(rng, d)
        attn(p + "self_attn.")
        ck[p + "final_layer_norm.weight"], ck[p + "final_layer_norm.bias"] = _ln(rng, d)
        ck[p + "fc1.weight"], ck[p + "fc1.bias"] = _lin(rng, dff, d, gain=1.4)
        ck[p + "fc2.weight"], ck[p + "fc2.bias"] = _lin(rng, d, dff)
    ck["model.encoder.layer_norm.weight"], ck["model.encoder.layer_norm.bias"] = _ln(rng, d)

    ck["model.decoder.embed_tokens.weight"] = rng.standard_normal((cfg.vocab, d), dtype=np.float32) * np.float32(1.0 / np.sqrt(d))
    ck["model.decoder.embed_positions.weight"] = rng.standard_normal((cfg.max_target_positions, d), dtype=np.float32) * np.float32(0.05)
    for i in range(cfg.n_dec_layers):
        p = f"model.decoder.layers.{i}."
        ck[p + "self_attn_layer_norm.weight"], ck[p + "self_attn_layer_norm.bias"] = _ln(rng, d)
        attn(p + "self_attn.")
        ck[p + "encoder_attn_layer_norm.weight"], ck[p + "encoder_attn_layer_norm.bias"] = _ln(rng, d)
        attn(p + "encoder_attn.")
        ck[p + "final_layer_norm.weight"], ck[p + "final_layer_norm.bias"] = _ln(rng, d)
        ck[p + "fc1.weight"], ck[p + "fc1.bias"] = _lin(rng, dff, d, gain=1.4)
        ck[p + "fc2.weight"], ck[p + "fc2.bias"] = _lin(rng, d, dff)
    ck["model.decoder.layer_norm.weight"], ck["model.decoder.layer_norm.bias"] = _ln(rng, d)
    # proj_out is tied to embed_tokens in HF Whisper (no separate tensor).
    return ck


def _synth_whisper_checkpoint_parallel(cfg: WhisperConfig, seed: int) -> dict:
    rng = _KeyedRng(seed)
    ck: dict[str, np.ndarray] = {}
    post = []
    d, dff = cfg.d_model, cfg.d_ffn

    def normal(key, shape, scale, offset=0.0):
        a = rng.standard_normal(shape, dtype=np.float32)
        ck[key] = a

        def fin(a=a, scale=np.float32(scale), offset=np.float32(offset)):
            a *= scale
            if offset:
                a += offset
        post.append(fin)

    def lin(prefix, out_f, in_f, bias=True, gain=1.0):
        normal(prefix + ".weight", (out_f, in_f), gain / np.sqrt(in_f))
        if bias:
            normal(prefix + ".bias", (out_f,), 0.1)

    def ln(prefix):
        normal(prefix + ".weight", (d,), 0.1, 1.0)
        normal(prefix + ".bias", (d,), 0.1)

    def attn(prefix):
        for name, has_bias in (("q_proj", True), ("k_proj", False), ("v_proj", True), ("out_proj", True)):
            lin(prefix + name, d, d, bias=has_bias)

    normal("model.encoder.conv1.weight", (d, cfg.n_mels, 3), 1.0 / np.sqrt(3 * cfg.n_mels))
    normal("model.encoder.conv1.bias", (d,), 0.1)
    normal("model.encoder.conv2.weight", (d, d, 3), 1.0 / np.sqrt(3 * d))
    normal("model.encoder.conv2.bias", (d,), 0.1)
    normal("model.encoder.embed_positions.weight", (cfg.max_source_positions, d), 0.1)
    for i in range(cfg.n_enc_layers):
        p = f"model.encoder.layers.{i}."
        ln(p + "self_attn_layer_norm")
        attn(p + "self_attn.")
        ln(p + "final_layer_norm")
        lin(p + "fc1", dff, d, gain=1.4)
        lin(p + "fc2", d, dff)
    ln("model.encoder.layer_norm")
    normal("model.decoder.embed_tokens.weight", (cfg.vocab, d), 1.0 / np.sqrt(d))
    normal("model.decoder.embed_positions.weight", (cfg.max_target_positions, d), 0.05)
    for i in range(cfg.n_dec_layers):
        p = f"model.decoder.layers.{i}."
        ln(p + "self_attn_layer_norm")
        attn(p + "self_attn.")
        ln(p + "encoder_attn_layer_norm")
        attn(p + "encoder_attn.")
        ln(p + "final_layer_norm")
        lin(p + "fc1", dff, d, gain=1.4)
        lin(p + "fc2", d, dff)
    ln("model.decoder.layer_norm")
    rng.fill(post)
    return ck


def synth_audio(kind: str, batch: int, n_samples: int, seed: int = 1234) -> np.ndarray:
    """Synthetic chunks, SURVEY.md section 8(d): int16-range values for the Kaldi
    front-ends ('kaldi'), [-1, 1] floats for Whisper/Qwen ('unit'). Shape (B, 1, L) f32."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((batch, 1, n_samples), dtype=np.float32)
    if kind == "kaldi":
        return np.round(np.clip(x * np.float32(3000.0), -32768, 32767)).astype(np.float32)
    if kind == "unit":
        return np.clip(x * np.float32(0.05), -1.0, 1.0).astype(np.float32)
    raise ValueError(kind)


def whisper_suppress_tokens(cfg: WhisperConfig) -> list:
    """Synthetic stand-in for generation_config.suppress_tokens (permanent -128 penalty, Export_Whisper.py:517-520):
    a fixed pseudo-random 1 % of the text vocabulary plus every special id except <|endoftext|> -- as in the real
    list, <|nospeech|> is among them (which is why NO_SPEECH_DETECTION re-adds +128, :338-345)."""
    rng = np.random.default_rng(77)
    n_text = min(cfg.eot_id, cfg.vocab)
    ids = set(int(i) for i in rng.choice(n_text, size=max(1, n_text // 100), replace=False))
    for i in (cfg.sot_id, cfg.transcribe_id, cfg.translate_id, cfg.no_speech_id):
        if 0 <= i < cfg.vocab:
            ids.add(int(i))
    return sorted(ids)


def whisper_begin_suppress_tokens(cfg: WhisperConfig) -> list:
    """generation_config.begin_suppress_tokens of Whisper = [" " token, <|endoftext|>]; synthetic: [220 % vocab, eot]."""
    return sorted({220 % cfg.vocab, cfg.eot_id})


def synth_paraformer_checkpoint(cfg, seed: int = 0) -> dict:
    """Random Paraformer-shaped checkpoint with FunASR state-dict names (Export_Paraformer.py:389-457,474-563)."""
    rng = np.random.default_rng(seed)
    ck: dict[str, np.ndarray] = {}
    d, dff, feat = cfg.d_model, cfg.d_ffn, cfg.feat_dim

    def enc_block(prefix, in_size):
        ck[prefix + "norm1.weight"], ck[prefix + "norm1.bias"] = _ln(rng, in_size)
        ck[prefix + "self_attn.linear_q_k_v.weight"], ck[prefix + "self_attn.linear_q_k_v.bias"] = _lin(rng, 3 * d, in_size)
        ck[prefix + "self_attn.fsmn_block.weight"] = rng.standard_normal((d, 1, cfg.fsmn_kernel), dtype=np.float32) * np.float32(0.2)
        ck[prefix + "self_attn.linear_out.weight"], ck[prefix + "self_attn.linear_out.bias"] = _lin(rng, d, d)
        ck[prefix + "norm2.weight"], ck[prefix + "norm2.bias"] = _ln(rng, d)
        ck[prefix + "feed_forward.w_1.weight"], ck[prefix + "feed_forward.w_1.bias"] = _lin(rng, dff, d, gain=1.4)
        ck[prefix + "feed_forward.w_2.weight"], ck[prefix + "feed_forward.w_2.bias"] = _lin(rng, d, dff)

    for i in range(cfg.n_enc0):
        enc_block(f"encoder.encoders0.{i}.", feat)
    for i in range(cfg.n_enc):
        enc_block(f"encoder.encoders.{i}.", d)
    ck["encoder.after_norm.weight"], ck["encoder.after_norm.bias"] = _ln(rng, d)
    ck["predictor.cif_conv1d.weight"] = rng.standard_normal((d, d, cfg.cif_kernel), dtype=np.float32) * np.float32(1.0 / np.sqrt(d * cfg.cif_kernel))
    ck["predictor.cif_conv1d.bias"] = rng.standard_normal((d,), dtype=np.float32) * np.float32(0.1)
    ck["predictor.cif_output.weight"] = rng.standard_normal((1, d), dtype=np.float32) * np.float32(2.0 / np.sqrt(d))
    ck["predictor.cif_output.bias"] = np.asarray([-0.6], dtype=np.float32)          # mean alpha ~ 0.37: a token every ~3 rows
    dd = cfg.d_dec_ffn
    for i in range(cfg.n_dec + cfg.n_dec3):
        p = f"decoder.decoders.{i}." if i < cfg.n_dec else f"decoder.decoders3.{i - cfg.n_dec}."
        ck[p + "norm1.weight"], ck[p + "norm1.bias"] = _ln(rng, d)
        ck[p + "feed_forward.w_1.weight"], ck[p + "feed_forward.w_1.bias"] = _lin(rng, dd, d, gain=1.4)
        ck[p + "feed_forward.norm.weight"], ck[p + "feed_forward.norm.bias"] = _ln(rng, dd)
        ck[p + "feed_forward.w_2.weight"] = _lin(rng, d, dd, bias=False)[0]
        if i < cfg.n_dec:
            ck[p + "norm2.weight"], ck[p + "norm2.bias"] = _ln(rng, d)
            ck[p + "norm3.weight"], ck[p + "norm3.bias"] = _ln(rng, d)
            ck[p + "self_attn.fsmn_block.weight"] = rng.standard_normal((d, 1, cfg.fsmn_kernel), dtype=np.float32) * np.float32(0.2)
            ck[p + "src_attn.linear_q.weight"], ck[p + "src_attn.linear_q.bias"] = _lin(rng, d, d)
            ck[p + "src_attn.linear_k_v.weight"], ck[p + "src_attn.linear_k_v.bias"] = _lin(rng, 2 * d, d)
            ck[p + "src_attn.linear_out.weight"], ck[p + "src_attn.linear_out.bias"] = _lin(rng, d, d)
    ck["decoder.after_norm.weight"], ck["decoder.after_norm.bias"] = _ln(rng, d)
    ck["decoder.output_layer.weight"], ck["decoder.output_layer.bias"] = _lin(rng, cfg.vocab, d)
    ck["frontend.cmvn_means"] = (-18.0 + rng.standard_normal((feat,), dtype=np.float32)).astype(np.float32)
    ck["frontend.cmvn_vars"] = (0.02 * (1.0 + 0.1 * rng.standard_normal((feat,), dtype=np.float32))).astype(np.float32)
    return ck


def synth_qwen_asr_checkpoint(cfg, seed: int = 0) -> dict:
    """Random Qwen3-ASR-shaped checkpoint with the Hugging Face state-dict names (Export_Qwen_ASR.py:311-516)."""
    rng = np.random.default_rng(seed)
    ck: dict[str, np.ndarray] = {}
    a, C, d = "thinker.audio_tower.", cfg.conv_channels, cfg.enc_d
    ck[a + "conv2d1.weight"] = (rng.standard_normal((C, 1, 3, 3), dtype=np.float32) * np.float32(1.0 / 3.0))
    ck[a + "conv2d1.bias"] = rng.standard_normal((C,), dtype=np.float32) * np.float32(0.1)
    for name in ("conv2d2", "conv2d3"):
        ck[a + name + ".weight"] = rng.standard_normal((C, C, 3, 3), dtype=np.float32) * np.float32(1.4 / np.sqrt(9 * C))
        ck[a + name + ".bias"] = rng.standard_normal((C,), dtype=np.float32) * np.float32(0.1)
    freq = (((cfg.n_mels + 1) // 2 + 1) // 2 + 1) // 2
    ck[a + "conv_out.weight"] = _lin(rng, d, C * freq, bias=False)[0]
    for i in range(cfg.n_enc_layers):
        p = f"{a}layers.{i}."
        for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
            ck[p + f"self_attn.{proj}.weight"], ck[p + f"self_attn.{proj}.bias"] = _lin(rng, d, d)
        ck[p + "self_attn_layer_norm.weight"], ck[p + "self_attn_layer_norm.bias"] = _ln(rng, d)
        ck[p + "fc1.weight"], ck[p + "fc1.bias"] = _lin(rng, cfg.enc_ffn, d, gain=1.4)
        ck[p + "fc2.weight"], ck[p + "fc2.bias"] = _lin(rng, d, cfg.enc_ffn)
        ck[p + "final_layer_norm.weight"], ck[p + "final_layer_norm.bias"] = _ln(rng, d)
    ck[a + "ln_post.weight"], ck[a + "ln_post.bias"] = _ln(rng, d)
    ck[a + "proj1.weight"], ck[a + "proj1.bias"] = _lin(rng, d, d, gain=1.4)
    ck[a + "proj2.weight"], ck[a + "proj2.bias"] = _lin(rng, cfg.d_model, d)
    t, h, hd = "thinker.model.", cfg.d_model, cfg.d_head
    ck[t + "embed_tokens.weight"] = rng.standard_normal((cfg.vocab, h), dtype=np.float32) * np.float32(0.5)
    for i in range(cfg.n_layers):
        p = f"{t}layers.{i}."
        ck[p + "self_attn.q_proj.weight"] = _lin(rng, cfg.n_heads * hd, h, bias=False)[0]
        ck[p + "self_attn.k_proj.weight"] = _lin(rng, cfg.n_kv_heads * hd, h, bias=False)[0]
        ck[p + "self_attn.v_proj.weight"] = _lin(rng, cfg.n_kv_heads * hd, h, bias=False)[0]
        ck[p + "self_attn.o_proj.weight"] = _lin(rng, h, cfg.n_heads * hd, bias=False)[0]
        ck[p + "self_attn.q_norm.weight"] = (1.0 + 0.1 * rng.standard_normal((hd,), dtype=np.float32)).astype(np.float32)
        ck[p + "self_attn.k_norm.weight"] = (1.0 + 0.1 * rng.standard_normal((hd,), dtype=np.float32)).astype(np.float32)
        ck[p + "mlp.gate_proj.weight"] = _lin(rng, cfg.d_ffn, h, bias=False, gain=1.4)[0]
        ck[p + "mlp.up_proj.weight"] = _lin(rng, cfg.d_ffn, h, bias=False)[0]
        ck[p + "mlp.down_proj.weight"] = _lin(rng, h, cfg.d_ffn, bias=False)[0]
        ck[p + "input_layernorm.weight"] = (1.0 + 0.1 * rng.standard_normal((h,), dtype=np.float32)).astype(np.float32)
        ck[p + "post_attention_layernorm.weight"] = (1.0 + 0.1 * rng.standard_normal((h,), dtype=np.float32)).astype(np.float32)
    ck[t + "norm.weight"] = (1.0 + 0.1 * rng.standard_normal((h,), dtype=np.float32)).astype(np.float32)
    ck["thinker.lm_head.weight"] = _lin(rng, cfg.vocab, h, bias=False)[0]
    return ck
