"""Run the REAL reference modules (build container only) to pin the oracle.

TEST INFRASTRUCTURE -- never imported by the product package.

`/root/reference` is a read-only mount that exists only in the build
container. Its export scripts execute on import and need packages that are not
installed (funasr, torchaudio, onnx), so -- as SURVEY.md Appendix A describes --
only the class/function *definitions* are compiled from the reference file where
it lies (nothing is copied into this repo) and run eagerly on CPU against a
stand-in module tree carrying our seeded synthetic checkpoint.

Used by `oracle/gen_golden*.py` (write tests/golden/*.npz) and by the live-reference tests in
`tests/test_oracle_*.py` (skipped when /root/reference is absent).
"""
from __future__ import annotations

import ast
import json
import os
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = os.environ.get("ASR_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "SenseVoice", "Export_SenseVoice.py"))


def _compile_defs(path: str, ns: dict, replacements=()):
    src = open(path, "r", encoding="utf-8").read()
    for old, new in replacements:
        src = src.replace(old, new)
    defs = [n for n in ast.parse(src).body if isinstance(n, (ast.ClassDef, ast.FunctionDef))]
    exec(compile(ast.Module(defs, []), os.path.basename(path) + ":defs", "exec"), ns)
    return ns


# --------------------------------------------------------------------------- SenseVoice
class _Layer(torch.nn.Module):
    def __init__(self, in_size, d, h, dk, dff, k):
        super().__init__()
        self.in_size, self.size = in_size, d
        self.norm1 = torch.nn.LayerNorm(in_size)
        self.norm2 = torch.nn.LayerNorm(d)
        sa = torch.nn.Module()
        sa.h, sa.d_k = h, dk
        sa.linear_q_k_v = torch.nn.Linear(in_size, 3 * d)
        sa.linear_out = torch.nn.Linear(d, d)
        sa.fsmn_block = torch.nn.Conv1d(d, d, k, groups=d, bias=False)
        self.self_attn = sa
        ff = torch.nn.Module()
        ff.w_1 = torch.nn.Linear(d, dff)
        ff.w_2 = torch.nn.Linear(dff, d)
        self.feed_forward = ff


def build_sensevoice_standin(cfg, ck: dict) -> torch.nn.Module:
    """FunASR-shaped module tree exposing exactly what SENSE_VOICE touches
    (SenseVoice/Export_SenseVoice.py:130-132,172-183,211-220), loaded from `ck`."""
    sv = torch.nn.Module()
    enc = torch.nn.Module()
    mk = lambda n, in_size: torch.nn.ModuleList(
        [_Layer(in_size, cfg.d_model, cfg.n_heads, cfg.d_head, cfg.d_ffn, cfg.fsmn_kernel) for _ in range(n)])
    enc.encoders0 = mk(cfg.n_enc0, cfg.feat_dim)
    enc.encoders = mk(cfg.n_enc, cfg.d_model)
    enc.tp_encoders = mk(cfg.n_tp, cfg.d_model)
    enc.after_norm = torch.nn.LayerNorm(cfg.d_model)
    enc.tp_norm = torch.nn.LayerNorm(cfg.d_model)
    sv.encoder = enc
    sv.embed = torch.nn.Embedding(cfg.embed_rows, cfg.feat_dim)
    ctc = torch.nn.Module()
    ctc.ctc_lo = torch.nn.Linear(cfg.d_model, cfg.vocab)
    sv.ctc = ctc
    sv.blank_id = cfg.blank_id
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in ck.items() if not k.startswith("frontend.")}
    missing, unexpected = sv.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return sv.eval()


def build_reference_sensevoice(cfg, ck: dict, kaldi_mel_banks_fn):
    """Instantiate the reference's SENSE_VOICE class on the synthetic checkpoint.

    Mirrors the script-level preparation at Export_SenseVoice.py:355-366:
    embed.weight *= sqrt(d_model); cmvn_vars *= sqrt(d_model)."""
    assert reference_available()
    ns = dict(torch=torch, json=json,
              kaldi=types.SimpleNamespace(get_mel_banks=kaldi_mel_banks_fn),
              LANGUAGE_PROMPT_TOKEN_IDS=tuple(cfg.language_prompt_token_ids))
    _compile_defs(os.path.join(REFERENCE_ROOT, "SenseVoice", "Export_SenseVoice.py"), ns)
    sv = build_sensevoice_standin(cfg, ck)
    factor = float(cfg.d_model) ** 0.5
    with torch.no_grad():
        sv.embed.weight.data *= factor
    means = torch.from_numpy(ck["frontend.cmvn_means"]).reshape(1, 1, -1)
    vars_ = (torch.from_numpy(ck["frontend.cmvn_vars"]) * factor).reshape(1, 1, -1)
    stft_len = cfg.n_frames(cfg.max_audio_len)
    lfr_len = (stft_len + cfg.lfr_n - 1) // cfg.lfr_n
    with torch.inference_mode():
        model = ns["SENSE_VOICE"](sv, cfg.d_model, cfg.nfft, cfg.win_length, cfg.hop_length, stft_len,
                                  cfg.n_mels, cfg.sample_rate, cfg.pre_emphasis, cfg.lfr_m, cfg.lfr_n,
                                  lfr_len, means, vars_, cfg.use_emo, False)
    return model.eval()


def reference_sensevoice_stages(model, audio: np.ndarray, language_idx: int) -> dict:
    """Re-trace SENSE_VOICE.forward (Export_SenseVoice.py:271-296) calling the
    reference's own sub-functions so intermediate tensors can be dumped."""
    with torch.inference_mode():
        a = torch.from_numpy(audio).reshape(1, 1, -1).float()
        lang = torch.tensor([language_idx], dtype=torch.int32)
        F = torch.nn.functional
        spectrum = F.conv1d(a, model.fbank_kernel, stride=model.hop_length)
        re, im = torch.split(spectrum * spectrum, model.fbank_freq, dim=1)
        power = (re + im).transpose(1, 2)
        mel = torch.matmul(power, model.mel_filters).clamp(min=model.log_eps).log()
        n_frames = mel.shape[1]
        _len = (n_frames + model.lfr_n - 1) // model.lfr_n
        idx = torch.minimum(model.indices_mel[:_len], torch.tensor(n_frames - 1))
        x = mel[:, idx].reshape(-1, model.feature_size)
        x = (x + model.cmvn_means) * model.cmvn_vars
        x = x + model.speech_position[:_len]
        x = torch.cat([model.language_embed[lang], model.system_embed, x], dim=0)
        enc_in = x.clone()
        layers = list(model.encoder.encoders0) + list(model.encoder.encoders)
        block0 = None
        for i, layer in enumerate(layers):
            x = model.sanm_block(x, layer)
            if i == 0:
                block0 = x.clone()
        x = model.layer_norm(x, model.encoder.after_norm)
        for layer in model.encoder.tp_encoders:
            x = model.sanm_block(x, layer)
        enc_out = model.layer_norm(x, model.encoder.tp_norm)
        logits = model.ctc_lo(enc_out)
        token_ids, num_id = model(a, lang)            # the reference's own end-to-end forward
    return dict(mel=mel[0].numpy(), enc_in=enc_in.numpy(), block0=block0.numpy(), enc_out=enc_out.numpy(),
                logits=logits.numpy(), token_ids=token_ids.numpy(), num_id=num_id.numpy())


# --------------------------------------------------------------------------- Whisper
def build_reference_whisper(cfg, ck: dict, use_fp16_kv=False, suppress_tokens=None):
    """Reference WHISPER_ENCODER / WHISPER_DECODER (+ embed/position shells) on a
    synthetic HF-layout checkpoint. Quantisation-only channel re-orderings are exact
    permutations (Export_Whisper.py:568-612) and are disabled."""
    assert reference_available()
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    from transformers.audio_utils import mel_filter_bank
    sys.path.insert(0, os.path.join(REFERENCE_ROOT, "Whisper"))
    try:
        from STFT_Process import STFT_Process  # the real, unmodified reference module
    finally:
        sys.path.pop(0)
    kv_dtype = torch.float16 if use_fp16_kv else torch.float32
    ns = dict(torch=torch, json=json, STFT_Process=STFT_Process,
              INPUT_AUDIO_DTYPE="F32", USE_FP16_KV=use_fp16_kv, COMPUTE_IN_F32=False, KV_DTYPE=kv_dtype,
              REORDER_DOWNPROJ_FOR_QUANT=False, REORDER_OPROJ_FOR_QUANT=False, REORDER_KEY="absmean",
              torchaudio=types.SimpleNamespace(functional=types.SimpleNamespace(
                  melscale_fbanks=lambda nf, lo, hi, nm, sr, norm, scale: torch.from_numpy(
                      mel_filter_bank(nf, nm, float(lo), float(hi), sr, norm=norm, mel_scale=scale)).float())))
    _compile_defs(os.path.join(REFERENCE_ROOT, "Whisper", "Export_Whisper.py"), ns,
                  replacements=[("hidden_states.shape[0].unsqueeze(0)", "hidden_states.shape[0]")])
    hf_cfg = WhisperConfig(d_model=cfg.d_model, encoder_layers=cfg.n_enc_layers, decoder_layers=cfg.n_dec_layers,
                           encoder_attention_heads=cfg.n_heads, decoder_attention_heads=cfg.n_heads,
                           encoder_ffn_dim=cfg.d_ffn, decoder_ffn_dim=cfg.d_ffn, num_mel_bins=cfg.n_mels,
                           vocab_size=cfg.vocab, max_source_positions=cfg.max_source_positions,
                           max_target_positions=cfg.max_target_positions, pad_token_id=0, bos_token_id=1,
                           eos_token_id=2, decoder_start_token_id=cfg.sot_id if cfg.sot_id < cfg.vocab else 1)
    model = WhisperForConditionalGeneration(hf_cfg).float().eval()
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in ck.items()}
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("embed_positions" in m or "proj_out" in m for m in missing), missing
    stft = STFT_Process("stft_B_power", n_fft=cfg.nfft, win_length=cfg.nfft, hop_len=cfg.hop_length, max_frames=0,
                        window_type="hann", pad_mode="reflect", center_pad=True, input_scale=1.0,
                        drop_last_frame=True).eval()
    with torch.inference_mode():
        enc = ns["WHISPER_ENCODER"](model.model, stft, cfg.nfft, cfg.n_mels, cfg.sample_rate, cfg.n_dec_layers).eval()
        dec = ns["WHISPER_DECODER"](model, list(suppress_tokens) if suppress_tokens is not None else None, cfg.n_dec_layers).eval()
    return dict(ns=ns, model=model, encoder=enc, decoder=dec, stft=stft, kv_dtype=kv_dtype)


# --------------------------------------------------------------------------- Paraformer (non-streaming)
def build_paraformer_standin(cfg, ck: dict) -> torch.nn.Module:
    """FunASR-shaped module tree with exactly the attributes PARAFORMER touches
    (Paraformer/Non-Streaming/Export_Paraformer.py:367-472)."""
    d, k = cfg.d_model, cfg.fsmn_kernel

    def enc_layer(in_size):
        l = _Layer(in_size, d, cfg.n_heads, cfg.d_head, cfg.d_ffn, k)
        l.self_attn.pad_fn = torch.nn.ConstantPad1d(((k - 1) // 2, (k - 1) // 2), 0.0)
        l.feed_forward.activation = torch.nn.ReLU()
        return l

    def dec_ff():
        ff = torch.nn.Module()
        ff.w_1 = torch.nn.Linear(d, cfg.d_dec_ffn)
        ff.norm = torch.nn.LayerNorm(cfg.d_dec_ffn)
        ff.w_2 = torch.nn.Linear(cfg.d_dec_ffn, d, bias=False)
        ff.activation = torch.nn.ReLU()
        return ff

    def dec_layer(full):
        l = torch.nn.Module()
        l.norm1 = torch.nn.LayerNorm(d)
        l.feed_forward = dec_ff()
        if full:
            l.norm2, l.norm3 = torch.nn.LayerNorm(d), torch.nn.LayerNorm(d)
            sa = torch.nn.Module()
            sa.fsmn_block = torch.nn.Conv1d(d, d, k, groups=d, bias=False)
            sa.pad_fn = torch.nn.ConstantPad1d(((k - 1) // 2, (k - 1) // 2), 0.0)
            l.self_attn = sa
            ca = torch.nn.Module()
            ca.h, ca.d_k = cfg.n_heads, cfg.d_head
            ca.linear_q, ca.linear_k_v, ca.linear_out = torch.nn.Linear(d, d), torch.nn.Linear(d, 2 * d), torch.nn.Linear(d, d)
            l.src_attn = ca
        return l

    m = torch.nn.Module()
    enc = torch.nn.Module()
    enc.encoders0 = torch.nn.Sequential(*[enc_layer(cfg.feat_dim) for _ in range(cfg.n_enc0)])
    enc.encoders = torch.nn.Sequential(*[enc_layer(d) for _ in range(cfg.n_enc)])
    enc.after_norm = torch.nn.LayerNorm(d)
    enc.embed = None
    m.encoder = enc
    pred = torch.nn.Module()
    pred.pad = torch.nn.ConstantPad1d((cfg.cif_kernel // 2, cfg.cif_kernel // 2), 0.0)
    pred.cif_conv1d = torch.nn.Conv1d(d, d, cfg.cif_kernel)
    pred.cif_output = torch.nn.Linear(d, 1)
    pred.tail_threshold = cfg.tail_threshold
    m.predictor = pred
    dec = torch.nn.Module()
    dec.decoders = torch.nn.Sequential(*[dec_layer(True) for _ in range(cfg.n_dec)])
    dec.decoders3 = torch.nn.Sequential(*[dec_layer(False) for _ in range(cfg.n_dec3)])
    dec.after_norm = torch.nn.LayerNorm(d)
    dec.output_layer = torch.nn.Linear(d, cfg.vocab)
    dec.embed = None
    m.decoder = dec
    sd = {k_: torch.from_numpy(np.ascontiguousarray(v)) for k_, v in ck.items() if not k_.startswith("frontend.")}
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return m.eval()


def build_reference_paraformer(cfg, ck: dict, kaldi_mel_banks_fn):
    """The reference's PARAFORMER + KaldiFbank classes on a synthetic checkpoint (script-level prep :580-584:
    cmvn_vars *= sqrt(d_model))."""
    assert reference_available()
    path = os.path.join(REFERENCE_ROOT, "Paraformer", "Non-Streaming", "Export_Paraformer.py")
    ns = dict(torch=torch, json=json, F=torch.nn.functional, kaldi=types.SimpleNamespace(get_mel_banks=kaldi_mel_banks_fn),
              DECODER_CROSS_KV_GROUP_SIZE=4)
    _compile_defs(path, ns)
    standin = build_paraformer_standin(cfg, ck)
    factor = float(cfg.d_model) ** 0.5
    means = torch.from_numpy(ck["frontend.cmvn_means"]).reshape(1, 1, -1)
    vars_ = (torch.from_numpy(ck["frontend.cmvn_vars"]) * factor).reshape(1, 1, -1)
    lfr_len = (cfg.n_frames(cfg.max_audio_len) + cfg.lfr_n - 1) // cfg.lfr_n
    with torch.inference_mode():
        fbank = ns["KaldiFbank"](cfg.nfft, cfg.win_length, cfg.hop_length, cfg.n_mels, cfg.sample_rate, "hamming", cfg.pre_emphasis).eval()
        model = ns["PARAFORMER"](standin, fbank, cfg.n_mels, cfg.lfr_m, cfg.lfr_n, lfr_len, means, vars_, cfg.d_model)
    return model.eval()


def reference_paraformer_stages(model, audio: np.ndarray) -> dict:
    """Run the reference's own forward; intermediates come from forward hooks (no re-statement)."""
    taps = {}
    hooks = [
        model.fbank_model.register_forward_hook(lambda m, i, o: taps.__setitem__("mel", o[0].detach().clone())),
        model.encoder.after_norm.register_forward_hook(lambda m, i, o: taps.__setitem__("enc_out", o[0].detach().clone())),
        model.predictor.cif_output.register_forward_hook(lambda m, i, o: taps.__setitem__("alpha_logit", o[0, :, 0].detach().clone())),
        model.decoder.output_layer.register_forward_hook(lambda m, i, o: taps.__setitem__("logits", o[0].detach().clone())),
    ]
    try:
        with torch.inference_mode():
            token_ids, num_id = model(torch.from_numpy(audio).reshape(1, 1, -1).float())
    finally:
        for h in hooks:
            h.remove()
    n = int(num_id[0])
    return dict(mel=taps["mel"].numpy(), enc_out=taps["enc_out"].numpy(), alphas=torch.sigmoid(taps["alpha_logit"]).numpy(),
                logits=taps["logits"].numpy()[:max(n, 1)], token_ids=token_ids[0].numpy().astype(np.int32),
                num_id=num_id.numpy().astype(np.int32))


# --------------------------------------------------------------------------- Paraformer (streaming)
class _SinusoidalEmbed(torch.nn.Module):
    """FunASR SinusoidalPositionEncoder.encode as the streaming exporter calls it (positions (1, n) -> (1, n, depth))."""

    def encode(self, positions, depth, dtype=torch.float32):
        positions = positions.to(torch.float32)
        log_inc = torch.log(torch.tensor([10000.0])) / (depth / 2 - 1)
        inv_ts = torch.exp(torch.arange(depth / 2).float() * (-log_inc)).reshape(1, -1)
        st = positions.reshape(1, -1, 1) * inv_ts.reshape(1, 1, -1)
        return torch.cat([torch.sin(st), torch.cos(st)], dim=2).to(dtype)


def build_reference_paraformer_streaming(cfg, ck: dict, kaldi_mel_banks_fn, chunk=8000, look_back_encoder=4, look_back_decoder=1,
                                         max_continue=502):
    """The reference's PARAFORMER_ENCODER / PARAFORMER_DECODER (streaming) on a synthetic checkpoint, float32 caches
    (USE_FP16_KV = False). Script-level prep :564-566: cmvn_vars *= sqrt(d_model)."""
    assert reference_available()
    from torch.onnx.operators import reshape_from_tensor_shape
    path = os.path.join(REFERENCE_ROOT, "Paraformer", "Streaming", "Export_Paraformer_Streaming.py")
    ns = dict(torch=torch, np=np, json=json, kaldi=types.SimpleNamespace(get_mel_banks=kaldi_mel_banks_fn),
              reshape_from_tensor_shape=reshape_from_tensor_shape, CACHE_DTYPE=torch.float32, COMPUTE_IN_F32=False,
              PREVENT_F16_OVERFLOW=False, USE_FP16_KV=False)
    _compile_defs(path, ns)
    standin = build_paraformer_standin(cfg, ck)
    standin.encoder.embed = _SinusoidalEmbed()
    for layer in standin.decoder.decoders:
        layer.self_attn.kernel_size = cfg.fsmn_kernel
    n_frames = (chunk - cfg.win_length) // cfg.hop_length + 1
    B = ((cfg.lfr_m - 1) // 2 + n_frames) // cfg.lfr_n + 1
    factor = float(cfg.d_model) ** 0.5
    means = torch.from_numpy(ck["frontend.cmvn_means"]).reshape(1, 1, -1)
    vars_ = (torch.from_numpy(ck["frontend.cmvn_vars"]) * factor).reshape(1, 1, -1)
    with torch.inference_mode():
        fbank = ns["KaldiFbank"](cfg.nfft, cfg.win_length, cfg.hop_length, cfg.n_mels, cfg.sample_rate, "hamming", cfg.pre_emphasis).eval()
        enc = ns["PARAFORMER_ENCODER"](standin, fbank, n_frames, cfg.lfr_m, cfg.lfr_n, B, means, vars_, cfg.d_model, cfg.d_model, cfg.feat_dim,
                                       0, B, B // 2, look_back_encoder, max_continue).eval()
        dec = ns["PARAFORMER_DECODER"](standin, B, B // 2, look_back_decoder, cfg.d_model, cfg.n_dec).eval()
    return dict(encoder=enc, decoder=dec, B=B, C=B // 2, chunk=chunk)


def reference_paraformer_streaming_run(ref, cfg, audio: np.ndarray) -> list:
    """Drive the two reference modules like Inference_Paraformer_Streaming_ONNX.py:401-449 (decoder state advances only when a
    frame fired)."""
    enc, dec = ref["encoder"], ref["decoder"]
    H, hd, d = cfg.n_heads, cfg.d_head, cfg.d_model
    n_en, n_de = cfg.n_enc0 + cfg.n_enc, cfg.n_dec
    keys = [torch.zeros(H, hd, 0) for _ in range(n_en)]
    vals = [torch.zeros(H, 0, hd) for _ in range(n_en)]
    prev = torch.zeros(1, ref["C"], cfg.feat_dim)
    cif_hidden, cif_alphas, start = torch.zeros(1, 1, d), torch.zeros(1), torch.zeros(1, dtype=torch.int64)
    de_fsmn = [torch.zeros(1, d, cfg.fsmn_kernel - 1) for _ in range(n_de)]
    de_k = [torch.zeros(H, hd, 0) for _ in range(n_de)]
    de_v = [torch.zeros(H, 0, hd) for _ in range(n_de)]
    out = []
    with torch.inference_mode():
        for s in range(0, audio.size - ref["chunk"] + 1, ref["chunk"]):
            chunk = torch.from_numpy(audio[s:s + ref["chunk"]]).reshape(1, 1, -1).float()
            res = enc(*[k.clone() for k in keys], *[v.clone() for v in vals], prev.clone(), cif_hidden.clone(), cif_alphas.clone(), start.clone(), chunk)
            keys, vals = [t.clone() for t in res[:n_en]], [t.clone() for t in res[n_en:2 * n_en]]
            prev, cif_hidden, cif_alphas, start, enc_out, list_frame, n = res[2 * n_en:]
            prev, cif_hidden, cif_alphas, start = prev.clone(), cif_hidden.clone(), cif_alphas.clone(), start.clone()
            n = int(n)
            rec = dict(enc_out=enc_out[0].numpy().copy(), n=n, list_frame=list_frame[0].numpy().copy(), cif_alphas=float(cif_alphas.reshape(-1)[0]))
            if n:
                taps = {}
                hook = dec.decoder.output_layer.register_forward_hook(lambda m, i, o: taps.__setitem__("logits", o[0].detach().clone()))
                try:
                    r2 = dec(*de_fsmn, *de_k, *de_v, enc_out, list_frame, torch.tensor(n))
                finally:
                    hook.remove()
                de_fsmn, de_k, de_v = [t.clone() for t in r2[:n_de]], [t.clone() for t in r2[n_de:2 * n_de]], [t.clone() for t in r2[2 * n_de:3 * n_de]]
                rec["token_ids"] = r2[-2].reshape(-1).numpy().astype(np.int32)
                rec["logits"] = taps["logits"].numpy()
            else:
                rec["token_ids"] = np.zeros(0, np.int32)
            out.append(rec)
    return out


# --------------------------------------------------------------------------- Qwen3-ASR
def build_reference_qwen_asr(cfg, ck: dict, head_ids, tail_ids, query_suffix_ids, max_seq_len=1024):
    """The reference's QWEN3_ASR_ENCODER / ROTARY_MASK_PREFILL / ROTARY_MASK_DECODE / DECODER_EMBED / DECODER_MAIN on a synthetic
    checkpoint (float32 KV cache, quantisation-only channel re-orderings disabled: they are exact permutations)."""
    assert reference_available()
    from typing import Dict, List, Sequence, Tuple
    from transformers import AutoConfig, AutoModel, AutoTokenizer
    from transformers.activations import ACT2FN
    from transformers.audio_utils import mel_filter_bank
    from transformers.configuration_utils import PretrainedConfig
    from transformers.generation import GenerationMixin
    from transformers.modeling_layers import GradientCheckpointingLayer
    from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS
    from transformers.modeling_utils import PreTrainedModel
    from torch.onnx import symbolic_helper
    import importlib.util
    spec = importlib.util.spec_from_file_location("qwen_stft_process", os.path.join(REFERENCE_ROOT, "Qwen_ASR", "STFT_Process.py"))
    stft_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(stft_mod)
    ns = dict(torch=torch, np=np, json=json, F=torch.nn.functional, nn=torch.nn, Tensor=torch.Tensor, Dict=Dict, List=List, Sequence=Sequence,
              Tuple=Tuple, AutoConfig=AutoConfig, AutoModel=AutoModel, AutoTokenizer=AutoTokenizer, ACT2FN=ACT2FN,
              PretrainedConfig=PretrainedConfig, GenerationMixin=GenerationMixin, GradientCheckpointingLayer=GradientCheckpointingLayer,
              ROPE_INIT_FUNCTIONS=ROPE_INIT_FUNCTIONS, PreTrainedModel=PreTrainedModel, symbolic_helper=symbolic_helper,
              STFT_Process=stft_mod.STFT_Process, INPUT_AUDIO_DTYPE="F32", USE_FP16_KV=False, COMPUTE_IN_F32=False,
              ROTARY_STORAGE_DTYPE=torch.float32, REORDER_DOWNPROJ_FOR_QUANT=False, REORDER_OPROJ_FOR_QUANT=False, REORDER_KEY="absmean",
              MAX_INPUT_AUDIO_LENGTH=cfg.max_audio_len, _MODEL_SAMPLE_RATE=cfg.sample_rate, _MODEL_WINDOW_TYPE="hann", _MODEL_NUM_MELS=cfg.n_mels,
              _MODEL_NFFT_STFT=cfg.nfft, _MODEL_WINDOW_LENGTH=cfg.nfft, _MODEL_HOP_LENGTH=cfg.hop_length, _MODEL_AUDIO_PCM_SCALE=32768,
              torchaudio=types.SimpleNamespace(functional=types.SimpleNamespace(
                  melscale_fbanks=lambda n_freqs, f_min, f_max, n_mels, sample_rate, norm, mel_scale: torch.from_numpy(
                      mel_filter_bank(n_freqs, n_mels, float(f_min), float(f_max), sample_rate, norm=norm, mel_scale=mel_scale)).float())))
    # transformers resolves the classes' string annotations through sys.modules[cls.__module__]: give the definitions a module
    mod = types.ModuleType("qwen_asr_reference_defs")
    mod.__dict__.update(ns)
    sys.modules[mod.__name__] = mod
    ns = mod.__dict__
    _compile_defs(os.path.join(REFERENCE_ROOT, "Qwen_ASR", "Export_Qwen_ASR.py"), ns)
    audio_cfg = dict(num_mel_bins=cfg.n_mels, encoder_layers=cfg.n_enc_layers, encoder_attention_heads=cfg.enc_heads, encoder_ffn_dim=cfg.enc_ffn,
                     d_model=cfg.enc_d, max_source_positions=cfg.max_source_positions, n_window=cfg.n_window, output_dim=cfg.d_model,
                     n_window_infer=cfg.n_window_infer, downsample_hidden_size=cfg.conv_channels, activation_function="gelu")
    text_cfg = dict(vocab_size=cfg.vocab, hidden_size=cfg.d_model, intermediate_size=cfg.d_ffn, num_hidden_layers=cfg.n_layers,
                    num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, head_dim=cfg.d_head, rms_norm_eps=cfg.rms_eps,
                    rope_theta=cfg.rope_theta, tie_word_embeddings=False, max_position_embeddings=4096)
    config = ns["Qwen3ASRConfig"](thinker_config=dict(audio_config=audio_cfg, text_config=text_cfg))
    with torch.inference_mode():
        model = ns["Qwen3ASRForConditionalGeneration"](config).float().eval()
        sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in ck.items()}
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        assert all("positional_embedding" in m or "inv_freq" in m for m in missing), missing
        ns["refresh_non_persistent_buffers"](model, config.thinker_config.text_config)
        enc = ns["QWEN3_ASR_ENCODER"](model.thinker.audio_tower, model.thinker.model.embed_tokens, head_ids, tail_ids, query_suffix_ids).eval()
        embed = ns["QWEN3_ASR_DECODER_EMBED"](model).eval()
        rot_p = ns["QWEN3_ASR_ROTARY_MASK_PREFILL"](model.thinker.model, max_seq_len).eval()
        rot_d = ns["QWEN3_ASR_ROTARY_MASK_DECODE"](model.thinker.model, max_seq_len).eval()
        main = ns["QWEN3_ASR_DECODER_MAIN"](model, cfg.n_heads, cfg.n_kv_heads, cfg.d_head, cfg.n_layers, cfg.d_model).eval()
    return dict(ns=ns, encoder=enc, embed=embed, rotary_prefill=rot_p, rotary_decode=rot_d, main=main, model=model)
