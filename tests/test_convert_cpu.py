"""CPU: the checkpoint converter (tools/convert_checkpoint.py) round-trips synthetic source-layout checkpoints through the
file formats a real conversion reads (torch pickle, safetensors, Kaldi am.mvn text) into the same arena bytes."""
import importlib.util
import os

import numpy as np
import torch

from conftest import sub
from helpers import sensevoice_setup

_spec = importlib.util.spec_from_file_location("convert_checkpoint", os.path.join(os.path.dirname(__file__), "..", "tools", "convert_checkpoint.py"))
cc = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(cc)


def _write_mvn(path, means, scales):
    with open(path, "w") as f:
        f.write("<Nnet>\n<Splice> 560 560\n[ 0 ]\n<AddShift> 560 560\n<LearnRateCoef> 0 [ " + " ".join(f"{v:.9g}" for v in means) + " ]\n")
        f.write("<Rescale> 560 560\n<LearnRateCoef> 0 [ " + " ".join(f"{v:.9g}" for v in scales) + " ]\n</Nnet>\n")


def test_sensevoice_pt_and_mvn_roundtrip(tmp_path):
    cfg, ck = sensevoice_setup("sensevoice_tiny")
    sd = {k: torch.from_numpy(v) for k, v in ck.items() if not k.startswith("frontend.")}
    torch.save({"state_dict": sd}, tmp_path / "model.pt")
    _write_mvn(tmp_path / "am.mvn", ck["frontend.cmvn_means"], ck["frontend.cmvn_vars"])
    means, scales = cc.load_kaldi_cmvn(str(tmp_path / "am.mvn"))
    assert np.allclose(means, ck["frontend.cmvn_means"], rtol=1e-6) and np.allclose(scales, ck["frontend.cmvn_vars"], rtol=1e-6)
    loaded = cc.load_state_dict(str(tmp_path / "model.pt"))
    got = cc.convert("sensevoice", loaded, str(tmp_path / "out"), 0, cmvn=(ck["frontend.cmvn_means"], ck["frontend.cmvn_vars"]))
    assert (got.d_model, got.n_enc, got.n_tp, got.vocab, got.d_ffn) == (cfg.d_model, cfg.n_enc, cfg.n_tp, cfg.vocab, cfg.d_ffn)
    info, blob = sub("ort_shim").load_model(str(tmp_path / "out" / "SenseVoiceSmall.asrmodel"))
    assert info["kind"] == "sensevoice" and np.array_equal(blob, sub("arena").build_sensevoice_arena(cfg, ck, 0))
    assert os.path.isfile(tmp_path / "out" / "ASR_Metadata.asrmodel")


def test_whisper_safetensors_roundtrip(tmp_path):
    from safetensors.numpy import save_file
    cfgm, ckm = sub("config"), sub("checkpoints")
    cfg = cfgm.whisper_tiny_test()
    ck = ckm.synth_whisper_checkpoint(cfg, 0)
    save_file({k: np.ascontiguousarray(v) for k, v in ck.items()}, str(tmp_path / "model.safetensors"))
    got = cc.convert("whisper", cc.load_state_dict(str(tmp_path / "model.safetensors")), str(tmp_path / "out"), 1)
    assert (got.d_model, got.n_enc_layers, got.n_dec_layers, got.n_mels, got.vocab) == (cfg.d_model, cfg.n_enc_layers, cfg.n_dec_layers,
                                                                                        cfg.n_mels, cfg.vocab)
    info, blob = sub("ort_shim").load_model(str(tmp_path / "out" / "Whisper.asrmodel"))
    assert info["kind"] == "whisper" and info["precision"] == 1 and blob.nbytes > 0


def test_paraformer_needs_tokens_and_cmvn(tmp_path):
    import pytest
    from test_oracle_paraformer import paraformer_setup
    cfg, ck = paraformer_setup("paraformer_tiny")
    toks = [f"t{i}" for i in range(cfg.vocab)]
    toks[0], toks[1], toks[2] = "<blank>", "<s>", "</s>"
    with pytest.raises(ValueError, match="token list"):
        cc.convert("paraformer", dict(ck), str(tmp_path / "o1"), 0)
    no_cmvn = {k: v for k, v in ck.items() if not k.startswith("frontend.")}
    with pytest.raises(ValueError, match="CMVN"):
        cc.convert("paraformer", no_cmvn, str(tmp_path / "o2"), 0, tokens=toks)
    got = cc.convert("paraformer", dict(ck), str(tmp_path / "o3"), 0, tokens=toks)
    assert (got.n_enc, got.n_dec, got.n_dec3, got.vocab) == (cfg.n_enc, cfg.n_dec, cfg.n_dec3, cfg.vocab)
    assert open(tmp_path / "o3" / "Vocab_Paraformer.txt", encoding="utf-8").read().splitlines()[:3] == ["<blank>", "<s>", "</s>"]


def test_qwen_asr_safetensors_roundtrip(tmp_path):
    import pytest
    from safetensors.numpy import save_file
    cfgm, ckm = sub("config"), sub("checkpoints")
    cfg = cfgm.qwen_asr_mid()
    ck = ckm.synth_qwen_asr_checkpoint(cfg, 0)
    save_file({k: np.ascontiguousarray(v) for k, v in ck.items()}, str(tmp_path / "model.safetensors"))
    sd = cc.load_state_dict(str(tmp_path / "model.safetensors"))
    with pytest.raises(ValueError, match="metadata"):
        cc.convert("qwen_asr", dict(sd), str(tmp_path / "o1"), 0)
    meta = {"audio_pcm_scale": "32768", "max_seq_len": "1024", "sample_rate": "16000", "special_token_ids": "{}", "supported_languages": "{}"}
    got = cc.convert("qwen_asr", sd, str(tmp_path / "out"), 0, tokens=meta)
    for f in ("enc_d", "enc_heads", "enc_ffn", "n_enc_layers", "conv_channels", "d_model", "n_heads", "n_kv_heads", "d_head", "d_ffn", "n_layers", "vocab"):
        assert getattr(got, f) == getattr(cfg, f), f
    info, blob = sub("ort_shim").load_model(str(tmp_path / "out" / "Qwen_ASR.asrmodel"))
    assert info["kind"] == "qwen_asr" and info["metadata"]["max_seq_len"] == "1024"
    assert np.array_equal(blob, sub("arena").build_qwen_asr_arena(got, ck, 0))
