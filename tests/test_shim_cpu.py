"""CPU: the onnxruntime-API shim's host logic and the ORT_IO restatement (no compute calls)."""
import importlib.util
import json
import os

import numpy as np
import pytest

from conftest import sub
from helpers import sensevoice_setup

REF_ORT_IO = "/root/reference/ORT_IO.py"


class _Meta:
    def __init__(self, name, shape, type_):
        self.name, self.shape, self.type = name, shape, type_


def test_ort_io_contract():
    io = sub("ort_io")
    audio = _Meta("audio", [1, 1, "audio_len"], "tensor(float)")
    a = io.array_for(audio, np.arange(6, dtype=np.int16), axes={0: 1, 1: 1, 2: 6})
    assert a.shape == (1, 1, 6) and a.dtype == np.float32 and a.flags.c_contiguous
    assert io.array_for(audio, np.zeros((1, 1, 9))).shape == (1, 1, 9)          # dynamic axis taken from the value
    with pytest.raises(ValueError, match="provide axes"):
        io.array_for(_Meta("kv", ["batch", 20, 64, "hist"], "tensor(float16)"), np.zeros((2, 20)))
    lang = _Meta("language_idx", [1], "tensor(int32)")
    assert io.filled_for(lang, 3, axes={0: 1}).tolist() == [3]
    assert io.filled_for(_Meta("k", ["batch", 20, 64, 0], "tensor(float16)"), axes={0: 2}).shape == (2, 20, 64, 0)
    assert io.scalar_for(_Meta("n", [], "tensor(int64)"), 7).shape == () and io.scalar_for(lang, 7).tolist() == [7]
    assert io.resolve_shape(audio, symbols={"audio_len": 5}) == (1, 1, 5)
    assert io.numpy_dtype("tensor(float16)") == np.float16 and io.is_dynamic_dim("x") and not io.is_dynamic_dim(np.int64(3))
    with pytest.raises(KeyError):
        io.numpy_dtype("tensor(string)")
    meta = {"supported_languages": json.dumps({"en": {"name": "English", "aliases": ["English", "en-US"], "selector_index": 2},
                                               "zh": {"aliases": ["Chinese", "中文"], "selector_index": 1}}),
            "special_token_ids": json.dumps({"eos": [1, 2]}), "ids": "1,2,,3", "sample_rate": "16000"}
    cat = io.load_supported_languages(meta)
    assert io.resolve_supported_language(cat, " EN ")[0] == "en" and io.resolve_supported_language(cat, "中文")[0] == "zh"
    assert cat["zh"]["name"] == "zh" and cat["zh"]["prompt_token_ids"] == []
    with pytest.raises(ValueError, match="Unsupported language"):
        io.resolve_supported_language(cat, "fr")
    assert io.metadata_int_list(meta, "ids") == [1, 2, 3] and io.metadata_int(meta, "sample_rate") == 16000
    assert io.load_special_token_ids(meta) == {"eos": [1, 2]}


@pytest.mark.skipif(not os.path.isfile(REF_ORT_IO), reason="reference only mounted in the build container")
def test_ort_io_matches_reference_module():
    spec = importlib.util.spec_from_file_location("ref_ort_io", REF_ORT_IO)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    io = sub("ort_io")
    rng = np.random.default_rng(0)
    metas = [_Meta("a", [1, 1, "L"], "tensor(float)"), _Meta("b", ["batch", 20, 64, "hist"], "tensor(float16)"),
             _Meta("c", [1], "tensor(int64)"), _Meta("d", [], "tensor(int32)")]
    for m in metas:
        for axes in (None, {0: 2}, {0: 1, 1: 1, 2: 4, 3: 0}):
            for val in (rng.standard_normal(4), np.zeros((2, 20, 64, 3)), 5):
                got = want = None
                try:
                    want = ref.array_for(m, val, axes=axes)
                except Exception as e:
                    want = type(e)
                try:
                    got = io.array_for(m, val, axes=axes)
                except Exception as e:
                    got = type(e)
                if isinstance(want, type):
                    assert got is want, (m.name, axes, np.shape(val), got, want)
                else:
                    assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want)
        assert np.array_equal(io.scalar_for(m, 3), ref.scalar_for(m, 3))


def test_model_bundle_and_metadata_session(tmp_path):
    shim, arena = sub("ort_shim"), sub("arena")
    cfg, ck = sensevoice_setup("sensevoice_tiny")
    meta = shim.sensevoice_metadata(cfg)
    blob = arena.build_sensevoice_arena(cfg, ck, 0)
    shim.save_model(str(tmp_path / "SenseVoiceSmall.asrmodel"), "sensevoice", cfg.to_dict(), blob, {}, 0)
    shim.save_model(str(tmp_path / "ASR_Metadata.asrmodel"), "metadata", None, None, meta)
    info, back = shim.load_model(str(tmp_path / "SenseVoiceSmall.onnx"))        # .onnx resolves to the sibling bundle
    assert info["kind"] == "sensevoice" and np.array_equal(back, blob)
    sess = shim.InferenceSession(str(tmp_path / "ASR_Metadata.onnx"), sess_options=shim.SessionOptions())
    m = sess.get_modelmeta().custom_metadata_map
    assert m["sample_rate"] == "16000" and m["audio_pcm_scale"] == "1"          # Kaldi front-end: int16-range floats
    io = sub("ort_io")
    cat = io.load_supported_languages(m)
    assert [cat[c]["selector_index"] for c in ("auto", "zh", "en", "yue", "ja", "ko", "nospeech")] == list(range(7))
    assert io.resolve_supported_language(cat, "Cantonese")[0] == "yue"
    assert sess.get_providers() == ["MI355XExecutionProvider"]
    with pytest.raises(ValueError, match="not an .asrmodel"):
        (tmp_path / "x.onnx").write_bytes(b"\x08\x07onnx-protobuf")
        shim.InferenceSession(str(tmp_path / "x.onnx"))


def test_install_as_onnxruntime_exposes_the_names_the_scripts_import():
    import sys
    shim = sub("ort_shim")
    saved = {k: sys.modules.get(k) for k in ("onnxruntime", "onnxruntime.capi", "onnxruntime.capi._pybind_state")}
    try:
        shim.install_as_onnxruntime()
        import onnxruntime
        from onnxruntime.capi import _pybind_state as C
        for name in ("InferenceSession", "SessionOptions", "RunOptions", "OrtValue", "ExecutionMode", "GraphOptimizationLevel"):
            assert hasattr(onnxruntime, name)
        dev = C.OrtDevice(C.OrtDevice.cuda(), C.OrtDevice.default_memory(), 0)
        assert dev.device_id == 0
        o = onnxruntime.SessionOptions()
        o.add_session_config_entry("session.set_denormal_as_zero", "1")
        r = onnxruntime.RunOptions()
        r.add_run_config_entry("disable_synchronize_execution_providers", "0")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_prepare_audio_and_window_plan():
    sv = sub("sensevoice")
    pcm = (np.arange(-5, 5, dtype=np.int16) * 1000).reshape(1, 1, -1)
    a = sv.prepare_audio_input(pcm, "F32", audio_pcm_scale=1)
    assert a.dtype == np.float32 and np.array_equal(a, pcm.astype(np.float32))          # no /32768 for the Kaldi front-end
    n = sv.prepare_audio_input(pcm, "F32", audio_pcm_scale=1, normalise=True)
    assert abs(float(np.sqrt(np.mean(n * n))) - 4096.0) < 1.0
    assert sv.prepare_audio_input(pcm, "INT16", audio_pcm_scale=1).dtype == np.int16
    x = np.ones((1, 1, 25), np.float32)
    assert sv.plan_windows(x, 25, 10, 10).shape[-1] == 30      # ceil((25-10)/10)+1 = 3 windows
    assert sv.plan_windows(x, 25, 40, 40).shape[-1] == 40
    assert sv.plan_windows(x, 25, 25, 25).shape[-1] == 25


def test_paraformer_decode_modes():
    pf = sub("paraformer")
    assert pf.decode_tokens(["hel@@", "lo", "wor@@", "ld"], "en") == "hello world"
    assert pf.decode_tokens(["你", "好"], "zh") == "你好"
    sp, langs = pf.build_tokenizer_metadata(["<blank>", "<s>", "</s>", "a", "<unk>"], "en", "en")
    assert sp == {"blank": 0, "eos": 2, "stop": [2], "unknown": 4, "bos": 1} and langs["en"]["decode_mode"] == "en"


def test_whisper_graph_io_names_follow_the_merged_graphs():
    """The reference host derives its binding plan from the NAMES and ORDER of the merged graphs' inputs / outputs
    (Whisper/Inference_Whisper_ONNX.py:323-392); `graph_io` must keep them for every role x strategy."""
    wg, cfg = sub("ort_shim_whisper"), sub("config").whisper_tiny_test()
    L = cfg.n_dec_layers
    for strategy in wg.STRATEGIES:
        for role in ("probe_prefill", "prefill", "decode"):
            ins, outs = wg.graph_io(cfg, role, strategy)
            names_in, names_out = [n for n, _, _ in ins], [n for n, _, _ in outs]
            assert names_in[:2 * L] == [f"in_de_key_layer_{i}" for i in range(L)] + [f"in_de_value_layer_{i}" for i in range(L)]
            assert not names_in[2 * L].startswith("in_de_")                       # the state block ends where the planner's loop breaks
            assert names_out[:2 * L] == [f"out_de_key_layer_{i}" for i in range(L)] + [f"out_de_value_layer_{i}" for i in range(L)]
            assert ("audio" in names_in) == (role == "probe_prefill")
            assert any(n.startswith("en_key_") for n in names_in) == (role != "probe_prefill")
            assert any(n.startswith("encoder_en_key_") for n in names_out) == (role == "probe_prefill")
            assert ("logits" in names_out) == (role != "decode")
            assert ("decode_kv_seq_len" in names_in) == (role == "decode") and ("prefill_history_len" in names_in) == (role != "decode")
            assert names_out[-1] == ("decode_kv_seq_len_next" if role == "decode" else "prefill_kv_seq_len")
            assert wg.MAX_OUT[strategy] in names_out
            if strategy == "penalty_greedy":
                assert "greedy_save_id_in" in names_in and "greedy_save_id_out" in names_out
                assert ("penalty_penalty_value" in names_in) == (role == "decode") == ("penalty_save_id_in" in names_in)
            if strategy == "sampling":
                assert all(n in names_in for n in wg.SAMPLING_INPUTS) and "sampling_previous_ids" in names_in and "sampling_save_id_out" in names_out
            if strategy == "greedy":
                assert not any(n.startswith(("greedy_", "penalty_", "sampling_")) for n in names_in + names_out)
    assert wg.graph_io(cfg, "no_speech", "greedy") == ([("logits", ["batch", cfg.vocab], np.float32)], [("no_speech_prob", ["batch"], np.float32)])


def test_qwen_graph_io_names_and_embedding_handles():
    """Qwen3-ASR host plans positionally after a leading `past_*` block (Inference_Qwen_ASR_ONNX.py:315-366); the Embed graph's output is
    an id-carrying tensor that survives the host's `array_for` conversions."""
    wq, cfg, io = sub("ort_shim_qwen"), sub("config").qwen_asr_tiny(), sub("ort_io")
    L = cfg.n_layers
    for strategy in wq.STRATEGIES:
        for role in ("prefill", "decode"):
            ins, outs = wq.graph_io(cfg, role, strategy)
            ni, no = [n for n, _, _ in ins], [n for n, _, _ in outs]
            assert ni[:2 * L] == [f"past_key_{i}" for i in range(L)] + [f"past_value_{i}" for i in range(L)] and not ni[2 * L].startswith("past_")
            assert no[:2 * L] == [f"present_key_{i}" for i in range(L)] + [f"present_value_{i}" for i in range(L)]
            assert len(no) - 2 * L == (2 if strategy == "greedy" else 3)             # max id, (save ids), kv_seq_len -- taken positionally
            assert ("hidden_states" in ni) == (role == "decode") and ("audio" in ni) == (role == "prefill")
            if strategy == "penalty_greedy":
                assert "penalty_greedy_save_id_in" in ni and ("penalty_save_id_in" in ni) == (role == "decode")
            if strategy == "sampling":
                assert "sampling_previous_ids" in ni and all(n in ni for n in wq.SAMPLING_INPUTS)
    e = wq.ids_as_embedding([3, 151935, 0], cfg.d_model)
    meta = sub("ort_shim").NodeArg("query_embed", [1, "query_len", cfg.d_model], np.float32)
    assert wq.embedding_as_ids(io.array_for(meta, e, axes={0: 1, 1: 3}), "query_embed") == [3, 151935, 0]
    assert wq.embedding_as_ids(np.zeros((1, 0, cfg.d_model), np.float32), "query_embed") == []


def test_paraformer_streaming_graph_io_names():
    """The streaming host wires the two graphs by NAME (encoder_feedback / decoder_feedback / encoder_decoder_bridge,
    Inference_Paraformer_Streaming_ONNX.py:296-338) and reads static sizes from the metadata (audio length, FSMN pad)."""
    ws, cfg = sub("ort_shim_paraformer_streaming"), sub("config").paraformer_large()
    ins, outs = ws.graph_io(cfg, "encoder", 8000, 9, 4)
    ni, no = {n: (s, d) for n, s, d in ins}, {n: (s, d) for n, s, d in outs}
    n_en = cfg.n_enc0 + cfg.n_enc
    for i in (0, n_en - 1):
        assert ni[f"in_en_key_{i}"][0][:2] == [4, 128] and isinstance(ni[f"in_en_key_{i}"][0][2], str)
        assert isinstance(ni[f"in_en_value_{i}"][0][1], str) and f"out_en_value_{i}" in no
    assert f"in_en_key_{n_en}" not in ni and ni["audio"][0] == [1, 1, 8000] and ni["in_previous_mel_features"][0] == [1, 4, 560]
    for a, b in (("in_previous_mel_features", "out_previous_mel_features"), ("in_cif_hidden", "out_cif_hidden"), ("in_cif_alphas", "out_cif_alphas"), ("start_idx", "end_idx")):
        assert ni[a] == no[b]
    assert no["encoder_out"][0] == [1, 13, 512] and no["list_frame_len"] == ([], np.int64)
    dins, douts = ws.graph_io(cfg, "decoder", 8000, 9, 4)
    di, do = {n: (s, d) for n, s, d in dins}, {n: (s, d) for n, s, d in douts}
    assert di["in_de_fsmn_0"][0] == [1, 512, 10] and f"in_de_fsmn_{cfg.n_dec - 1}" in di and f"in_de_fsmn_{cfg.n_dec}" not in di
    for n in ("encoder_out", "list_frame", "list_frame_len"):
        assert di[n] == no[n]
    assert do["max_logit_ids"][1] == np.int32 and do["num_id"] == ([1], np.int32)
