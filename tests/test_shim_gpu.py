"""GPU: the reference-shaped host loop (bind -> run_with_iobinding -> read ids) through the onnxruntime-API shim."""
import numpy as np
import pytest

from conftest import sub
from helpers import golden_cases, kaldi_audio, load_golden, sensevoice_setup

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model_folder(tmp_path_factory):
    folder = tmp_path_factory.mktemp("SenseVoice_ONNX")
    cfg, ck = sensevoice_setup("sensevoice_tiny")
    sub("sensevoice").export_sensevoice(str(folder), cfg, ck, precision=1)      # fp32 mode: token-exact vs goldens
    return str(folder)


@pytest.mark.parametrize("device_type", ["cpu", "cuda"])
def test_transcriber_matches_goldens(model_folder, device_type):
    g = load_golden("sensevoice_tiny")
    tr_by_lang = {}
    for _, c in golden_cases(g):
        code = ["auto", "zh", "en", "yue", "ja", "ko", "nospeech"][int(c["lang"])]
        tr = tr_by_lang.setdefault(code, sub("sensevoice").SenseVoiceTranscriber(model_folder, code, device_type=device_type))
        assert tr.selector_index == int(c["lang"])
        pcm = kaldi_audio(c["audio_seed"], c["n_samples"]).astype(np.int16)      # synthetic audio is integer valued
        out = tr.transcribe(pcm)
        assert out["windows"] == 1 and out["rtf"] > 0
        if (c["margin"] > 2e-3).all():
            assert np.array_equal(out["token_ids"][0], c["token_ids"])
        assert out["token_ids"][0].dtype == np.int32 and out["token_ids"][0].ndim == 1


def test_session_io_contract_and_batch_extension(model_folder):
    shim = sub("ort_shim")
    io = sub("ort_io")
    sess = shim.InferenceSession(model_folder + "/SenseVoiceSmall.onnx")
    ins, outs = sess.get_inputs(), sess.get_outputs()
    assert [(a.name, a.type) for a in ins] == [("audio", "tensor(float)"), ("language_idx", "tensor(int32)")]
    assert ins[0].shape == [1, 1, "audio_len"] and [a.name for a in outs] == ["token_ids", "num_id"]
    a0, a1 = kaldi_audio(1, 16000), kaldi_audio(2, 16000)
    single = [sess.run(None, {"audio": io.array_for(ins[0], a, axes={0: 1, 1: 1, 2: a.size}), "language_idx": np.array([2], np.int32)})
              for a in (a0, a1)]
    assert single[0][0].ndim == 1 and single[0][1].shape == (1,)
    tok, num = sess.run(["token_ids", "num_id"], {"audio": np.stack([a0, a1])[:, None, :], "language_idx": np.array([2, 2], np.int32)})
    for b in range(2):
        assert np.array_equal(tok[b, :num[b]], single[b][0])
    # sliding windows: each window is an independent run of the same graph
    tr = sub("sensevoice").SenseVoiceTranscriber(model_folder, "en")
    pcm = np.concatenate([a0, a1]).astype(np.int16)
    # window == whole clip (dynamic axis) unless SLIDING_WINDOW is set; emulate 2 windows by two calls
    w0 = tr.transcribe(pcm[:16000])["token_ids"][0]
    assert np.array_equal(w0, single[0][0])
    b = sess.io_binding()
    with pytest.raises(ValueError, match="not an input"):
        b.bind_cpu_input("audio_typo", a0)
    b.bind_cpu_input("audio", a0.reshape(1, 1, -1))
    b._iobinding.bind_output("token_ids", None)
    with pytest.raises(ValueError, match="not bound"):
        sess.run_with_iobinding(b)


# ------------------------------------------------------------------------------------------ Paraformer
def _vocab(n):
    toks = [f"t{i}" for i in range(n)]
    toks[0], toks[1], toks[2], toks[n - 1] = "<blank>", "<s>", "</s>", "<unk>"
    return toks


@pytest.mark.parametrize("device_type", ["cpu", "cuda"])
def test_paraformer_transcriber_matches_goldens(tmp_path, device_type):
    from test_oracle_paraformer import paraformer_setup
    g = load_golden("paraformer_tiny")
    cfg, ck = paraformer_setup("paraformer_tiny")
    pf = sub("paraformer")
    vocab = _vocab(cfg.vocab)
    pf.export_paraformer(str(tmp_path), cfg, ck, vocab, "zh", "zh", precision=1)
    tr = pf.ParaformerTranscriber(str(tmp_path), device_type=device_type)
    assert tr.stop_token_ids == [2] and tr.decode_mode == "zh" and tr.language == "zh"
    sess = tr.session
    assert [(a.name, a.type, a.shape) for a in sess.get_inputs()] == [("audio", "tensor(float)", [1, 1, "audio_len"])]
    assert [(a.name, a.shape) for a in sess.get_outputs()] == [("token_ids", [1, "num_token"]), ("num_id", [1])]
    checked = 0
    for _, c in golden_cases(g):
        pcm = kaldi_audio(c["audio_seed"], c["n_samples"]).astype(np.int16)
        out = tr.transcribe(pcm)
        assert out["windows"] == 1
        if c["cif_slack"] > 2e-4 and int(c["num_id"][0]) and (c["margin"] > 2e-3).all():
            ids = np.asarray(c["token_ids"]).reshape(-1)
            keep = ids[~np.isin(ids, [2])]
            assert np.array_equal(out["token_ids"][0], keep)
            assert out["text"] == "".join(vocab[i] for i in keep)
            checked += 1
    assert checked >= 1
    # raw graph contract: (1, num_token) ids and (1,) count; batch extension pads to the longest
    a = kaldi_audio(5, 16000)
    tok, num = sess.run(None, {"audio": a.reshape(1, 1, -1)})
    assert tok.ndim == 2 and tok.shape == (1, int(num[0])) and tok.dtype == np.int32
    tb, nb = sess.run(None, {"audio": np.stack([a, a])[:, None, :]})
    assert np.array_equal(tb[0, :nb[0]], tok[0]) and np.array_equal(tb[1, :nb[1]], tok[0])
