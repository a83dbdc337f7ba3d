"""GPU: streaming Paraformer's two graphs through the onnxruntime-API shim, driven by the reference host's per-window sequence
(Paraformer/Streaming/Inference_Paraformer_Streaming_ONNX.py:296-449): persistent input buffers allocated once (zero-length K/V
histories, zero carried rows / CIF state / FSMN caches), `audio` updated in place per window, encoder run, decoder run when
`list_frame_len != 0`, encoder_feedback / decoder_feedback / encoder_decoder_bridge transfers. Checked against the goldens minted from
the reference's streaming classes (f32 mode)."""
import numpy as np
import pytest

from conftest import sub
from helpers import kaldi_audio, load_golden
from test_oracle_paraformer_streaming import streaming_cases, streaming_setup

pytestmark = pytest.mark.gpu

F32, TOL = 1, 1e-3


def host_loop(folder, audio, chunk):
    ort, io, ws = sub("ort_shim"), sub("ort_io"), sub("ort_shim_paraformer_streaming")
    enc, dec = ort.InferenceSession(f"{folder}/{ws.ENCODER_FILE}.onnx"), ort.InferenceSession(f"{folder}/{ws.DECODER_FILE}.onnx")
    ei, eo, di, do = (io.metadata_by_name(x) for x in (enc.get_inputs(), enc.get_outputs(), dec.get_inputs(), dec.get_outputs()))
    in_e, out_e, in_d, out_d = ([v.name for v in x] for x in (enc.get_inputs(), enc.get_outputs(), dec.get_inputs(), dec.get_outputs()))
    n_en = len([n for n in in_e if n.startswith("in_en_key_")])
    n_de = len([n for n in in_d if n.startswith("in_de_fsmn_")])
    device = ort.OrtDevice(ort.OrtDevice.cuda(), ort.OrtDevice.default_memory(), 0)
    ov = lambda a: ort.OrtValue.ortvalue_from_numpy(np.ascontiguousarray(a), "cuda", 0)
    pairs = lambda a, b, n: [(f"{a}{i}", f"{b}{i}") for i in range(n)]
    enc_fb = pairs("in_en_key_", "out_en_key_", n_en) + pairs("in_en_value_", "out_en_value_", n_en) + [
        ("in_previous_mel_features", "out_previous_mel_features"), ("in_cif_hidden", "out_cif_hidden"), ("in_cif_alphas", "out_cif_alphas"), ("start_idx", "end_idx")]
    dec_fb = pairs("in_de_fsmn_", "out_de_fsmn_", n_de) + pairs("in_de_key_", "out_de_key_", n_de) + pairs("in_de_value_", "out_de_value_", n_de)
    L = int(ei["audio"].shape[2])
    assert L == chunk
    pad = int(di["in_de_fsmn_0"].shape[2])
    bufs = {}
    for i in range(n_en):
        bufs[f"in_en_key_{i}"] = ov(io.filled_for(ei[f"in_en_key_{i}"], axes={2: 0}))
        bufs[f"in_en_value_{i}"] = ov(io.filled_for(ei[f"in_en_value_{i}"], axes={1: 0}))
    for n in ("in_previous_mel_features", "in_cif_hidden", "in_cif_alphas"):
        bufs[n] = ov(io.filled_for(ei[n]))
    bufs["start_idx"] = ov(io.scalar_for(ei["start_idx"], 0))
    bufs["audio"] = ov(io.filled_for(ei["audio"], axes={2: L}))
    dbufs = {}
    for i in range(n_de):
        dbufs[f"in_de_fsmn_{i}"] = ov(io.filled_for(di[f"in_de_fsmn_{i}"], axes={2: pad}))
        dbufs[f"in_de_key_{i}"] = ov(io.filled_for(di[f"in_de_key_{i}"], axes={2: 0}))
        dbufs[f"in_de_value_{i}"] = ov(io.filled_for(di[f"in_de_value_{i}"], axes={1: 0}))
    be, bd = enc.io_binding(), dec.io_binding()
    for n in in_e:
        be.bind_ortvalue_input(n, bufs[n])
    for n in out_e:
        be._iobinding.bind_output(n, device)
    for n, v in dbufs.items():
        bd.bind_ortvalue_input(n, v)
    audio = np.asarray(audio, dtype=np.float32).reshape(1, 1, -1)
    start, pieces, fired_counts = 0, [], []
    while True:
        bufs["audio"].update_inplace(io.array_for(ei["audio"], audio[:, :, start:start + L], axes={2: L}))
        start += L
        enc.run_with_iobinding(be)
        o_e = dict(zip(out_e, be.get_outputs()))
        n = int(o_e["list_frame_len"].numpy().reshape(-1)[0])
        fired_counts.append(n)
        more = start + L <= audio.shape[-1]
        if n:
            for name in ("encoder_out", "list_frame", "list_frame_len"):
                bd.bind_ortvalue_input(name, o_e[name])
            for name in out_d:
                bd._iobinding.bind_output(name, device)
            dec.run_with_iobinding(bd)
            o_d = dict(zip(out_d, bd.get_outputs()))
            ids = o_d["max_logit_ids"].numpy().reshape(-1)
            assert int(o_d["num_id"].numpy().reshape(-1)[0]) == ids.size == n
            pieces.append(ids)
        if more:
            for cin, pout in enc_fb:
                be.bind_ortvalue_input(cin, o_e[pout])
            for name in out_e:
                be._iobinding.bind_output(name, device)
        if n:
            if not more:
                break
            for cin, pout in dec_fb:
                bd.bind_ortvalue_input(cin, o_d[pout])
        elif not more:
            break
    return (np.concatenate(pieces) if pieces else np.zeros(0, np.int32)), fired_counts, (enc, dec, be, bd, o_e)


@pytest.mark.parametrize("fixture", ["paraformer_streaming_tiny", "paraformer_streaming_sparse"])
def test_reference_host_loop_matches_goldens(fixture, tmp_path):
    g = load_golden(fixture)
    cfg, ck = streaming_setup(g)
    chunk = int(g["chunk"])
    meta = {"sample_rate": "16000", "audio_pcm_scale": "1", "special_token_ids": '{"stop": [2]}', "supported_languages": "{}"}
    sub("ort_shim_paraformer_streaming").export_paraformer_streaming_folder(str(tmp_path), cfg, ck, meta, precision=F32, chunk=chunk)
    ws = sub("ort_shim_paraformer_streaming")
    ws._SHARED.clear()
    last = None
    for i, c in streaming_cases(g):                                  # clip after clip through the same sessions: fresh caches restart the state
        audio = kaldi_audio(c["audio_seed"], int(c["n_chunks"]) * chunk)
        toks, fired, last = host_loop(str(tmp_path), audio, chunk)
        assert fired == [int(v) for v in c["n_fired"]], i
        if (c["margin"] > 2 * TOL).all():
            assert np.array_equal(toks, c["token_ids"]), i
    enc, dec, be, bd, o_e = last
    with pytest.raises(ValueError, match="once per window|latest encoder run|stale decoder state"):
        dec.run_with_iobinding(bd)                                    # a second decoder run for the same window
    stale = dict(o_e)
    be.bind_ortvalue_input("in_cif_hidden", stale["out_cif_hidden"])
    with pytest.raises(ValueError, match="mix values|stale"):
        enc.run_with_iobinding(be)
