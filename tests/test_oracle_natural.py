"""CPU: the oracles are pinned on NON-NOISE audio too (oracle/natural_audio.py: digital silence, DC + clipped square, an 80 dB chirp, three clips of
the reference's example speech) against goldens minted from the reference's own classes on these clips (oracle/gen_golden_natural.py). The clips take
the branches a flat spectrum never reaches: the log floor of the Kaldi front-end, Whisper's clamp / global-maximum rule. Tiny geometries (seconds on CPU);
the full-size fixtures of the same clips are the GPU suite's (tests/test_natural_audio_gpu.py)."""
import numpy as np

from helpers import load_golden, sensevoice_setup
from oracle import natural_audio as na

F32_TOL = 1e-4


def test_clips_are_what_the_generator_saw():
    clips = na.load_clips()
    assert list(clips) == list(na.ORDER)
    assert clips["silence"].dtype == np.int16 and not clips["silence"].any() and clips["silence"].size == na.N8
    assert int(clips["dc_clip"].max()) == 32767 and int(clips["dc_clip"].min()) == -28000       # saturated on top, DC-shifted below
    c = clips["chirp80"].astype(np.float64)
    assert 75.0 < 20 * np.log10(np.abs(c[:1600]).max() / max(np.abs(c[-1600:]).max(), 1.0)) < 85.0
    for name, (_, _, n) in na.SPEECH.items():
        assert clips[name].size == n and clips[name].dtype == np.int16 and np.abs(clips[name].astype(np.int32)).max() > 1000, name


def test_sensevoice_oracle_on_natural_clips():
    from oracle.sensevoice_oracle import SenseVoiceOracle
    g = load_golden("sensevoice_tiny_natural")
    cfg, ck = sensevoice_setup("sensevoice_tiny", int(g["ckpt_seed"]))
    orc = SenseVoiceOracle(cfg, ck)
    floor_seen = False
    for n, pcm in na.load_clips().items():
        st = orc.stages(na.kaldi_input(pcm), int(g[n + "_lang"]))
        floor_seen |= bool((g[n + "_mel"] < -15.9).any())
        for k in ("mel", "enc_in", "logits"):
            assert st[k].shape == g[n + "_" + k].shape, (n, k)
            assert np.abs(st[k] - g[n + "_" + k]).max() < F32_TOL * max(1.0, np.abs(g[n + "_" + k]).max()), (n, k)
        safe = g[n + "_margin"] > 1e-3
        assert np.array_equal(st["frame_ids"][safe], g[n + "_frame_ids"][safe]), n
        if safe.all():
            assert np.array_equal(st["token_ids"], g[n + "_token_ids"]), n
    assert floor_seen                           # the clamp(FLT_EPS) branch is really in the fixture


def test_paraformer_oracle_on_natural_clips():
    from oracle.paraformer_oracle import ParaformerOracle
    from test_oracle_paraformer import paraformer_setup
    g = load_golden("paraformer_tiny_natural")
    cfg, ck = paraformer_setup(str(g["cfg_name"]), int(g["ckpt_seed"]))
    orc = ParaformerOracle(cfg, ck)
    for n, pcm in na.load_clips().items():
        st = orc.stages(na.kaldi_input(pcm))
        assert np.abs(st["alphas"] - g[n + "_alphas"]).max() < F32_TOL, n
        assert np.abs(st["enc_out"] - g[n + "_enc_out"]).max() < F32_TOL, n
        if g[n + "_cif_slack"] > 2e-4:
            nid = int(g[n + "_num_id"][0])
            assert np.array_equal(st["num_id"], g[n + "_num_id"]), n
            assert np.abs(st["logits"][:max(nid, 1)] - g[n + "_logits"]).max() < F32_TOL, n


def test_whisper_oracle_on_natural_clips():
    from oracle.whisper_oracle import WhisperOracle
    from test_oracle_whisper import whisper_setup
    g = load_golden("whisper_tiny_natural")
    cfg, ck, sup, beg = whisper_setup(str(g["cfg_name"]), int(g["ckpt_seed"]))
    orc = WhisperOracle(cfg, ck, sup, beg)
    clips = na.load_clips()
    for n, pcm in clips.items():
        # the front-end alone (Export_Whisper.py:424-427), not only through 2 layers of encoder: log-mel with the 1e-10 clamp and the clip-wise max - 8 floor
        import torch
        mel = orc.log_mel(torch.as_tensor(na.unit_input(pcm))).t().numpy()
        assert mel.shape == g[n + "_mel"].shape and np.abs(mel - g[n + "_mel"]).max() < F32_TOL, n
        r = orc.greedy([na.unit_input(pcm)], [g["prompt"].tolist()], int(g["n_new"]))
        k, v = r["cross"][0]
        assert np.abs(k - g[n + "_cross_k"]).max() < F32_TOL and np.abs(v - g[n + "_cross_v"]).max() < F32_TOL, n
        assert np.abs(r["logits"][0] - g[n + "_logits"]).max() < F32_TOL, n
        if (g[n + "_margin"] > 1e-3).all():
            assert np.array_equal(r["token_ids"][0], g[n + "_token_ids"]), n
    # silence next to speech in ONE batch: the global-maximum clamp is per clip, so batching must not change either
    pair = orc.greedy([na.unit_input(clips["silence"]), na.unit_input(clips["zh_1"])], [g["prompt"].tolist()] * 2, 2)
    for b, n in enumerate(("silence", "zh_1")):
        assert np.abs(pair["logits"][b] - g[n + "_logits"][:2]).max() < F32_TOL, n


def test_qwen_asr_oracle_on_natural_clips():
    from oracle.qwen_asr_oracle import QwenAsrOracle
    from test_oracle_qwen_asr import qwen_setup
    g = load_golden("qwen_asr_tiny_natural")
    cfg, ck = qwen_setup(g)
    orc = QwenAsrOracle(cfg, ck, g["head_ids"].tolist(), g["tail_ids"].tolist(), g["suffix_ids"].tolist())
    for n, pcm in na.load_clips().items():
        r = orc.greedy(na.unit_input(pcm), int(g["n_new"]), [], [77, 540])
        assert r["audio_hidden"].shape == g[n + "_audio_hidden"].shape and r["ids_len"] == int(g[n + "_ids_len"]), n
        assert np.abs(r["audio_hidden"] - g[n + "_audio_hidden"]).max() < F32_TOL, n
        assert np.abs(r["logits"] - g[n + "_logits"]).max() < 5 * F32_TOL, n
        if (g[n + "_margin"] > 1e-3).all():
            assert np.array_equal(r["token_ids"], g[n + "_token_ids"]), n


def test_paraformer_streaming_oracle_on_natural_clips():
    """The streaming graphs on composite clips with digital silence between speech (oracle/natural_audio.py: streaming_clips): carried rows, K/V / FSMN
    histories and the CIF state across chunks in which the front-end sits at its log floor."""
    from oracle.paraformer_streaming_oracle import ParaformerStreamingOracle
    from test_oracle_paraformer_streaming import streaming_cases, streaming_setup
    g = load_golden("paraformer_streaming_tiny_natural")
    cfg, ck = streaming_setup(g)
    orc = ParaformerStreamingOracle(cfg, ck, chunk=int(g["chunk"]))
    clips = na.streaming_clips()
    assert list(clips) == [str(x) for x in g["clips"]]
    for (i, c), (name, pcm) in zip(streaming_cases(g), clips.items()):
        assert pcm.size == int(c["n_chunks"]) * na.STREAM_CHUNK
        recs = orc.run(na.kaldi_input(pcm))
        assert [r["n"] for r in recs] == c["n_fired"].tolist(), name
        assert np.abs(np.asarray([r["cif_alphas"] for r in recs]) - c["cif_alphas"]).max() < 2e-4, name
        small = cfg.d_model <= 128                       # (the generator keeps every 8th column of the rows / every 37th logit of larger geometries)
        for r, k in zip(recs, c["chunks"]):
            assert np.abs((r["enc_out"] if small else r["enc_out"][:, ::8]) - k["enc_out"]).max() < 2e-4, name
            if r["n"]:
                assert np.abs((r["logits"] if small else r["logits"][:, ::37]) - k["logits"]).max() < 2e-4, name
        if (c["margin"] > 1e-3).all():
            assert np.array_equal(np.concatenate([r["token_ids"] for r in recs]), c["token_ids"]), name
