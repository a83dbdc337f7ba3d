"""GPU parity: the HIP streaming-Paraformer path (per-stream state in HBM, chunk steps) through the C ABI vs goldens minted from
the reference's PARAFORMER_ENCODER / PARAFORMER_DECODER and vs the streaming oracle."""
import numpy as np
import pytest

from conftest import sub
from helpers import kaldi_audio, load_golden
from oracle.paraformer_streaming_oracle import ParaformerStreamingOracle
from test_oracle_paraformer_streaming import streaming_cases, streaming_setup

pytestmark = pytest.mark.gpu

BF16, F32 = 0, 1
TOL_F32 = 1e-3


@pytest.mark.parametrize("fixture", ["paraformer_streaming_tiny", "paraformer_streaming_sparse", "paraformer_streaming_large"])
def test_f32_mode_matches_reference_goldens(fixture):
    """All clips of a fixture run as concurrent streams of ONE session (shorter clips drop out when they end); per chunk the
    encoder output, the fired-frame count, the logits and the token ids are compared."""
    g = load_golden(fixture)
    cfg, ck = streaming_setup(g)
    chunk = int(g["chunk"])
    cases = [c for _, c in streaming_cases(g)]
    audios = [kaldi_audio(c["audio_seed"], int(c["n_chunks"]) * chunk) for c in cases]
    sess = sub("engine").ParaformerStreamSession(cfg, ck, precision=F32, chunk=chunk, max_streams=4)
    sess.taps(True)
    small = cfg.d_model <= 128
    toks = [[] for _ in cases]
    for k in range(max(int(c["n_chunks"]) for c in cases)):
        live = [i for i, c in enumerate(cases) if k < int(c["n_chunks"])]
        sids = [i + 1 for i in live]                                  # stream ids need not be dense or start at 0
        out = sess.step(np.stack([audios[i][k * chunk:(k + 1) * chunk] for i in live]), sids)
        enc, logits = sess.tap("enc_out"), sess.tap("logits")
        for slot, i in enumerate(live):
            ref = cases[i]["chunks"][k]
            e = enc[16 * slot:16 * slot + 13]
            assert np.abs((e if small else e[:, ::8]) - ref["enc_out"]).max() < TOL_F32, (i, k)
            n = int(cases[i]["n_fired"][k])
            assert out[slot].size == n, (i, k)
            if n:
                lg = logits[16 * slot:16 * slot + n]
                assert np.abs((lg if small else lg[:, ::37]) - ref["logits"]).max() < TOL_F32, (i, k)
            toks[i].append(out[slot])
    for i, c in enumerate(cases):
        got = np.concatenate(toks[i]) if toks[i] else np.zeros(0, np.int32)
        if (c["margin"] > 2 * TOL_F32).all():
            assert np.array_equal(got, c["token_ids"]), i


def test_bf16_streams_vs_oracle_reset_and_errors():
    g = load_golden("paraformer_streaming_large")
    cfg, ck = streaming_setup(g)
    chunk = int(g["chunk"])
    orc = ParaformerStreamingOracle(cfg, ck, chunk=chunk)
    audio = [kaldi_audio(3400 + i, 5 * chunk) for i in range(3)]
    want = [orc.run(a) for a in audio]
    sess = sub("engine").ParaformerStreamSession(cfg, ck, precision=BF16, chunk=chunk, max_streams=3)
    sess.taps(True)
    fired_ok = total = 0
    for k in range(5):
        out = sess.step(np.stack([a[k * chunk:(k + 1) * chunk] for a in audio]), [2, 0, 1])
        enc = sess.tap("enc_out")
        for slot in range(3):
            ref = want[slot][k]
            assert np.abs(enc[16 * slot:16 * slot + 13] - ref["enc_out"]).max() < 0.15
            fired_ok += int(out[slot].size == ref["n"])
            total += 1
    assert fired_ok / total >= 0.8                              # bf16 alphas may move a fire across a chunk boundary
    # a reset stream restarts exactly like a fresh session; the others keep their state
    sess.reset(2)
    first = sess.step(audio[0][:chunk][None], [2])
    fresh = sub("engine").ParaformerStreamSession(cfg, ck, precision=BF16, chunk=chunk, max_streams=1)
    assert np.array_equal(first[0], fresh.step(audio[0][:chunk][None], [0])[0])
    with pytest.raises(Exception, match="repeated"):
        sess.step(np.stack([audio[0][:chunk]] * 2), [1, 1])
    with pytest.raises(Exception, match="invalid"):
        sess.step(audio[0][:chunk][None], [3])


def test_stream_transcriber_follows_the_reference_host_loop():
    """Host loop: noise padding to whole chunks, decoder only on fired chunks, stop ids stripped, pieces joined; several clips
    advance as concurrent streams and give what they give alone."""
    g = load_golden("paraformer_streaming_tiny")
    cfg, ck = streaming_setup(g)
    chunk = int(g["chunk"])
    ps = sub("paraformer_streaming")
    vocab = [f"t{i}" for i in range(cfg.vocab)]
    vocab[2] = "</s>"
    sess = sub("engine").ParaformerStreamSession(cfg, ck, precision=F32, chunk=chunk, max_streams=2)
    tr = ps.ParaformerStreamTranscriber(sess, vocab, stop_token_ids=[2], decode_mode="zh")
    cases = [c for _, c in streaming_cases(g)]
    clips = [kaldi_audio(c["audio_seed"], int(c["n_chunks"]) * chunk).astype(np.int16) for c in cases]
    both, stats = tr.transcribe_many(clips)
    assert stats["chunks"] == max(int(c["n_chunks"]) for c in cases) and all(r > 0 for r in stats["rtf_per_chunk"])
    for o, c, clip in zip(both, cases, clips):
        want = c["token_ids"][~np.isin(c["token_ids"], [2])]
        if (c["margin"] > 2e-3).all():
            assert np.array_equal(o["token_ids"], want) and o["text"] == "".join(vocab[i] for i in want)
        alone, _ = tr.transcribe(clip)
        assert np.array_equal(alone["token_ids"], o["token_ids"])
    padded = ps.pad_to_chunks(np.ones((1, 1, chunk + 10), np.float32), chunk, np.random.default_rng(0))
    assert padded.shape[-1] == 2 * chunk and np.all(padded[0, 0, :chunk + 10] == 1) and padded[0, 0, chunk + 10:].std() > 0.5
    assert ps.pad_to_chunks(np.ones((1, 1, 2 * chunk), np.float32), chunk).shape[-1] == 2 * chunk


def test_fused_launches_vs_per_launch_path(monkeypatch):
    """bf16 sessions run encoder layers 1.. and every decoder block of a chunk step as ONE launch each (csrc/stream_layers.hip, stream_dec.hip: clusters of four
    workgroups per stream). Same rounding points as the per-launch path (ASR_STREAM_FUSED=0), different accumulation order and bf16 attention weights: 11 streams
    (not a multiple of the 8 clusters a group holds) over 6 chunks, a subset step, a reset in between -- encoder rows, fired counts, logits of the fired rows and
    tokens of the two paths side by side, and the launch names of the fused session."""
    g = load_golden("paraformer_streaming_large")
    cfg, ck = streaming_setup(g)
    chunk, S, n_chunks = int(g["chunk"]), 11, 6
    audio = [kaldi_audio(5100 + i, n_chunks * chunk) for i in range(S)]
    monkeypatch.setenv("ASR_STREAM_FUSED", "0")
    plain = sub("engine").ParaformerStreamSession(cfg, ck, precision=BF16, chunk=chunk, max_streams=S)
    monkeypatch.delenv("ASR_STREAM_FUSED")
    fused = sub("engine").ParaformerStreamSession(cfg, ck, precision=BF16, chunk=chunk, max_streams=S)
    plain.taps(True)
    fused.taps(True)
    same = total = 0
    worst_enc = worst_logit = 0.0
    for k in range(n_chunks):
        sids = list(range(S)) if k != 3 else [9, 2, 4, 0, 7]                 # one step advances a subset only (slots != stream ids)
        if k == 4:
            plain.reset(2)
            fused.reset(2)
        chunks = np.stack([audio[s][k * chunk:(k + 1) * chunk] for s in sids])
        a, b = plain.step(chunks, sids), fused.step(chunks, sids)
        ea, eb = plain.tap("enc_out"), fused.tap("enc_out")
        la, lb = plain.tap("logits"), fused.tap("logits")
        for slot in range(len(sids)):
            worst_enc = max(worst_enc, float(np.abs(ea[16 * slot:16 * slot + 13] - eb[16 * slot:16 * slot + 13]).max()))
            total += 1
            if a[slot].size == b[slot].size:
                n = a[slot].size
                if n:
                    worst_logit = max(worst_logit, float(np.abs(la[16 * slot:16 * slot + n] - lb[16 * slot:16 * slot + n]).max()))
                same += int(np.array_equal(a[slot], b[slot]))
    print("fused vs per-launch: enc_out", worst_enc, "logits", worst_logit, "equal token lists", same, "/", total)
    assert worst_enc < 0.05 and worst_logit < 0.1                             # (measured 0.024 / 0.040)
    assert same / total >= 0.9                                                # bf16 noise may move a fire across a chunk boundary or flip a near-tie
    fused.taps(False)
    fused.profile(True)
    fused.profile_reset()
    fused.step(np.stack([audio[s][:chunk] for s in range(S)]), list(range(S)))
    names = fused.profile_read()
    fused.profile(False)
    assert names["stream_layers"]["launches"] == 1 and names["stream_dec"]["launches"] == 1
    assert sum(v["launches"] for v in names.values()) <= 24                   # (520 launches per step on the per-launch path)
