"""GPU parity of BASELINE.json configs[4]'s "mixed batch": Qwen3-ASR beam search and streaming-Paraformer chunk steps running CONCURRENTLY on one GPU
(two sessions, two HIP streams, two host threads -- what bench.py --workload mixed only times). Both families are checked against their reference-minted
goldens while the other one is in flight, pass after pass: kernels of the two sessions interleave on the chip, so any state shared by accident
(workspaces, tickets, per-process statics, the block kernel's co-residency fallback) would show as a wrong token, a wrong score or a hang."""
import threading

import numpy as np
import pytest

from conftest import sub
from helpers import golden_cases, kaldi_audio, load_golden
from test_oracle_paraformer_streaming import streaming_cases, streaming_setup
from test_oracle_qwen_asr import qwen_setup, unit_audio
from test_qwen_asr_gpu import _beam_arrays, _prompts

pytestmark = pytest.mark.gpu

F32 = 1
TOL = 1e-3


def test_qwen_beam_search_and_paraformer_streams_concurrently_match_their_goldens():
    eng = sub("engine")
    # ---- Qwen3-ASR: width-3 beam search of the mid-size fixture (hypothesis rows, ancestry attention, device-side ranking)
    gq = load_golden("qwen_asr_mid")
    qcfg, qck = qwen_setup(gq)
    qsess = eng.QwenAsrSession.from_checkpoint(qcfg, qck, precision=F32)
    width, max_new = (int(v) for v in gq["beam"])
    qcases = [c for _, c in golden_cases(gq) if "beam_tokens" in c]
    qaudios = [unit_audio(c["audio_seed"], c["n_samples"]) for c in qcases]
    pre, post = _prompts(gq, qcases)
    # ---- streaming Paraformer: every clip of the large fixture as concurrent streams, chunk by chunk
    gp = load_golden("paraformer_streaming_large")
    pcfg, pck = streaming_setup(gp)
    chunk = int(gp["chunk"])
    pcases = [c for _, c in streaming_cases(gp)]
    paudios = [kaldi_audio(c["audio_seed"], int(c["n_chunks"]) * chunk) for c in pcases]
    psess = eng.ParaformerStreamSession(pcfg, pck, precision=F32, chunk=chunk, max_streams=4)
    n_iter = 6
    errors = []

    def qwen_worker():
        try:
            for it in range(n_iter):
                qsess.prefill(qaudios, pre, post)
                got = qsess.beam_search(width, max_new)
                for b, c in enumerate(qcases):
                    toks, scores = _beam_arrays(got[b], max_new)
                    assert np.array_equal(toks, c["beam_tokens"]), ("qwen tokens", it, b)
                    assert np.abs(scores - c["beam_scores"]).max() < 5 * TOL, ("qwen scores", it, b)
        except Exception as e:          # noqa: BLE001 -- reported on the main thread
            errors.append(e)

    def stream_worker():
        try:
            for it in range(n_iter):
                for i in range(len(pcases)):
                    psess.reset(i + 1)
                toks = [[] for _ in pcases]
                for k in range(max(int(c["n_chunks"]) for c in pcases)):
                    live = [i for i, c in enumerate(pcases) if k < int(c["n_chunks"])]
                    out = psess.step(np.stack([paudios[i][k * chunk:(k + 1) * chunk] for i in live]), [i + 1 for i in live])
                    for slot, i in enumerate(live):
                        assert out[slot].size == int(pcases[i]["n_fired"][k]), ("stream fire count", it, i, k)
                        toks[i].append(out[slot])
                for i, c in enumerate(pcases):
                    got = np.concatenate(toks[i]) if toks[i] else np.zeros(0, np.int32)
                    if (c["margin"] > 2 * TOL).all():
                        assert np.array_equal(got, c["token_ids"]), ("stream tokens", it, i)
        except Exception as e:          # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=qwen_worker), threading.Thread(target=stream_worker)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a worker hung"
    assert not errors, errors[:2]


@pytest.mark.parametrize("share_rule", ["1", "0"])
def test_bf16_cluster_launches_beside_a_qwen_session(share_rule, monkeypatch):
    """configs[4] in the precision the bench times it in: full-size streaming Paraformer (bf16: the two CLUSTER launches per chunk step, four workgroups per stream
    that wait on each other) next to a bf16 Qwen3-ASR session doing prefill + beam search in a loop on the same GPU (the f32 test above never runs
    a cluster kernel). share_rule "1" (default): while the Qwen session is inside a call the streaming steps take the per-launch path (counted: stream_stats), which
    leaves CUs to the other tenant; checked against the ORACLE with the bf16 bars of tests/test_paraformer_streaming_gpu.py (encoder rows < 0.15, fired counts right
    in >= 80 % of the chunk steps). share_rule "0": the cluster launches run under contention; every step snapshots first (another session exists), so a give-up is
    restored and redone -- the run must neither hang nor fail, and where no give-up happened the tokens are bit for bit those of the same session run alone.
    Qwen3-ASR's beams must be those of its own solo run, pass after pass."""
    from oracle.paraformer_streaming_oracle import ParaformerStreamingOracle
    BF16 = 0
    monkeypatch.setenv("ASR_STREAM_SHARE", share_rule)
    eng = sub("engine")
    gq = load_golden("qwen_asr_mid")
    qcfg, qck = qwen_setup(gq)
    width, max_new = (int(v) for v in gq["beam"])
    qcases = [c for _, c in golden_cases(gq) if "beam_tokens" in c]
    qaudios = [unit_audio(c["audio_seed"], c["n_samples"]) for c in qcases]
    pre, post = _prompts(gq, qcases)
    gp = load_golden("paraformer_streaming_large")
    pcfg, pck = streaming_setup(gp)
    chunk, S, n_chunks = int(gp["chunk"]), 16, 6
    paudio = [kaldi_audio(9300 + i, n_chunks * chunk) for i in range(S)]
    orc = ParaformerStreamingOracle(pcfg, pck, chunk=chunk)
    want = [orc.run(a) for a in paudio[:4]]                                     # (the oracle is the slow part: four of the sixteen streams)

    def stream_pass(sess, taps):
        sess.reset(-1)
        out, encs = [], []
        for k in range(n_chunks):
            out.append(sess.step(np.stack([a[k * chunk:(k + 1) * chunk] for a in paudio]), list(range(S))))
            if taps:
                encs.append(sess.tap("enc_out")[:16 * 4].copy())
        return out, encs

    psess = eng.ParaformerStreamSession(pcfg, pck, precision=BF16, chunk=chunk, max_streams=S)
    solo, _ = stream_pass(psess, False)                                          # nobody else computing: the cluster launches (snapshots only if sessions of earlier tests still exist)
    assert psess.stream_stats()["shared_steps"] == 0 and psess.stream_stats()["giveups"] == 0
    qsess = eng.QwenAsrSession.from_checkpoint(qcfg, qck, precision=BF16)
    qsess.prefill(qaudios, pre, post)
    qsolo = qsess.beam_search(width, max_new)
    psess.taps(True)
    errors, passes = [], []
    stop = threading.Event()

    def qwen_worker():
        try:
            while not stop.is_set():
                qsess.prefill(qaudios, pre, post)
                got = qsess.beam_search(width, max_new)
                for b in range(len(qcases)):
                    for (ta, sa), (tb, sb) in zip(got[b], qsolo[b]):
                        assert np.array_equal(ta, tb) and sa == sb, ("qwen beams changed under concurrency", b)
        except Exception as e:          # noqa: BLE001
            errors.append(e)

    def stream_worker():
        try:
            for _ in range(4):
                passes.append(stream_pass(psess, True))
        except Exception as e:          # noqa: BLE001
            errors.append(e)
        finally:
            stop.set()

    threads = [threading.Thread(target=qwen_worker), threading.Thread(target=stream_worker)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    stop.set()
    assert not any(t.is_alive() for t in threads), "a worker hung"
    assert not errors, errors[:2]
    st = psess.stream_stats()
    print("share rule", share_rule, "stream stats", st)
    fired_ok = total = 0
    for out, encs in passes:
        for k in range(n_chunks):
            for s in range(4):
                assert np.abs(encs[k][16 * s:16 * s + 13] - want[s][k]["enc_out"]).max() < 0.15
                fired_ok += int(out[k][s].size == want[s][k]["n"])
                total += 1
    assert fired_ok / total >= 0.8
    if share_rule == "1":
        assert st["shared_steps"] > 0                                              # the other tenant was seen and given room
    else:
        assert st["shared_steps"] == 0 and st["snapshots"] > 0
        if st["giveups"] == 0:
            bad = [(pi, k, si, solo[k][si].tolist(), out[k][si].tolist()) for pi, (out, _) in enumerate(passes) for k in range(n_chunks) for si in range(S)
                   if not np.array_equal(out[k][si], solo[k][si])]
            assert not bad, ("(pass, chunk, stream, solo tokens, tokens under contention)", bad[:8], len(bad))
