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
