"""GPU parity: the HIP Qwen3-ASR path (log-mel, Conv2d chunk stem, windowed encoder, prompt assembly, Qwen3 decoder with an
in-place KV cache, arg-max) through the C ABI vs goldens minted from the reference's classes and vs the oracle."""
import numpy as np
import pytest

from conftest import sub
from helpers import golden_cases, load_golden
from oracle.qwen_asr_oracle import QwenAsrOracle
from test_oracle_qwen_asr import qwen_setup, unit_audio

pytestmark = pytest.mark.gpu

BF16, F32 = 0, 1
TOL_F32 = 1e-3


def _prompts(g, cases):
    head, tail, suffix = g["head_ids"].tolist(), g["tail_ids"].tolist(), g["suffix_ids"].tolist()
    pre = [head + c["query_ids"].tolist() + suffix for c in cases]
    post = [tail + c["language_tail_ids"].tolist() for c in cases]
    return pre, post


def _stepwise(sess, audios, pre, post, n_new):
    nxt, logits, ids_len = sess.prefill(audios, pre, post)
    steps_logits, steps_ids = [logits], [nxt]
    for _ in range(n_new - 1):
        nxt, logits = sess.decode(None, want_logits=True)          # ids fed back on the device
        steps_logits.append(logits)
        steps_ids.append(nxt)
    return np.stack(steps_logits, 1), np.stack(steps_ids, 1), ids_len


@pytest.mark.parametrize("fixture,order", [("qwen_asr_tiny", (0, 1, 2, 3)), ("qwen_asr_tiny", (2, 3, 1)), ("qwen_asr_tiny", (3,)),
                                           ("qwen_asr_mid", (0, 1, 2)), ("qwen_asr_mid", (2, 0)), ("qwen_asr_0p6b", (0, 1))])
def test_f32_mode_matches_reference_goldens(fixture, order):
    """Ragged batches (different clip lengths, prompts and language tails per sequence) against the reference's own outputs.
    qwen_asr_0p6b is the real Qwen3-ASR-0.6B geometry (18-layer audio tower, 28-layer decoder, 151936-entry vocabulary) with
    sub-sampled goldens: the model BASELINE.json configs[4] names."""
    g = load_golden(fixture)
    cfg, ck = qwen_setup(g)
    sess = sub("engine").QwenAsrSession.from_checkpoint(cfg, ck, precision=F32)
    allc = [c for _, c in golden_cases(g)]
    cases = [allc[i] for i in order]
    audios = [unit_audio(c["audio_seed"], c["n_samples"]) for c in cases]
    pre, post = _prompts(g, cases)
    sess.taps(True)
    got_logits, got_ids, ids_len = _stepwise(sess, audios, pre, post, int(g["n_new"]))
    assert ids_len.tolist() == [int(c["ids_len"]) for c in cases]
    sess.taps(True)
    sess.prefill(audios, pre, post)
    sub_sampled = "top1" in cases[0]
    for b, (h, c) in enumerate(zip(sess.audio_hidden([int(c["n_samples"]) for c in cases]), cases)):
        if sub_sampled:
            h = h[:, ::8]
        assert h.shape == c["audio_hidden"].shape, b
        assert np.abs(h - c["audio_hidden"]).max() < TOL_F32, b
    for b, c in enumerate(cases):
        if sub_sampled:
            assert np.abs(got_logits[b][:, ::151] - c["logits"]).max() < TOL_F32, b
            assert np.abs(np.sort(got_logits[b], axis=1)[:, -1] - c["top1"]).max() < TOL_F32, b
        else:
            assert np.abs(got_logits[b] - c["logits"]).max() < TOL_F32, b
        if (c["margin"] > 2 * TOL_F32).all():
            assert np.array_equal(got_ids[b], c["token_ids"]), b


@pytest.mark.parametrize("fixture,no_fuse", [("qwen_asr_tiny", "0"), ("qwen_asr_mid", "0"), ("qwen_asr_mid", "1")])
def test_generate_equals_stepwise_and_oracle_bf16(fixture, no_fuse, monkeypatch):
    """bf16 mode: generate() == explicit prefill / decode; logits stay within the bf16 budget of the reference; host-fed ids == device-fed.
    no_fuse = 1 runs the unfused kernels (separate RMSNorm / RoPE / per-head attention) instead of the fused decode step and the MFMA
    prefill attention."""
    monkeypatch.setenv("ASR_QWEN_NO_FUSE", no_fuse)
    g = load_golden(fixture)
    cfg, ck = qwen_setup(g)
    sess = sub("engine").QwenAsrSession.from_checkpoint(cfg, ck, precision=BF16)
    cases = [c for _, c in golden_cases(g)]
    audios = [unit_audio(c["audio_seed"], c["n_samples"]) for c in cases]
    pre, post = _prompts(g, cases)
    n_new = 8
    # teacher-forced with the reference's ids (random weights have small margins; a flipped arg-max would change the later steps)
    forced = np.stack([c["token_ids"] for c in cases])
    _, lg, _ = sess.prefill(audios, pre, post)
    steps = [lg]
    for t in range(forced.shape[1] - 1):
        steps.append(sess.decode(forced[:, t], want_logits=True)[1])
    for b, c in enumerate(cases):
        ref = c["logits"]
        assert np.abs(np.stack(steps, 1)[b] - ref).max() < 0.06 * np.abs(ref).max() + 0.05, b
    logits, ids, _ = _stepwise(sess, audios, pre, post, n_new)
    sess.prefill(audios, pre, post, want_logits=False)
    gen = sess.generate(n_new, stop_ids=())
    for b in range(len(cases)):
        assert np.array_equal(gen[b], ids[b]), b
    # a stop id ends its own sequence only, and is not emitted
    stop = int(ids[0, 2])
    sess.prefill(audios, pre, post, want_logits=False)
    gen = sess.generate(n_new, stop_ids=(stop,))
    for b in range(len(cases)):
        k = np.flatnonzero(ids[b] == stop)
        want = ids[b, :k[0]] if k.size else ids[b]
        assert np.array_equal(gen[b], want), b
    # host-fed ids
    nxt, _, _ = sess.prefill(audios, pre, post, want_logits=False)
    for t in range(1, 4):
        nxt, _ = sess.decode(nxt)
        assert np.array_equal(nxt, ids[:, t])


@pytest.mark.parametrize("prec", [BF16, F32])
def test_paged_kv_cache_equals_extents_and_returns_pages_out_of_order(prec, monkeypatch):
    """The KV cache is paged (16-position pages behind one block table for all layers, a free list on the host): logits bit for bit those of the extent layout
    (ASR_QWEN_KV_PAGED=0) over a prefill and 40 steps -- every sequence crosses page boundaries, the pool grows on the way --, the same with the free list
    scrambled (ASR_KV_PAGE_SHUFFLE=1: a table that is not monotone); sequences that finish EARLY in generate() give their pages back while the others go on
    (out-of-order completion: the third finishes first), the running ones take them over, tokens unchanged; beam search reads a paged prompt."""
    g = load_golden("qwen_asr_mid")
    cfg, ck = qwen_setup(g)
    cases = [c for _, c in golden_cases(g)]
    audios = [unit_audio(c["audio_seed"], c["n_samples"]) for c in cases]
    pre, post = _prompts(g, cases)
    n_new = 41
    eng = sub("engine")
    runs = {}
    for name, env in (("extents", {"ASR_QWEN_KV_PAGED": "0"}), ("paged", {}), ("shuffled", {"ASR_KV_PAGE_SHUFFLE": "1"})):
        for k in ("ASR_QWEN_KV_PAGED", "ASR_KV_PAGE_SHUFFLE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sess = eng.QwenAsrSession.from_checkpoint(cfg, ck, precision=prec)
        logits, ids, ids_len = _stepwise(sess, audios, pre, post, n_new)
        st = sess.kv_stats()
        # generate with per-sequence stop ids chosen from the stepwise run so that the sequences end at different steps, the LAST one of the batch first
        B = len(cases)
        ends = [30, 18, 6][:B] + [12] * max(0, B - 3)
        stops = [int(ids[b, ends[b]]) for b in range(B)]
        sess.prefill(audios, pre, post, want_logits=False)
        held0 = sess.kv_stats()["held"]
        gen = sess.generate(n_new, stop_ids=tuple(stops))
        st2 = sess.kv_stats()
        # generation in two calls: a sequence the first call finished (its pages are back in the pool) stays finished, the others go on (this used to abort the batch)
        sess.prefill(audios, pre, post, want_logits=False)
        g1 = sess.generate(10, stop_ids=tuple(stops))
        g2 = sess.generate(5, stop_ids=tuple(stops))
        if len(gen[0]) >= 14:
            assert np.array_equal(np.concatenate([g1[0], g2[0][1:]]), gen[0][:14]) and g2[0][0] == g1[0][-1]      # (a call starts by emitting the pick the last one ended on)
        if B >= 3:
            assert len(g1[2]) == len(gen[2]) < 10 and len(g2[2]) == 0
        beam = None
        if name != "shuffled":
            sess.prefill(audios, pre, post, want_logits=False)
            beam = sess.beam_search(3, 6, stop_ids=())
        runs[name] = (logits, ids, st, gen, held0, st2, beam, [int(x) for x in ids_len])
        del sess
    ext, pag, shf = runs["extents"], runs["paged"], runs["shuffled"]
    assert not ext[2]["paged"] and pag[2]["paged"] and shf[2]["paged"]
    for other in (pag, shf):
        assert np.array_equal(other[0], ext[0]) and np.array_equal(other[1], ext[1])       # logits and ids, every step, bit for bit
        for a, b in zip(other[3], ext[3]):
            assert np.array_equal(a, b)                                                     # generate() with early finishers
    for ua, ub in zip(pag[6], ext[6]):                                                     # beam search over a paged prompt: the same n-best lists
        for (ta, sa), (tb, sb) in zip(ua, ub):
            assert np.array_equal(ta, tb) and sa == sb
    # accounting: a sequence holds pages for its positions (+ at most two ahead), nothing like max_seq_len; finished sequences hold none
    pps, lens = (cfg.max_seq_len + 15) // 16, pag[7]
    assert pag[2]["held"] <= sum((n + n_new + 15) // 16 + 2 for n in lens) < len(lens) * pps
    assert pag[4] == sum((n + 1 + 15) // 16 for n in lens)        # after a prefill: the prompt's pages (+ the first generated position's)
    assert pag[5]["held"] < pag[5]["high_water"]                  # pages came back before the batch ended ...
    assert pag[5]["high_water"] <= pag[2]["high_water"]           # ... and the peak was no higher than when nobody finished early


@pytest.mark.parametrize("prec", [F32, BF16])
def test_paged_prompt_of_exactly_max_seq_len(prec, monkeypatch):
    """A prompt of exactly max_seq_len positions (prefill accepts it; max_seq_len is a multiple of the 16-position page) needs max_seq_len / 16 pages, not one
    more for a next position that does not exist (the fill loop wrote one table entry past the row -- into the NEXT sequence's row, or past the
    vector). First AND last sequence of the batch at the limit, a short one between them: logits bit for bit those of the extent layout, the page count exact,
    and a decode step fails loudly for want of room."""
    g = load_golden("qwen_asr_tiny")
    cfg, ck = qwen_setup(g)
    assert cfg.max_seq_len % 16 == 0
    eng = sub("engine")
    a = unit_audio(77, 16000)
    runs = {}
    for name, env in (("extents", {"ASR_QWEN_KV_PAGED": "0"}), ("paged", {}), ("shuffled", {"ASR_KV_PAGE_SHUFFLE": "1"})):
        for k in ("ASR_QWEN_KV_PAGED", "ASR_KV_PAGE_SHUFFLE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sess = eng.QwenAsrSession.from_checkpoint(cfg, ck, precision=prec)
        n_audio = sess.audio_tokens(len(a))
        full = [3 + (i % 7) for i in range(cfg.max_seq_len - n_audio - 2)]
        pre, post = [full, [1, 2, 3], full], [[4, 5], [4, 5], [4, 5]]
        nxt, logits, ids_len = sess.prefill([a, a, a], pre, post)
        assert [int(x) for x in ids_len] == [cfg.max_seq_len, n_audio + 5, cfg.max_seq_len]
        short_len = n_audio + 5
        runs[name] = (nxt, logits, sess.kv_stats())
        with pytest.raises(Exception, match="max_seq_len"):
            sess.decode(None)
        del sess
    pps = cfg.max_seq_len // 16
    for name in ("paged", "shuffled"):
        assert np.array_equal(runs[name][1], runs["extents"][1]) and np.array_equal(runs[name][0], runs["extents"][0])
        assert runs[name][2]["held"] == 2 * pps + (short_len + 1 + 15) // 16                # full rows: exactly a row of pages each
    assert np.array_equal(runs["paged"][1][0], runs["paged"][1][2])       # the two full-length rows are the same prompt: the same logits


def test_bad_arguments_fail_loudly():
    g = load_golden("qwen_asr_tiny")
    cfg, ck = qwen_setup(g)
    sess = sub("engine").QwenAsrSession.from_checkpoint(cfg, ck, precision=F32)
    a = unit_audio(1, 16000)
    with pytest.raises(Exception, match="prefill first"):
        sess.batch = 1
        sess.decode(None)
    with pytest.raises(Exception, match="out of range"):
        sess.prefill([a], [[cfg.vocab + 5]], [[1]])
    with pytest.raises(Exception, match="n_fft"):
        sess.prefill([a[:100]], [[1]], [[1]])
    with pytest.raises(Exception, match="max_seq_len"):
        sess.prefill([a], [[1] * cfg.max_seq_len], [[1]])


def test_transcriber_matches_reference_goldens():
    """The host mirror (prompt assembly from the metadata map, language tail, generation limit, stop handling) drives the same
    prompts the goldens were minted with: [head | system-prompt ids | suffix | audio | tail + "language " | language tail]."""
    g = load_golden("qwen_asr_tiny")
    cfg, ck = qwen_setup(g)
    sess = sub("engine").QwenAsrSession.from_checkpoint(cfg, ck, precision=F32)
    special = {"stop": [1, 521], "asr_text": [540], "audio_start": 524, "audio_end": 520, "audio_pad": 525, "im_start": 510, "im_end": 521,
               "system": 511, "user": 523, "assistant": 522, "newline": 512, "language_prefix": [530, 531]}
    langs = {"en": {"name": "English", "aliases": ["english"], "prompt_token_ids": [77, 540]},
             "zh": {"name": "Chinese", "aliases": ["chinese"], "prompt_token_ids": [78, 540]}}
    meta = {"audio_pcm_scale": "32768", "max_seq_len": str(cfg.max_seq_len), "sample_rate": "16000", "special_token_ids": special,
            "supported_languages": langs}
    q = sub("qwen_asr")
    tr = q.QwenAsrTranscriber(cfg, sess, meta)
    assert (tr.head_ids, tr.suffix_ids, tr.tail_ids) == (g["head_ids"].tolist(), g["suffix_ids"].tolist(), g["tail_ids"].tolist())
    cases = [c for _, c in golden_cases(g)]
    lang_of = {(): "", (77, 540): "English", (78, 540): "zh"}
    clips = [np.round(unit_audio(c["audio_seed"], c["n_samples"]) * 32768.0).astype(np.int16) for c in cases]
    out, stats = tr.transcribe([c for c in clips], task_prompts=[c["query_ids"].tolist() for c in cases],
                               language_prompts=[lang_of[tuple(c["language_tail_ids"].tolist())] for c in cases], max_new=int(g["n_new"]))
    assert stats["rtf"] > 0
    for b, c in enumerate(cases):
        assert out[b]["prompt_tokens"] == int(c["ids_len"]), b
    # exact-float run for the token comparison
    pre = [tr.head_ids + c["query_ids"].tolist() + tr.suffix_ids for c in cases]
    post = [tr.tail_ids + c["language_tail_ids"].tolist() for c in cases]
    sess.prefill([unit_audio(c["audio_seed"], c["n_samples"]) for c in cases], pre, post, want_logits=False)
    toks = sess.generate(int(g["n_new"]), stop_ids=tr.stop)
    for b, c in enumerate(cases):
        if (c["margin"] > 2 * TOL_F32).all():
            want = c["token_ids"].tolist()
            cut = next((i for i, t in enumerate(want) if t in tr.stop), len(want))
            assert toks[b].tolist() == want[:cut], b


def test_transcriber_beam_mode_returns_the_best_hypothesis():
    """QwenAsrTranscriber(beam_size = 3): the host mirror's beam mode hands back the first entry of the device's n-best list, cut at the
    generation limit; it refuses to combine with the repeat penalty."""
    g = load_golden("qwen_asr_tiny")
    cfg, ck = qwen_setup(g)
    sess = sub("engine").QwenAsrSession.from_checkpoint(cfg, ck, precision=F32)
    special = {"stop": [1, 521], "asr_text": [540], "audio_start": 524, "audio_end": 520, "audio_pad": 525, "im_start": 510, "im_end": 521,
               "system": 511, "user": 523, "assistant": 522, "newline": 512, "language_prefix": [530, 531]}
    meta = {"audio_pcm_scale": "32768", "max_seq_len": str(cfg.max_seq_len), "sample_rate": "16000", "special_token_ids": special,
            "supported_languages": {"en": {"name": "English", "aliases": [], "prompt_token_ids": [77, 540]}}}
    q = sub("qwen_asr")
    width, max_new = (int(v) for v in g["beam"])
    tr = q.QwenAsrTranscriber(cfg, sess, meta, beam_size=width)
    cases = [c for _, c in golden_cases(g) if "beam_tokens" in c and len(c["language_tail_ids"]) == 0 and len(c["query_ids"]) == 0]
    assert cases
    clips = [np.round(unit_audio(c["audio_seed"], c["n_samples"]) * 32768.0).astype(np.int16) for c in cases]
    out, _ = tr.transcribe(clips, max_new=max_new)
    for b, c in enumerate(cases):          # int16 round trip of the clip: compare where the golden ranking gaps allow it
        if float(c["beam_margin"]) > 0.01:
            assert out[b]["tokens"].tolist() == c["beam_tokens"][0][:int(c["beam_lens"][0])].tolist(), b
    with pytest.raises(ValueError, match="REPEAT_PENALTY"):
        q.QwenAsrTranscriber(cfg, sess, meta, beam_size=width, repeat_penalty=0.8)


@pytest.mark.parametrize("fixture", ["qwen_asr_tiny", "qwen_asr_mid"])
def test_penalty_and_sampling_heads_match_reference_goldens(fixture):
    """f32 mode: penalty-greedy (window = save_id[-range:] of what exists, decode steps only) and top-k / top-p sampling with the
    committed uniforms reproduce the reference's head classes; generate() runs the selected head; a new prefill restarts the history."""
    g = load_golden(fixture)
    cfg, ck = qwen_setup(g)
    sess = sub("engine").QwenAsrSession.from_checkpoint(cfg, ck, precision=F32)
    cases = [c for _, c in golden_cases(g) if "penalty_token_ids" in c]
    audios = [unit_audio(c["audio_seed"], c["n_samples"]) for c in cases]
    pre, post = _prompts(g, cases)
    value, rng = float(g["penalty"][0]), int(g["penalty"][1])
    n = len(cases[0]["penalty_token_ids"])
    sess.set_penalty(value, rng)
    for rep in range(2):                                           # twice: the id history restarts at the prefill
        logits, ids, _ = _stepwise(sess, audios, pre, post, n)
        for b, c in enumerate(cases):
            assert np.abs(logits[b][:, ::7] - c["penalty_logits"]).max() < TOL_F32, (rep, b)
            if (c["penalty_margin"] > 2 * TOL_F32).all():
                assert np.array_equal(ids[b], c["penalty_token_ids"]), (rep, b)
    sess.prefill(audios, pre, post, want_logits=False)
    gen = sess.generate(n, stop_ids=())
    for b in range(len(cases)):
        assert np.array_equal(gen[b], ids[b]), b
    t, k, p, rp = (float(v) for v in g["sampling_params"])
    sess.set_sampling(True, t, int(k), p, rp, seed=1)
    steps = len(cases[0]["sampling_token_ids"])
    got = []
    for step in range(steps):
        sess.set_sampling_noise(np.stack([c["sampling_noise"][step] for c in cases]))
        nxt = sess.prefill(audios, pre, post, want_logits=False)[0] if step == 0 else sess.decode(None)[0]
        got.append(nxt)
    got = np.stack(got, 1)
    for b, c in enumerate(cases):
        assert np.array_equal(got[b], c["sampling_token_ids"]), b
    # device generator: reproducible for a seed, and back to arg-max afterwards
    runs = []
    for _ in range(2):
        sess.set_sampling(True, t, int(k), p, rp, seed=99)
        sess.prefill(audios, pre, post, want_logits=False)
        runs.append(np.stack(sess.generate(6, stop_ids=())))
    assert np.array_equal(runs[0], runs[1])
    sess.set_sampling(False)
    sess.set_penalty(1.0)
    _, ids_plain, _ = _stepwise(sess, audios, pre, post, 4)
    for b, c in enumerate(cases):
        if (c["margin"] > 2 * TOL_F32).all():
            assert np.array_equal(ids_plain[b], c["token_ids"][:4]), b


def test_edge_lengths_vs_oracle():
    """f32 mode against the oracle on the lengths the goldens do not hold: the shortest clip the front-end accepts (one token), exact
    chunk / window multiples, and max_audio_len (30 s: 30 chunks, 8 windows of 4 at the tiny geometry, 390 audio tokens)."""
    g = load_golden("qwen_asr_tiny")
    cfg, ck = qwen_setup(g)
    head, tail, suffix = g["head_ids"].tolist(), g["tail_ids"].tolist(), g["suffix_ids"].tolist()
    orc = QwenAsrOracle(cfg, ck, head, tail, suffix)
    sess = sub("engine").QwenAsrSession.from_checkpoint(cfg, ck, precision=F32)
    lens = [cfg.nfft, 16000, 64000, cfg.max_audio_len, 64160]
    audios = [unit_audio(9000 + i, n) for i, n in enumerate(lens)]
    pre, post = [head + suffix], [tail]
    sess.taps(True)
    nxt, logits, ids_len = sess.prefill(audios, pre, post)
    hidden = sess.audio_hidden(lens)
    steps = [logits]
    for _ in range(2):
        steps.append(sess.decode(None, want_logits=True)[1])
    for b, a in enumerate(audios):
        r = orc.greedy(a, 3)
        assert r["ids_len"] == int(ids_len[b]) and hidden[b].shape == r["audio_hidden"].shape == (sess.audio_tokens(lens[b]), cfg.d_model), b
        assert np.abs(hidden[b] - r["audio_hidden"]).max() < TOL_F32, b
        srt = np.sort(r["logits"], axis=1)
        ok = True
        for t in range(3):
            if not ok:
                break                                               # a flipped near-tie changes the later steps
            assert np.abs(steps[t][b] - r["logits"][t]).max() < TOL_F32, (b, t)
            ok = srt[t, -1] - srt[t, -2] > 2 * TOL_F32


@pytest.mark.parametrize("prec", [F32, BF16])
def test_large_batch_equals_small_batches(prec):
    """70 sequences (more than the 64 the weight-streaming decode kernels take: the tiled kernels run instead) against the same
    clips decoded three at a time. f32: identical ids (every row is computed independently of the batch); bf16: logits within
    the operand-rounding budget."""
    g = load_golden("qwen_asr_mid")
    cfg, ck = qwen_setup(g)
    sess = sub("engine").QwenAsrSession.from_checkpoint(cfg, ck, precision=prec)
    head, tail, suffix = g["head_ids"].tolist(), g["tail_ids"].tolist(), g["suffix_ids"].tolist()
    lens = [8000 + 1600 * (i % 17) for i in range(70)]
    audios = [unit_audio(5000 + i, n) for i, n in enumerate(lens)]
    pre = [head + ([40, 41] if i % 3 == 0 else []) + suffix for i in range(70)]
    post = [tail + ([77, 540] if i % 2 else []) for i in range(70)]
    nxt, logits, ids_len = sess.prefill(audios, pre, post)
    big = [nxt.copy()]
    big_logits = [logits]
    for _ in range(3):
        n, lg = sess.decode(None, want_logits=True)
        big.append(n.copy())
        big_logits.append(lg)
    big = np.stack(big, 1)
    for lo in (0, 33, 67):
        sel = list(range(lo, lo + 3))
        n0, lg0, il = sess.prefill([audios[i] for i in sel], [pre[i] for i in sel], [post[i] for i in sel])
        assert il.tolist() == ids_len[sel].tolist()
        small, small_logits = [n0.copy()], [lg0]
        for _ in range(3):
            n, lg = sess.decode(None, want_logits=True)
            small.append(n.copy())
            small_logits.append(lg)
        small = np.stack(small, 1)
        if prec == F32:
            assert np.array_equal(small, big[sel])
            assert np.abs(small_logits[0] - big_logits[0][sel]).max() < 1e-4
        else:
            ref = small_logits[0]
            assert np.abs(ref - big_logits[0][sel]).max() < 0.06 * np.abs(ref).max() + 0.05


def test_long_prompt_bf16_vs_f32_mode():
    """30 s clips: ~400-position prompts, i.e. the MFMA causal attention walks several key chunks and the decode attention several key
    blocks. The bf16 path must stay within the operand-rounding budget of the f32 verification mode (itself pinned to the oracle)."""
    g = load_golden("qwen_asr_mid")
    cfg, ck = qwen_setup(g)
    eng = sub("engine")
    head, tail, suffix = g["head_ids"].tolist(), g["tail_ids"].tolist(), g["suffix_ids"].tolist()
    audios = [unit_audio(8800 + i, n) for i, n in enumerate((cfg.max_audio_len, 300000, 47000))]
    pre, post = [head + suffix], [tail + [77, 540]]
    out = {}
    for prec in (F32, BF16):
        s = eng.QwenAsrSession.from_checkpoint(cfg, ck, precision=prec)
        nxt, logits, ids_len = s.prefill(audios, pre, post)
        steps = [logits]
        forced = nxt.copy() if prec == F32 else out[F32][2]                # teacher-force the bf16 run with the f32 run's ids
        for t in range(3):
            ids = forced if prec == F32 else out[F32][3][t]
            nx, lg = s.decode(ids, want_logits=True)
            steps.append(lg)
            if prec == F32:
                out.setdefault("ids", []).append(nx.copy())
                forced = nx
        if prec == F32:
            out[F32] = (np.stack(steps), ids_len, nxt.copy(), [nxt.copy()] + out["ids"][:2])
        else:
            out[BF16] = (np.stack(steps), ids_len)
    assert out[F32][1].tolist() == out[BF16][1].tolist() and int(out[F32][1].max()) > 380
    ref, got = out[F32][0], out[BF16][0]
    assert np.abs(got - ref).max() < 0.06 * np.abs(ref).max() + 0.05


def _beam_arrays(hyps, max_new):
    toks = np.full((len(hyps), max_new), -1, np.int32)
    for r, (t, _) in enumerate(hyps):
        toks[r, :t.size] = t
    return toks, np.asarray([s for _, s in hyps], np.float32)


@pytest.mark.parametrize("fixture", ["qwen_asr_tiny", "qwen_asr_mid"])
def test_beam_search_f32_matches_goldens(fixture):
    """Width-3 search on the device (hypothesis rows, ancestry-following attention, device-side ranking) == the search over the reference
    classes' logits: the n-best token lists and their scores, with and without a stop id, as one ragged batch."""
    g = load_golden(fixture)
    cfg, ck = qwen_setup(g)
    sess = sub("engine").QwenAsrSession.from_checkpoint(cfg, ck, precision=F32)
    width, max_new = (int(v) for v in g["beam"])
    cases = [c for _, c in golden_cases(g) if "beam_tokens" in c]
    audios = [unit_audio(c["audio_seed"], c["n_samples"]) for c in cases]
    pre, post = _prompts(g, cases)
    sess.prefill(audios, pre, post)
    got = sess.beam_search(width, max_new)
    for b, c in enumerate(cases):
        toks, scores = _beam_arrays(got[b], max_new)
        assert np.array_equal(toks, c["beam_tokens"]), b
        assert np.abs(scores - c["beam_scores"]).max() < 5 * TOL_F32, b
    greedy = sess.generate(max_new)                                # the greedy state of the session is untouched by the search
    for b, c in enumerate(cases):
        assert np.array_equal(greedy[b], c["token_ids"][:max_new]), b
    for b, c in enumerate(cases):                                  # stop ids differ per clip: one utterance per call
        sess.prefill([audios[b]], [pre[b]], [post[b]])
        toks, scores = _beam_arrays(sess.beam_search(width, max_new, c["beamstop_stop"].tolist())[0], max_new)
        assert np.array_equal(toks, c["beamstop_tokens"]), b
        assert np.abs(scores - c["beamstop_scores"]).max() < 5 * TOL_F32, b


@pytest.mark.parametrize("prec", [F32, BF16])
def test_beam_width_one_is_greedy_and_wider_beams_score_higher(prec):
    """Size-independent properties on a batch of 6 ragged clips: width 1 == generate(); every width-5 list is sorted and holds distinct
    hypotheses; an utterance's result does not depend on its batch neighbours."""
    g = load_golden("qwen_asr_mid")
    cfg, ck = qwen_setup(g)
    sess = sub("engine").QwenAsrSession.from_checkpoint(cfg, ck, precision=prec)
    lens = [128000, 30000, 64000, 9000, 100000, 48000]
    audios = [unit_audio(5100 + i, n) for i, n in enumerate(lens)]
    head, tail, suffix = g["head_ids"].tolist(), g["tail_ids"].tolist(), g["suffix_ids"].tolist()
    pre, post = [head + suffix] * len(lens), [tail] * len(lens)
    max_new = 8
    sess.prefill(audios, pre, post)
    greedy = sess.generate(max_new)
    sess.prefill(audios, pre, post)
    one = sess.beam_search(1, max_new)
    for b in range(len(lens)):
        assert np.array_equal(one[b][0][0], greedy[b]), b
    sess.prefill(audios, pre, post)
    five = sess.beam_search(5, max_new)
    for b in range(len(lens)):
        scores = [s for _, s in five[b]]
        assert scores == sorted(scores, reverse=True), b
        assert len({tuple(t.tolist()) for t, _ in five[b]}) == 5, b           # distinct hypotheses
    if prec == F32:
        sess.prefill(audios[2:4], pre[2:4], post[2:4])
        pair = sess.beam_search(5, max_new)
        for j, b in enumerate((2, 3)):
            for (t1, s1), (t2, s2) in zip(pair[j], five[b]):
                assert np.array_equal(t1, t2) and abs(s1 - s2) < 1e-3, b


def test_beam_search_bad_arguments():
    g = load_golden("qwen_asr_tiny")
    cfg, ck = qwen_setup(g)
    eng = sub("engine")
    sess = eng.QwenAsrSession.from_checkpoint(cfg, ck, precision=F32)
    with pytest.raises(Exception, match="prefill first"):
        sess.beam_search(3, 4)
    c = [c for _, c in golden_cases(g)][0]
    pre, post = _prompts(g, [c])
    sess.prefill([unit_audio(c["audio_seed"], c["n_samples"])], pre, post)
    with pytest.raises(Exception, match="beam width"):
        sess.beam_search(9, 4)
    sess.set_penalty(0.8, 5)
    with pytest.raises(Exception, match="do not combine"):
        sess.beam_search(3, 4)


def test_beam_search_finished_utterance_stands_while_neighbours_continue():
    """Two clips, one stop-id set: the clip whose best hypothesis ends early must keep its n-best list while the other clip's search goes
    on (the device freezes a finished utterance), i.e. the batch result equals the two single-clip results; plus the edges max_new = 1
    and the widest beam."""
    g = load_golden("qwen_asr_tiny")
    cfg, ck = qwen_setup(g)
    sess = sub("engine").QwenAsrSession.from_checkpoint(cfg, ck, precision=F32)
    cases = [c for _, c in golden_cases(g) if "beam_tokens" in c]
    audios = [unit_audio(c["audio_seed"], c["n_samples"]) for c in cases]
    pre, post = _prompts(g, cases)
    stop = cases[0]["beamstop_stop"].tolist()                      # ends clip 0 after two ids; clip 1 meets it late or never
    width, max_new = 3, 8
    alone = []
    for b in range(len(cases)):
        sess.prefill([audios[b]], [pre[b]], [post[b]])
        alone.append(sess.beam_search(width, max_new, stop)[0])
    assert len({len(h[0][0]) for h in alone}) > 1                   # the clips do end at different steps
    sess.prefill(audios, pre, post)
    both = sess.beam_search(width, max_new, stop)
    for b in range(len(cases)):
        for (t1, s1), (t2, s2) in zip(both[b], alone[b]):
            assert np.array_equal(t1, t2) and abs(s1 - s2) < 1e-4, b
    sess.prefill(audios, pre, post)
    one = sess.beam_search(width, 1)
    for b, c in enumerate(cases):
        assert [len(t) for t, _ in one[b]] == [1] * width and one[b][0][0][0] == c["token_ids"][0]
    sess.prefill(audios, pre, post)
    wide = sess.beam_search(8, 4)
    for b in range(len(cases)):
        assert len(wide[b]) == 8 and [s for _, s in wide[b]] == sorted((s for _, s in wide[b]), reverse=True)


def test_0p6b_bf16_batch64_vs_golden_and_oracle():
    """Qwen3-ASR-0.6B at its real size, bf16, 64 x 8 s in one batch (the single-GPU share of BASELINE.json configs[4]): 256 x 256 prefill
    tiles, GQA attention over 8 kv heads, RMSNorm-folded skinny GEMMs at 64 rows and the 151936-column lm_head -- kernels / shapes
    the small configurations never dispatch. Slot 0 (and its duplicate in slot 63) is the reference-minted golden clip; slot 1 is
    checked against the f32 oracle run here; prefill + 3 decode steps."""
    g = load_golden("qwen_asr_0p6b")
    cfg, ck = qwen_setup(g)
    sess = sub("engine").QwenAsrSession.from_checkpoint(cfg, ck, precision=BF16)
    c0 = [c for _, c in golden_cases(g)][0]
    head, tail, suffix = g["head_ids"].tolist(), g["tail_ids"].tolist(), g["suffix_ids"].tolist()
    B, n_new = 64, int(g["n_new"])
    audios = [unit_audio(8100 + i, 128000) for i in range(B)]
    audios[0] = unit_audio(c0["audio_seed"], c0["n_samples"])
    audios[63] = audios[0].copy()
    pre = [head + c0["query_ids"].tolist() + suffix] * B
    post = [tail + c0["language_tail_ids"].tolist()] * B
    got, ids, ids_len = _stepwise(sess, audios, pre, post, n_new)
    assert int(ids_len[0]) == int(c0["ids_len"])
    assert np.array_equal(got[0], got[63]) and np.array_equal(ids[0], ids[63])          # batch invariance, bit for bit
    scale = max(float(np.abs(c0["top1"]).max()), 1.0)
    assert np.abs(got[0][:, ::151] - c0["logits"]).max() < 0.03 * scale + 0.1
    assert np.abs(np.sort(got[0], axis=1)[:, -1] - c0["top1"]).max() < 0.03 * scale + 0.1
    orc = QwenAsrOracle(cfg, ck, head, tail, suffix)
    r = orc.greedy(audios[1], 2, c0["query_ids"].tolist(), c0["language_tail_ids"].tolist())
    for t in range(2):
        assert np.abs(got[1][t] - r["logits"][t]).max() < 0.03 * scale + 0.1


def test_0p6b_two_granule_gate_up_gemm_agrees_with_one_granule(monkeypatch):
    """Round 6: at 64 sequences the gate|up GEMM (6144 columns = 384 granules, wider than the chip) takes TWO 16-column granules per workgroup (the default, what the
    test above runs); ASR_SKINNY_NT=1 is the one-granule form it replaced. Same batch, the prefill (identical: it does not take that kernel) + ONE decode step fed with the prefill's picks: its logits differ by summation order only."""
    g = load_golden("qwen_asr_0p6b")
    cfg, ck = qwen_setup(g)
    c0 = [c for _, c in golden_cases(g)][0]
    head, tail, suffix = g["head_ids"].tolist(), g["tail_ids"].tolist(), g["suffix_ids"].tolist()
    B = 64
    audios = [unit_audio(8300 + i, 128000) for i in range(B)]
    pre = [head + c0["query_ids"].tolist() + suffix] * B
    post = [tail + c0["language_tail_ids"].tolist()] * B
    out = {}
    for nt in ("2", "1"):
        monkeypatch.setenv("ASR_SKINNY_NT", nt)
        sess = sub("engine").QwenAsrSession.from_checkpoint(cfg, ck, precision=BF16)
        out[nt] = _stepwise(sess, audios, pre, post, 2)[0]
        del sess
    assert np.array_equal(out["2"][:, 0], out["1"][:, 0])
    scale = max(float(np.abs(out["1"]).max()), 1.0)
    diff = float(np.abs(out["2"] - out["1"]).max())
    print(f"qwen 0.6b, 64 sequences: two-granule vs one-granule gate|up GEMM, logits differ by {diff:.4f} (|max| {scale:.1f})")
    assert diff < 0.02 * scale + 0.05


def test_0p6b_bf16_beam5_batch64_vs_oracle_rule():
    """The Qwen3-ASR half of BASELINE.json configs[4] as written: 0.6B dimensions, bf16, beam = 5, 64 x 8 s in one batch = 320 hypothesis
    rows per step (tiled decode GEMMs at 320 rows, the ancestry-table attention over 8 kv heads, device-side top-k over 151936 columns).
    Size-independent properties on all 64 utterances + the search rule of the oracle on one of them."""
    g = load_golden("qwen_asr_0p6b")
    cfg, ck = qwen_setup(g)
    sess = sub("engine").QwenAsrSession.from_checkpoint(cfg, ck, precision=BF16)
    c0 = [c for _, c in golden_cases(g)][0]
    head, tail, suffix = g["head_ids"].tolist(), g["tail_ids"].tolist(), g["suffix_ids"].tolist()
    B, max_new = 64, 4
    audios = [unit_audio(8300 + i, 128000) for i in range(B)]
    audios[63] = audios[0].copy()
    pre = [head + c0["query_ids"].tolist() + suffix] * B
    post = [tail + c0["language_tail_ids"].tolist()] * B
    sess.prefill(audios, pre, post)
    greedy = sess.generate(max_new)
    sess.prefill(audios, pre, post)
    one = sess.beam_search(1, max_new)
    sess.prefill(audios, pre, post)
    five = sess.beam_search(5, max_new)
    for b in range(B):
        assert np.array_equal(one[b][0][0], greedy[b]), b                        # width 1 is greedy
        scores = [s for _, s in five[b]]
        assert scores == sorted(scores, reverse=True), b
        assert len({tuple(t.tolist()) for t, _ in five[b]}) == 5, b               # five distinct hypotheses
        assert five[b][0][1] >= one[b][0][1] - 1e-3, b                            # the wider beam never scores below the greedy path
    for (t0, s0), (t1, s1) in zip(five[0], five[63]):                              # batch invariance, bit for bit
        assert np.array_equal(t0, t1) and s0 == s1
    orc = QwenAsrOracle(cfg, ck, head, tail, suffix)
    ref = orc.beam(audios[1], 5, max_new, c0["query_ids"].tolist(), c0["language_tail_ids"].tolist())
    scale = max(float(np.abs(c0["top1"]).max()), 1.0)
    tol = max_new * (0.03 * scale + 0.1)                                            # summed log-probabilities of max_new bf16 steps
    assert abs(five[1][0][1] - ref[0][1]) < tol, (five[1][0][1], ref[0][1])
    ref_sets = {tuple(t.tolist()) for t, _ in ref}
    assert len(ref_sets & {tuple(t.tolist()) for t, _ in five[1]}) >= 3             # most of the oracle's list survives bf16 rank flips
