"""GPU parity: the HIP Whisper path (encoder, cross-KV, KV-cache decoder, greedy heads) through the C ABI vs goldens
minted from the reference's WHISPER_ENCODER / WHISPER_DECODER and vs the oracle."""
import numpy as np
import pytest

from conftest import sub
from helpers import golden_cases, load_golden
from oracle.whisper_oracle import WhisperOracle
from test_oracle_whisper import unit_audio, whisper_setup

pytestmark = pytest.mark.gpu

BF16, F32 = 0, 1
LOGIT_TOL_F32 = 1e-3


def _session(cfg_name, prec):
    cfg, ck, sup, beg = whisper_setup(cfg_name)
    sess = sub("engine").WhisperSession.from_checkpoint(cfg, ck, precision=prec, suppress_tokens=sup, begin_suppress_tokens=beg)
    return cfg, ck, sup, beg, sess


@pytest.mark.parametrize("fixture", ["whisper_tiny", "whisper_mid", "whisper_large_v3"])
def test_f32_mode_matches_reference_goldens(fixture):
    """One ragged batch: encoder cross-KV, prefill logits, per-step decode logits and greedy ids. whisper_large_v3 is the real
    geometry (d = 1280, 32 + 32 layers, 20 heads, 51866-entry vocabulary) on a 30 s + 8 s ragged batch: BASELINE.json configs[2]'s
    model, sub-sampled goldens minted from the reference's own classes."""
    g = load_golden(fixture)
    cfg, ck, sup, beg, sess = _session(str(g["cfg_name"]), F32)
    cases = [c for _, c in golden_cases(g)]
    n_new = int(g["n_new"])
    audios = [unit_audio(c["audio_seed"], c["n_samples"]) for c in cases]
    sess.taps(True)
    npos = sess.encode(audios)
    assert [int(t) for t in npos] == [cfg.n_enc_pos(int(c["n_samples"])) for c in cases]
    sub_sampled = "top1" in cases[0]
    for (k, v), c in zip(sess.cross_kv(npos), cases):
        if sub_sampled:
            k, v = k[:, ::5, ::16], v[:, ::5, ::16]
        assert np.abs(k - c["cross_k"]).max() < LOGIT_TOL_F32 and np.abs(v - c["cross_v"]).max() < LOGIT_TOL_F32
    prompts = np.stack([c["prompt"] for c in cases])
    nxt, logits = sess.prefill(prompts)
    steps_logits, steps_ids = [logits], [nxt]
    for _ in range(n_new - 1):
        nxt, logits = sess.decode(None, want_logits=True)          # ids fed back on the device
        steps_logits.append(logits)
        steps_ids.append(nxt)
    got_logits = np.stack(steps_logits, 1)                         # (B, n_new, V)
    got_ids = np.stack(steps_ids, 1)
    for b, c in enumerate(cases):
        lg = got_logits[b][:, ::53] if sub_sampled else got_logits[b]
        assert np.abs(lg - c["logits"]).max() < LOGIT_TOL_F32, b
        if sub_sampled:
            assert np.abs(np.sort(got_logits[b], axis=1)[:, -1] - c["top1"]).max() < LOGIT_TOL_F32
        if (c["margin"] > 2 * LOGIT_TOL_F32).all():
            assert np.array_equal(got_ids[b], c["token_ids"]), b


def test_generate_equals_stepwise_and_oracle_bf16():
    """bf16 mode: generate() == explicit prefill/decode; logits stay within the bf16 budget of the oracle."""
    cfg, ck, sup, beg, sess = _session("whisper_mid_test", BF16)
    orc = WhisperOracle(cfg, ck, sup, beg)
    audios = [unit_audio(31, 64000), unit_audio(32, 25600)]
    prompt = [cfg.sot_id, cfg.first_language_id, cfg.transcribe_id, cfg.no_timestamps_id]
    ref = orc.greedy(audios, [prompt, prompt], 6)
    sess.encode(audios)
    nxt, logits = sess.prefill(np.array([prompt, prompt], np.int32))
    for b in range(2):
        err = np.abs(logits[b] - ref["logits"][b][0]).max()
        assert err < 0.15, err
    ids = [nxt.copy()]
    for _ in range(5):
        nxt, _ = sess.decode(None)
        ids.append(nxt.copy())
    stepwise = np.stack(ids, 1)
    sess.encode(audios)
    sess.prefill(np.array([prompt, prompt], np.int32), want_logits=False)
    gen = sess.generate(6, eos_id=-1)
    for b in range(2):
        assert np.array_equal(gen[b], stepwise[b])
    # stop token: generation halts per sequence and the stop token is not emitted
    sess.encode(audios)
    sess.prefill(np.array([prompt, prompt], np.int32), want_logits=False)
    stop = int(stepwise[0][2])
    gen2 = sess.generate(6, eos_id=stop)
    assert stop not in gen2[0].tolist() and len(gen2[0]) <= 2


def test_erf_vs_tanh_gelu_and_errors():
    cfg, ck, sup, beg = whisper_setup("whisper_tiny_test")
    eng, _lib = sub("engine"), sub("_lib")
    a = [unit_audio(40, 16000)]
    prompt = np.array([[cfg.sot_id, cfg.first_language_id, cfg.transcribe_id, cfg.no_timestamps_id]], np.int32)
    outs = {}
    for kind in (False, True):
        s = eng.WhisperSession.from_checkpoint(cfg, ck, precision=F32, suppress_tokens=sup, begin_suppress_tokens=beg, gelu_tanh=kind)
        s.encode(a)
        outs[kind] = s.prefill(prompt)[1]
        orc = WhisperOracle(cfg, ck, sup, beg, gelu="tanh" if kind else "erf")
        ref = orc.greedy(a, [prompt[0].tolist()], 1)["logits"][0][0]
        assert np.abs(outs[kind][0] - ref).max() < LOGIT_TOL_F32
    assert np.abs(outs[False] - outs[True]).max() > 1e-5          # the two GELUs are distinguishable (max gap 4.7e-4 per activation)
    s = eng.WhisperSession.from_checkpoint(cfg, ck, precision=BF16, suppress_tokens=sup, begin_suppress_tokens=beg)
    with pytest.raises(_lib.AsrError):
        s.prefill(prompt)                                         # no encoded batch yet
    with pytest.raises(_lib.AsrError):
        s.encode([unit_audio(1, 399)])                            # shorter than n_fft
    s.encode(a)
    with pytest.raises(_lib.AsrError):
        s.prefill(np.array([[cfg.vocab + 5]], np.int32))          # token id out of range


def test_penalty_greedy_head_matches_reference_goldens():
    """set_penalty(): APPLY_PENALTY + GREEDY_SEARCH on the device (history and counter never leave HBM), f32 mode, ragged
    batch, through generate() and through explicit decode steps; switching back to 1.0 restores plain greedy."""
    g = load_golden("whisper_tiny")
    cfg, ck, sup, beg, sess = _session(str(g["cfg_name"]), F32)
    cases = [c for _, c in golden_cases(g)]
    value, rng = float(g["penalty_value"]), int(g["penalty_range"])
    audios = [unit_audio(c["audio_seed"], c["n_samples"]) for c in cases]
    prompts = np.stack([c["prompt"] for c in cases])
    n = cases[0]["penalty_token_ids"].size
    sess.encode(audios)
    sess.set_penalty(value, rng)
    sess.prefill(prompts, want_logits=False)
    gen = sess.generate(n, eos_id=-1)
    first, _ = sess.prefill(prompts, want_logits=False)       # the id history restarts with every prefill
    nxt = [first]
    for _ in range(n - 1):
        nxt.append(sess.decode(None)[0])
    step = np.stack(nxt, 1)
    sess.set_penalty(1.0, rng)
    sess.prefill(prompts, want_logits=False)
    plain = sess.generate(n, eos_id=-1)
    for b, c in enumerate(cases):
        safe = c["penalty_margin"] > 2e-3
        k = int(np.argmin(safe)) if not safe.all() else n
        assert k > rng + 1, "golden too fragile to exercise the penalty"
        assert np.array_equal(gen[b][:k], c["penalty_token_ids"][:k]), b
        assert np.array_equal(step[b][:k], c["penalty_token_ids"][:k]), b
        m = int(np.argmin(c["margin"] > 2e-3)) if not (c["margin"] > 2e-3).all() else c["token_ids"].size
        assert np.array_equal(plain[b][:m], c["plain_token_ids"][:m]), b
    with pytest.raises(Exception, match="range"):
        sess.set_penalty(0.8, 65)


def test_sampling_head_matches_reference_goldens_and_is_reproducible():
    """set_sampling(): with the reference's own uniforms the device head reproduces the reference module's picks (f32 mode, ragged
    batch, one step at a time); with the built-in generator the same seed gives the same tokens and another seed differs."""
    g = load_golden("whisper_tiny")
    cfg, ck, sup, beg, sess = _session(str(g["cfg_name"]), F32)
    cases = [c for _, c in golden_cases(g)]
    t, k, p, rp = (float(v) for v in g["sampling_params"])
    k = int(k)
    audios = [unit_audio(c["audio_seed"], c["n_samples"]) for c in cases]
    prompts = np.stack([c["prompt"] for c in cases])
    steps = cases[0]["sampling_token_ids"].size
    sess.encode(audios)
    sess.set_sampling(True, t, k, p, rp, seed=1)
    got = []
    for s in range(steps):
        sess.set_sampling_noise(np.stack([c["sampling_noise"][s] for c in cases]))
        got.append(sess.prefill(prompts, want_logits=False)[0] if s == 0 else sess.decode(None)[0])
    got = np.stack(got, 1)
    for b, c in enumerate(cases):
        assert np.array_equal(got[b], c["sampling_token_ids"]), b

    def run(seed):
        sess.set_sampling(True, t, k, p, rp, seed=seed)
        sess.prefill(prompts, want_logits=False)
        return np.stack(sess.generate(12, eos_id=-1))
    a, b2, c2 = run(7), run(7), run(8)
    assert np.array_equal(a, b2) and not np.array_equal(a, c2)
    sess.set_sampling(True, t, 1, p, 1.0, seed=3)                  # top_k = 1 degenerates to greedy
    sess.prefill(prompts, want_logits=False)
    k1 = sess.generate(6, eos_id=-1)
    sess.set_sampling(False)
    sess.prefill(prompts, want_logits=False)
    greedy = sess.generate(6, eos_id=-1)
    for x, y, c in zip(k1, greedy, cases):
        m = int(np.argmin(c["margin"] > 2e-3)) if not (c["margin"] > 2e-3).all() else 6
        assert np.array_equal(x[:m], y[:m])
    with pytest.raises(Exception, match="top_k"):
        sess.set_sampling(True, t, 65, p, rp)


def test_maximum_30s_window_with_a_short_one_f32():
    """The 30 s maximum (1500 encoder positions, multi-chunk attention, 1500-key cross-attention) next to a 1 s clip: prefill and
    greedy steps against the batch-1 oracle; clips beyond the maximum are refused."""
    cfg, ck, sup, beg, sess = _session("whisper_tiny_test", F32)
    orc = WhisperOracle(cfg, ck, sup, beg)
    lens = [cfg.max_audio_len, 16000]
    audios = [unit_audio(820 + i, n) for i, n in enumerate(lens)]
    npos = sess.encode(audios)
    assert [int(t) for t in npos] == [1500, 50]
    prompt = [cfg.sot_id, cfg.first_language_id, cfg.transcribe_id, cfg.no_timestamps_id]
    nxt, logits = sess.prefill(np.asarray([prompt, prompt], np.int32))
    steps_l, steps_i = [logits], [nxt]
    for _ in range(3):
        nxt, logits = sess.decode(None, want_logits=True)
        steps_l.append(logits)
        steps_i.append(nxt)
    got_l, got_i = np.stack(steps_l, 1), np.stack(steps_i, 1)
    want = orc.greedy(audios, [prompt, prompt], 4)
    for b in range(2):
        assert np.abs(got_l[b] - want["logits"][b]).max() < LOGIT_TOL_F32, b
        srt = np.sort(want["logits"][b], axis=1)
        if ((srt[:, -1] - srt[:, -2]) > 2 * LOGIT_TOL_F32).all():
            assert np.array_equal(got_i[b], want["token_ids"][b])
    with pytest.raises(Exception, match="max_audio_len|samples"):
        sess.encode([unit_audio(1, cfg.max_audio_len + 160)])


def test_concurrent_sessions_on_separate_streams_match_sequential_runs():
    """Serving mode: independent sessions driven by separate host threads (own HIP stream and state, ctypes releases the GIL) -- each must
    produce exactly what it produces alone."""
    import threading
    cfg, ck, sup, beg, _ = _session("whisper_mid_test", BF16)
    eng = sub("engine")
    prompt = np.tile(np.array([[cfg.sot_id, cfg.first_language_id, cfg.transcribe_id, cfg.no_timestamps_id]], np.int32), (3, 1))
    jobs = [[unit_audio(700 + 10 * j + i, 48000 + 16000 * i) for i in range(3)] for j in range(3)]

    def run(sess, audios, out, k, reps):
        for _ in range(reps):
            sess.encode(audios)
            sess.prefill(prompt, want_logits=False)
            out[k] = sess.generate(12, eos_id=-1)

    sessions = [eng.WhisperSession.from_checkpoint(cfg, ck, precision=BF16, suppress_tokens=sup, begin_suppress_tokens=beg) for _ in jobs]
    alone, together = [None] * 3, [None] * 3
    for k, (s, a) in enumerate(zip(sessions, jobs)):
        run(s, a, alone, k, 1)
    ths = [threading.Thread(target=run, args=(s, a, together, k, 4)) for k, (s, a) in enumerate(zip(sessions, jobs))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for k in range(3):
        for b in range(3):
            assert np.array_equal(alone[k][b], together[k][b]), (k, b)
    # the same through the pool: batches go to whichever session is free, results come back in submission order
    def transcribe(sess, audios):
        sess.encode(audios)
        sess.prefill(prompt, want_logits=False)
        return sess.generate(12, eos_id=-1)

    it = iter(sessions)
    with sub("pool").SessionPool(lambda i: next(it), n=3) as pool:
        got = pool.map(transcribe, [(jobs[k % 3],) for k in range(7)])
        bad = pool.submit(lambda sess: sess.prefill(np.zeros((5, 1), np.int32)))
        with pytest.raises(Exception):
            bad.result()
    for k in range(7):
        for b in range(3):
            assert np.array_equal(got[k][b], alone[k % 3][b]), (k, b)


def test_large_v3_bf16_batch32x30s_vs_golden_and_oracle():
    """BASELINE.json configs[2] at its real size: Whisper-large-v3 bf16, 32 x 30 s in one batch, prefill + 3 decode steps. This is the
    batch that dispatches the 256 x 256 encoder tiles, the 20-head attention over 1500 keys and the skinny / tiled decode GEMMs at
    K = 1280 / 5120 -- no smaller test reaches them. Utterance 0 (and its duplicate in slot 31) is the 30 s clip of the reference-minted
    golden; utterance 1 is checked against the f32 oracle run here; the duplicate must match bit for bit (batch invariance)."""
    g = load_golden("whisper_large_v3")
    cfg, ck, sup, beg, sess = _session("whisper_large_v3", BF16)
    c0 = [c for _, c in golden_cases(g)][0]
    assert int(c0["n_samples"]) == 480000
    B, n_new = 32, int(g["n_new"])
    audios = [unit_audio(9000 + i, 480000) for i in range(B)]
    audios[0] = unit_audio(c0["audio_seed"], c0["n_samples"])
    audios[31] = audios[0].copy()
    prompt = c0["prompt"]
    prompts = np.tile(prompt[None], (B, 1))
    npos = sess.encode(audios)
    assert all(int(t) == 1500 for t in npos)
    # (the cross-K/V slabs of this batch are 7.9 GB: they are pinned at full dimensions by the f32 test above, here through the logits)
    nxt, logits = sess.prefill(prompts)
    steps = [logits]
    for _ in range(n_new - 1):
        nxt, logits = sess.decode(None, want_logits=True)
        steps.append(logits)
    got = np.stack(steps, 1)                                        # (B, n_new, V)
    assert np.array_equal(got[0], got[31])
    scale = float(np.abs(c0["top1"]).max())
    err = np.abs(got[0][:, ::53] - c0["logits"]).max()
    assert err < 2e-3 * max(scale, 50.0), (err, scale)              # logits are O(100) with these weights: 2e-3 relative
    assert np.abs(np.sort(got[0], axis=1)[:, -1] - c0["top1"]).max() < 2e-3 * max(scale, 50.0)
    orc = WhisperOracle(cfg, ck, sup, beg)
    ref = orc.greedy([audios[1]], [prompt.tolist()], 2)
    for s in range(2):
        assert np.abs(got[1][s] - ref["logits"][0][s]).max() < 2e-3 * max(scale, 50.0)


def test_large_v3_bf16_batch64x8s_vs_golden_and_oracle():
    """The north-star's batch-64 point: Whisper-large-v3 bf16, 64 x 8 s in one batch, prefill + 3 decode steps. 64 decoder rows take the
    four-row-tile instances of the decode GEMM (4 x 1 and 4 x 2 tiles, LayerNorm folded in) and the tiled split-K pass for fc2 -- no smaller
    batch dispatches them. Slot 0 (duplicated in slot 63) is the 8 s clip of the reference-minted golden; slot 1 is checked against the
    f32 oracle run here; the duplicate must match bit for bit."""
    g = load_golden("whisper_large_v3")
    cfg, ck, sup, beg, sess = _session("whisper_large_v3", BF16)
    c1 = [c for _, c in golden_cases(g)][1]
    assert int(c1["n_samples"]) == 128000
    B, n_new = 64, int(g["n_new"])
    audios = [unit_audio(9500 + i, 128000) for i in range(B)]
    audios[0] = unit_audio(c1["audio_seed"], c1["n_samples"])
    audios[63] = audios[0].copy()
    prompt = c1["prompt"]
    prompts = np.tile(prompt[None], (B, 1))
    npos = sess.encode(audios)
    assert all(int(t) == 400 for t in npos)
    nxt, logits = sess.prefill(prompts)
    steps = [logits]
    for _ in range(n_new - 1):
        nxt, logits = sess.decode(None, want_logits=True)
        steps.append(logits)
    got = np.stack(steps, 1)                                        # (B, n_new, V)
    assert np.array_equal(got[0], got[63])
    scale = max(float(np.abs(c1["top1"]).max()), 50.0)
    assert np.abs(got[0][:, ::53] - c1["logits"]).max() < 2e-3 * scale
    assert np.abs(np.sort(got[0], axis=1)[:, -1] - c1["top1"]).max() < 2e-3 * scale
    orc = WhisperOracle(cfg, ck, sup, beg)
    ref = orc.greedy([audios[1]], [prompt.tolist()], 2)
    for s in range(2):
        assert np.abs(got[1][s] - ref["logits"][0][s]).max() < 2e-3 * scale


@pytest.mark.parametrize("prec", [BF16, F32])
def test_paged_self_kv_cache_equals_the_contiguous_extents_bit_for_bit(prec, monkeypatch):
    """The decoder's self-KV cache is PAGED (the default): 16-position pages from a pool, one block table per batch shared by the layers, pages handed
    out a generation at a time as the sequences grow (csrc/whisper.hip: ensure_kv_pages; the attention kernels address rows through the table). Three
    sessions on the same batch -- paged, paged with the page ids of every generation permuted (ASR_KV_PAGE_SHUFFLE=1: the kernels must follow the table,
    not an assumed order), and one contiguous max_target_positions extent per sequence and head (ASR_KV_PAGED=0, the layout of rounds 1-2) -- must
    return the same logits bit for bit over a prefill and 75 teacher-forced steps: that crosses four page boundaries and the first growth of the pool
    (prompt + 48 positions = 4 generations, doubled at position 64: old pages copied, table rewritten, graph re-captured)."""
    cfg, ck, sup, beg = whisper_setup("whisper_d256_test")
    eng = sub("engine")
    audios = [unit_audio(501, 64000), unit_audio(502, 25600), unit_audio(503, 128000)]
    prompt = [cfg.sot_id, cfg.first_language_id, cfg.transcribe_id, cfg.no_timestamps_id]
    prompts = np.array([prompt] * 3, np.int32)
    rng = np.random.default_rng(77)
    forced = rng.integers(0, cfg.eot_id, (3, 75)).astype(np.int32)
    out = {}
    for mode, env in (("paged", {}), ("shuffled", {"ASR_KV_PAGE_SHUFFLE": "1"}), ("contiguous", {"ASR_KV_PAGED": "0"})):
        for k in ("ASR_KV_PAGE_SHUFFLE", "ASR_KV_PAGED"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sess = eng.WhisperSession.from_checkpoint(cfg, ck, precision=prec, suppress_tokens=sup, begin_suppress_tokens=beg)
        sess.encode(audios)
        _, logits = sess.prefill(prompts)
        steps = [logits]
        for s in range(forced.shape[1]):
            _, logits = sess.decode(np.ascontiguousarray(forced[:, s:s + 1]), want_logits=True)
            steps.append(logits)
        first = np.stack(steps, 1)
        sess.encode(audios[::-1])                        # a second batch on the same session reuses pool and table
        _, again = sess.prefill(prompts)
        out[mode] = (first, again)
        del sess
    for mode in ("shuffled", "contiguous"):
        assert np.array_equal(out["paged"][0], out[mode][0]), mode
        assert np.array_equal(out["paged"][1], out[mode][1]), mode
    assert np.isfinite(out["paged"][0]).all()


@pytest.mark.parametrize("prec", [BF16, 2])
def test_twin_sequences_in_one_batch_get_identical_logits(prec):
    """A sequence's logits must not depend on its position in the batch: 40 sequences of which 0 / 35 and 7 / 33 are the same audio, over a prefill and 20
    graph-replayed steps with the penalty head on, give identical logits and picks for the twins. prec 2 = FP8W: the cross-K/V scale rows are addressed per
    head over the WHOLE batch (DecAttnArgs::scale_ld). (Round 5's decode chains, which this test was written for, were removed in round 6.)"""
    name = "whisper_d256_test"
    cfg, ck, sup, beg = whisper_setup(name)
    eng = sub("engine")
    B = 40
    audios = [unit_audio(800 + i, (25600, 64000, 128000)[i % 3]) for i in range(B)]
    audios[35] = audios[0].copy()
    audios[33] = audios[7].copy()
    prompt = [cfg.sot_id, cfg.first_language_id, cfg.transcribe_id, cfg.no_timestamps_id]
    prompts = np.array([prompt] * B, np.int32)
    sess = eng.WhisperSession.from_checkpoint(cfg, ck, precision=prec, suppress_tokens=sup, begin_suppress_tokens=beg)
    sess.set_penalty(0.9, 4)
    sess.encode(audios)
    nxt, logits = sess.prefill(prompts)
    ids, steps = [nxt], [logits]
    for _ in range(20):
        nxt, logits = sess.decode(None, want_logits=True)
        ids.append(nxt)
        steps.append(logits)
    ids, lg = np.stack(ids, 1), np.stack(steps, 1)[..., :cfg.vocab]
    assert np.isfinite(lg).all()
    assert np.array_equal(lg[0], lg[35]) and np.array_equal(lg[7], lg[33])
    assert np.array_equal(ids[0], ids[35]) and np.array_equal(ids[7], ids[33])




@pytest.mark.parametrize("B", [20, 40])
def test_decode_step_forms_of_round_6_agree_with_the_forms_they_replaced(monkeypatch, B):
    """Round 6 changed HOW three pieces of a decode step run, not what they compute: the decode GEMM multiplies a narrow output as two row blocks side by side instead
    of one block or a K split (ASR_DECODE_RB=1: off; 20 sequences = 16 + 4 rows, 40 = 32 + 8), the single-token cross-attention runs in one pass with a running
    soft-max (ASR_DECODE_ATTN_ONLINE=0: two passes). Both sessions see the same ragged batch over a prefill and 6 teacher-forced steps: the logits may differ by
    summation order only (a few bf16 ulps of an O(100) logit), and the new forms are what every other test of this file runs on."""
    name = "whisper_d256_test"
    cfg, ck, sup, beg = whisper_setup(name)
    eng = sub("engine")
    audios = [unit_audio(1500 + i, (25600, 64000, 128000)[i % 3]) for i in range(B)]
    prompt = [cfg.sot_id, cfg.first_language_id, cfg.transcribe_id, cfg.no_timestamps_id]
    prompts = np.array([prompt] * B, np.int32)
    out = {}
    for key, env in (("new", {}), ("old", {"ASR_DECODE_RB": "1", "ASR_DECODE_ATTN_ONLINE": "0"})):
        for k in ("ASR_DECODE_RB", "ASR_DECODE_ATTN_ONLINE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sess = eng.WhisperSession.from_checkpoint(cfg, ck, precision=BF16, suppress_tokens=sup, begin_suppress_tokens=beg)
        sess.encode(audios)
        nxt, logits = sess.prefill(prompts)
        steps, forced = [logits], out["new"][1] if key == "old" else []
        for s in range(6):
            ids = np.ascontiguousarray(forced[s]) if key == "old" else nxt
            if key == "new":
                forced.append(nxt.copy())
            nxt, logits = sess.decode(ids.reshape(B, 1), want_logits=True)
            steps.append(logits)
        out[key] = (np.stack(steps, 1)[..., :cfg.vocab], forced)
        del sess
    a, b = out["new"][0], out["old"][0]
    scale = float(np.abs(b).max())
    diff = float(np.abs(a - b).max())
    print(f"whisper_d256, {B} sequences: new vs old decode-step forms differ by {diff:.4f} on logits of |max| {scale:.1f}")
    assert np.isfinite(a).all() and diff < 1e-2 * scale
    assert (a.argmax(-1) == b.argmax(-1)).mean() > 0.97
