"""GPU: precision mode ASR_PRECISION_MXFP4W (include/asr_mi355x.h) -- Whisper decoder projections as OCP MXFP4 (e2m1 elements, one e8m0 scale per 32 input
channels; the 4-bit counterpart of the reference's MatMulNBits Q4 decoders, /root/reference/README.md:70 q4f32), cross-K/V as in FP8W.

What pins it:
  * the block quantiser against a numpy restatement of the OCP Microscaling rule (shared exponent = floor(log2(amax)) - 2, round to nearest with ties to the
    even code, saturation at 6, zero blocks at scale 1), nibble order and scale bytes included;
  * the nibble-weight decode GEMM against the SAME kernel over the dequantised bf16 weights, bit for bit (e2m1 x 2^e is exact in bf16, so the widening
    instruction adds nothing), and against a float64 product;
  * a whole MXFP4W session against the same session with ASR_FP8_FAKE=1 (identical quantisation, bf16 kernels throughout), bit for bit on the logits;
  * the quantisation error itself against the f32 oracle, with the budget written down next to FP8W's and bf16's on the same input."""
import os

import numpy as np
import pytest

from conftest import sub
from oracle.whisper_oracle import WhisperOracle
from test_oracle_whisper import unit_audio, whisper_setup
from test_whisper_fp8_gpu import _run, _session

pytestmark = pytest.mark.gpu

BF16, FP8W, MXFP4W = 0, 2, 4
LEVELS = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])


def _mxfp4_reference(wb):
    """OCP MX v1.0, e2m1 + e8m0 over blocks of 32 along the last axis: (codes [N][K] uint8 incl. the sign bit, scale bytes [N][K / 32], dequantised f64)."""
    N, K = wb.shape
    blk = wb.reshape(N, K // 32, 32).astype(np.float64)
    amax = np.abs(blk).max(axis=2)
    e = np.where(amax > 0, np.floor(np.log2(np.where(amax > 0, amax, 1.0))) - 2, 0.0)
    sb = np.clip(e + 127, 1, 254).astype(np.int64)
    scale = 2.0 ** (sb - 127.0)
    v = np.abs(blk) / scale[..., None]
    # round to nearest level, ties to the even code; above 6 saturates
    code = np.zeros(v.shape, np.int64)
    for lo, hi, c_lo in ((0.0, 0.5, 0), (0.5, 1.0, 1), (1.0, 1.5, 2), (1.5, 2.0, 3), (2.0, 3.0, 4), (3.0, 4.0, 5), (4.0, 6.0, 6)):
        mid = 0.5 * (lo + hi)
        up = (v > mid) | ((v == mid) & (c_lo % 2 == 1))                       # a tie goes to the even code
        code = np.where((v >= lo) & up, c_lo + 1, code)
        code = np.where((v >= lo) & (v < hi) & ~up, c_lo, code)
    code = np.where(v >= 6.0, 7, code)
    sign = (np.signbit(blk)).astype(np.int64)
    dq = np.where(sign == 1, -1.0, 1.0) * LEVELS[code] * scale[..., None]
    return (code | (sign << 3)).reshape(N, K).astype(np.uint8), sb.astype(np.uint8), dq.reshape(N, K)


def test_block_quantiser_is_ocp_mxfp4():
    probe = sub("_probe")
    rng = np.random.default_rng(11)
    w = (rng.standard_normal((40, 256)) * np.exp(rng.uniform(-6, 3, (40, 1)))).astype(np.float32)
    w[3] = 0.0                                                                # all-zero blocks keep scale 1
    w[4, 32:64] = 0.0
    # ties, saturation, the sign of zero, tiny next to large: block 0 of row 5 has amax 6 -> scale 1
    w[5, :32] = [6.0, -6.0, 5.0, -5.0, 3.5, 2.5, 1.75, 1.25, 0.75, 0.25, -0.25, 0.2, 0.3, 4.9, 5.1, 7.5 * 0 + 5.5,
                 -0.0, 0.0, 1.0, -1.0, 1.5, 2.0, 3.0, 4.0, 0.5, -0.5, 1e-3, -1e-3, 2.4, 2.6, 3.4, 3.6]
    w[6, :32] = np.linspace(-7.9, 7.9, 32)                                    # amax 7.9: scale 1, |x| > 6 saturates at 6
    q, sc, dq = probe.quantize_mxfp4(w)
    wb = probe._bf16_to_f32(probe._bf16_bits(w))                              # the kernel sees bf16-rounded weights
    want_code, want_sc, want_dq = _mxfp4_reference(wb)
    assert np.array_equal(sc, want_sc)
    got_code = np.stack([q & 15, q >> 4], axis=2).reshape(q.shape[0], -1)     # element 2 i in the low nibble of byte i
    zero = (want_code & 7) == 0                                               # (the sign of a zero code carries no value)
    assert np.array_equal(got_code & 7, want_code & 7)
    assert np.array_equal((got_code >> 3)[~zero], (want_code >> 3)[~zero])
    assert np.array_equal(dq.astype(np.float64), want_dq)                     # and the dequantisation is exact in bf16
    amax = np.abs(wb.reshape(40, -1, 32)).max(axis=2)
    s = 2.0 ** (sc.astype(np.float64) - 127)
    assert (((amax / s >= 4.0) & (amax / s < 8.0)) | (amax == 0)).all()


@pytest.mark.parametrize("M,N,K,fold", [(32, 768, 256, True), (7, 256, 1024, False), (64, 1280, 1280, True), (32, 1280, 5120, False),
                                         (64, 5120, 1280, True), (48, 5120, 1280, False), (1, 3840, 1280, True), (16, 3840, 1280, True)])
def test_nibble_weight_decode_gemm_equals_the_bf16_kernel_over_dequantised_weights(M, N, K, fold):
    probe = sub("_probe")
    rng = np.random.default_rng(M + N + K)
    a = rng.standard_normal((M, K)).astype(np.float32) * 1.5 + (0.3 if fold else 0.0)
    w = (rng.standard_normal((N, K)) * 0.04 * np.exp(rng.uniform(-2, 2, (N, 1)))).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    q, sc, dq = probe.quantize_mxfp4(w)
    nib_path = probe.decode_gemm_mxfp4(a, q, sc, w_dq=dq, bias=bias, fold=fold)
    bf16_path = probe.decode_gemm(a, w=dq, bias=bias, fold=fold)
    assert np.array_equal(nib_path, bf16_path)               # bit for bit (incl. the split-K hand-over of K = 5120 and the two-column-tile instances of N = 5120)
    ab = probe._bf16_to_f32(probe._bf16_bits(a)).astype(np.float64)
    if fold:
        mu, var = ab.mean(1, keepdims=True), ab.var(1, keepdims=True)
        ab = (ab - mu) / np.sqrt(var + 1e-5)
    want = ab @ dq.astype(np.float64).T + bias
    assert np.abs(nib_path - want).max() < 2e-3 * np.abs(want).max()


def test_mxfp4_session_equals_fake_quantised_bf16_session_and_stays_within_budget_of_the_oracle():
    name = "whisper_d256_test"
    cfg, ck, sup, beg, s4 = _session(name, MXFP4W)
    _, _, _, _, sfake = _session(name, MXFP4W, {"ASR_FP8_FAKE": "1"})
    _, _, _, _, s8 = _session(name, FP8W)
    audios = [unit_audio(71, 64000), unit_audio(72, 25600), unit_audio(73, 128000)]
    prompt = [cfg.sot_id, cfg.first_language_id, cfg.transcribe_id, cfg.no_timestamps_id]
    prompts = np.array([prompt] * 3, np.int32)
    orc = WhisperOracle(cfg, ck, sup, beg)
    ref = orc.greedy(audios, [prompt] * 3, 5)
    forced = np.stack([np.asarray(ref["token_ids"][b][:4], np.int32) for b in range(3)])
    want = np.stack([np.stack(ref["logits"][b][:5]) for b in range(3)])
    l4, lf, l8 = _run(s4, audios, prompts, forced), _run(sfake, audios, prompts, forced), _run(s8, audios, prompts, forced)
    V = cfg.vocab
    assert np.array_equal(l4[..., :V], lf[..., :V])          # the nibble kernels add no error of their own
    scale = float(np.abs(want).max())
    e_4, e_8 = float(np.abs(l4[..., :V] - want).max()), float(np.abs(l8[..., :V] - want).max())
    print(f"whisper_d256 logits |max| {scale:.2f}: fp8w error {e_8:.4f}, mxfp4w error {e_4:.4f}")
    assert e_4 < 6e-2 * scale                                 # e2m1 keeps 2 significant bits: ~10 x e4m3's error per weight
    assert e_4 > e_8                                          # (sanity: 4 bits cost more than 8)
    assert np.array_equal(_run(s4, audios, prompts, forced), l4)              # deterministic
    with pytest.raises(RuntimeError):
        _session("whisper_mid_test", MXFP4W)                  # d_model 384 is not a multiple of 256


def test_mxfp4_batch64_generate_matches_stepwise_decode():
    """64 sequences, the graph-replayed generate() path against explicit decode steps in MXFP4W mode (the 33..64-row fc2 takes the nibble GEMM too)."""
    cfg, ck, sup, beg, s4 = _session("whisper_d256_test", MXFP4W)
    B = 64
    audios = [unit_audio(8600 + b, 40000 + 997 * b) for b in range(B)]
    prompts = np.array([[cfg.sot_id, cfg.first_language_id, cfg.transcribe_id, cfg.no_timestamps_id]] * B, np.int32)
    s4.encode(audios)
    nxt, _ = s4.prefill(prompts)
    steps = [nxt.copy()]
    for _ in range(5):
        nxt, _ = s4.decode(None)
        steps.append(nxt.copy())
    want = np.stack(steps, 1).reshape(B, -1)
    s4.encode(audios)
    s4.prefill(prompts, want_logits=False)
    toks = s4.generate(6, eos_id=-1)
    for b in range(B):
        assert np.array_equal(toks[b], want[b]), b
