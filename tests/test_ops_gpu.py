"""Per-kernel parity: each HIP kernel, through the C ABI, against a plain torch fp32 statement of the op."""
import numpy as np
import pytest
import torch

from conftest import sub

pytestmark = pytest.mark.gpu

BF16, F32 = 0, 1


def _bf16_round(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


@pytest.mark.parametrize("M,N,K", [(137, 128, 64), (300, 384, 576), (1000, 1536, 512), (260, 512, 2048)])
@pytest.mark.parametrize("prec", [BF16, F32])
def test_gemm_matches_torch(M, N, K, prec):
    eng = sub("engine")
    rng = np.random.default_rng(M + N + K)
    # asymmetric operands: catches transposed fragments / C-layout swaps
    a = rng.standard_normal((M, K), dtype=np.float32) + np.linspace(-1, 1, K, dtype=np.float32)[None, :]
    w = rng.standard_normal((N, K), dtype=np.float32) * 0.1 + np.linspace(0, 0.3, N, dtype=np.float32)[:, None]
    bias = rng.standard_normal((N,), dtype=np.float32)
    out = eng.op_gemm(a, w, bias, act=1, precision=prec)
    if prec == BF16:
        a, w = _bf16_round(a), _bf16_round(w)
    ref = torch.relu(torch.from_numpy(a).double() @ torch.from_numpy(w).double().t() + torch.from_numpy(bias).double()).numpy()
    tol = 2e-3 if prec == BF16 else 2e-4     # bf16: operands pre-rounded, only f32 accumulation order differs
    assert np.abs(out - ref).max() <= tol * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("act", [0, 2, 3])
def test_gemm_activations(act):
    eng = sub("engine")
    rng = np.random.default_rng(act)
    a = rng.standard_normal((130, 128), dtype=np.float32)
    w = rng.standard_normal((128, 128), dtype=np.float32) * 0.2
    out = eng.op_gemm(a, w, None, act=act, precision=F32)
    z = torch.from_numpy(a) @ torch.from_numpy(w).t()
    ref = {0: z, 2: torch.nn.functional.gelu(z), 3: torch.nn.functional.gelu(z, approximate="tanh")}[act].numpy()
    assert np.abs(out - ref).max() < 1e-4


@pytest.mark.parametrize("D", [512, 560, 1280])
def test_layernorm(D):
    eng = sub("engine")
    rng = np.random.default_rng(D)
    x = rng.standard_normal((77, D), dtype=np.float32) * 3 + 1
    g = rng.standard_normal((D,), dtype=np.float32)
    b = rng.standard_normal((D,), dtype=np.float32)
    ref = torch.nn.functional.layer_norm(torch.from_numpy(x), (D,), torch.from_numpy(g), torch.from_numpy(b), 1e-5).numpy()
    out = eng.op_layernorm(x, g, b, 1e-5, precision=F32)
    assert np.abs(out - ref).max() < 2e-5
    out16 = eng.op_layernorm(x, g, b, 1e-5, precision=BF16)
    assert np.abs(out16 - ref).max() < 0.02 * np.abs(ref).max()
    ref_na = torch.nn.functional.layer_norm(torch.from_numpy(x), (D,), None, None, 1e-5).numpy()
    assert np.abs(eng.op_layernorm(x, None, None, 1e-5, precision=F32) - ref_na).max() < 2e-5


def _attn_ref(q, k, v, lens, H, D):
    out = np.zeros_like(q)
    r = 0
    for T in lens:
        qq = torch.from_numpy(q[r:r + T]).double().reshape(T, H, D).transpose(0, 1)
        kk = torch.from_numpy(k[r:r + T]).double().reshape(T, H, D).transpose(0, 1)
        vv = torch.from_numpy(v[r:r + T]).double().reshape(T, H, D).transpose(0, 1)
        p = torch.softmax(qq @ kk.transpose(1, 2), dim=-1)
        out[r:r + T] = (p @ vv).transpose(0, 1).reshape(T, H * D).float().numpy()
        r += T
    return out


@pytest.mark.parametrize("prec,H,D,lens", [
    (BF16, 4, 128, [137, 45, 1, 64, 129, 300]),
    (BF16, 2, 64, [400, 33, 128]),
    (F32, 4, 128, [137, 45, 1, 64, 129]),
    (F32, 2, 64, [200, 33]),
])
def test_attention_ragged(prec, H, D, lens):
    eng = sub("engine")
    rng = np.random.default_rng(sum(lens))
    n = sum(lens)
    scale = D ** -0.25
    q = rng.standard_normal((n, H * D), dtype=np.float32) * scale
    k = rng.standard_normal((n, H * D), dtype=np.float32) * scale
    v = rng.standard_normal((n, H * D), dtype=np.float32)
    # one spiked key per sequence: forces the online-softmax running-max rescale branch
    r = 0
    for T in lens:
        k[r + T // 2] *= 6.0
        r += T
    if prec == BF16:
        q, k, v = _bf16_round(q), _bf16_round(k), _bf16_round(v)
    out = eng.op_attention(q, k, v, lens, H, D, precision=prec)
    ref = _attn_ref(q, k, v, lens, H, D)
    tol = 2e-2 if prec == BF16 else 2e-5
    assert np.isfinite(out).all()
    assert np.abs(out - ref).max() < tol


@pytest.mark.parametrize("prec", [BF16, F32])
def test_fsmn_zero_padding_inside_each_utterance(prec):
    eng = sub("engine")
    rng = np.random.default_rng(5)
    lens = [137, 3, 20, 1]
    C, K = 256, 11
    v = rng.standard_normal((sum(lens), C), dtype=np.float32)
    w = rng.standard_normal((C, K), dtype=np.float32)
    b = rng.standard_normal((C,), dtype=np.float32)
    if prec == BF16:
        v = _bf16_round(v)
    out = eng.op_fsmn(v, w, b, lens, precision=prec)
    r = 0
    for T in lens:
        x = torch.from_numpy(v[r:r + T]).t().unsqueeze(0)
        ref = torch.nn.functional.conv1d(x, torch.from_numpy(w).unsqueeze(1), torch.from_numpy(b), padding=5, groups=C)[0].t().numpy()
        assert np.abs(out[r:r + T] - ref).max() < 1e-4
        r += T


def test_ctc_collapse_edge_cases():
    """Circular next-neighbour rule (SenseVoice/Export_SenseVoice.py:291-292)."""
    eng = sub("engine")
    from oracle.sensevoice_oracle import SenseVoiceOracle
    cases = [
        [0, 0, 0, 0],                   # all blank
        [5, 5, 5, 5],                   # one run touching both ends: circular compare drops everything
        [7, 0, 3, 3, 0, 7],             # first == last: final token dropped by the wrap-around
        [1, 2, 2, 0, 2, 3],
        [4],                            # single frame: compared with itself
        list(np.random.default_rng(0).integers(0, 4, size=700)),   # > 256 frames: multi-pass compaction
    ]
    lens = [len(c) for c in cases]
    flat = np.concatenate([np.asarray(c, dtype=np.int32) for c in cases])
    got = eng.op_ctc_collapse(flat, lens, blank_id=0)
    for c, g in zip(cases, got):
        want = SenseVoiceOracle.ctc_collapse(torch.tensor(c, dtype=torch.int64), 0).numpy()
        assert np.array_equal(g, want), (c[:10], g, want)


@pytest.mark.parametrize("M,N,K", [(32, 384, 1280), (7, 128, 256), (64, 256, 5120), (1, 1280, 1280)])
def test_skinny_gemm_decode_shapes(M, N, K):
    """Decode shapes: up to 32 rows stream the weights (skinny kernel: 8-way K split, LDS reduction); 33..64 rows take the tiled
    split-K pass (activation rows shared across 64 columns)."""
    eng = sub("engine")
    rng = np.random.default_rng(M * N + K)
    a = _bf16_round(rng.standard_normal((M, K), dtype=np.float32) + np.linspace(-1, 1, K, dtype=np.float32)[None, :])
    w = _bf16_round(rng.standard_normal((N, K), dtype=np.float32) * 0.05 + np.linspace(0, 0.1, N, dtype=np.float32)[:, None])
    bias = rng.standard_normal((N,), dtype=np.float32)
    out = eng.op_gemm(a, w, bias, act=2, precision=BF16)
    ref = torch.nn.functional.gelu(torch.from_numpy(a).double() @ torch.from_numpy(w).double().t() + torch.from_numpy(bias).double()).numpy()
    assert np.abs(out - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("affine", [False, True])
def test_skinny_gemm_with_fused_layernorm(affine):
    eng = sub("engine")
    rng = np.random.default_rng(9)
    M, N, K = 24, 256, 1280
    x = rng.standard_normal((M, K), dtype=np.float32) * 2 + 0.5
    w = _bf16_round(rng.standard_normal((N, K), dtype=np.float32) * 0.05)
    bias = rng.standard_normal((N,), dtype=np.float32)
    g = rng.standard_normal((K,), dtype=np.float32) if affine else None
    b = rng.standard_normal((K,), dtype=np.float32) if affine else None
    out = eng.op_gemm_ln(x, w, bias, g, b)
    ln = torch.nn.functional.layer_norm(torch.from_numpy(x), (K,), torch.from_numpy(g) if affine else None, torch.from_numpy(b) if affine else None, 1e-5)
    ref = (ln.to(torch.bfloat16).double() @ torch.from_numpy(w).double().t() + torch.from_numpy(bias).double()).numpy()
    assert np.abs(out - ref).max() <= 3e-3 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("variant", [5, 6])
@pytest.mark.parametrize("M,N,K,act", [(1440, 512, 512, 0), (1447, 256, 576, 1), (2304, 128, 2048, 0)])
def test_gemm_144_row_tiles(variant, M, N, K, act):
    """144 x 128 tiles (8 waves, K-halves summed through LDS), 4-stage and 2-stage rings, ragged last tile."""
    eng = sub("engine")
    rng = np.random.default_rng(M + N + K)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    try:
        eng.op_gemm_set_variant(variant)
        got = eng.op_gemm(a, w, b, act=act, precision=0)
    finally:
        eng.op_gemm_set_variant(-1)
    ref = _bf16_round(a).astype(np.float64) @ _bf16_round(w).astype(np.float64).T + b
    if act == 1:
        ref = np.maximum(ref, 0)
    assert np.abs(got - ref).max() < 2e-3 * max(1.0, np.abs(ref).max())
    base = eng.op_gemm(a, w, b, act=act, precision=0)              # default kernel: same products, other summation order
    assert np.abs(got - base).max() < 2e-3 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("M,N,K,act", [(300, 256, 128, 0), (2500, 512, 576, 1), (4096, 1024, 64, 0)])
def test_gemm_256_tiles(M, N, K, act):
    """256 x 256 tiles (8 waves of 64 x 128, two 64 KiB stages), ragged last row tile with clamped operand rows."""
    eng = sub("engine")
    rng = np.random.default_rng(M + N + K)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    try:
        eng.op_gemm_set_variant(7)
        got = eng.op_gemm(a, w, b, act=act, precision=0)
    finally:
        eng.op_gemm_set_variant(-1)
    ref = _bf16_round(a).astype(np.float64) @ _bf16_round(w).astype(np.float64).T + b
    if act == 1:
        ref = np.maximum(ref, 0)
    assert np.abs(got - ref).max() < 2e-3 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("M,N,K,act", [(137, 512, 2048, 0), (1024, 512, 2048, 1), (144, 128, 4096, 0)])
def test_tiled_split_k_for_small_grids(M, N, K, act):
    """A few tiles walking a long K: the launcher splits K across workgroups (f32 partials + an ordered reduce / epilogue launch)."""
    eng = sub("engine")
    rng = np.random.default_rng(M + N + K)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    got = eng.op_gemm(a, w, b, act=act, precision=0)
    again = eng.op_gemm(a, w, b, act=act, precision=0)
    ref = _bf16_round(a).astype(np.float64) @ _bf16_round(w).astype(np.float64).T + b
    if act == 1:
        ref = np.maximum(ref, 0)
    assert np.abs(got - ref).max() < 2e-3 * max(1.0, np.abs(ref).max())
    assert np.array_equal(got, again)                             # fixed summation order


def test_skinny_split_k_hand_over_is_exact(monkeypatch):
    """Opt-in split-K across workgroups of the skinny (weight-streaming) GEMM: partial sums handed over through agent-scope
    atomics, last-arriver reduction in split order => bit-reproducible, and a single 137-row window through the 9-row-tile
    instance agrees with the default tiled path."""
    from helpers import kaldi_audio, sensevoice_setup
    cfg, ck = sensevoice_setup("sensevoice_small")
    eng = sub("engine")
    a = [kaldi_audio(901, 128000)]
    base = eng.SenseVoiceSession.from_checkpoint(cfg, ck, precision=0)
    base.taps(True)
    t0 = base.run(a, [0])
    l0 = base.tap("logits").copy()
    monkeypatch.setenv("ASR_SKINNY_SPLITK", "1")
    monkeypatch.setenv("ASR_SKINNY_M144", "1")
    # the switches are re-read at session creation
    sess = eng.SenseVoiceSession.from_checkpoint(cfg, ck, precision=0)
    sess.taps(True)
    t1 = sess.run(a, [0])
    l1 = sess.tap("logits").copy()
    t2 = sess.run(a, [0])
    assert np.array_equal(t1[0], t2[0]) and np.array_equal(l1, sess.tap("logits"))
    assert np.abs(l1 - l0).max() < 0.2


# ---- the kernels the batch-64 headline dispatches to (wide tiles), at the sizes that select them -------------------------------
def _ln_ref(a, w, bias):
    x = torch.from_numpy(_bf16_round(a)).double()
    h = torch.nn.functional.layer_norm(x, (x.shape[1],), None, None, 1e-5).float()
    return torch.relu(h @ torch.from_numpy(_bf16_round(w)).t() + torch.from_numpy(bias)).numpy()


@pytest.mark.parametrize("M,want", [(9216, "t288w"),        # 64 windows of 144 rows: BASELINE configs[1], 32 x 8 = 256 tiles
                                    (16272, "t288w"),       # 113 windows: 56.5 row tiles -> ragged last 288-row tile
                                    (13824, "t144w"),       # 96 windows: 384 tiles of 288 x 256 is not whole rounds, 768 of 144 x 256 is
                                    (1296, "t144")])        # 9 windows: the tiling the small-batch tests already reach
def test_gemm_layernorm_folded_wide_tiles(M, want):
    """FFN-1 as the product runs it at batch 64: LayerNorm evaluated inside the GEMM from handed-over row statistics and the column
    sums of the bf16 weights, bias + ReLU, bf16 store -- through the dispatcher, which must pick the 288 x 256 / 144 x 256 tiles."""
    probe = sub("_probe")
    N, K = 2048, 512
    rng = np.random.default_rng(M)
    a = (rng.standard_normal((M, K)) * (1.0 + 0.5 * rng.random((M, 1))) + 0.3 * rng.standard_normal((M, 1))).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K) + np.linspace(-0.01, 0.01, N)[:, None]).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) * 0.1
    out, kern = probe.gemm(a, w, bias, act=1, ln=True)
    assert kern == want, kern
    ref = _ln_ref(a, w, bias)
    err = np.abs(out["lo"] - ref)
    assert (err <= 2.0 ** -7 * np.abs(ref) + 6e-3).all(), float(err.max())          # bf16 store (2^-8 relative) + f32 summation order
    # the same product through the 144 x 128 tiles (variant 6): other tiling, other summation order, same function
    base, kern6 = probe.gemm(a[:2304], w, bias, act=1, ln=True, variant=6)
    assert kern6 == "t144"
    assert np.abs(base["lo"] - out["lo"][:2304]).max() < 2e-2


@pytest.mark.parametrize("M", [9216, 9353])
@pytest.mark.parametrize("pp", [1, 0])
def test_gemm_argmax_wide_tiles(M, pp, monkeypatch):
    """CTC head at batch 64: the persistent ping-pong kernel (the default since round 5) or -- ASR_GEMM_AMAX_PP=0 -- the 288 x 256 tiles, with the per-slab
    arg-max epilogue + reduce; ids must equal the arg-max of the f32 product wherever the top-2 margin exceeds the accumulation-order noise, and equal the
    128 x 128-tile path's ids there."""
    probe = sub("_probe")
    monkeypatch.setenv("ASR_GEMM_AMAX_PP", str(pp))
    N, K, V = 25088, 512, 25055
    rng = np.random.default_rng(M)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = (rng.standard_normal(N) * 0.5).astype(np.float32)
    out, kern = probe.gemm(a, w, bias, argmax=True, n_valid=V)
    assert kern == ("pp_amax" if pp else "t288w_amax"), kern
    z = torch.from_numpy(_bf16_round(a)) @ torch.from_numpy(_bf16_round(w)).t() + torch.from_numpy(bias)
    z[:, V:] = -np.inf
    top2 = torch.topk(z, 2, dim=1)
    safe = (top2.values[:, 0] - top2.values[:, 1]).numpy() > 1e-3
    assert safe.mean() > 0.99
    assert np.array_equal(out["ids"][safe], top2.indices[:, 0].numpy()[safe])
    assert (out["ids"] < V).all()
    base, kern2 = probe.gemm(a, w, bias, argmax=True, n_valid=V, variant=2)
    assert kern2 == "pipe"
    assert np.array_equal(base["ids"][safe], out["ids"][safe])


def test_gemm_producer_epilogue_at_batch64():
    """Out-projection at batch 64 (M = 9216, N = K = 512): additive term, f32 + bf16 stores and the row statistics of the bf16 values
    (what the LayerNorm-folded consumers read) from the 144-row tiles with the 4-stage ring."""
    probe = sub("_probe")
    M, N, K = 9216, 512, 512
    rng = np.random.default_rng(7)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    add = rng.standard_normal((M, N)).astype(np.float32)
    out, kern = probe.gemm(a, w, None, add=add, want=("lo", "f32", "stats"))
    assert kern == "t144", kern
    ref = (torch.from_numpy(_bf16_round(a)) @ torch.from_numpy(_bf16_round(w)).t()).numpy() + add
    assert np.abs(out["f32"] - ref).max() < 2e-3 * max(1.0, np.abs(ref).max())
    assert np.array_equal(out["lo"], _bf16_round(out["f32"]))
    lo = out["lo"].reshape(M, N // 32, 32).astype(np.float64)
    assert np.abs(out["stats"][..., 0] - lo.sum(-1)).max() < 1e-3
    assert np.abs(out["stats"][..., 1] - (lo * lo).sum(-1)).max() < 1e-2


def test_skinny_gemm_four_row_tiles(monkeypatch):
    """33..64 rows: plain GEMMs take the tiled split-K pass by default; ASR_SKINNY_MAX_M=64 keeps them on the weight-streaming
    kernel's 4-row-tile instance (the one Qwen3-ASR's RMSNorm-folded projections use), which must give the same product."""
    eng = sub("engine")
    rng = np.random.default_rng(48)
    for M, N, K in [(48, 384, 512), (64, 256, 1024)]:
        a = rng.standard_normal((M, K)).astype(np.float32)
        w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        ref = _bf16_round(a).astype(np.float64) @ _bf16_round(w).astype(np.float64).T + b
        monkeypatch.setenv("ASR_SKINNY_MAX_M", "64")
        got = eng.op_gemm(a, w, b, act=0, precision=0)
        monkeypatch.delenv("ASR_SKINNY_MAX_M")
        tiled = eng.op_gemm(a, w, b, act=0, precision=0)
        for o in (got, tiled):
            assert np.abs(o - ref).max() < 2e-3 * max(1.0, np.abs(ref).max())
