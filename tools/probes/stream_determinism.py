"""Is a bf16 streaming-Paraformer pass a pure function of its inputs? Same 16 streams x 6 chunks, pass after pass (reset in between): alone with the graph path,
alone with taps on (eager launches), and (argv[1] == 'qwen') beside a Qwen3-ASR session that loops prefill + beam search on the same GPU, then beside the same session idle."""
import importlib, sys, threading
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import sub
from helpers import golden_cases, kaldi_audio, load_golden
from test_oracle_paraformer_streaming import streaming_setup
from test_oracle_qwen_asr import qwen_setup, unit_audio
eng = sub("engine")
gp = load_golden("paraformer_streaming_large")
pcfg, pck = streaming_setup(gp)
chunk, S, n_chunks = int(gp["chunk"]), 16, 6
paudio = [kaldi_audio(9300 + i, n_chunks * chunk) for i in range(S)]
PREC = int(__import__("os").environ.get("PROBE_PREC", "0"))
psess = eng.ParaformerStreamSession(pcfg, pck, precision=PREC, chunk=chunk, max_streams=S)
TAPS = ("enc_in", "s0_ck", "s0_cv", "s0_len") + tuple("s%d_%s" % (i, n) for i in (0, 3) for n in ("h", "qkv", "ctx", "mem", "xb", "ffn", "xa")) + ("enc_out",)
def stream_pass(taps):
    psess.taps(taps)
    psess.reset(-1)
    out, tp = [], []
    for k in range(n_chunks):
        out.append(psess.step(np.stack([a[k * chunk:(k + 1) * chunk] for a in paudio]), list(range(S))))
        if taps:
            tp.append({n: psess.tap(n, dtype=(np.int32 if n == 's0_len' else np.uint16 if PREC == 0 and n.split('_')[-1] in ('h', 'qkv', 'ctx', 'ffn', 'ck', 'cv') and n.startswith('s') else np.float32)).copy().astype(np.float64) for n in TAPS})
    return out, tp
def diff(a, b):
    return [(k, s, a[k][s].tolist(), b[k][s].tolist()) for k in range(n_chunks) for s in range(S) if not np.array_equal(a[k][s], b[k][s])]
solo, _ = stream_pass(False)
_, solo_taps = stream_pass(True)
def series(name, taps, n=12):
    bad = 0
    worst = {t: 0.0 for t in TAPS}
    for p in range(n):
        out, tp = stream_pass(taps)
        d = diff(solo, out)
        bad += bool(d)
        if d: print("  ", name, "pass", p, d[:3])
        if taps:
            for k in range(n_chunks):
                for t in TAPS:
                    a, b = solo_taps[k][t], tp[k][t]
                    if a.shape == b.shape:
                        e = float(np.abs(a - b).max()) if a.size else 0.0
                        if t == "s0_ctx" and e > 0:
                            def f32(u): return (u.astype(np.uint32) << 16).view(np.float32)
                            d = np.argwhere(a != b)
                            rows_ = sorted(set(d[:, 0].tolist())); cols_ = sorted(set(d[:, 1].tolist()))
                            print("      s0_ctx pass", p, "chunk", k, ":", len(d), "elements differ; rows", rows_[:12], "cols", cols_[:6], "..", cols_[-3:], "n_cols", len(cols_),
                                  "| first:", d[0].tolist(), float(f32(a[d[0][0], d[0][1]].astype(np.uint16))), "->", float(f32(b[d[0][0], d[0][1]].astype(np.uint16))), flush=True)
                        if e > worst[t]:
                            worst[t] = e
                            if e > 0: print("    ", name, "pass", p, "chunk", k, "tap", t, "max |diff| vs the solo pass", e, "at row", int(np.argmax(np.abs(a - b).reshape(a.shape[0], -1).max(1))), flush=True)
                    else:
                        print("    ", name, "pass", p, "chunk", k, "tap", t, "shape", a.shape, b.shape)
    print(name, ":", bad, "of", n, "passes differ from the first pass", psess.stream_stats(), {t: v for t, v in worst.items() if v > 0}, flush=True)
series("alone, graph", False)
series("alone, taps on (eager)", True)
if len(sys.argv) > 1 and sys.argv[1] == "sensevoice":
    # another kind of co-tenant: a SenseVoiceSmall session looping 16 x 8 s batches (four-launch path and tile GEMMs: no cluster kernels, no Qwen code)
    from test_oracle_sensevoice import sensevoice_setup
    scfg, sck = sensevoice_setup("sensevoice_small")
    ssess = eng.SenseVoiceSession.from_checkpoint(scfg, sck, precision=0)
    saud = [kaldi_audio(7000 + i, 128000) for i in range(16)]
    stop = threading.Event()
    def sworker():
        while not stop.is_set():
            ssess.run(saud, [0] * 16)
    t = threading.Thread(target=sworker); t.start()
    series("beside a SenseVoice session (16 x 8 s batches), taps on", True, 24)
    stop.set(); t.join()
if len(sys.argv) > 1 and sys.argv[1] == "qwen":
    gq = load_golden("qwen_asr_mid")
    qcfg, qck = qwen_setup(gq)
    width, max_new = (int(v) for v in gq["beam"])
    qcases = [c for _, c in golden_cases(gq) if "beam_tokens" in c]
    qaudios = [unit_audio(c["audio_seed"], c["n_samples"]) for c in qcases]
    head, tail, suffix = gq["head_ids"].tolist(), gq["tail_ids"].tolist(), gq["suffix_ids"].tolist()
    pre = [head + c["query_ids"].tolist() + suffix for c in qcases]
    post = [tail + c["language_tail_ids"].tolist() for c in qcases]
    qsess = eng.QwenAsrSession.from_checkpoint(qcfg, qck, precision=0)
    series("qwen session idle, graph (snapshots on)", False)
    series("qwen session idle, taps on", True)
    stop = threading.Event()
    mode = {"beam": True}
    def worker():
        while not stop.is_set():
            qsess.prefill(qaudios, pre, post)
            if mode["beam"]: qsess.beam_search(width, max_new)
    for name, taps, beam in (("beside qwen prefill + beam search, taps on", True, True), ("beside qwen PREFILL ONLY (heavy kernels), taps on", True, False), ("beside qwen prefill only, graph", False, False)):
        mode["beam"] = beam
        t = threading.Thread(target=worker); stop.clear(); t.start()
        series(name, taps, 24)
        stop.set(); t.join()
