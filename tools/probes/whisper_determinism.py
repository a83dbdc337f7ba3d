"""Whisper decode steps at 64 rows (decode GEMM at 32-column granules: 69.6 KB of dynamic LDS, a size that shares a CU) alone and beside a busy SenseVoice session:
logits of prefill + N graph-replayed steps must be bit-identical (the check that found stream_attn_kernel's fault, applied to the other > 64 KB kernel that can share a CU)."""
import sys, threading
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import sub
from helpers import kaldi_audio, sensevoice_setup
from test_oracle_whisper import unit_audio, whisper_setup
eng = sub("engine")
name = sys.argv[1] if len(sys.argv) > 1 else "whisper_d256_test"
cfg, ck, sup, beg = whisper_setup(name)
B, steps = 64, 12
audios = [unit_audio(900 + i, (25600, 64000, 128000)[i % 3]) for i in range(B)]
prompt = [cfg.sot_id, cfg.first_language_id, cfg.transcribe_id, cfg.no_timestamps_id]
prompts = np.array([prompt] * B, np.int32)
sess = eng.WhisperSession.from_checkpoint(cfg, ck, precision=0, suppress_tokens=sup, begin_suppress_tokens=beg)
def run():
    sess.encode(audios)
    nxt, logits = sess.prefill(prompts)
    out = [logits]
    for _ in range(steps):
        nxt, logits = sess.decode(None, want_logits=True)
        out.append(logits)
    return np.stack(out, 1)
first = run()
def series(label, n):
    bad, worst = 0, 0.0
    for p in range(n):
        got = run()
        if not np.array_equal(got, first):
            bad += 1
            worst = max(worst, float(np.abs(got - first).max()))
    print(label, ":", bad, "of", n, "runs differ from the first run; max |logit diff|", worst, flush=True)
series("alone", 6)
scfg, sck = sensevoice_setup("sensevoice_small")
ssess = eng.SenseVoiceSession.from_checkpoint(scfg, sck, precision=0)
saud = [kaldi_audio(7000 + i, 128000) for i in range(16)]
stop = threading.Event()
def worker():
    while not stop.is_set():
        ssess.run(saud, [0] * 16)
t = threading.Thread(target=worker); t.start()
series("beside a SenseVoice session (16 x 8 s batches, four-launch path)", 16)
stop.set(); t.join()
