"""taps off -> taps on -> taps off on one streaming session, alone on the GPU (probe for a give-up seen in stream_determinism.py)"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import sub
from helpers import kaldi_audio, load_golden
from test_oracle_paraformer_streaming import streaming_setup
eng = sub("engine")
gp = load_golden("paraformer_streaming_large")
pcfg, pck = streaming_setup(gp)
chunk, S, n_chunks = int(gp["chunk"]), 16, 6
paudio = [kaldi_audio(9300 + i, n_chunks * chunk) for i in range(S)]
psess = eng.ParaformerStreamSession(pcfg, pck, precision=0, chunk=chunk, max_streams=S)
def stream_pass(taps, read):
    psess.taps(taps)
    psess.reset(-1)
    out = []
    for k in range(n_chunks):
        out.append(psess.step(np.stack([a[k * chunk:(k + 1) * chunk] for a in paudio]), list(range(S))))
        if read:
            for n in read: psess.tap(n)
    return out
seq = sys.argv[1] if len(sys.argv) > 1 else "0 1 0"
read = tuple(sys.argv[2].split(",")) if len(sys.argv) > 2 else ()
first = None
for i, t in enumerate(seq.split()):
    try:
        out = stream_pass(t == "1", read if t == "1" else ())
        first = first or out
        same = all(np.array_equal(a, b) for ka, kb in zip(out, first) for a, b in zip(ka, kb))
        print("pass", i, "taps", t, "ok, tokens equal the first pass:", same, psess.stream_stats(), flush=True)
    except Exception as e:
        print("pass", i, "taps", t, "FAILED:", str(e)[:160], flush=True)
        break
