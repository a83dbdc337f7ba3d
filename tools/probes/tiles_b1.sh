#!/bin/bash
# SenseVoiceSmall bf16, small batches of 8 s windows, ms per batch: the tile kernel (ASR_SANM_TILES_OPT 0 default, 1 no L2 warm-up, 2 a tile's four heads on one XCD) against the four-launch path (ASR_SANM_TILES=0).
for t in "1 0" "1 1" "1 2" "0 0"; do
  set -- $t
  echo "== ASR_SANM_TILES=$1 ASR_SANM_TILES_OPT=$2"
  ASR_SANM_TILES=$1 ASR_SANM_TILES_OPT=$2 python - <<'PY'
import importlib, sys, time, numpy as np
sys.path.insert(0, '.')
PKG = "automatic-speech-recognition-asr-onnx_amd"
cfgm = importlib.import_module(PKG + ".config"); ckm = importlib.import_module(PKG + ".checkpoints"); eng = importlib.import_module(PKG + ".engine")
cfg = cfgm.sensevoice_small(); ck = ckm.synth_sensevoice_checkpoint(cfg, seed=0)
sess = eng.SenseVoiceSession.from_checkpoint(cfg, ck, precision=0)
for B in (1, 2, 4, 7):
    audio = [ckm.synth_audio("kaldi", 1, 128000, seed=50 + i)[0, 0] for i in range(B)]
    for _ in range(4): sess.run(audio, [0] * B)
    t0 = time.perf_counter()
    for _ in range(30): sess.run(audio, [0] * B)
    print("B", B, "ms per batch", round((time.perf_counter() - t0) / 30 * 1e3, 3))
PY
done
