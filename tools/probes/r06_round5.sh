# round 6, fifth measurement: the small-batch tile kernel with the out-projection and FFN-2 split over K (three meetings per block instead of five)
set -x
mkdir -p gpurun_out/r06e
python -m pytest tests/test_sensevoice_gpu.py tests/test_paraformer_gpu.py tests/test_shim_gpu.py tests/test_natural_audio_gpu.py -m gpu -q -x --durations=5 > gpurun_out/r06e/pytest.txt 2>&1
tail -12 gpurun_out/r06e/pytest.txt
for k in 1 0; do
  echo "== ASR_SANM_TILES_KSPLIT=$k"
  ASR_SANM_TILES_KSPLIT=$k python - <<'PY'
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
done > gpurun_out/r06e/tiles_small_batches.txt 2>&1
cat gpurun_out/r06e/tiles_small_batches.txt
for k in 1 0; do echo "== phase clock ASR_SANM_TILES_KSPLIT=$k (B = 1, then 7)"; ASR_SANM_TILES_KSPLIT=$k bash tools/probes/tiles_clock.sh; done > gpurun_out/r06e/tiles_clock.txt 2>&1
cat gpurun_out/r06e/tiles_clock.txt
