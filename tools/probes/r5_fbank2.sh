set -x
mkdir -p gpurun_out/r5e
python -m pytest tests/test_sensevoice_gpu.py tests/test_natural_audio_gpu.py tests/test_ops_gpu.py -q -x > gpurun_out/r5e/pytest.txt 2>&1
tail -3 gpurun_out/r5e/pytest.txt
python -m pytest tests/test_whisper_gpu.py tests/test_qwen_asr_gpu.py -q -x -k "f32_mode or chains or natural or golden" > gpurun_out/r5e/pytest2.txt 2>&1
tail -3 gpurun_out/r5e/pytest2.txt
line() { python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d.get('kernels', {}); print(d['ms_per_step'], 'ms per step,', d['value'], 'audio-s/s;', {n: v.get('ms_per_step') for n, v in k.items()} if isinstance(k, dict) else '')"; }
for v in 1 2; do
  echo "run $v: $(python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | line)"
done > gpurun_out/r5e/ab.txt 2>&1
cat gpurun_out/r5e/ab.txt
