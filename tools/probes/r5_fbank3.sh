set -x
mkdir -p gpurun_out/r5f
python -m pytest tests/test_qwen_fp8_gpu.py -q -x -s > gpurun_out/r5f/pytest_qfp8.txt 2>&1
tail -8 gpurun_out/r5f/pytest_qfp8.txt
python -m pytest tests/test_sensevoice_gpu.py tests/test_natural_audio_gpu.py tests/test_ops_gpu.py -q -x > gpurun_out/r5f/pytest.txt 2>&1
tail -3 gpurun_out/r5f/pytest.txt
line() { python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d.get('kernels', {}); print(d['ms_per_step'], 'ms per step,', d['value'], 'audio-s/s;', d.get('decode_ms_per_token'), {n: v.get('ms_per_step') for n, v in list(k.items())[:6]} if isinstance(k, dict) else '')"; }
for v in 1 2; do
  echo "run $v: $(python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | line)"
done > gpurun_out/r5f/ab.txt 2>&1
echo "qwen bf16: $(python bench.py --workload qwen --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | line)" >> gpurun_out/r5f/ab.txt
echo "qwen fp8w: $(python bench.py --workload qwen --fp8 --steps 6 --warmup 2 --no-cpu-baseline 2>gpurun_out/r5f/qfp8.err | line)" >> gpurun_out/r5f/ab.txt
cat gpurun_out/r5f/ab.txt
