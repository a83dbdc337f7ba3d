set -x
mkdir -p gpurun_out/r5g
python -m pytest tests/test_qwen_fp8_gpu.py -q -s > gpurun_out/r5g/pytest_qfp8.txt 2>&1
tail -8 gpurun_out/r5g/pytest_qfp8.txt
line() { python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d.get('kernels', {}); print(d['ms_per_step'], 'ms per step; fbank', k['fbank']['ms_per_step'])"; }
for v in 0 1 2 4 8 3 7 0; do
  echo "ASR_FBANK_DBG=$v: $(ASR_FBANK_DBG=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | line)"
done > gpurun_out/r5g/fbank_ablations.txt 2>&1
cat gpurun_out/r5g/fbank_ablations.txt
