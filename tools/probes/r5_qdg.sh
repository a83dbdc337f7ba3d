set -x
mkdir -p gpurun_out/r5n
python -m pytest tests/test_qwen_fp8_gpu.py tests/test_qwen_asr_gpu.py tests/test_shim_qwen_gpu.py tests/test_mixed_gpu.py -q -x -s > gpurun_out/r5n/pytest.txt 2>&1
tail -4 gpurun_out/r5n/pytest.txt
line() { python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms per batch,', d['value'], 'audio-s/s,', d.get('decode_ms_per_token'), 'ms per token', d.get('ms'))"; }
{
for v in 1 0 1 0; do echo "ASR_QWEN_DECODE_GEMM=$v greedy: $(ASR_QWEN_DECODE_GEMM=$v python bench.py --workload qwen --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | line)"; done
echo "fp8w: $(python bench.py --workload qwen --fp8 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | line)"
echo "beam 5: $(python bench.py --workload qwen --beam 5 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | line)"
} > gpurun_out/r5n/bench.txt 2>&1
cat gpurun_out/r5n/bench.txt
