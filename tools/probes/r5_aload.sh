# activation fragments requested in batches (decode_gemm.hip, gemm.hip: gemm_bf16_skinny): parity tests of the GEMM families, then the decode benches
set -x
mkdir -p gpurun_out/r5i
python -m pytest tests/test_ops_gpu.py tests/test_whisper_fp8_gpu.py tests/test_qwen_fp8_gpu.py -q -x > gpurun_out/r5i/pytest_ops.txt 2>&1
tail -3 gpurun_out/r5i/pytest_ops.txt
python -m pytest tests/test_whisper_gpu.py tests/test_qwen_asr_gpu.py -q -x > gpurun_out/r5i/pytest_wq.txt 2>&1
tail -3 gpurun_out/r5i/pytest_wq.txt
line() { python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms per batch,', d['value'], 'audio-s/s,', d.get('decode_ms_per_token'), 'ms per token', d.get('ms'))"; }
{
echo "whisper B=64 x 8 s: $(python bench.py --workload whisper --batch 64 --steps 5 --warmup 3 --no-cpu-baseline 2>/dev/null | line)"
echo "whisper B=32 x 8 s: $(python bench.py --workload whisper --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | line)"
echo "whisper B=1 x 8 s: $(python bench.py --workload whisper --batch 1 --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | line)"
echo "whisper B=32 x 30 s: $(python bench.py --workload whisper --seconds 30 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | line)"
echo "whisper B=32 x 30 s fp8: $(python bench.py --workload whisper --fp8 --seconds 30 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | line)"
echo "qwen greedy: $(python bench.py --workload qwen --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | line)"
echo "qwen fp8w: $(python bench.py --workload qwen --fp8 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | line)"
echo "qwen beam 5: $(python bench.py --workload qwen --beam 5 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | line)"
echo "paraformer-streaming 256: $(python bench.py --workload paraformer-streaming --batch 256 --steps 16 --warmup 8 --no-cpu-baseline 2>/dev/null | line)"
} > gpurun_out/r5i/bench.txt 2>&1
cat gpurun_out/r5i/bench.txt
