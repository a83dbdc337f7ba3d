set -x
mkdir -p gpurun_out/r5k
python tools/probes/decode_gemm_clock.py > gpurun_out/r5k/decode_gemm_clock.txt 2>&1
grep -v amdgpu.ids gpurun_out/r5k/decode_gemm_clock.txt | grep "^M=" 
python -m pytest tests/test_ops_gpu.py tests/test_whisper_fp8_gpu.py -q -x > gpurun_out/r5k/pytest_ops.txt 2>&1
tail -3 gpurun_out/r5k/pytest_ops.txt
line() { python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms per batch,', d['value'], 'audio-s/s,', d.get('decode_ms_per_token'), 'ms per token', d.get('ms'))"; }
{
echo "whisper B=64 x 8 s: $(python bench.py --workload whisper --batch 64 --steps 5 --warmup 3 --no-cpu-baseline 2>/dev/null | line)"
echo "whisper B=32 x 8 s: $(python bench.py --workload whisper --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | line)"
echo "whisper B=1 x 8 s: $(python bench.py --workload whisper --batch 1 --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | line)"
echo "whisper B=32 x 30 s: $(python bench.py --workload whisper --seconds 30 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | line)"
} > gpurun_out/r5k/bench.txt 2>&1
cat gpurun_out/r5k/bench.txt
