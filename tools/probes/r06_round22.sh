# round 6, twenty-second / -third call: decode GEMM with two row blocks (32 + 32 or 16 + 16 rows) side by side for narrow outputs (ASR_DECODE_RB=1: off): tests + A/B
set -x
mkdir -p gpurun_out/r06w
python -m pytest tests/test_whisper_gpu.py tests/test_whisper_fp8_gpu.py tests/test_whisper_mxfp4_gpu.py tests/test_qwen_asr_gpu.py tests/test_qwen_fp8_gpu.py tests/test_shim_whisper_gpu.py tests/test_shim_qwen_gpu.py -m gpu -q -x > gpurun_out/r06w/pytest.txt 2>&1; tail -n 4 gpurun_out/r06w/pytest.txt
for args in "whisper --batch 32" "whisper --seconds 30 --batch 32" "whisper --fp8 --seconds 30 --batch 32" "whisper --batch 24" "whisper --batch 64"; do
  for v in 0 1 0 1; do
    echo "$args ASR_DECODE_RB=$v: $(ASR_DECODE_RB=$v python bench.py --workload $args --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per batch,', d['value'], 'audio-s/s, dec_gemm', d['kernels']['dec_gemm'])")"
  done
done > gpurun_out/r06w/decode_rb_ab.txt 2>&1
grep "ASR_DECODE_RB" gpurun_out/r06w/decode_rb_ab.txt | grep -v "^+"
