# round 6, twentieth call: skinny GEMM with two 16-column granules per workgroup where the output is wider than the chip (Qwen3-ASR gate|up at 64 sequences; ASR_SKINNY_NT=1: off)
set -x
mkdir -p gpurun_out/r06t
python -m pytest tests/test_qwen_asr_gpu.py tests/test_qwen_fp8_gpu.py tests/test_shim_qwen_gpu.py tests/test_whisper_gpu.py tests/test_sensevoice_gpu.py -m gpu -q -x > gpurun_out/r06t/pytest.txt 2>&1; tail -n 4 gpurun_out/r06t/pytest.txt
for args in "" "--fp8" "--mxfp4" "--beam 5"; do
  for v in 2 1 2 1; do
    echo "qwen $args ASR_SKINNY_NT=$v: $(ASR_SKINNY_NT=$v python bench.py --workload qwen $args --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per batch,', d['value'], 'audio-s/s, dec_gemm', d['kernels']['dec_gemm'])")"
  done
done > gpurun_out/r06t/skinny_nt_ab.txt 2>&1
grep "^qwen" gpurun_out/r06t/skinny_nt_ab.txt
