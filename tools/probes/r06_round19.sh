# round 6, nineteenth call: single-token cross-attention in one pass with a running soft-max (ASR_DECODE_ATTN_ONLINE=0: the two-pass kernel): Whisper tests + A/B
set -x
mkdir -p gpurun_out/r06s
python -m pytest tests/test_whisper_gpu.py tests/test_whisper_fp8_gpu.py tests/test_whisper_mxfp4_gpu.py tests/test_shim_whisper_gpu.py tests/test_whisper_host.py tests/test_transcribe_gpu.py -m gpu -q -x > gpurun_out/r06s/pytest_whisper.txt 2>&1; tail -n 4 gpurun_out/r06s/pytest_whisper.txt
for args in "--batch 32" "--batch 64" "--seconds 30 --batch 32" "--fp8 --seconds 30 --batch 32"; do
  for v in 1 0 1 0; do
    echo "$args ASR_DECODE_ATTN_ONLINE=$v: $(ASR_DECODE_ATTN_ONLINE=$v python bench.py --workload whisper $args --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per batch,', d['value'], 'audio-s/s, cross', d['kernels']['dec_cross_attn'])")"
  done
done > gpurun_out/r06s/cross_attn_ab.txt 2>&1
grep "ONLINE" gpurun_out/r06s/cross_attn_ab.txt
