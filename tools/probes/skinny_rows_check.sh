# the skinny GEMM after a change: its parity tests (ops, Qwen3-ASR incl. FP8W, the small-batch SenseVoice / Paraformer paths) and Qwen3-ASR decode at one and at 64 sequences
mkdir -p gpurun_out/sk
python -m pytest tests/test_ops_gpu.py tests/test_qwen_asr_gpu.py tests/test_qwen_fp8_gpu.py tests/test_sensevoice_gpu.py -q -x > gpurun_out/sk/pytest.txt 2>&1
tail -3 gpurun_out/sk/pytest.txt
for b in 1 64; do echo "qwen B=$b: $(python bench.py --workload qwen --batch $b --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('decode_ms_per_token'))")"; done
