set -x
mkdir -p gpurun_out/r5c
python -m pytest tests/test_sensevoice_gpu.py -m gpu -x -q > gpurun_out/r5c/pytest_sv.txt 2>&1
tail -15 gpurun_out/r5c/pytest_sv.txt
for f in 1 0 1 0; do echo "FFN22=$f: $(ASR_SANM_BLOCK_FFN22=$f python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per step,', d['value'], 'audio-s/s, block', d['roofline']['avg_block_us'], 'us')")"; done > gpurun_out/r5c/ffn22_ab.txt 2>&1
cat gpurun_out/r5c/ffn22_ab.txt
ASR_SANM_BLOCK_DBG=10 python tools/probes/sanm_block_clock.py > gpurun_out/r5c/clock_ffn22.txt 2>&1
python -m pytest tests/test_paraformer_streaming_gpu.py tests/test_mixed_gpu.py tests/test_natural_audio_gpu.py tests/test_qwen_asr_gpu.py tests/test_paraformer_gpu.py -m gpu -q > gpurun_out/r5c/pytest_rest.txt 2>&1
tail -30 gpurun_out/r5c/pytest_rest.txt
