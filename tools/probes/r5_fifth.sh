set -x
mkdir -p gpurun_out/r5e
for o in 0 1024 0 1024 256; do echo "ASR_SANM_BLOCK8_OPT=$o: $(ASR_SANM_BLOCK8_OPT=$o python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per step,', d['value'], 'audio-s/s, block', d['roofline']['avg_block_us'], 'us')")"; done > gpurun_out/r5e/opt1024_ab.txt 2>&1
cat gpurun_out/r5e/opt1024_ab.txt
ASR_SANM_BLOCK8_OPT=1024 ASR_SANM_BLOCK_DBG=10 python tools/probes/sanm_block_clock.py 2>&1 | grep -v amdgpu.ids | tail -17 > gpurun_out/r5e/clock_opt1024.txt
ASR_SANM_BLOCK8_OPT=1024 python -m pytest tests/test_sensevoice_gpu.py -m gpu -q -k "ragged or headline_dispatch" > gpurun_out/r5e/pytest_opt1024.txt 2>&1
tail -5 gpurun_out/r5e/pytest_opt1024.txt
python -m pytest tests/test_paraformer_streaming_gpu.py tests/test_mixed_gpu.py tests/test_natural_audio_gpu.py -m gpu -q > gpurun_out/r5e/pytest_rest.txt 2>&1
tail -12 gpurun_out/r5e/pytest_rest.txt
