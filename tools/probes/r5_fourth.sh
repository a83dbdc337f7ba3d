set -x
mkdir -p gpurun_out/r5d
python -m pytest tests/test_sensevoice_gpu.py -m gpu -q -k "ragged or headline_dispatch" > gpurun_out/r5d/pytest_block.txt 2>&1
tail -8 gpurun_out/r5d/pytest_block.txt
ASR_SANM_BLOCK_FFN22=1 python -m pytest tests/test_sensevoice_gpu.py tests/test_paraformer_gpu.py -m gpu -q -k "headline_dispatch or block_kernel or trained_margins" > gpurun_out/r5d/pytest_ffn22.txt 2>&1
tail -8 gpurun_out/r5d/pytest_ffn22.txt
for f in 1 0 1 0; do echo "FFN22=$f: $(ASR_SANM_BLOCK_FFN22=$f python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per step,', d['value'], 'audio-s/s, block', d['roofline']['avg_block_us'], 'us')")"; done > gpurun_out/r5d/ffn22_ab.txt 2>&1
ASR_SANM_BLOCK_FFN22=1 ASR_SANM_BLOCK_DBG=10 python tools/probes/sanm_block_clock.py 2>&1 | grep -v amdgpu.ids | tail -17 > gpurun_out/r5d/clock_ffn22.txt
ASR_SANM_BLOCK_DBG=10 python tools/probes/sanm_block_clock.py 2>&1 | grep -v amdgpu.ids | tail -17 > gpurun_out/r5d/clock_r4form.txt
ASR_SANM_BLOCK8_OPT=2048 ASR_SANM_BLOCK_DBG=10 python tools/probes/sanm_block_clock.py 2>&1 | grep -v amdgpu.ids | tail -23 > gpurun_out/r5d/clock_inside_A.txt
python -m pytest tests/test_paraformer_streaming_gpu.py tests/test_mixed_gpu.py tests/test_natural_audio_gpu.py -m gpu -q > gpurun_out/r5d/pytest_rest.txt 2>&1
tail -30 gpurun_out/r5d/pytest_rest.txt
python -m pytest tests/test_qwen_asr_gpu.py -m gpu -q -k "max_seq_len or paged" > gpurun_out/r5d/pytest_qwen.txt 2>&1
tail -5 gpurun_out/r5d/pytest_qwen.txt
