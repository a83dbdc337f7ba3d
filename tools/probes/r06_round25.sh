# round 6: the agreement tests of the late decode-step forms with the forms they replaced
set -x
mkdir -p gpurun_out/r06y
python -m pytest tests/test_whisper_gpu.py tests/test_qwen_asr_gpu.py -m gpu -q -s -k "decode_step_forms or two_granule" > gpurun_out/r06y/pytest.txt 2>&1; grep -n "differ\|passed\|failed\|Error\|assert" gpurun_out/r06y/pytest.txt | tail -12
