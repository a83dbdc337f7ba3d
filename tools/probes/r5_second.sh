set -x
mkdir -p gpurun_out/r5b
python -m pytest tests/test_paraformer_streaming_gpu.py tests/test_mixed_gpu.py tests/test_natural_audio_gpu.py tests/test_qwen_asr_gpu.py -m gpu -x -q > gpurun_out/r5b/pytest.txt 2>&1
tail -30 gpurun_out/r5b/pytest.txt
bash tools/probes/stream_fused_sweep.sh > gpurun_out/r5b/stream_fused_sweep.txt 2>&1
python bench.py --workload mixed --beam 5 --steps 6 --warmup 1 > gpurun_out/r5b/mixed.json 2> gpurun_out/r5b/mixed.err
