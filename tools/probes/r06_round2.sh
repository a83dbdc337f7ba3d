# round 6, second measurement: the tests that failed in the first (inline-asm v_max3 behind an MFMA: results depended on timing), then the Whisper encoder again
set -x
mkdir -p gpurun_out/r06b
python -m pytest tests/test_dist_gpu.py tests/test_mixed_gpu.py tests/test_ops_gpu.py tests/test_qwen_asr_gpu.py tests/test_qwen_fp8_gpu.py tests/test_transcribe_gpu.py tests/test_whisper_fp8_gpu.py tests/test_whisper_gpu.py tests/test_whisper_host.py tests/test_natural_audio_gpu.py -m gpu -q --durations=5 > gpurun_out/r06b/pytest.txt 2>&1
tail -25 gpurun_out/r06b/pytest.txt
python bench.py --workload whisper --batch 64 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r06b/bench_whisper_b64.json 2> gpurun_out/r06b/bench_whisper_b64.err
python bench.py --workload whisper --seconds 30 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r06b/bench_whisper30.json 2> gpurun_out/r06b/bench_whisper30.err
for f in gpurun_out/r06b/bench_*.json; do python -c "
import json
d = json.loads(open('$f').read().strip().splitlines()[-1])
k = d['kernels']
print('$f', d['ms_per_step'], d['value'], 'attention', k.get('attention'), d['roofline']['frac'])
"; done
