# round 6, third measurement: the new tests (foreign-kernel gate, prototype head, MXFP4W), then Whisper in FP8W and MXFP4W next to bf16
set -x
mkdir -p gpurun_out/r06c
python -m pytest tests/test_dist_gpu.py tests/test_whisper_mxfp4_gpu.py tests/test_transcribe_gpu.py "tests/test_whisper_fp8_gpu.py::test_batch64_32_steps_token_for_token_on_a_prototype_head" -m gpu -q -s --durations=5 > gpurun_out/r06c/pytest.txt 2>&1
tail -30 gpurun_out/r06c/pytest.txt
grep -h "prototype head\|mxfp4w error\|passes beside" gpurun_out/r06c/pytest.txt
for mode in "" "--fp8" "--mxfp4"; do
  python bench.py --workload whisper --seconds 30 --steps 3 --warmup 2 --no-cpu-baseline $mode > gpurun_out/r06c/bench_whisper30$mode.json 2> gpurun_out/r06c/bench_whisper30$mode.err
  python bench.py --workload whisper --batch 64 --steps 4 --warmup 2 --no-cpu-baseline $mode > gpurun_out/r06c/bench_whisper_b64$mode.json 2> gpurun_out/r06c/bench_whisper_b64$mode.err
done
for f in gpurun_out/r06c/bench_*.json; do python -c "
import json
d = json.loads(open('$f').read().strip().splitlines()[-1])
k = d['kernels']
print('$f', d['ms_per_step'], d['value'], 'ms/token', d.get('decode_ms_per_token'), 'dec_gemm', k.get('dec_gemm'), 'dec', d.get('roofline_decode', {}).get('frac'))
"; done
