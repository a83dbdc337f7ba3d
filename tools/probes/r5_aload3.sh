set -x
mkdir -p gpurun_out/r5m
python tools/probes/decode_gemm_clock.py > gpurun_out/r5m/decode_gemm_clock.txt 2>&1
grep -v amdgpu.ids gpurun_out/r5m/decode_gemm_clock.txt | grep "^M=  1" 
python -m pytest tests/test_ops_gpu.py tests/test_whisper_fp8_gpu.py tests/test_whisper_gpu.py tests/test_shim_whisper_gpu.py -q -x > gpurun_out/r5m/pytest.txt 2>&1
tail -3 gpurun_out/r5m/pytest.txt
python bench.py --workload whisper --steps 6 --warmup 3 --inflight 3 > gpurun_out/r5m/bench_whisper.json 2> /dev/null
python bench.py --workload whisper --batch 64 --steps 5 --warmup 3 --inflight 2 --no-cpu-baseline > gpurun_out/r5m/bench_whisper_b64.json 2> /dev/null
python bench.py --workload whisper --seconds 30 --steps 3 --warmup 2 --inflight 3 --no-cpu-baseline > gpurun_out/r5m/bench_whisper30.json 2> /dev/null
python bench.py --workload whisper --fp8 --seconds 30 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r5m/bench_whisper30_fp8.json 2> /dev/null
python bench.py --workload whisper --fp8 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r5m/bench_whisper_fp8.json 2> /dev/null
python bench.py --workload whisper --fp8mm --batch 64 --steps 5 --warmup 3 --inflight 2 --no-cpu-baseline > gpurun_out/r5m/bench_whisper_b64_fp8mm.json 2> /dev/null
python bench.py --workload whisper --fp8mm --seconds 30 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r5m/bench_whisper30_fp8mm.json 2> /dev/null
python bench.py --workload whisper --batch 1 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r5m/bench_whisper_b1.json 2> /dev/null
for f in gpurun_out/r5m/bench_*.json; do python -c "import sys, json; d = json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'], d.get('decode_ms_per_token'), d.get('ms'))"; done
