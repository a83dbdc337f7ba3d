set -x
mkdir -p gpurun_out/r5q
python -m pytest tests -m gpu -q > gpurun_out/r5q/pytest_all.txt 2>&1
tail -6 gpurun_out/r5q/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5q/smoke.txt 2>&1
tail -2 gpurun_out/r5q/smoke.txt
python bench.py --workload paraformer-streaming --steps 16 --warmup 8 > gpurun_out/r5q/bench_paraformer_streaming.json 2> /dev/null
python bench.py --workload paraformer-streaming --batch 256 --steps 16 --warmup 8 --no-cpu-baseline > gpurun_out/r5q/bench_paraformer_streaming_256.json 2> /dev/null
python bench.py --workload mixed --beam 5 --steps 6 --warmup 1 > gpurun_out/r5q/bench_mixed_beam5.json 2> /dev/null
python bench.py --workload qwen --steps 6 --warmup 2 --inflight 3 > gpurun_out/r5q/bench_qwen.json 2> /dev/null
python bench.py --workload qwen --beam 5 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r5q/bench_qwen_beam5.json 2> /dev/null
python bench.py --workload qwen --fp8 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r5q/bench_qwen_fp8.json 2> /dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/r5q/bench_sensevoice.json 2> /dev/null
for f in gpurun_out/r5q/bench_*.json; do python -c "import sys, json; d = json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'], d.get('decode_ms_per_token'), d.get('concurrent'), d.get('tenant_slowdown'))"; done
