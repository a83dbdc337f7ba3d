# the whole -m gpu suite + smoke() on the current build, then the streaming / mixed bench lines
set -x
mkdir -p gpurun_out/full
python -m pytest tests -m gpu -q > gpurun_out/full/pytest_all.txt 2>&1
tail -6 gpurun_out/full/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full/smoke.txt 2>&1
tail -2 gpurun_out/full/smoke.txt
python bench.py --workload paraformer-streaming --steps 16 --warmup 8 --no-cpu-baseline > gpurun_out/full/bench_paraformer_streaming.json 2> /dev/null
python bench.py --workload paraformer-streaming --batch 256 --steps 16 --warmup 8 --no-cpu-baseline > gpurun_out/full/bench_paraformer_streaming_256.json 2> /dev/null
python bench.py --workload mixed --beam 5 --steps 6 --warmup 1 > gpurun_out/full/bench_mixed_beam5.json 2> /dev/null
for f in gpurun_out/full/bench_*.json; do python -c "import sys, json; d = json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'], d.get('concurrent'), d.get('tenant_slowdown'))"; done
