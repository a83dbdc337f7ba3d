# round 6, twenty-first call: the whole -m gpu suite + smoke on the tree with the one-pass cross-attention and the two-granule skinny GEMM; the Whisper / Qwen bench lines they move
set -x
mkdir -p gpurun_out/r06u
python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r06u/pytest_all.txt 2>&1
tail -10 gpurun_out/r06u/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06u/smoke.txt 2>&1; tail -1 gpurun_out/r06u/smoke.txt
python bench.py --workload qwen --steps 6 --warmup 2 --inflight 3 > gpurun_out/r06u/bench_qwen.json 2> /dev/null
python bench.py --workload qwen --fp8 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r06u/bench_qwen_fp8.json 2> /dev/null
python bench.py --workload qwen --mxfp4 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r06u/bench_qwen_mxfp4.json 2> /dev/null
python bench.py --workload whisper --steps 6 --warmup 3 --inflight 3 > gpurun_out/r06u/bench_whisper.json 2> /dev/null
python bench.py --workload whisper --batch 64 --steps 5 --warmup 3 --inflight 2 --no-cpu-baseline > gpurun_out/r06u/bench_whisper_b64.json 2> /dev/null
python bench.py --workload whisper --fp8 --seconds 30 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r06u/bench_whisper30_fp8.json 2> /dev/null
python bench.py --workload whisper --mxfp4 --seconds 30 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r06u/bench_whisper30_mxfp4.json 2> /dev/null
python bench.py --workload whisper --fp8 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r06u/bench_whisper_fp8.json 2> /dev/null
python bench.py --workload whisper --fp8mm --batch 64 --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/r06u/bench_whisper_b64_fp8mm.json 2> /dev/null
python bench.py --workload mixed --beam 5 --steps 6 --warmup 1 > gpurun_out/r06u/bench_mixed_beam5.json 2> /dev/null
for f in gpurun_out/r06u/bench_*.json; do python -c "
import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
