# round 6, first measurement: whole -m gpu suite, then the Whisper encoder with the rewritten attention soft-max (kernels.hip: attn_bf16_kernel)
set -x
mkdir -p gpurun_out/r06a
python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r06a/pytest_all.txt 2>&1
tail -25 gpurun_out/r06a/pytest_all.txt
python bench.py --workload whisper --batch 64 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r06a/bench_whisper_b64.json 2> gpurun_out/r06a/bench_whisper_b64.err
python bench.py --workload whisper --seconds 30 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r06a/bench_whisper30.json 2> gpurun_out/r06a/bench_whisper30.err
for f in gpurun_out/r06a/bench_*.json; do python -c "
import json
d = json.loads(open('$f').read().strip().splitlines()[-1])
k = d['kernels']
print('$f', d['ms_per_step'], d['value'], 'attention', k.get('attention'), 'enc', d.get('roofline_encode'), d['roofline']['frac'])
"; done
