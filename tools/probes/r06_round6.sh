# round 6, sixth measurement: whole -m gpu suite on the final tree + smoke, the block kernel A/B after the attention-body change, Qwen3-ASR in MXFP4W
set -x
mkdir -p gpurun_out/r06f
python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r06f/pytest_all.txt 2>&1
tail -25 gpurun_out/r06f/pytest_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06f/smoke.txt 2>&1; tail -2 gpurun_out/r06f/smoke.txt
for k in 1 0 1; do
  ASR_SANM_BLOCK_FFNK=$k python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r06f/bench_ffnk$k.json 2> gpurun_out/r06f/bench_ffnk$k.err
  python -c "
import json
d = json.loads(open('gpurun_out/r06f/bench_ffnk$k.json').read().strip().splitlines()[-1])
print('FFNK=$k', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('avg_block_us'))
"
done
ASR_SANM_BLOCK_DBG=10 python tools/probes/sanm_block_clock.py 2>&1 | grep -v amdgpu.ids | tail -17 > gpurun_out/r06f/phase_clock.txt; cat gpurun_out/r06f/phase_clock.txt
for m in "" "--fp8" "--mxfp4"; do python bench.py --workload qwen --steps 6 --warmup 2 --no-cpu-baseline $m > gpurun_out/r06f/bench_qwen$m.json 2> gpurun_out/r06f/bench_qwen$m.err; python -c "
import json
d = json.loads(open('gpurun_out/r06f/bench_qwen$m.json').read().strip().splitlines()[-1])
print('qwen $m', d['ms_per_step'], d['value'], d.get('decode_ms_per_token'))
"; done
