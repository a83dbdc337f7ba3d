# round 6, fourth measurement: the block kernel with FFN-2 split over K (f16 partials exchanged, hid stays in LDS) against the round-4 form
set -x
mkdir -p gpurun_out/r06d
python -m pytest tests/test_sensevoice_gpu.py tests/test_paraformer_gpu.py tests/test_shim_gpu.py -m gpu -q -x --durations=5 > gpurun_out/r06d/pytest.txt 2>&1
tail -12 gpurun_out/r06d/pytest.txt
for k in 1 0 1 0; do
  ASR_SANM_BLOCK_FFNK=$k python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r06d/bench_ffnk$k.json 2> gpurun_out/r06d/bench_ffnk$k.err
  python -c "
import json
d = json.loads(open('gpurun_out/r06d/bench_ffnk$k.json').read().strip().splitlines()[-1])
print('FFNK=$k', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('avg_block_us'), d['kernels'].get('sanm_block'))
"
done
for k in 1 0; do echo "=== phase clock FFNK=$k, 64 windows"; ASR_SANM_BLOCK_FFNK=$k ASR_SANM_BLOCK_DBG=10 python tools/probes/sanm_block_clock.py 2>&1 | grep -v amdgpu.ids | tail -17; done > gpurun_out/r06d/phase_clock.txt 2>&1
cat gpurun_out/r06d/phase_clock.txt
for k in 1 0; do echo "=== phase clock FFNK=$k, one window (idle chip)"; CLOCK_B=1 ASR_SANM_BLOCK_MIN=1 ASR_SANM_BLOCK_FFNK=$k ASR_SANM_BLOCK_DBG=10 python tools/probes/sanm_block_clock.py 2>&1 | grep -v amdgpu.ids | tail -17; done > gpurun_out/r06d/phase_clock_b1.txt 2>&1
cat gpurun_out/r06d/phase_clock_b1.txt
