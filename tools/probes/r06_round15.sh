# round 6, fifteenth call: offline Paraformer decoder with every layer's cross K / V projection in two GEMMs (ASR_PF_KV_BATCH=0: per layer): parity tests + A/B
set -x
mkdir -p gpurun_out/r06o
python -m pytest tests/test_paraformer_gpu.py tests/test_shim_gpu.py tests/test_natural_audio_gpu.py -m gpu -q -x > gpurun_out/r06o/pytest_paraformer.txt 2>&1; tail -n 4 gpurun_out/r06o/pytest_paraformer.txt
for v in 1 0 1 0; do
  echo "ASR_PF_KV_BATCH=$v: $(ASR_PF_KV_BATCH=$v python bench.py --workload paraformer --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per step,', d['value'], 'audio-s/s, gemm_dec', d['kernels']['gemm_dec'])")"
done > gpurun_out/r06o/pf_kv_batch_ab.txt 2>&1
grep "^ASR" gpurun_out/r06o/pf_kv_batch_ab.txt
python bench.py --workload paraformer --steps 10 2>/dev/null | tail -1 > gpurun_out/r06o/bench_paraformer.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06o/bench_sensevoice.json
python -c "
import json; d = json.load(open('gpurun_out/r06o/bench_sensevoice.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['rocprof'])"
