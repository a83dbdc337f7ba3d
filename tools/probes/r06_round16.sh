# round 6, sixteenth call: Paraformer decoder -- norm2 + FSMN + next LayerNorm as one launch: bit-exactness test + A/B
set -x
mkdir -p gpurun_out/r06p
python -m pytest tests/test_paraformer_gpu.py tests/test_shim_gpu.py tests/test_natural_audio_gpu.py -m gpu -q -x -s > gpurun_out/r06p/pytest_paraformer.txt 2>&1; grep -n "paraformer decoder\|passed\|failed\|Error" gpurun_out/r06p/pytest_paraformer.txt | tail -8
for v in 1 0 1 0; do
  echo "ASR_PF_FSMN_FUSE=$v: $(ASR_PF_FSMN_FUSE=$v python bench.py --workload paraformer --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per step,', d['value'], 'audio-s/s, fsmn', d['kernels']['fsmn'], 'layernorm', d['kernels']['layernorm'])")"
done > gpurun_out/r06p/pf_fsmn_fuse_ab.txt 2>&1
grep "^ASR" gpurun_out/r06p/pf_fsmn_fuse_ab.txt
python bench.py --workload paraformer --steps 10 2>/dev/null | tail -1 > gpurun_out/r06p/bench_paraformer.json
