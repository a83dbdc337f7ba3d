# fbank_split_kernel with four basis steps in flight (csrc/kernels.hip), the arg-max head on the ping-pong GEMM, front-end parity tests
set -x
mkdir -p gpurun_out/r5d
python -m pytest tests/test_sensevoice_gpu.py tests/test_natural_audio_gpu.py tests/test_ops_gpu.py tests/test_paraformer_gpu.py -q -x > gpurun_out/r5d/pytest.txt 2>&1
tail -5 gpurun_out/r5d/pytest.txt
line() { python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d.get('kernels', {}); print(d['ms_per_step'], 'ms per step,', d['value'], 'audio-s/s;', {n: v for n, v in k.items()} if isinstance(k, dict) else '')"; }
for v in "ASR_GEMM_AMAX_PP=0" "ASR_GEMM_AMAX_PP=1" "ASR_GEMM_AMAX_PP=0" "ASR_GEMM_AMAX_PP=1"; do
  echo "$v: $(env $v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | line)"
done > gpurun_out/r5d/ab.txt 2>&1
cat gpurun_out/r5d/ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5d/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --profile-steps 1 > $GRAFT_REPO_ROOT/gpurun_out/r5d/stats.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r5d/stats -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-150
find $GRAFT_REPO_ROOT/gpurun_out/r5d/stats -name "*kernel_trace.csv" -delete
