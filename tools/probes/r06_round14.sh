# round 6, fourteenth call: look-ahead chunk requests in the block kernel's A / B / C loops (ASR_SANM_BLOCK8_OPT=1024: one chunk ahead, 2048: two; default: all three in front of the loop)
set -x
mkdir -p gpurun_out/r06n
for o in 0 1024 2048 0 1024 2048; do
  echo "ASR_SANM_BLOCK8_OPT=$o: $(ASR_SANM_BLOCK8_OPT=$o python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per step,', d['value'], 'audio-s/s,', d['roofline']['avg_block_us'], 'us per block')")"
done > gpurun_out/r06n/lookahead_ab.txt 2>&1
grep "^ASR" gpurun_out/r06n/lookahead_ab.txt
for o in 1024 2048; do ASR_SANM_BLOCK8_OPT=$o python -m pytest tests/test_sensevoice_gpu.py -m gpu -q -k "block_kernel or trained_margins or two_block" > gpurun_out/r06n/pytest_opt$o.txt 2>&1; tail -n 3 gpurun_out/r06n/pytest_opt$o.txt; done
for o in 0 1024; do ASR_SANM_BLOCK8_OPT=$o ASR_SANM_BLOCK_DBG=10 python tools/probes/sanm_block_clock.py 2>&1 | grep -v amdgpu.ids | tail -17; done > gpurun_out/r06n/phase_clock.txt 2>&1
cat gpurun_out/r06n/phase_clock.txt
