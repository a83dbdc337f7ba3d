# round 6, thirteenth call: the block kernel's f32 rows (x1 in phase B, x in phase D) stored behind the exchange count-in (ASR_SANM_BLOCK8_OPT=1024)
set -x
mkdir -p gpurun_out/r06m
for o in 0 1024 0 1024 0 1024; do
  echo "ASR_SANM_BLOCK8_OPT=$o: $(ASR_SANM_BLOCK8_OPT=$o python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per step,', d['value'], 'audio-s/s,', d['roofline']['avg_block_us'], 'us per block')")"
done > gpurun_out/r06m/late_f32_ab.txt 2>&1
grep "^ASR" gpurun_out/r06m/late_f32_ab.txt
ASR_SANM_BLOCK8_OPT=1024 python -m pytest tests/test_sensevoice_gpu.py tests/test_paraformer_gpu.py -m gpu -q -k "block_kernel or trained_margins or two_block or paraformer" > gpurun_out/r06m/pytest_opt1024.txt 2>&1; tail -n 3 gpurun_out/r06m/pytest_opt1024.txt
for o in 0 1024; do ASR_SANM_BLOCK8_OPT=$o ASR_SANM_BLOCK_DBG=10 python tools/probes/sanm_block_clock.py 2>&1 | grep -v amdgpu.ids | tail -17; done > gpurun_out/r06m/phase_clock.txt 2>&1
cat gpurun_out/r06m/phase_clock.txt
