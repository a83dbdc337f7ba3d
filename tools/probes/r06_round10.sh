# round 6, tenth measurement: exchange payloads read through the XCD's L2 (ASR_SANM_BLOCK8_OPT=32768: workgroup-scope L1 invalidate + ordinary loads when the cluster
# shares an XCD) instead of sc1 loads served by memory; + 256: the payloads by ordinary stores as well
set -x
mkdir -p gpurun_out/r06j
for o in 0 32768 33024 0 32768 33024; do
  echo "ASR_SANM_BLOCK8_OPT=$o: $(ASR_SANM_BLOCK8_OPT=$o python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per step,', d['value'], 'audio-s/s,', d['roofline']['avg_block_us'], 'us per block')")"
done > gpurun_out/r06j/l2_loads_ab.txt 2>&1
grep "^ASR" gpurun_out/r06j/l2_loads_ab.txt
for o in 32768 33024; do ASR_SANM_BLOCK8_OPT=$o python -m pytest tests/test_sensevoice_gpu.py -m gpu -q -k "block_kernel or trained_margins or two_block" > gpurun_out/r06j/pytest_opt$o.txt 2>&1; tail -3 gpurun_out/r06j/pytest_opt$o.txt; done
for o in 0 32768 33024; do ASR_SANM_BLOCK8_OPT=$o ASR_SANM_BLOCK_DBG=10 python tools/probes/sanm_block_clock.py 2>&1 | grep -v amdgpu.ids | tail -17; done > gpurun_out/r06j/phase_clock.txt 2>&1
cat gpurun_out/r06j/phase_clock.txt
