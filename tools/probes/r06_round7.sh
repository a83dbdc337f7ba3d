# round 6, seventh measurement: which payloads of the block kernel's exchanges go out write-through (default) and which by ordinary stores inside an XCD
# (ASR_SANM_BLOCK8_OPT: 256 = every payload, 512 = the f16 partial images only -- one reader each)
set -x
mkdir -p gpurun_out/r06g
for o in 0 512 256 768 0 512 256 768; do
  echo "ASR_SANM_BLOCK8_OPT=$o: $(ASR_SANM_BLOCK8_OPT=$o python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per step,', d['value'], 'audio-s/s,', d['roofline']['avg_block_us'], 'us per block')")"
done > gpurun_out/r06g/store_policy_ab.txt 2>&1
cat gpurun_out/r06g/store_policy_ab.txt
for o in 512 768; do ASR_SANM_BLOCK8_OPT=$o python -m pytest tests/test_sensevoice_gpu.py -m gpu -q -k "block_kernel or trained_margins or two_block" > gpurun_out/r06g/pytest_opt$o.txt 2>&1; tail -3 gpurun_out/r06g/pytest_opt$o.txt; done
